#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ramp
rm -rf $O
rocprofv3 --output-format csv --kernel-trace -d $O -o r -- python3 tools/ramp.py > $O.log 2>&1
python3 - <<PY
import csv, glob
rows=[]
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tsqrmi" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void tsqrmi::","")[:22], int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
rows.sort()
ap=[d/1e3 for _,k,d in rows if k.startswith("apply")]
gr=[d/1e3 for _,k,d in rows if k.startswith("gram_bf16")]
ch=[d/1e3 for _,k,d in rows if k.startswith("chol")]
print("apply first 36:", " ".join("%.0f"%x for x in ap[:36]))
print("gram  first 36:", " ".join("%.0f"%x for x in gr[:36]))
print("chol  first 36:", " ".join("%.0f"%x for x in ch[:36]))
import statistics as st
print("apply median calls 100..400: %.1f  gram %.1f chol %.1f" % (st.median(ap[100:]), st.median(gr[100:]), st.median(ch[100:])))
PY
tail -6 $O.log
