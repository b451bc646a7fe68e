#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ldsp
rm -rf $O
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $O -o p -- python3 tools/lds_pattern.py > $O.log 2>&1
tail -1 $O.log
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lds_b128" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(rows): print(k, rows[k])
PY
