#!/bin/bash
# usage: gpu_pmc.sh <label> "<counters of pass 1>" ["<counters of pass 2>" ...] -- <prof_run args>
# one rocprofv3 run per counter group (--kernel-trace + --pmc only), merged per kernel -> gpurun_out/pmc_<label>.json
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
L=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/pmc_$L.d; mkdir -p $O/pmc_$L.d
i=0; dirs=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
	rocprofv3 --output-format csv --kernel-trace --pmc $1 -d $O/pmc_$L.d/p$i -o b -- python3 tools/prof_run.py "${@:$(( $(printf '%s\n' "$@" | grep -n -x -- '--' | head -1 | cut -d: -f1) + 1 ))}" > $O/pmc_$L.d/p$i.log 2>&1 || { tail -5 $O/pmc_$L.d/p$i.log; exit 1; }
	dirs="$dirs $O/pmc_$L.d/p$i"; i=$((i + 1)); shift
done
python3 tools/pmc_sq.py $O/pmc_$L.json $dirs > /dev/null
python3 - $O/pmc_$L.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    print(k)
    for x, y in v.items():
        print("    %-40s %s" % (x, ("%.4f" % y) if y < 10 else ("%.0f" % y)))
PY
rm -rf $O/pmc_$L.d
