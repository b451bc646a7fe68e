#!/bin/bash
# usage: gpu_pmc_run.sh <prof_run args...>  -- LDS counters + duration per engine kernel for one workload
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/pr
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/pr -o b -- python3 tools/prof_run.py "$@" > $O/pr.log 2>&1
python3 tools/pmc_sq.py $O/pr.json $O/pr > /dev/null
python3 -c "
import json
d=json.load(open('$O/pr.json'))
for k,v in d['kernels'].items():
    print(k, {x: round(y,1) for x,y in v.items() if not x.startswith('derived')})
"
