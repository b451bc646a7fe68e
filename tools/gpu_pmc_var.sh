#!/bin/bash
# usage: gpu_pmc_var.sh "<flags>" ...  -- rebuild with each flag set; LDS counters + duration of the one-panel kernels at 2^20 x 128
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
for f in "$@"; do
  make -C tsqr_gpu_amd/csrc -B -s libtsqr_mi.so EXTRA="$f" 2>&1 | grep -E "error" || true
  rm -rf $O/pv
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/pv -o b -- python3 tools/prof_run.py fp32_tc_cor 3 --n 128 > $O/pv.log 2>&1
  echo "== $f"
  python3 tools/pmc_sq.py $O/pv.json $O/pv > /dev/null
  python3 -c "
import json
d=json.load(open('$O/pv.json'))
for k,v in d['kernels'].items():
    if 'gram_wide' in k: print(k, {x: round(y,1) for x,y in v.items() if not x.startswith('derived')})
"
done
