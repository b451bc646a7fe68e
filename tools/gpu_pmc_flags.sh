#!/bin/bash
# usage: gpu_pmc_flags.sh "<flags>" ...  -- rebuild with each flag set; LDS conflict counters of the headline kernels + bench line
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
for f in "$@"; do
  make -C tsqr_gpu_amd/csrc -B -s libtsqr_mi.so EXTRA="$f" 2>&1 | grep -E "error" || true
  rm -rf $O/pf
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/pf -o b -- python3 tools/prof_run.py fp32_tc_cor 4 > $O/pf.log 2>&1
  python3 tools/pmc_sq.py $O/pf.json $O/pf > /dev/null
  echo "== $f"
  python3 -c "
import json
d=json.load(open('$O/pf.json'))
for k,v in d['kernels'].items():
    if 'apply' in k: print(k, {x: round(y,1) for x,y in v.items() if not x.startswith('derived')})
"
  python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "[$f]"
done
