#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py tests/test_gpu_configs.py -m gpu -x -q -k "engines or householder or rccl or dist_driver" 2>&1 | tail -6
for mode in fp32_tc_cor fp32_notc; do
  python bench.py --steps 10 --no-cpu-baseline --policy 1 --mode $mode 2>/dev/null | python tools/bench_line.py hh_$mode
done
TSQR_MI_FOLD_COR=0 python bench.py --steps 10 --no-cpu-baseline --policy 1 2>/dev/null | python tools/bench_line.py hh_tc_cor_fp32refl
TSQR_MI_FOLD_TREE=0 python bench.py --steps 10 --no-cpu-baseline --policy 1 --mode fp32_notc 2>/dev/null | python tools/bench_line.py hh_notc_oldtree
