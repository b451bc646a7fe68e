#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "128 or 100 or 200 or 130 or c3" 2>&1 | tail -4
for mode in fp32_tc_cor fp32_notc; do
  python bench.py --steps 10 --no-cpu-baseline --n 128 --mode $mode 2>/dev/null | python tools/bench_line.py c3_$mode
done
