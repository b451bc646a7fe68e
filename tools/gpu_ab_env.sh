#!/bin/bash
# usage: gpu_ab_env.sh VAR   -- headline bench line with VAR=1 and VAR=0, twice each (A/B of one of the TSQR_MI_* environment switches), then the parity tests
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 1 0; do
  export $1=$v
  python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "$1=$v"
done
done
unset $1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
