#!/bin/bash
# usage: gpu_ab_flags.sh "<flags A>" "<flags B>" [bench args]  -- rebuild with each flag set on the GPU box, headline bench line twice each
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in "$1" "$2"; do
  make -C tsqr_gpu_amd/csrc -B -s libtsqr_mi.so EXTRA="$f" 2>&1 | grep -E "error" || true
  python bench.py --steps 20 --no-cpu-baseline $3 2>/dev/null | python tools/bench_line.py "[$f]"
done
done
