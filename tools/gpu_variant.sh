#!/bin/bash
# usage: gpu_variant.sh "<EXTRA flags>" label [bench args]   -- rebuild libtsqr_mi.so with the flags on the GPU box, print the bench summary
cd $GRAFT_REPO_ROOT
make -C tsqr_gpu_amd/csrc -B -s libtsqr_mi.so EXTRA="$1" 2>&1 | grep -E "error" || true
python bench.py --steps 20 --no-cpu-baseline $3 2>/dev/null | python tools/bench_line.py "$2"
