#!/bin/bash
# usage: gpu_wide_var.sh "<flags>" ...   -- rebuild with each flag set, kernel stats of 4 calls at 2^20 x 128
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  make -C tsqr_gpu_amd/csrc -B -s libtsqr_mi.so EXTRA="$f" 2>&1 | grep -E "error" || true
  echo "== $f"
  bash tools/gpu_kt.sh var fp32_tc_cor 4 --n 128 | grep -E "gram_wide|apply_wide|schur|zwide"
done
