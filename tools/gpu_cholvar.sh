#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in "" "-DTSQR_CHOL_NEWTON=1" "-DTSQR_CHOL_ZLDS" "-DTSQR_CHOL_ZLDS -DTSQR_CHOL_NEWTON=1"; do
  make -C tsqr_gpu_amd/csrc -B -s libtsqr_selftest.so EXTRA="$f" 2>&1 | grep -E " error" || true
  echo "== $f"; python tools/chol_bench.py 2>&1 | grep "n=64\|n=16"
done
