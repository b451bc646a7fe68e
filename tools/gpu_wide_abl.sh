#!/bin/bash
# ablations of gram_wide_kernel: rebuild with -DWIDE_ABL=k on the GPU box, kernel stats of 4 calls at 2^20 x 128
cd $GRAFT_REPO_ROOT
for k in $@; do
  make -C tsqr_gpu_amd/csrc -B -s libtsqr_mi.so EXTRA="-DWIDE_ABL=$k" 2>&1 | grep -E "error" || true
  bash tools/gpu_kt.sh abl$k fp32_tc_cor 4 --n 128 | grep -E "gram_wide|apply_wide"
done
