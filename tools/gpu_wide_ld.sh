#!/bin/bash
cd $GRAFT_REPO_ROOT
make -C tsqr_gpu_amd/csrc -B -s libtsqr_mi.so EXTRA="-DWIDE_ABL=$1" 2>&1 | grep -E "error" || true
bash tools/gpu_kt.sh ld0 fp32_tc_cor 4 --n 128 | grep -E "gram_wide"
bash tools/gpu_kt.sh ld1 fp32_tc_cor 4 --n 128 --lda 1049600 | grep -E "gram_wide"
bash tools/gpu_kt.sh ld2 fp32_tc_cor 4 --n 128 --lda 1048640 | grep -E "gram_wide"
bash tools/gpu_kt.sh ld3 fp32_tc_cor 4 --n 96 | grep -E "gram_wide"
bash tools/gpu_kt.sh ld4 fp32_tc_cor 4 --n 128 --m 524288 | grep -E "gram_wide"
