#!/bin/bash
# reorthogonalised-call tests (parity, configs, dist, fuzz), then the c5 / reorth bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_fuzz.py tests/test_gpu_chol.py tests/test_gpu_async.py -m gpu -x -q > gpurun_out/r04_reorth_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r04_reorth_pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/r04_ab.sh "--workload c5" - || exit 1
bash tools/r04_ab.sh "--reorth 1" -
