#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_parity.py -x -q > gpurun_out/bal_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/bal_pytest.log
[ $rc -ne 0 ] && exit $rc
tools/gpu_timeline.sh bal_d3 -- 400 1048576 64 fp32_tc_cor 0 0 3 > /dev/null; cat gpurun_out/timeline_bal_d3.txt
tools/gpu_timeline.sh bal_d1 -- 400 1048576 64 fp32_tc_cor 0 0 1 > /dev/null; cat gpurun_out/timeline_bal_d1.txt
for d in 1 3; do timeout -k 10 120 python tools/loop_run.py 1000 1048576 64 fp32_tc_cor 0 0 $d | sed "s/^/depth $d: /"; done
