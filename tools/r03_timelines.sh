#!/bin/bash
# the per-call timelines of tools/r03_collect.sh alone
cd "$GRAFT_REPO_ROOT" || exit 1
tools/gpu_timeline.sh r03_c2 -- 400
tools/gpu_timeline.sh r03_c2_two_in_flight -- 400 1048576 64 fp32_tc_cor 0 0 2
tools/gpu_timeline.sh r03_c2_blocking -- 400 1048576 64 fp32_tc_cor 0 0 1
tools/gpu_timeline.sh r03_c2_notc -- 400 1048576 64 fp32_notc
tools/gpu_timeline.sh r03_c3 -- 200 1048576 128 fp32_tc_cor
tools/gpu_timeline.sh r03_c3_blocking -- 200 1048576 128 fp32_tc_cor 0 0 1
tools/gpu_timeline.sh r03_c3_notc -- 200 1048576 128 fp32_notc
tools/gpu_timeline.sh r03_reorth -- 200 1048576 64 fp32_tc_cor 1
tools/gpu_timeline.sh r03_policy1_tc_cor -- 100 1048576 64 fp32_tc_cor 0 1
tools/gpu_timeline.sh r03_policy1_notc -- 100 1048576 64 fp32_notc 0 1
tools/gpu_timeline.sh r03_2p23 -- 120 8388608 64 fp32_tc_cor
