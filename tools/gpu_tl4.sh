#!/bin/bash
# timelines of the stream of calls against blocking calls: headline and the 128-column workload
cd "$GRAFT_REPO_ROOT" || exit 1
tools/gpu_timeline.sh c2_stream -- 400 1048576 64 fp32_tc_cor 0 0 2 && cat gpurun_out/timeline_c2_stream.txt
tools/gpu_timeline.sh c2_blocking -- 400 1048576 64 fp32_tc_cor 0 0 1 && cat gpurun_out/timeline_c2_blocking.txt
tools/gpu_timeline.sh c3_stream -- 200 1048576 128 fp32_tc_cor 0 0 2 && cat gpurun_out/timeline_c3_stream.txt
tools/gpu_timeline.sh c3_blocking -- 200 1048576 128 fp32_tc_cor 0 0 1 && cat gpurun_out/timeline_c3_blocking.txt
