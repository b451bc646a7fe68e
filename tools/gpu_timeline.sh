#!/bin/bash
# usage: gpu_timeline.sh <label> [ENV=val ...] -- [loop_run args]   rocprofv3 kernel trace of tools/loop_run.py -> gpurun_out/timeline_<label>.txt
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
L=$1; shift
while [ $# -gt 0 ] && [ "$1" != "--" ]; do export "$1"; shift; done
[ "$1" = "--" ] && shift
O=$GRAFT_REPO_ROOT/gpurun_out/tl_$L
rm -rf $O; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace -d $O -o t -- python3 tools/loop_run.py "$@" > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
tail -1 $O/run.log
python3 tools/timeline.py $O 60 gpurun_out/timeline_$L.txt
head -3 $O/*/*kernel_trace.csv > gpurun_out/tl_head_$L.txt 2>&1; ls -R $O >> gpurun_out/tl_head_$L.txt; rm -rf $O
