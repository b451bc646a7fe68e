#!/bin/bash
# usage: gpu_kt_any.sh <label> <python script> [args]  -- rocprofv3 kernel trace + stats of any python driver, engine kernels only
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/kt_$1
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
L=$1; shift
rocprofv3 --output-format csv --kernel-trace --stats -d $O -o x -- python3 "$@" > $O/run.log 2>&1
python3 tools/kstats.py $O/x_kernel_stats.csv $O/summary.csv "$L: $*"
find $O -name "*agent_info*" -delete; find $O -name "*domain_stats*" -delete; find $O -name "*kernel_trace*" -delete
