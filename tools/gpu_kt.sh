#!/bin/bash
# usage: gpu_kt.sh <label> <prof_run args...>   -- rocprofv3 kernel trace + stats of tools/prof_run.py, filtered summary on stdout
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/kt_$1
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
L=$1; shift
rocprofv3 --output-format csv --kernel-trace --stats -d $O -o x -- python3 tools/prof_run.py "$@" > $O/run.log 2>&1
python3 tools/kstats.py $O/x_kernel_stats.csv $O/summary.csv "$L: prof_run.py $*"
find $O -name "*agent_info*" -delete; find $O -name "*domain_stats*" -delete
