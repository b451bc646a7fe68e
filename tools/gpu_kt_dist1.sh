#!/bin/bash
# kernel stats of the one-rank row-partitioned bench over raw RCCL (the rank's environment is exported here: rocprofv3 needs the program
# itself behind `--`, no launcher in between)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
rm -rf $O/kt_dist1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_dist1 -o d -- python3 bench.py --gpus 1 --force-dist --dist-comm rccl --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_dist1.json 2> $O/kt_dist1.err || { tail -5 $O/kt_dist1.err; exit 1; }
python3 tools/kstats.py $O/kt_dist1/d_kernel_stats.csv $O/r03_kstats_dist1_rccl_2p20x64.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --force-dist --dist-comm rccl --steps 20 --warmup 5 --no-cpu-baseline (RANK=0 WORLD_SIZE=1 exported): the row-partitioned call on one rank over raw RCCL, 2^20 x 64 fp32_tc_cor; value window = chained stream (Cholesky of call i inside the Gram launch of call i + 1)"
cat $O/r03_kstats_dist1_rccl_2p20x64.csv
rm -rf $O/kt_dist1
