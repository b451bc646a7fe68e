#!/bin/bash
# Round-3 evidence, all on ONE GPU box: kernel stats of the driver's bench command, per-call timelines, kernel stats of the other
# workloads, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, kernel trace only) and SQ counters of the Householder
# engine.  Writes gpurun_out/r03_*; copy what is to be judged into profiles/.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
step() { echo "== $*"; }

step "kernel stats of: python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
rm -rf $O/r03_kb
rocprofv3 --output-format csv --kernel-trace --stats -d $O/r03_kb -o b -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r03_bench_under_rocprof.json 2> $O/r03_kb.err || { tail -5 $O/r03_kb.err; exit 1; }
python3 tools/kstats.py $O/r03_kb/b_kernel_stats.csv $O/r03_rocprofv3_kernel_stats_bench.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline (2^20 x 64 fp32_tc_cor: 1 checked call + first window 5+20 + 20 under HIP events + window 5+20 + blocking-calls window 5+20 = 96 calls)"
rm -rf $O/r03_kb

step "per-call timelines (tools/loop_run.py, 400 calls; the loop entry's default schedule unless a depth is given: 1 = blocking calls, 2 = two in flight)"
bash tools/r03_timelines.sh

step "HBM traffic of the headline kernels (two --pmc passes)"
for cnt in FETCH_SIZE WRITE_SIZE; do
	rm -rf $O/r03_pmc_$cnt
	rocprofv3 --output-format csv --kernel-trace --pmc $cnt -d $O/r03_pmc_$cnt -o p -- python3 tools/prof_run.py fp32_tc_cor 3 > $O/r03_pmc_$cnt.log 2>&1 || { tail -3 $O/r03_pmc_$cnt.log; exit 1; }
done
python3 tools/pmc_traffic.py $O/r03_pmc_FETCH_SIZE $O/r03_pmc_WRITE_SIZE $O/r03_pmc_hbm_traffic.json 3
rm -rf $O/r03_pmc_FETCH_SIZE $O/r03_pmc_WRITE_SIZE

step "SQ counters of the Householder engine (policy 1, fp32_tc_cor), two passes"
rm -rf $O/r03_sq1 $O/r03_sq2
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/r03_sq1 -o s -- python3 tools/prof_run.py fp32_tc_cor 3 --policy 1 > $O/r03_sq1.log 2>&1 || { tail -3 $O/r03_sq1.log; exit 1; }
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU -d $O/r03_sq2 -o s -- python3 tools/prof_run.py fp32_tc_cor 3 --policy 1 > $O/r03_sq2.log 2>&1 || { tail -3 $O/r03_sq2.log; exit 1; }
python3 tools/pmc_sq.py $O/r03_pmc_sq_counters_policy1_householder.json $O/r03_sq1 $O/r03_sq2 > $O/r03_sq_summary.txt
rm -rf $O/r03_sq1 $O/r03_sq2
step done
