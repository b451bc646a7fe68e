#!/bin/bash
# Round-4 evidence, all on ONE GPU box: kernel stats of the driver's bench command, per-call timelines of every schedule, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, kernel trace only) of the blocking, chained and rotating-buffer windows and of the
# shapes beyond the Infinity Cache, kernel stats + SQ counters of the panel-coupling kernels (2^20 x 192) and of C5.
# Writes gpurun_out/r04_*; copy what is to be judged into profiles/.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
step() { echo "== $*"; }

step "kernel stats of: python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
rm -rf $O/r04_kb
rocprofv3 --output-format csv --kernel-trace --stats -d $O/r04_kb -o b -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_under_rocprof.json 2> $O/r04_kb.err || { tail -5 $O/r04_kb.err; exit 1; }
python3 tools/kstats.py $O/r04_kb/b_kernel_stats.csv $O/r04_rocprofv3_kernel_stats_bench.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline (2^20 x 64 fp32_tc_cor; windows: 1 checked call, first_window, stream_same_a (+ its event leg), stream_rotating / two_in_flight_rotating / blocking_rotating over 4 triples, K blocking calls under events, value = W + K blocking calls)" > /dev/null
rm -rf $O/r04_kb

step "per-call timelines (tools/loop_run.py; depth 1 = blocking calls, 2 = two in flight, 3 = chained where it applies; rot4 = four different matrices through the batch entry)"
tools/gpu_timeline.sh r04_c2_blocking -- 400 1048576 64 fp32_tc_cor 0 0 1 || exit 1
tools/gpu_timeline.sh r04_c2 -- 400 || exit 1
tools/gpu_timeline.sh r04_c2_two_in_flight -- 400 1048576 64 fp32_tc_cor 0 0 2 || exit 1
tools/gpu_timeline.sh r04_c2_rot4_batch -- 400 1048576 64 fp32_tc_cor 0 0 3 rot4 || exit 1
tools/gpu_timeline.sh r04_c2_rot4_blocking -- 400 1048576 64 fp32_tc_cor 0 0 1 rot4 || exit 1
tools/gpu_timeline.sh r04_c2_rot4_chained TSQR_MI_CHAIN_MAX_MIB=100000 -- 400 1048576 64 fp32_tc_cor 0 0 3 rot4 || exit 1
tools/gpu_timeline.sh r04_2p18_rot4_chained -- 400 262144 64 fp32_tc_cor 0 0 3 rot4 || exit 1
tools/gpu_timeline.sh r04_2p18_rot4_two_in_flight -- 400 262144 64 fp32_tc_cor 0 0 2 rot4 || exit 1
tools/gpu_timeline.sh r04_c3 -- 200 1048576 128 fp32_tc_cor || exit 1
tools/gpu_timeline.sh r04_c3_blocking -- 200 1048576 128 fp32_tc_cor 0 0 1 || exit 1
tools/gpu_timeline.sh r04_reorth -- 200 1048576 64 fp32_tc_cor 1 || exit 1
tools/gpu_timeline.sh r04_2p23 -- 120 8388608 64 fp32_tc_cor || exit 1
tools/gpu_timeline.sh r04_policy1_tc_cor -- 100 1048576 64 fp32_tc_cor 0 1 || exit 1

step "kernel stats: C5 (latms cond 1e8, reorth), panel coupling at 2^20 x 192"
rm -rf $O/r04_k1; rocprofv3 --output-format csv --kernel-trace --stats -d $O/r04_k1 -o b -- python3 tools/prof_run.py fp32_tc_cor 8 --cond 1e8 --reorth > $O/r04_k1.log 2>&1 || { tail -5 $O/r04_k1.log; exit 1; }
python3 tools/kstats.py $O/r04_k1/b_kernel_stats.csv $O/r04_kstats_c5_cond1e8_reorth.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_tc_cor 8 --cond 1e8 --reorth (2^20 x 64; 8 calls at process start)" > /dev/null; rm -rf $O/r04_k1
rm -rf $O/r04_k2; rocprofv3 --output-format csv --kernel-trace --stats -d $O/r04_k2 -o b -- python3 tools/prof_run.py fp32_tc_cor 8 --n 192 > $O/r04_k2.log 2>&1 || { tail -5 $O/r04_k2.log; exit 1; }
python3 tools/kstats.py $O/r04_k2/b_kernel_stats.csv $O/r04_kstats_coupling_2p20x192.csv "rocprofv3 --kernel-trace --stats -- python3 tools/prof_run.py fp32_tc_cor 8 --n 192 (2^20 x 192: three 64-column panels, cross_kernel + apply<UPD> couple them)" > /dev/null; rm -rf $O/r04_k2

step "SQ counters of the coupling kernels (2^20 x 192), two passes"
rm -rf $O/r04_sq1 $O/r04_sq2
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/r04_sq1 -o s -- python3 tools/prof_run.py fp32_tc_cor 3 --n 192 > $O/r04_sq1.log 2>&1 || { tail -3 $O/r04_sq1.log; exit 1; }
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $O/r04_sq2 -o s -- python3 tools/prof_run.py fp32_tc_cor 3 --n 192 > $O/r04_sq2.log 2>&1 || { tail -3 $O/r04_sq2.log; exit 1; }
python3 tools/pmc_sq.py $O/r04_pmc_sq_counters_coupling_2p20x192.json $O/r04_sq1 $O/r04_sq2 > $O/r04_sq_summary.txt
rm -rf $O/r04_sq1 $O/r04_sq2

step "HBM traffic (two --pmc passes each): blocking calls, chained stream, four rotating matrices (two in flight), 2^20 x 128 chained, 2^23 x 64"
pmc() {  # label m n loop_run-args...
	local L=$1 M=$2 N=$3; shift 3
	for cnt in FETCH_SIZE WRITE_SIZE; do
		rm -rf $O/r04_pmc_$cnt
		rocprofv3 --output-format csv --kernel-trace --pmc $cnt -d $O/r04_pmc_$cnt -o p -- python3 tools/loop_run.py "$@" > $O/r04_pmc_$cnt.log 2>&1 || { tail -3 $O/r04_pmc_$cnt.log; exit 1; }
	done
	python3 tools/pmc_traffic.py $O/r04_pmc_FETCH_SIZE $O/r04_pmc_WRITE_SIZE $O/r04_pmc_hbm_traffic$L.json 4 $M $N "python3 tools/loop_run.py $*"
	rm -rf $O/r04_pmc_FETCH_SIZE $O/r04_pmc_WRITE_SIZE
}
pmc "" 1048576 64 40 1048576 64 fp32_tc_cor 0 0 1
pmc _chained 1048576 64 40 1048576 64 fp32_tc_cor 0 0 3
pmc _rot4 1048576 64 40 1048576 64 fp32_tc_cor 0 0 3 rot4
pmc _c3_chained 1048576 128 30 1048576 128 fp32_tc_cor 0 0 3
pmc _2p23 8388608 64 12 8388608 64 fp32_tc_cor 0 0 2
step done
