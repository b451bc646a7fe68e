#!/bin/bash
# round-2 evidence: rocprofv3 kernel stats of the bench command and of the other workloads, PMC passes, bench line, C++ caller
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02p
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
P="rocprofv3 --output-format csv --kernel-trace"
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
$P --stats -d $O/kt_bench -o b -- python3 bench.py --steps 20 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt_bench.log
$P --stats -d $O/kt_policy1 -o p1 -- python3 tools/prof_run.py fp32_tc_cor 4 --policy 1 > $O/kt_policy1.log 2>&1
$P --stats -d $O/kt_policy1n -o p1n -- python3 tools/prof_run.py fp32_notc 4 --policy 1 > $O/kt_policy1n.log 2>&1
$P --stats -d $O/kt_c3 -o c3 -- python3 tools/prof_run.py fp32_tc_cor 6 --n 128 > $O/kt_c3.log 2>&1
$P --stats -d $O/kt_c3n -o c3n -- python3 tools/prof_run.py fp32_notc 6 --n 128 > $O/kt_c3n.log 2>&1
$P --stats -d $O/kt_c3p -o c3p -- python3 tools/prof_run.py fp32_tc_cor 4 --n 128 --policy 5 > $O/kt_c3p.log 2>&1
$P --stats -d $O/kt_c3r -o c3r -- python3 tools/prof_run.py fp32_tc_cor 4 --n 128 --reorth > $O/kt_c3r.log 2>&1
$P --stats -d $O/kt_c4 -o c4 -- python3 tools/prof_run.py fp32_tc_cor 4 --m 8388608 > $O/kt_c4.log 2>&1
$P --stats -d $O/kt_notc -o nc -- python3 tools/prof_run.py fp32_notc 6 > $O/kt_notc.log 2>&1
echo "kt half"
$P --stats -d $O/kt_c5 -o c5 -- python3 tools/prof_run.py fp32_tc_cor 4 --reorth --cond 1e8 > $O/kt_c5.log 2>&1
$P --stats -d $O/kt_c5n -o c5n -- python3 tools/prof_run.py fp32_tc_cor 4 --cond 1e8 > $O/kt_c5n.log 2>&1
$P --stats -d $O/kt_reorth -o ro -- python3 tools/prof_run.py fp32_tc_cor 4 --reorth > $O/kt_reorth.log 2>&1
echo "kt done"
$P --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 tools/prof_run.py fp32_tc_cor 3 > $O/pmc_fetch.log 2>&1
$P --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 tools/prof_run.py fp32_tc_cor 3 > $O/pmc_write.log 2>&1
$P --pmc FETCH_SIZE -d $O/pmc_fetch_c3 -o f -- python3 tools/prof_run.py fp32_tc_cor 3 --n 128 > $O/pmc_fetch_c3.log 2>&1
$P --pmc WRITE_SIZE -d $O/pmc_write_c3 -o w -- python3 tools/prof_run.py fp32_tc_cor 3 --n 128 > $O/pmc_write_c3.log 2>&1
echo "pmc traffic done"
$P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/sq_a -o a -- python3 tools/prof_run.py fp32_tc_cor 3 > $O/sq_a.log 2>&1
$P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS -d $O/sq_b -o b -- python3 tools/prof_run.py fp32_tc_cor 3 > $O/sq_b.log 2>&1
$P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/sq_p1a -o a -- python3 tools/prof_run.py fp32_tc_cor 3 --policy 1 > $O/sq_p1a.log 2>&1
$P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS -d $O/sq_p1b -o b -- python3 tools/prof_run.py fp32_tc_cor 3 --policy 1 > $O/sq_p1b.log 2>&1
echo "sq done"
make -C tests/cpp -s 2>&1 | tail -2
(cd tests/cpp && ./speed_blockqr > $O/cpp_speed.csv 2> $O/cpp_speed.err; ./sample_blockqr > $O/cpp_sample.log 2>&1)
bash tools/gpu_dist1.sh > $O/dist_one_rank.txt 2>&1
python tools/wide_check.py --big > $O/wide_check.txt 2>&1
for w in "--m 1048576 --n 64" "--m 1048576 --n 64 --mode fp32_notc" "--m 1048576 --n 64 --reorth 1" "--m 1048576 --n 128" "--m 1048576 --n 128 --mode fp32_notc" "--m 8388608 --n 64"; do
  python bench.py --steps 20 --no-cpu-baseline $w 2>/dev/null | python tools/bench_line.py "$w" >> $O/bench_workloads.txt
done
find $O -name "*agent_info*" -delete; find $O -name "*domain_stats*" -delete
du -sh $O; ls $O
