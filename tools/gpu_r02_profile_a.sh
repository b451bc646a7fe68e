#!/bin/bash
# round 2, first GPU pass: rocprofv3 kernel stats for the workloads round 1 left un-profiled (Householder engine, C3, C5)
set -e
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_policy1 -o p1 -- python3 tools/prof_run.py fp32_tc_cor 4 --policy 1 > $O/kt_policy1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_c3 -o c3 -- python3 tools/prof_run.py fp32_tc_cor 4 --n 128 > $O/kt_c3.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_c3n -o c3n -- python3 tools/prof_run.py fp32_notc 4 --n 128 > $O/kt_c3n.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_c5 -o c5 -- python3 tools/prof_run.py fp32_tc_cor 4 --reorth --cond 1e8 > $O/kt_c5.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt_notc -o notc -- python3 tools/prof_run.py fp32_notc 4 --policy 4 > $O/kt_notc.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/sq_p1_a -o a -- python3 tools/prof_run.py fp32_tc_cor 3 --policy 1 > $O/sq_p1_a.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_MISC -d $O/sq_p1_b -o b -- python3 tools/prof_run.py fp32_tc_cor 3 --policy 1 > $O/sq_p1_b.log 2>&1
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
ls -R $O | head -50
