#!/bin/bash
# SQ counter passes for the one-panel path at 2^20 x 128 (two --pmc passes, kernel trace only) -> gpurun_out/sq_c3{a,b}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/sq_c3a $O/sq_c3b
P="rocprofv3 --output-format csv --kernel-trace"
$P --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/sq_c3a -o a -- python3 tools/prof_run.py fp32_tc_cor 3 --n 128 > $O/sq_c3a.log 2>&1
$P --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS -d $O/sq_c3b -o b -- python3 tools/prof_run.py fp32_tc_cor 3 --n 128 > $O/sq_c3b.log 2>&1
find $O/sq_c3a $O/sq_c3b -name "*agent_info*" -delete
python3 tools/pmc_sq.py $O/sq_c3.json $O/sq_c3a $O/sq_c3b | tail -3
