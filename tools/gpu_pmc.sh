#!/bin/bash
# usage: gpu_pmc.sh <outdir-name> [prof_run args...]  -- two SQ counter passes + kernel trace on tools/prof_run.py
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $O; cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/a -o a -- python3 tools/prof_run.py "$@" > $O/a.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS -d $O/b -o b -- python3 tools/prof_run.py "$@" > $O/b.log 2>&1
python3 tools/pmc_sq.py $O/sq.json $O/a $O/b > /dev/null
python3 - $O/sq.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,e in d['kernels'].items():
    if 'gram' in k or 'apply' in k or 'chol' in k:
        print(k)
        for c,v in e.items(): print('   %-36s %.4g'%(c,v))
PY
