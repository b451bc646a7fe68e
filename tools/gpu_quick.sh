#!/bin/bash
# quick GPU check: parity tests + headline bench line; output also kept under gpurun_out/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/quick_pytest.log
python bench.py --steps 20 --no-cpu-baseline 2>gpurun_out/quick_bench.err | python tools/bench_line.py "$1"
