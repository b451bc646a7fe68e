#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_chol.py -x -q -s 2>&1 | tail -40
python tools/chol_bench.py 2>&1 | tail -8
