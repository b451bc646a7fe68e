#!/bin/bash
# quick GPU check: parity tests + headline bench line (+ optional extra command)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py "$1"
