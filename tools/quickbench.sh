#!/bin/bash
# usage: tools/quickbench.sh "<EXTRA flags>" label   -- rebuilds the library with the flags and prints the bench summary
set -e
cd "$(dirname "$0")/.."
make -C tsqr_gpu_amd/csrc -B -s libtsqr_mi.so EXTRA="$1" 2>&1 | grep -E "error" || true
python bench.py --steps 8 --no-cpu-baseline $3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$2', 'GF %.0f ms %.3f orth %.2e res %.2e' % (d['value'], d['ms_per_step'], d['orth_fro'], d['residual']), {k: round(v,4) for k,v in d['roofline']['kernel_ms_per_step'].items()})"
