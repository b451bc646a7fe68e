#!/bin/bash
# round 4: packed rows in the 16-wave Cholesky -- tests (bitwise against the four-wave body through the chained schedules), stamps, bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_chol.py tests/test_gpu_async.py tests/test_gpu_wide.py -m gpu -x -q > gpurun_out/r04_step4_pytest.log 2>&1; rc=$?
tail -6 gpurun_out/r04_step4_pytest.log
[ $rc -ne 0 ] && exit $rc
python tools/chol_stamps.py 64 > gpurun_out/r04_step4_chol_stamps.txt 2>&1 || exit 1; cat gpurun_out/r04_step4_chol_stamps.txt
for w in "" "--workload c3" "--workload c5" "--reorth 1"; do
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rotate 0 $w > gpurun_out/r04_step4_bench.json 2> gpurun_out/r04_step4_bench.err || { tail -20 gpurun_out/r04_step4_bench.err; exit 1; }
python - "$w" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04_step4_bench.json").read().strip().splitlines()[-1])
print("[%s] value(blocking) %.4f ms  first %.4f  stream %.4f  %.1f TF/s orth %.2e res %.2e" % (sys.argv[1], d["ms_per_step"], d["first_window"]["ms_per_step"], d["stream_same_a"]["ms_per_step"], d["value"] / 1e3, d["orth_fro"], d["residual"]), {a: round(b * 1e3, 1) for a, b in d["roofline"]["kernel_ms_per_step"].items()})
PY
done
