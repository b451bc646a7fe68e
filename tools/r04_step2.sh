#!/bin/bash
# round 4: Cholesky owner section with the up-front diagonal-block broadcast -- tests, in-kernel stamps, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_chol.py tests/test_gpu_async.py tests/test_gpu_parity.py tests/test_gpu_wide.py -m gpu -x -q > gpurun_out/r04_step2_pytest.log 2>&1; rc=$?
tail -6 gpurun_out/r04_step2_pytest.log
[ $rc -ne 0 ] && exit $rc
python tools/chol_stamps.py 64 > gpurun_out/r04_step2_chol_stamps.txt 2>&1; cat gpurun_out/r04_step2_chol_stamps.txt
for w in "" "--workload c3" "--workload c5"; do
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $w > gpurun_out/r04_step2_bench.json 2> gpurun_out/r04_step2_bench.err || { tail -20 gpurun_out/r04_step2_bench.err; exit 1; }
python - "$w" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04_step2_bench.json").read().strip().splitlines()[-1])
def w(k):
    o = d.get(k)
    if not o: return "%s: -" % k
    r = o.get("roofline") or {}
    return "%s: %.4f ms  kernels %s" % (k, o["ms_per_step"], {a: round(b * 1e3, 1) for a, b in (r.get("kernel_ms_per_step") or {}).items()})
print("[%s] value(blocking) %.4f ms  %.1f TF/s orth %.2e res %.2e" % (sys.argv[1], d["ms_per_step"], d["value"] / 1e3, d["orth_fro"], d["residual"]), {a: round(b * 1e3, 1) for a, b in d["roofline"]["kernel_ms_per_step"].items()})
for k in ("first_window", "stream_same_a", "stream_rotating", "two_in_flight_rotating", "blocking_rotating"):
    print("   ", w(k))
PY
done
