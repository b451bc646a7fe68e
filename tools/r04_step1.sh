#!/bin/bash
# round 4, first GPU step: the new batch / alias / dist-vote tests, then the bench line with its new windows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/r04_step1_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r04_step1_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_step1_bench.json 2> gpurun_out/r04_step1_bench.err || { tail -20 gpurun_out/r04_step1_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_step1_bench.json").read().strip().splitlines()[-1])
def w(k):
    o = d.get(k)
    if not o: return "%s: -" % k
    r = o.get("roofline") or {}
    return "%s: %.4f ms  apply %.1f us frac %.2f kernels %s" % (k, o["ms_per_step"], r.get("avg_launch_us", 0), r.get("frac", 0), {a: round(b * 1e3, 1) for a, b in (r.get("kernel_ms_per_step") or {}).items()})
print("value(blocking) %.4f ms  %.1f TF/s orth %.2e res %.2e" % (d["ms_per_step"], d["value"] / 1e3, d["orth_fro"], d["residual"]))
print("roofline", d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"])
for k in ("first_window", "stream_same_a", "stream_rotating", "two_in_flight_rotating", "blocking_rotating"):
    print(w(k))
PY
