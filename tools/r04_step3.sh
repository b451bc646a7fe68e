#!/bin/bash
# round 4: shifted CholeskyQR3 on the bf16-split Gram matrix for reorthogonalised calls -- tests, then c5 / reorth bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_fuzz.py tests/test_gpu_harness.py tests/test_gpu_chol.py -m gpu -x -q > gpurun_out/r04_step3_pytest.log 2>&1; rc=$?
tail -25 gpurun_out/r04_step3_pytest.log
[ $rc -ne 0 ] && exit $rc
for w in "--workload c5" "--reorth 1"; do
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rotate 0 $w > gpurun_out/r04_step3_bench.json 2> gpurun_out/r04_step3_bench.err || { tail -20 gpurun_out/r04_step3_bench.err; exit 1; }
python - "$w" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r04_step3_bench.json").read().strip().splitlines()[-1])
print("[%s] value(blocking) %.4f ms  %.1f TF/s orth %.2e res %.2e engine %s" % (sys.argv[1], d["ms_per_step"], d["value"] / 1e3, d["orth_fro"], d["residual"], d["config"]["engine"]), {a: round(b * 1e3, 1) for a, b in d["roofline"]["kernel_ms_per_step"].items()})
print("    first_window %.4f" % d["first_window"]["ms_per_step"])
PY
done
exit $rc
