#!/bin/bash
# GPU box: the submit / finish tests, then the headline bench twice (value = stream of calls; blocking_calls beside it)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_async.py -x -q > gpurun_out/async_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/async_pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
	timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/async_bench_$rep.json 2> gpurun_out/async_bench_$rep.err || { tail -5 gpurun_out/async_bench_$rep.err; exit 1; }
	python - gpurun_out/async_bench_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4f ms  blocking %.4f ms  first_window %.4f  orth %.2e" % (d["ms_per_step"], d["blocking_calls"]["ms_per_step"], d["first_window"]["ms_per_step"], d["orth_fro"]))
PY
done
timeout -k 10 300 python bench.py --workload c3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/async_bench_c3.json 2> gpurun_out/async_bench_c3.err || { tail -5 gpurun_out/async_bench_c3.err; exit 1; }
python - gpurun_out/async_bench_c3.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c3 value %.4f ms  blocking %.4f ms" % (d["ms_per_step"], d["blocking_calls"]["ms_per_step"]))
PY
tests/cpp/speed_blockqr 1048576 64 64
