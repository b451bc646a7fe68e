#!/bin/bash
# GPU box: the chained stream schedule -- kernel-level test, loop tests, headline bench twice, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_chol.py -x -q -k chained > gpurun_out/chain_pytest.log 2>&1; rc=$?
tail -12 gpurun_out/chain_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_async.py -x -q > gpurun_out/chain_pytest2.log 2>&1; rc=$?
tail -12 gpurun_out/chain_pytest2.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
	timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/chain_bench_$rep.json 2> gpurun_out/chain_bench_$rep.err || { tail -5 gpurun_out/chain_bench_$rep.err; exit 1; }
	python - gpurun_out/chain_bench_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4f ms  blocking %.4f ms  first_window %.4f  orth %.2e res %.2e" % (d["ms_per_step"], d["blocking_calls"]["ms_per_step"], d["first_window"]["ms_per_step"], d["orth_fro"], d["residual"]))
PY
done
tools/gpu_timeline.sh c2_chained -- 400 1048576 64 fp32_tc_cor 0 0 3 && cat gpurun_out/timeline_c2_chained.txt
