#!/bin/bash
# GPU box: submit / finish + row-partitioned loop tests, headline bench, one-rank RCCL bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_async.py tests/test_gpu_dist.py tests/test_gpu_configs.py -x -q > gpurun_out/async_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/async_pytest.log
[ $rc -ne 0 ] && exit $rc
show() { python - $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s: value %.4f ms  blocking %.4f ms  first_window %.4f  orth %.2e  %s" % (sys.argv[1], d["ms_per_step"], d["blocking_calls"]["ms_per_step"], d["first_window"]["ms_per_step"], d["orth_fro"], d["config"].get("dist_transport")))
PY
}
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/async_bench_1.json 2> gpurun_out/async_bench_1.err || { tail -5 gpurun_out/async_bench_1.err; exit 1; }
show gpurun_out/async_bench_1.json
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --dist-comm rccl --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/async_bench_dist1.json 2> gpurun_out/async_bench_dist1.err || { echo "dist1 failed"; tail -8 gpurun_out/async_bench_dist1.err; exit 1; }
show gpurun_out/async_bench_dist1.json
