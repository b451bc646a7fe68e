#!/bin/bash
# GPU box, round 4: the whole GPU test suite, smoke(), then the bench lines of every workload (c2 headline, c3 both modes, c5, well-conditioned
# reorth, Householder engine, fp16 I/O, 2^23 x 64 = strong scaling at N = 1) and the one-rank row-partitioned driver over raw RCCL.
# Everything lands under gpurun_out/r04_verify_*.  SKIP_TESTS=1: the bench lines only.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04_verify_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r04_verify_pytest.log
[ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
: > gpurun_out/r04_verify_bench_all.jsonl
run() { n=$1; shift; timeout -k 10 500 python bench.py "$@" > gpurun_out/r04_verify_bench_$n.json 2> gpurun_out/r04_verify_bench_$n.err || { echo "bench $n failed"; tail -5 gpurun_out/r04_verify_bench_$n.err; exit 1; }
	tail -1 gpurun_out/r04_verify_bench_$n.json >> gpurun_out/r04_verify_bench_all.jsonl
	python - gpurun_out/r04_verify_bench_$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
g = lambda k: ("%.4f" % d[k]["ms_per_step"]) if k in d else "-"
print("%-8s value(blocking) %.4f ms  first %s  stream %s  rot: stream %s 2fl %s blocking %s  %.1f TF/s  orth %.2e res %.2e  %s %.1f us frac %.2f | %s" % (
      sys.argv[2], d["ms_per_step"], g("first_window"), g("stream_same_a"), g("stream_rotating"), g("two_in_flight_rotating"), g("blocking_rotating"),
      d["value"] / 1e3, d["orth_fro"], d["residual"], r.get("kernel"), r.get("avg_launch_us", 0), r.get("frac", 0), d["config"]["engine"]))
PY
}
run c2 --steps 20 --warmup 5
run c3 --workload c3 --steps 20 --warmup 5 --no-cpu-baseline
run c3notc --workload c3 --mode fp32_notc --steps 20 --warmup 5 --no-cpu-baseline
run c5 --workload c5 --steps 20 --warmup 5 --no-cpu-baseline
run reorth --reorth 1 --steps 20 --warmup 5 --no-cpu-baseline
run notc --mode fp32_notc --steps 20 --warmup 5 --no-cpu-baseline
run hh --policy 1 --steps 20 --warmup 5 --no-cpu-baseline --rotate 0
run c2h --workload c2h --steps 20 --warmup 5
run strong1 --scaling strong --steps 10 --warmup 3 --no-cpu-baseline --rotate 0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --dist-comm rccl --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_verify_bench_dist1.json 2> gpurun_out/r04_verify_bench_dist1.err || { echo "dist1 failed"; tail -8 gpurun_out/r04_verify_bench_dist1.err; exit 1; }
tail -1 gpurun_out/r04_verify_bench_dist1.json >> gpurun_out/r04_verify_bench_all.jsonl
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_verify_bench_dist1.json").read().strip().splitlines()[-1])
print("dist x1 over %s (%s): value(blocking) %.4f ms  stream %.4f  orth %.2e" % (d["config"]["dist_transport"], d["config"]["dist_exchange"][:14], d["ms_per_step"], d["stream_same_a"]["ms_per_step"], d["orth_fro"]))
PY
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-cpu-baseline --m 262144 > gpurun_out/r04_verify_bench_gloo2.json 2> gpurun_out/r04_verify_bench_gloo2.err || { echo "gloo2 failed"; tail -8 gpurun_out/r04_verify_bench_gloo2.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_verify_bench_gloo2.json").read().strip().splitlines()[-1])
print("2 ranks on one GPU over gloo callbacks (rehearsal of the multi-rank path, 2^18 rows per rank): value(blocking) %.4f ms  stream %.4f  orth %.2e  transport %s" % (d["ms_per_step"], d["stream_same_a"]["ms_per_step"], d["orth_fro"], d["config"]["dist_transport"]))
PY
