#!/bin/bash
# GPU box: the whole GPU test suite, then the bench lines of every workload (c2 headline, c3 both modes, c5, Householder engine, fp16 I/O) and the one-rank
# row-partitioned driver over raw RCCL.  Everything lands under gpurun_out/verify_*.  SKIP_TESTS=1: the bench lines only.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/verify_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/verify_pytest.log
[ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
run() { n=$1; shift; timeout -k 10 400 python bench.py "$@" > gpurun_out/verify_bench_$n.json 2> gpurun_out/verify_bench_$n.err || { echo "bench $n failed"; tail -5 gpurun_out/verify_bench_$n.err; exit 1; }
	python - gpurun_out/verify_bench_$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("%-8s %.4f ms (blocking calls %s, first window %s)  %.1f TF/s  orth %.2e res %.2e  dominant %s %.1f us frac %.2f  | %s" % (sys.argv[2], d["ms_per_step"],
      "%.4f" % d["blocking_calls"]["ms_per_step"] if "blocking_calls" in d else "-",
      "%.4f" % d["first_window"]["ms_per_step"] if "first_window" in d else "-", d["value"] / 1e3, d["orth_fro"], d["residual"],
      r.get("kernel"), r.get("avg_launch_us", 0), r.get("frac", 0), d["config"]["engine"]))
PY
}
run c2 --steps 20 --warmup 5
run c3 --workload c3 --steps 20 --warmup 5 --no-cpu-baseline
run c3notc --workload c3 --mode fp32_notc --steps 20 --warmup 5 --no-cpu-baseline
run c5 --workload c5 --steps 20 --warmup 5 --no-cpu-baseline
run hh --policy 1 --steps 20 --warmup 5 --no-cpu-baseline
run c2h --workload c2h --steps 20 --warmup 5
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-dist --dist-comm rccl --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/verify_bench_dist1.json 2> gpurun_out/verify_bench_dist1.err || { echo "dist1 failed"; tail -8 gpurun_out/verify_bench_dist1.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/verify_bench_dist1.json").read().strip().splitlines()[-1])
print("dist x1 over %s: %.4f ms  orth %.2e" % (d["config"]["dist_transport"], d["ms_per_step"], d["orth_fro"]))
PY
