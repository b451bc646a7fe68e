#!/bin/bash
# GPU box: headline bench A/B over environment switches (each run = a fresh process), then the GPU test suite.
# usage: tools/gpu_ab.sh "<env assignments A>" "<env assignments B>" ...   ("-" = no switch); output -> gpurun_out/ab_*.json
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
i=0
for v in "$@"; do
	[ "$v" = "-" ] && v=""
	for rep in $(seq 1 ${REPS:-2}); do
		env $v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_${i}_${rep}.json 2> gpurun_out/ab_${i}_${rep}.err || { echo "bench failed for [$v]"; tail -5 gpurun_out/ab_${i}_${rep}.err; exit 1; }
		python - "$v" gpurun_out/ab_${i}_${rep}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d["roofline"]["kernel_ms_per_step"]
print("[%s] value %.4f ms  first_window %.4f ms  orth %.2e  kernels(us): %s" % (sys.argv[1], d["ms_per_step"], d.get("first_window", {}).get("ms_per_step", float("nan")),
      d["orth_fro"], {a: round(b * 1e3, 1) for a, b in k.items()}))
PY
	done
	i=$((i + 1))
done
