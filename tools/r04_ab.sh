#!/bin/bash
# round 4: A/B of environment switches on ONE box: tools/r04_ab.sh "<bench args>" "<env A>" "<env B>" ...  ("-" = no switch); REPS runs each, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
args="$1"; shift
for rep in $(seq 1 ${REPS:-2}); do
	i=0
	for v in "$@"; do
		[ "$v" = "-" ] && v=""
		env $v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rotate 0 --no-first-window $args > gpurun_out/ab_${i}_${rep}.json 2> gpurun_out/ab_${i}_${rep}.err || { echo "bench failed for [$v]"; tail -5 gpurun_out/ab_${i}_${rep}.err; exit 1; }
		python - "$v" gpurun_out/ab_${i}_${rep}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d["roofline"]["kernel_ms_per_step"]
print("[%s] value %.4f ms  stream %.4f  orth %.2e res %.2e  kernels(us): %s" % (sys.argv[1], d["ms_per_step"], d.get("stream_same_a", {}).get("ms_per_step", float("nan")),
      d["orth_fro"], d["residual"], {a: round(b * 1e3, 1) for a, b in k.items()}))
PY
		i=$((i + 1))
	done
done
