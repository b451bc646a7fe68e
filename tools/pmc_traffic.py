#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) into
profiles/rNN_pmc_hbm_traffic.json: HBM bytes per launch and per kernel, corrected as MI355X_MICROARCH.md (section HBM)
prescribes for gfx950 (FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE counts 128-B requests at 64 B for wide streaming reads -> x2).
usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [round] [m] [n] [command line that was profiled]"""
import csv, glob, json, sys, collections

CLASS = {"apply_wg_kernel": "apply", "apply_wg_gramq_kernel": "apply", "gram_bf16_kernel": "gram", "gram_kernel": "gram", "gram_blk_kernel": "gram",
         "gram_blk_chain_kernel": "gram", "apply_wide_kernel": "apply_wide", "apply_wide_f32_kernel": "apply_wide", "gram_wide_kernel": "gram_wide",
         "gram_wide_chain_kernel": "gram_wide", "cross_kernel": "cross"}
M = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 20
N = int(sys.argv[6]) if len(sys.argv) > 6 else 64
NP = 64 if N <= 64 else (128 if N <= 128 else 64)         # columns a launch of the class streams (panels of 64 beyond 128 columns)
ALG = {"apply": 8 * M * min(N, 64), "gram": 4 * M * min(N, 64), "apply_wide": 8 * M * N, "gram_wide": 4 * M * N, "cross": 8 * M * 64}


def collect(d, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and "tsqrmi" in r["Kernel_Name"]:
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in per.items()}


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"round": int(sys.argv[4]) if len(sys.argv) > 4 else 1,
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- " + (sys.argv[7] if len(sys.argv) > 7 else "python3 tools/prof_run.py fp32_tc_cor 3"),
       "units": "FETCH_SIZE / WRITE_SIZE are KiB (x1024 -> bytes); gfx950 correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE x2 "
                "for 16-B-per-lane streaming reads, WRITE_SIZE exact; Infinity-Cache hits are counted (memory-side requests of the L2)",
       "workload": "%d x %d fp32_tc_cor, auto policy (bf16-split Gram), per launch" % (M, N), "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    short = k.split("(")[0].replace("void ", "")
    e = {"fetch_size_raw_kib": fetch.get(k, 0.0), "write_size_raw_kib": write.get(k, 0.0)}
    e["read_bytes_corrected"] = 2 * 1024 * e["fetch_size_raw_kib"]
    e["write_bytes"] = 1024 * e["write_size_raw_kib"]
    e["hbm_bytes"] = e["read_bytes_corrected"] + e["write_bytes"]
    for pat, cls in CLASS.items():
        if ("::" + pat + "<") in short + "<":
            e["class"] = cls
            e["algorithmic_bytes"] = ALG[cls]
            e["traffic_over_algorithmic"] = e["hbm_bytes"] / ALG[cls]
    out["kernels"][short] = e
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: (v.get("class"), round(v["hbm_bytes"] / 1e6, 1)) for k, v in out["kernels"].items()}))
