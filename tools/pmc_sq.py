#!/usr/bin/env python3
"""Merge rocprofv3 --pmc SQ_* passes (one directory per pass, --kernel-trace only) into profiles/rNN_pmc_sq_counters.json:
per-kernel averages per launch plus derived MFMA utilisation.  usage: pmc_sq.py <out.json> <dir> [<dir> ...]"""
import csv, glob, json, sys, collections

per = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tsqrmi" in r["Kernel_Name"]:
                per[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tsqrmi" in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
CLOCK_HZ, SIMDS = 2.4e9, 256 * 4
out = {"note": "rocprofv3 --pmc SQ_* passes on tools/prof_run.py (2^20 x 64 fp32_tc_cor), averages per launch; SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / "
               "SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_VALU_MFMA_BUSY_CYCLES are in quad-cycles summed over SEs/waves as the counter defines",
       "kernels": {}}
for k, cs in sorted(per.items()):
    e = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    if dur.get(k):
        e["avg_duration_us_under_pmc"] = sum(dur[k]) / len(dur[k]) / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:      # cycles summed over SIMDs / (duration x 2.4 GHz x 1024 SIMDs)
            e["derived_mfma_pipe_utilisation"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["avg_duration_us_under_pmc"] * 1e-6 * CLOCK_HZ * SIMDS)
    if e.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in e:
        e["derived_wait_any_over_wave_cycles"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
    out["kernels"][k] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: {c: v for c, v in e.items() if c.startswith("derived")} for k, e in out["kernels"].items()}, indent=1))
