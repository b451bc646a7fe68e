#!/usr/bin/env python3
"""Filter a rocprofv3 --stats kernel_stats.csv to the engine's own kernels (tsqrmi::*) and print / save a compact CSV.
usage: kstats.py <kernel_stats.csv> [out.csv] [note]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tsqrmi" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = [("kernel", "calls", "avg_us", "min_us", "max_us", "total_us", "pct_of_engine_kernels")]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].split("(")[0].replace("void ", "")
    out.append((name, r["Calls"], "%.2f" % (float(r["AverageNs"]) / 1e3), "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3),
                "%.1f" % (float(r["TotalDurationNs"]) / 1e3), "%.2f" % (100 * float(r["TotalDurationNs"]) / tot)))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        if len(sys.argv) > 3:
            f.write("# " + sys.argv[3] + "\n")
        csv.writer(f).writerows(out)
for o in out:
    print("%-60s %6s %10s %10s %10s %12s %8s" % o)
