#!/usr/bin/env python3
"""Per-call timeline from a rocprofv3 --kernel-trace CSV of tools/loop_run.py: median duration of every kernel of a call and the median
gaps between them (end of one kernel -> start of the next, including the gap from the last kernel of call i to the first kernel of
call i+1).  usage: timeline.py <dir with *kernel_trace.csv> [skip_calls] [out.txt] [flags]"""
import csv, glob, statistics as st, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tsqrmi" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tsqrmi::", "")))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 60
# a call ends with host_flag_kernel -- or, inside tsqr_mi_qr_f32_loop's stream of calls (two in flight, the completion word of call i
# raised by the first kernel of call i + 1), where a Gram / fold kernel follows an apply kernel.  argv[4] = "flags": flag kernels only.
flags_only = len(sys.argv) > 4 and sys.argv[4] == "flags"
calls, cur = [], []
for r in rows:
    if cur and not flags_only and cur[-1][2].startswith("apply_") and r[2].startswith(("gram_blk", "gram_bf16", "gram_wide", "gram_kernel", "gram_h_", "fold_kernel")):
        calls.append(cur); cur = []
    cur.append(r)
    if r[2].startswith("host_flag"):
        calls.append(cur); cur = []
calls = calls[skip:]
shape = max(set(tuple(k[2] for k in c) for c in calls), key=lambda s: sum(1 for c in calls if tuple(k[2] for k in c) == s))
calls = [c for c in calls if tuple(k[2] for k in c) == shape]
out = ["calls analysed: %d (after skipping %d); kernels per call: %d" % (len(calls), skip, len(shape))]
tot = 0.0
for i, name in enumerate(shape):
    d = st.median((c[i][1] - c[i][0]) / 1e3 for c in calls)
    if i + 1 < len(shape):
        g = st.median((c[i + 1][0] - c[i][1]) / 1e3 for c in calls)
        gl = "gap to next kernel"
    else:
        g = st.median((calls[j + 1][0][0] - calls[j][i][1]) / 1e3 for j in range(len(calls) - 1))
        gl = "gap to the first kernel of the next call"
    tot += d + g
    out.append("%-52s %8.2f us   %s %6.2f us" % (name[:52], d, gl, g))
periods = [(calls[j + 1][0][0] - calls[j][0][0]) / 1e3 for j in range(len(calls) - 1)]
per = st.median(periods)
out.append("sum of medians %.2f us; median call period (start to start) %.2f us" % (tot, per))
# outliers: calls whose period is more than twice the median -- where the time went (longest kernel, longest gap inside the call,
# gap to the next call's first kernel)
slow = [j for j, p in enumerate(periods) if p > 2 * per]
out.append("calls with a period above twice the median: %d of %d" % (len(slow), len(periods)) + ("" if not slow else " -- " + "; ".join(
    "call %d: %.0f us (longest kernel %s %.0f us, largest gap inside the call %.0f us, gap to the next call %.0f us)" % (
        skip + j, periods[j], *max(((c[2], (c[1] - c[0]) / 1e3) for c in calls[j]), key=lambda t: t[1]),
        max([(calls[j][i + 1][0] - calls[j][i][1]) / 1e3 for i in range(len(calls[j]) - 1)] or [0.0]),
        (calls[j + 1][0][0] - calls[j][-1][1]) / 1e3) for j in slow[:6])))
print("\n".join(out))
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(out) + "\n")
