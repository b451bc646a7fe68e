#!/usr/bin/env python3
"""profiles/<round>_roofline_table.md (usage: roofline_table.py [r03|r04]) from the committed per-call timelines (profiles/<round>_timeline_*.txt, tools/timeline.py: medians per
kernel over the steady-state calls of a rocprofv3 kernel trace): per kernel the algorithmic bytes / flops of a launch (DESIGN.md
section 4) over its median duration, against 8 TB/s HBM and the 157.3 TF/s fp32 matrix peak the whole path is priced on."""
import os, re, sys
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r04"          # profiles/<ROUND>_timeline_*.txt -> profiles/<ROUND>_roofline_table.md
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
WORK = [  # (timeline file, label, m, n)
    (ROUND + "_timeline_c2.txt", "2^20 x 64 fp32_tc_cor (headline; stream of calls, chained schedule)", 1 << 20, 64),
    (ROUND + "_timeline_c2_two_in_flight.txt", "2^20 x 64 fp32_tc_cor (stream of calls, two in flight)", 1 << 20, 64),
    (ROUND + "_timeline_c2_blocking.txt", "2^20 x 64 fp32_tc_cor (blocking calls)", 1 << 20, 64),
    (ROUND + "_timeline_c2_notc.txt", "2^20 x 64 fp32_notc", 1 << 20, 64),
    (ROUND + "_timeline_c3.txt", "2^20 x 128 fp32_tc_cor (one panel; stream of calls, chained schedule)", 1 << 20, 128),
    (ROUND + "_timeline_c3_blocking.txt", "2^20 x 128 fp32_tc_cor (one panel; blocking calls)", 1 << 20, 128),
    (ROUND + "_timeline_c3_notc.txt", "2^20 x 128 fp32_notc (one panel)", 1 << 20, 128),
    (ROUND + "_timeline_reorth.txt", "2^20 x 64 fp32_tc_cor, reorth", 1 << 20, 64),
    (ROUND + "_timeline_policy1_tc_cor.txt", "2^20 x 64 fp32_tc_cor, Householder engine", 1 << 20, 64),
    (ROUND + "_timeline_policy1_notc.txt", "2^20 x 64 fp32_notc, Householder engine", 1 << 20, 64),
    (ROUND + "_timeline_2p23.txt", "2^23 x 64 fp32_tc_cor", 1 << 23, 64),
    (ROUND + "_timeline_c5.txt", "2^20 x 64 fp32_tc_cor, latms cond 1e8, reorth (C5)", 1 << 20, 64),
    (ROUND + "_timeline_c2_rot4_blocking.txt", "2^20 x 64 fp32_tc_cor, four rotating matrices (cache-cold across calls), blocking calls", 1 << 20, 64),
    (ROUND + "_timeline_c2_rot4_batch.txt", "2^20 x 64 fp32_tc_cor, four rotating matrices, batch entry (two in flight)", 1 << 20, 64),
]
def algorithmic(kernel, m, n):
    """(bytes, flops) of one launch, or (None, None) for the n^3-scale one-workgroup kernels"""
    if kernel.startswith(("gram_blk", "gram_bf16", "gram_wide", "gram_kernel", "fold_kernel")):      # (incl. the chained launches gram_*_chain_kernel)
        return 4.0 * m * n, 2.0 * m * n * n / 2 if not kernel.startswith("fold_kernel") else 2.0 * m * n * n
    if kernel.startswith(("apply_wg", "apply_wide")):
        return 8.0 * m * n, 2.0 * m * n * n / (2 if "f32" not in kernel and n == 128 else 1) if n == 128 else 2.0 * m * n * n
    return None, None
out = ["# Round-" + ROUND[1:].lstrip("0") + " roofline table: every kernel of every workload, from the committed per-call timelines (`profiles/" + ROUND + "_timeline_*.txt`,",
       "medians over the steady-state calls under `rocprofv3 --kernel-trace`; tools/timeline.py, tools/roofline_table.py).  Algorithmic bytes / flops",
       "per launch as in DESIGN.md section 4 (Gram: 4MN bytes, MN^2 flops; apply: 8MN bytes, 2MN^2 flops -- MN^2 for the triangular 128-column Z);",
       "peaks: HBM 8 TB/s, fp32 matrix 157.3 TF/s (MI355X_MICROARCH.md).", "",
       "| workload | kernel | median us | algorithmic MB | TB/s | frac of 8 TB/s | algorithmic GF | TF/s | share of the call |", "|---|---|---|---|---|---|---|---|---|"]
for fn, label, m, n in WORK:
    path = os.path.join(R, fn)
    if not os.path.exists(path):
        continue
    ks, period, ssum, gap = [], None, None, 0.0
    for line in open(path):
        mt = re.search(r"gap to the first kernel of the next call\s+([0-9.]+) us", line)
        if mt:
            gap = float(mt.group(1))
        mt = re.match(r"^(\S.*?)\s+([0-9.]+) us\s+gap to", line)
        if mt:
            ks.append((mt.group(1).strip(), float(mt.group(2))))
        mt = re.match(r"^sum of medians ([0-9.]+) us; median call period \(start to start\) ([0-9.]+) us", line)
        if mt:
            ssum, period = float(mt.group(1)), float(mt.group(2))
    for k, us in ks:
        by, fl = algorithmic(k, m, n)
        out.append("| %s | `%s` | %.1f | %s | %s | %s | %s | %s | %.0f %% |" % (label, k, us, "%.0f" % (by / 1e6) if by else "", "%.2f" % (by / us / 1e6) if by else "",
                   "%.2f" % (by / us / 1e6 / 8.0) if by else "", "%.1f" % (fl / 1e9) if fl else "", "%.1f" % (fl / us / 1e6) if fl else "", 100.0 * us / period))
    fqr = 4.0 * m * n * n - 4.0 / 3 * n ** 3
    out.append("| %s | **whole call** (period under the profiler, incl. %.1f us between calls) | %.1f | %.0f | %.2f | %.2f | %.1f | **%.1f** = %.0f %% of 157.3 | |" % (
        label, gap, period, 4.0 * (2 * m * n + n * n) / 1e6, 4.0 * (2 * m * n + n * n) / period / 1e6,
        4.0 * (2 * m * n + n * n) / period / 1e6 / 8.0, fqr / 1e9, fqr / period / 1e6, 100.0 * fqr / period / 1e6 / 157.3))
open(os.path.join(R, ROUND + "_roofline_table.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
