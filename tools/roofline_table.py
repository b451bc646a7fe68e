#!/usr/bin/env python3
"""profiles/rNN_roofline_table.md from the committed rocprofv3 kernel stats: per kernel, algorithmic bytes (or flops) per launch
 divided by the average launch duration of the trace, against 8 TB/s HBM and the dense matrix peaks."""
import csv, sys, os
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
M = 1 << 20
def rows(name):
    out = {}
    with open(os.path.join(R, name)) as f:
        lines = [l for l in f if not l.startswith("#")]
    for r in csv.DictReader(lines):
        out[r["kernel"]] = float(r["avg_us"])
    return out
def find(d, key):
    for k, v in d.items():
        if key in k: return v
    return None
T = []
def add(workload, kernel, us, byts=None, flops=None, peak_tf=None, note=""):
    if us is None: return
    gbs = byts / us / 1e3 if byts else None
    tf = flops / us / 1e6 if flops else None
    T.append((workload, kernel, us, byts, gbs, gbs / 8000.0 if gbs else None, tf, (tf / peak_tf) if (tf and peak_tf) else None, note))
b = rows("r02_rocprofv3_kernel_stats_bench.csv")
add("2^20 x 64 tc_cor", "apply_wg_kernel<1,4,false,64>", find(b, "apply_wg_kernel<1, 4, false, 64"), 8.0 * M * 64, 2.0 * M * 64 * 64, 2500.0, "read A + write Q; bf16 MFMA (6 products per tile executed)")
add("2^20 x 64 tc_cor", "gram_bf16_kernel<4>", find(b, "gram_bf16_kernel"), 4.0 * M * 64, 2.0 * M * 64 * 64 * 10 / 16, 2500.0, "read A; 10 of 16 tiles")
n = rows("r02_kstats_2p20x64_notc.csv")
add("2^20 x 64 notc", "apply_wg_kernel<0,4,false,128>", find(n, "apply_wg_kernel<0"), 8.0 * M * 64, 2.0 * M * 64 * 64, 157.3, "exact fp32 MFMA")
c = rows("r02_kstats_c3_tc_cor_2p20x128_one_panel.csv")
add("2^20 x 128 tc_cor", "gram_wide_kernel", find(c, "gram_wide_kernel"), 4.0 * M * 128, 2.0 * M * 128 * 128 * 36 / 64, 2500.0, "read A; 36 of 64 tiles")
add("2^20 x 128 tc_cor", "apply_wide_kernel<1>", find(c, "apply_wide_kernel<1>"), 8.0 * M * 128, 2.0 * M * 128 * 128, 2500.0, "")
cn = rows("r02_kstats_c3_notc_2p20x128_one_panel.csv")
add("2^20 x 128 notc", "apply_wide_f32_kernel", find(cn, "apply_wide_f32_kernel"), 8.0 * M * 128, 2.0 * M * 128 * 128 * 144 / 256, 157.3, "flops executed (triangular Z): 19.3 GF")
c4 = rows("r02_kstats_2p23x64_tc_cor.csv")
add("2^23 x 64 tc_cor", "apply_wg_kernel<1,4,false,64>", find(c4, "apply_wg_kernel"), 8.0 * 8 * M * 64, None, None, "beyond the Infinity Cache")
add("2^23 x 64 tc_cor", "gram_bf16_kernel<4>", find(c4, "gram_bf16_kernel"), 4.0 * 8 * M * 64, None, None, "beyond the Infinity Cache")
p = rows("r02_kstats_c3_tc_cor_2p20x128_panels_policy5.csv")
add("2^20 x 128, panel path", "cross_kernel", find(p, "cross_kernel"), 8.0 * M * 64, 2.0 * M * 64 * 64, 2500.0, "reads Qb and Ap")
add("2^20 x 128, panel path", "apply_wg_kernel<1,4,true,128>", find(p, "apply_wg_kernel<1, 4, true"), 12.0 * M * 64, 2.0 * M * 64 * 64, 2500.0, "reads Qb, Ap; writes Ap")
h = rows("r02_kstats_policy1_householder_tc_cor_after.csv")
add("2^20 x 64, Householder engine", "fold_kernel<4,false,true>", find(h, "fold_kernel"), 4.0 * M * 64, 2.0 * M * 64 * 64 - 2.0 / 3 * 64 ** 3, 157.3, "level 0; VALU panel + MFMA block reflectors (fraction of the fp32-matrix peak)")
g4 = rows("r02_kstats_policy4_fp64gram_2p20x64.csv") if os.path.exists(os.path.join(R, "r02_kstats_policy4_fp64gram_2p20x64.csv")) else {}
add("2^20 x 64, fp64 Gram level", "gram_kernel<4>", find(g4, "gram_kernel"), 4.0 * M * 64, 2.0 * M * 64 * 64 * 10 / 16, 78.6, "fp64 MFMA")
with open(os.path.join(R, "r02_roofline_table.md"), "w") as f:
    f.write("# Round-2 roofline table (tools/roofline_table.py from the committed rocprofv3 kernel stats)\n\n")
    f.write("HBM peak 8 TB/s; dense matrix peaks: bf16 2.5 PF/s, fp32 157.3 TF/s, fp64 78.6 TF/s. Bytes and flops are ALGORITHMIC per launch.\n\n")
    f.write("| workload | kernel | avg launch µs | algorithmic MB | GB/s | of 8 TB/s | TFLOP/s | of its matrix peak | note |\n|---|---|---|---|---|---|---|---|---|\n")
    for w, k, us, by, gbs, fr, tf, ft, note in T:
        f.write("| %s | `%s` | %.1f | %s | %s | %s | %s | %s | %s |\n" % (w, k, us, "%.0f" % (by / 1e6) if by else "", "%.0f" % gbs if gbs else "", "%.2f" % fr if fr else "",
                                                               "%.1f" % tf if tf else "", "%.3f" % ft if ft else "", note))
print(open(os.path.join(R, "r02_roofline_table.md")).read())
