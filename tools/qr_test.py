#!/usr/bin/env python3
"""The reference's test driver (src/main.cu:13-121) on this engine: accuracy / speed / condition-number sweeps for the
modes gfx950 implements (fp16_notc, fp16_tc_nocor, fp32_notc, fp32_tc_nocor, fp32_tc_cor: the order of src/main.cu), with and without
re-orthogonalisation, printing the reference's CSV
schema so that its scripts/*/mk_*.py plotters read the output.  Defaults are reduced sweeps that finish in minutes;
--full selects the reference's own lists (m = 2^10..2^15, n = 2^10..m; cond sweep at 2^15 x 2^7, c = 2^2..2^15)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from tsqr_gpu_amd import blockqr as bq, harness  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--medium", action="store_true", help="the reference's shape family (n = 2^10 .. m) up to m = 2^13")
    ap.add_argument("--count", type=int, default=0, help="matrices per configuration (reference: 16)")
    ap.add_argument("--what", default="accuracy,speed,cond")
    args = ap.parse_args()
    what = args.what.split(",")
    modes = [bq.compute_mode.fp16_notc, bq.compute_mode.fp16_tc_nocor, bq.compute_mode.fp32_notc, bq.compute_mode.fp32_tc_nocor, bq.compute_mode.fp32_tc_cor]
    if args.full:
        C = args.count or 16
        sizes = [(1 << m, 1 << n, 1.0) for m in range(10, 16) for n in range(10, m + 1)]
        conds = [(1 << 15, 1 << 7, float(1 << c)) for c in range(2, 16)]
    elif args.medium:
        C = args.count or 2
        sizes = [(1 << m, 1 << n, 1.0) for m in range(10, 14) for n in range(10, m + 1)]
        conds = [(1 << 13, 1 << 7, float(1 << c)) for c in (2, 8, 14)]
    else:
        C = args.count or 4
        sizes = [(1 << 12, 1 << 10, 1.0), (1 << 14, 1 << 7, 1.0), (1 << 15, 1 << 10, 1.0), (1 << 20, 64, 1.0)]
        conds = [(1 << 15, 1 << 7, float(1 << c)) for c in (2, 6, 10, 14)]
    import torch
    print("# device: %s, torch %s, libtsqr_mi %s" % (torch.cuda.get_device_name(0), torch.__version__, bq.lib().tsqr_mi_version()))
    if "accuracy" in what:
        print("# accuracy test")
        for reorth in (False, True):
            for mode in modes:
                harness.accuracy(sizes, C, mode, reorth)
        for dt in (torch.float32, torch.float64):               # the place cusolver_accuracy<float/double> has in src/main.cu:35-36
            harness.rocsolver_accuracy(sizes, C, dt)
    if "speed" in what:
        for reorth in (False, True):
            for mode in modes:
                harness.speed(sizes, C, mode, reorth)
        for dt in (torch.float32, torch.float64):
            harness.rocsolver_speed(sizes, C, dt)
    if "cond" in what:
        print("# condition number test")
        for reorth in (False, True):
            for mode in modes:
                harness.accuracy_cond(conds, C, mode, reorth)


if __name__ == "__main__":
    main()
