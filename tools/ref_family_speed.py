#!/usr/bin/env python3
"""The shape family of the reference's own standard test (src/main.cu:93-101: m = 2^10 .. 2^15, n = 2^10 .. m) through harness.speed /
harness.rocsolver_speed (the reference's protocol and CSV schema), fp32_tc_cor and fp32_notc without re-orthogonalisation, plus the accuracy
of the first call per shape.  usage: ref_family_speed.py [max log2 m, default 15]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsqr_gpu_amd import blockqr as bq, harness
top = int(sys.argv[1]) if len(sys.argv) > 1 else 15
sizes = [(1 << m, 1 << n, 1.0) for m in range(10, top + 1) for n in range(10, m + 1)]
print("# device: %s, libtsqr_mi %s" % (torch.cuda.get_device_name(0), bq.lib().tsqr_mi_version()))
for mode in (bq.compute_mode.fp32_tc_cor, bq.compute_mode.fp32_notc):
    harness.speed(sizes, 2, mode, False)
harness.rocsolver_speed(sizes, 2, torch.float32)
print("# accuracy test")
harness.accuracy(sizes, 1, bq.compute_mode.fp32_tc_cor, False)
