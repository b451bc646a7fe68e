#!/usr/bin/env python3
"""Speed of the fp16 I/O modes next to the fp32 modes they run on (harness.speed: the reference's protocol, src/test.cu:257-343)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tsqr_gpu_amd import blockqr as bq, harness
sizes = [(1 << 20, 64, 1.0), (1 << 16, 64, 1.0), (1 << 20, 128, 1.0)]
head = True
for mode in (bq.compute_mode.fp16_notc, bq.compute_mode.fp16_tc_nocor, bq.compute_mode.fp32_notc, bq.compute_mode.fp32_tc_nocor, bq.compute_mode.fp32_tc_cor):
    harness.speed(sizes, 32, mode, False, head=head)
    head = False
