#!/usr/bin/env python3
"""Small driver for rocprofv3: a few mtk::qr::qr calls at the headline size (2^20 x 64, fp32_tc_cor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsqr_gpu_amd import blockqr as bq
m, n = 1 << 20, 64
mode = bq.compute_mode[sys.argv[1]] if len(sys.argv) > 1 else bq.compute_mode.fp32_tc_cor
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
g = torch.Generator(device="cuda"); g.manual_seed(0)
d_a = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
d_q = torch.empty(n, m, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
bf = bq.buffer(mode, False); bf.allocate(m, n)
for _ in range(steps):
    assert bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf) == 0
torch.cuda.synchronize()
print("done")
