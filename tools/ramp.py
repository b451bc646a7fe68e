#!/usr/bin/env python3
"""Per-call wall clock of the first calls of a fresh process at the headline size (clock / cache ramp-up)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tsqr_gpu_amd import blockqr as bq
m, n = 1 << 20, 64
a = torch.rand(n, m, device='cuda') * 2 - 1
q = torch.empty(n, m, device='cuda'); r = torch.zeros(n, n, device='cuda')
bf = bq.buffer(bq.compute_mode.fp32_tc_cor, False); bf.allocate(m, n)
call = bq.bind(q, m, r, n, a, m, m, n, bf)
torch.cuda.synchronize()
ts = []
for _ in range(400):
    t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e6)
print("first 30:", " ".join("%.0f" % t for t in ts[:30]))
for lo in (30, 50, 100, 200, 300):
    seg = sorted(ts[lo:lo + 50 if lo < 300 else 400]); print("calls %d..: median %.1f" % (lo, seg[len(seg) // 2]))
