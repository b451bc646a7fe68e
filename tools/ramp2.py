#!/usr/bin/env python3
"""ramp.py with different activities in front of the timed calls: what removes the slow phase of calls 10..30?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsqr_gpu_amd import blockqr as bq
m, n = 1 << 20, 64
a = torch.rand(n, m, device='cuda') * 2 - 1
q = torch.empty(n, m, device='cuda'); r = torch.zeros(n, n, device='cuda')
bf = bq.buffer(bq.compute_mode.fp32_tc_cor, False); bf.allocate(m, n)
call = bq.bind(q, m, r, n, a, m, m, n, bf)
call(); torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode == "gpu_fp64":                      # ~what bench.py's accuracy leg does
    for _ in range(3):
        q64 = q.double(); g = q64 @ q64.T; del q64
    torch.cuda.synchronize()
elif mode == "gpu_copy":
    b = torch.empty_like(a)
    for _ in range(100): b.copy_(a)
    torch.cuda.synchronize()
elif mode == "cpu_spin":
    t = time.perf_counter()
    while time.perf_counter() - t < 0.05: pass
elif mode == "sleep":
    time.sleep(0.2)
ts = []
for _ in range(120):
    t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e6)
def med(x): x = sorted(x); return x[len(x) // 2]
print("%-9s calls 1-9 %.1f | 10-30 %.1f | 31-60 %.1f | 61-120 %.1f   first 12: %s" % (mode, med(ts[:9]), med(ts[9:30]), med(ts[30:60]), med(ts[60:]), " ".join("%.0f" % t for t in ts[:12])))
