#!/usr/bin/env python3
"""Headline workload in a plain loop for profilers: loop_run.py [calls] [m] [n] [mode] [reorth] [policy] [loop depth]
(default 400 calls of 2^20 x 64 fp32_tc_cor through the C-side loop tsqr_mi_qr_f32_loop, its default schedule; depth 1 = blocking calls, 2 = two in flight)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsqr_gpu_amd import blockqr as bq
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 400
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
mode = bq.compute_mode[sys.argv[4]] if len(sys.argv) > 4 else bq.compute_mode.fp32_tc_cor
reorth = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
bq.set_policy(int(sys.argv[6]) if len(sys.argv) > 6 else 0)
bq.set_loop_depth(int(sys.argv[7]) if len(sys.argv) > 7 else 3)
g = torch.Generator(device="cuda"); g.manual_seed(0)
a = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
keep = a.clone() if n > 64 else None
q = torch.empty(n, m, device="cuda"); r = torch.zeros(n, n, device="cuda")
bf = bq.buffer(mode, reorth); bf.allocate(m, n)
loop = bq.bind_loop(q, m, r, n, a, m, m, n, bf)
assert loop(3) == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
assert loop(calls) == 0
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / calls
print("loop_run: %d calls of %d x %d %s reorth=%d: %.2f us per call, engine %d" % (calls, m, n, mode.name, reorth, dt * 1e6, bq.last_engine()))
