#!/usr/bin/env python3
"""One fp16 I/O workload in a plain loop for profilers: f16_loop.py [mode] [calls] [m] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsqr_gpu_amd import blockqr as bq
mode = bq.compute_mode[sys.argv[1]] if len(sys.argv) > 1 else bq.compute_mode.fp16_tc_nocor
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
n = int(sys.argv[4]) if len(sys.argv) > 4 else 64
g = torch.Generator(device="cuda"); g.manual_seed(0)
a = (torch.rand(n, m, generator=g, device="cuda") * 2 - 1).half()
q = torch.empty(n, m, dtype=torch.float16, device="cuda"); r = torch.zeros(n, n, dtype=torch.float16, device="cuda")
bf = bq.buffer(mode, False); bf.allocate(m, n)
loop = bq.bind_loop(q, m, r, n, a, m, m, n, bf)                 # K blocking calls from one C loop (tsqr_mi_qr_f16_loop)
assert loop(60) == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
assert loop(calls) == 0
torch.cuda.synchronize()
print("f16_loop: %d calls of %d x %d %s: %.2f us per call, engine %d" % (calls, m, n, mode.name, (time.perf_counter() - t0) / calls * 1e6, bq.last_engine()))
