#!/usr/bin/env python3
"""Headline workload in a plain loop for profilers: loop_run.py [calls] [m] [n] [mode] [reorth] [policy] [loop depth]
(default 400 calls of 2^20 x 64 fp32_tc_cor through the C-side loop tsqr_mi_qr_f32_loop, its default schedule; depth 1 = blocking calls, 2 = two in flight).
An eighth argument "c5" factors the latms cond 1e8 matrix of bench.py's c5 workload instead of U(-1,1); "rot<R>" rotates over R different
U(-1,1) matrices through the batch entry (tsqr_mi_qr_f32_batch)."""
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
if os.environ.get("TSQR_TUNE_APPLY_WAVES") or os.environ.get("TSQR_TUNE_GRAM_WAVES"):   # grid experiments (tsqr_mi_set_tuning2)
    bq.lib().tsqr_mi_set_tuning2(int(os.environ.get("TSQR_TUNE_GRAM_WAVES", "0")), int(os.environ.get("TSQR_TUNE_APPLY_WAVES", "0")))
kind = sys.argv[8] if len(sys.argv) > 8 else ""
g = torch.Generator(device="cuda"); g.manual_seed(0)
if kind == "c5":
    from tsqr_gpu_amd import harness
    a = harness.get_rand_matrix_with_cond_number(m, n, 1e8, seed=5, device="cuda")
else:
    a = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
keep = a.clone() if n > 64 else None
q = torch.empty(n, m, device="cuda"); r = torch.zeros(n, n, device="cuda")
bf = bq.buffer(mode, reorth); bf.allocate(m, n)
if kind.startswith("rot"):
    R = int(kind[3:])
    trip = [(a, q, r)] + [(torch.rand(n, m, generator=g, device="cuda") * 2 - 1, torch.empty(n, m, device="cuda"), torch.zeros(n, n, device="cuda")) for _ in range(R - 1)]
    binds = {}
    def loop(k):
        if k not in binds:
            seq = [trip[i % R] for i in range(k)]
            binds[k] = bq.bind_batch([t[1] for t in seq], m, [t[2] for t in seq], n, [t[0] for t in seq], m, m, n, bf)
        return binds[k]()[0]
else:
    loop = bq.bind_loop(q, m, r, n, a, m, m, n, bf)
assert loop(3) == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
assert loop(calls) == 0
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / calls
print("loop_run: %d calls of %d x %d %s reorth=%d: %.2f us per call, engine %d" % (calls, m, n, mode.name, reorth, dt * 1e6, bq.last_engine()))
