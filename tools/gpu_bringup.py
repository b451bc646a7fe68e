#!/usr/bin/env python3
"""GPU bring-up: run the engine on many shapes/modes and print metrics (no asserts). Writes gpurun_out/bringup.log."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import ref_oracle as ro
from tsqr_gpu_amd import blockqr as bq

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", "bringup.log"), "w")
def P(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True); log.write(s + "\n"); log.flush()

def run(a, mode, reorth, lda_pad=0):
    m, n = a.shape
    lda = m + lda_pad
    buf = np.zeros((n, lda), np.float32); buf[:, :m] = a.T
    d_a = torch.from_numpy(buf).cuda()
    d_q = torch.full((n, lda), float('nan'), dtype=torch.float32, device="cuda")
    d_r = torch.zeros(n, n, dtype=torch.float32, device="cuda")
    bf = bq.buffer(mode, reorth); bf.allocate(m, n)
    torch.cuda.synchronize()
    t = time.time()
    st = bq.qr(d_q, lda, d_r, n, d_a, lda, m, n, bf)
    dt = time.time() - t
    q = d_q.cpu().numpy()[:, :m].T; r = d_r.cpu().numpy().T
    return st, q, r, dt

cases = [(64, 16), (128, 16), (100, 7), (33, 16), (20, 7), (64, 64), (200, 64), (4096, 64), (9211, 51), (4096, 32),
         (65536, 64), (1 << 20, 64), (4096, 128), (9000, 100), (1 << 17, 128)]
for (m, n) in cases:
    a = ro.uniform_matrix(m, n, seed=1)
    for mode in (bq.compute_mode.fp32_notc, bq.compute_mode.fp32_tc_cor):
        for reorth in (False, True):
            try:
                st, q, r, dt = run(a, mode, reorth, lda_pad=(3 if m == 9211 else 0))
                nanq = int(np.isnan(q).sum()); low = float(np.abs(np.tril(r, -1)).max()) if n > 1 else 0.0
                P("%8d x %3d %-12s reorth=%d st=%d  res %.3e  orthF %.3e  nanQ %d lowR %.1e  %.1f ms" % (
                    m, n, mode.name, reorth, st, ro.residual(a, q, r), ro.orthogonality_fro(q), nanq, low, dt * 1e3))
            except Exception as e:
                P("%8d x %3d %s reorth=%d EXC %s" % (m, n, mode.name, reorth, e)); traceback.print_exc()
# ill-conditioned
for cond in (1e4, 1e8):
    a = ro.matrix_with_cond(1 << 15, 64, cond, seed=1)
    for mode in (bq.compute_mode.fp32_notc, bq.compute_mode.fp32_tc_cor):
        for reorth in (False, True):
            st, q, r, dt = run(a, mode, reorth)
            P("cond %.0e %-12s reorth=%d  res %.3e  orthF %.3e" % (cond, mode.name, reorth, ro.residual(a, q, r), ro.orthogonality_fro(q)))
# timing at the headline size
m, n = 1 << 20, 64
a = ro.uniform_matrix(m, n, seed=0)
for mode in (bq.compute_mode.fp32_notc, bq.compute_mode.fp32_tc_cor):
    d_a0 = torch.from_numpy(np.ascontiguousarray(a.T)).cuda(); d_a = d_a0.clone()
    d_q = torch.empty(n, m, dtype=torch.float32, device="cuda"); d_r = torch.zeros(n, n, dtype=torch.float32, device="cuda")
    bf = bq.buffer(mode, False); bf.allocate(m, n)
    bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf)
    ts = []
    for i in range(10):
        d_a.copy_(d_a0); torch.cuda.synchronize(); t = time.time()
        bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf); ts.append(time.time() - t)
    P("timing 2^20x64 %s: min %.1f us median %.1f us  -> %.2f TFLOP/s (F_QR=4MN^2-4/3N^3)" % (
        mode.name, min(ts) * 1e6, sorted(ts)[5] * 1e6, (4 * m * n * n - 4 / 3 * n ** 3) / min(ts) / 1e12))
P("bringup done")
