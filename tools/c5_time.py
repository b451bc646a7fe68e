"""Timing of BASELINE config C5 (latms cond 1e8, 2^20 x 64, Reorthogonalize = true) and which engines ran."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tsqr_gpu_amd import blockqr as bq, harness
m, n = 1 << 20, 64
for cond_exp in (4, 6, 8):
    s = torch.logspace(0, -cond_exp, n, dtype=torch.float64)
    a = harness.latms(m, n, n, s, seed=5)
    for mode in (bq.compute_mode.fp32_tc_cor, bq.compute_mode.fp32_notc):
        q = torch.empty(n, m, device='cuda'); r = torch.zeros(n, n, device='cuda')
        bf = bq.buffer(mode, True); bf.allocate(m, n)
        call = bq.bind(q, m, r, n, a, m, m, n, bf)
        call(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8): call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        bq.profile_enable(True)                        # per-kernel-class times in a separate pass (event bookkeeping perturbs the wall clock)
        for _ in range(8): call()
        torch.cuda.synchronize()
        prof = bq.profile_read(); bq.profile_enable(False)
        print('cond 1e%d %-12s %.3f ms  engine %s  orth %.2e  res %.2e  %s' % (cond_exp, mode.name, dt * 1e3, bq.ENGINE_NAMES[bq.last_engine()],
              harness.orthogonality_fro(q, m, n), harness.residual(q, r, a, m, n), {k: round(v[0] / 8, 3) for k, v in prof.items() if v[1]}), flush=True)
