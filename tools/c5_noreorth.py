import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tsqr_gpu_amd import blockqr as bq, harness
m, n = 1 << 20, 64
s = torch.logspace(0, -8, n, dtype=torch.float64)
a = harness.latms(m, n, n, s, seed=5)
for mode in (bq.compute_mode.fp32_tc_cor,):
    q = torch.empty(n, m, device='cuda'); r = torch.zeros(n, n, device='cuda')
    bf = bq.buffer(mode, False); bf.allocate(m, n)
    call = bq.bind(q, m, r, n, a, m, m, n, bf)
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8): call()
    torch.cuda.synchronize()
    print('cond 1e8 no reorth: %.3f ms engine %s orth %.2e res %.2e' % ((time.perf_counter() - t0) / 8 * 1e3, bq.ENGINE_NAMES[bq.last_engine()], harness.orthogonality_fro(q, m, n), harness.residual(q, r, a, m, n)))
