import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import ref_oracle as ro
from tsqr_gpu_amd import blockqr as bq
a = ro.uniform_matrix(6000, 64, seed=9)
m, n = a.shape
d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda(); d_q = torch.empty(n, m, device='cuda'); d_r = torch.zeros(n, n, device='cuda')
bf = bq.buffer(bq.compute_mode.fp32_tc_cor, False); bf.allocate(m, n)
bq.set_policy(bq.POLICY_GRAM_BF16)
outs = []
for it in range(6):
    if it == 3: bf.dwr.fill_(float('nan')); bf.dwq.fill_(float('nan'))
    if it == 4: bf.dwr.fill_(1e30); bf.dwq.fill_(1e30)
    bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf)
    outs.append(d_r.cpu().numpy().copy())
    print(it, 'same as run0:', np.array_equal(outs[0], outs[-1]), 'finite', np.isfinite(outs[-1]).all(), 'maxdiff %.2e' % np.abs(outs[-1] - outs[0]).max())
bq.set_policy(0)
