import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import ref_oracle as ro
from tsqr_gpu_amd import blockqr as bq
a = ro.uniform_matrix(6000, 64, seed=9)
m, n = a.shape
d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda(); d_q = torch.empty(n, m, device='cuda'); d_r = torch.zeros(n, n, device='cuda')
bf = bq.buffer(bq.compute_mode.fp32_tc_cor, False); bf.allocate(m, n)
bq.set_policy(bq.POLICY_GRAM_BF16)
parts = []
nblocks = 24
for it in range(4):
    bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf)
    parts.append(bf.dwr[: nblocks * 2560 * 2].cpu().numpy().view(np.float64).reshape(nblocks, 10, 4, 64).copy())
    if it:
        d = parts[it] != parts[0]
        idx = np.argwhere(d)
        print(it, 'differing partial entries:', d.sum(), 'blocks', sorted(set(idx[:, 0].tolist()))[:10], 'tiles', sorted(set(idx[:, 1].tolist())), 'regs', sorted(set(idx[:, 2].tolist())), 'lanes', sorted(set(idx[:, 3].tolist()))[:20])
        if d.sum():
            i = idx[0]; print('   e.g.', parts[0][tuple(i)], parts[it][tuple(i)])
bq.set_policy(0)
