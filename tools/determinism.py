import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import ref_oracle as ro
from tsqr_gpu_amd import blockqr as bq
def run(a, mode, pol):
    bq.set_policy(pol)
    m, n = a.shape
    d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda(); d_q = torch.empty(n, m, device='cuda'); d_r = torch.zeros(n, n, device='cuda')
    bf = bq.buffer(mode, False); bf.allocate(m, n)
    bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf)
    return d_q.cpu().numpy(), d_r.cpu().numpy(), bq.last_engine()
a = ro.uniform_matrix(6000, 64, seed=9)
for pol in (bq.POLICY_HOUSEHOLDER, bq.POLICY_GRAM_F64, bq.POLICY_GRAM_BF16):
    q1, r1, e1 = run(a, bq.compute_mode.fp32_tc_cor, pol)
    q2, r2, e2 = run(a, bq.compute_mode.fp32_tc_cor, pol)
    print('policy', pol, 'engine', e1, 'repeat bitwise R', np.array_equal(r1, r2), 'Q', np.array_equal(q1, q2))
    for s in (2.0 ** -20, 2.0 ** 12, 2.0 ** -2):
        q3, r3, e3 = run((a * s).astype(np.float32), bq.compute_mode.fp32_tc_cor, pol)
        d = np.abs(r3 - (r1 * s).astype(np.float32)).max() / np.abs(r1 * s).max()
        print('   scale 2^%d: R exact %s (rel diff %.1e)  Q exact %s' % (int(np.log2(s)), np.array_equal(r3, (r1 * s).astype(np.float32)), d, np.array_equal(q3, q1)))
bq.set_policy(0)
