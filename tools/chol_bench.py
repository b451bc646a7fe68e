import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
L.tsqr_selftest_chol_time.restype = ctypes.c_float
NT = 4; ntri = 10
sub = torch.zeros(16, ntri * 256, dtype=torch.float64, device='cuda')
# G = 4*I + small: tile (ti,ti) diagonal entries; f64 layout row=(l>>4)+4*reg, col=l&15
g = np.zeros((ntri, 4, 64))
idx = 0
for ti in range(4):
    for tj in range(ti, 4):
        if ti == tj:
            for reg in range(4):
                for l in range(64):
                    row = (l >> 4) + 4 * reg; col = l & 15
                    g[idx, reg, l] = 4.0 if row == col else 0.01
        else:
            g[idx] = 0.01
        idx += 1
sub[0] = torch.from_numpy(g.reshape(-1)).cuda()
r = torch.zeros(64 * 64, device='cuda'); z = torch.zeros(64 * 64, device='cuda'); st = torch.zeros(4, dtype=torch.int32, device='cuda')
for n in (4, 16, 32, 48, 64):
    ms = L.tsqr_selftest_chol_time(ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(st.data_ptr()),
                                   ctypes.c_void_p(sub.data_ptr()), n, NT, 50)
    print('n=%d: %.1f us  status %s' % (n, ms * 1e3, st[:1].tolist()))
if hasattr(L, 'tsqr_selftest_chol_stamps'):
    buf = (ctypes.c_longlong * 8)()
    L.tsqr_selftest_chol_stamps(buf)
    t = list(buf)
    print('cycle stamps (100 MHz refclk or shader clock, see below):', [t[i] - t[0] for i in range(8)])
