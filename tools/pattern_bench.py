"""Access-pattern study: copy a 2^20 x 64 column-major matrix with different wave/workgroup tile shapes, chunk assignments
and leading dimensions (power-of-two column stride vs padded)."""
import ctypes, torch
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
L.tsqr_selftest_copy_pat.restype = ctypes.c_float
L.tsqr_selftest_copy_pat.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t] + [ctypes.c_int] * 4
m, n = 1 << 20, 64
for pad in (0, 1056):
    ld = m + pad
    a = torch.rand(n, ld, device='cuda'); q = torch.zeros(n, ld, device='cuda')
    for pat in (0, 2, 3):
        for inter in (0, 1):
            res = []
            for waves in (2048, 4096, 8192):
                ms = L.tsqr_selftest_copy_pat(q.data_ptr(), a.data_ptr(), ld, m, pat, inter, waves, 20)
                res.append('%5d: %6.1f us %.2f TB/s' % (waves, ms * 1e3, 8 * m * n / ms / 1e9))
            assert torch.equal(a[:, :m], q[:, :m])
            print('pad %5d pat %d inter %d | ' % (pad, pat, inter) + ' | '.join(res), flush=True)
    del a, q
