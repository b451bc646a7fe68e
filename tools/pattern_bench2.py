"""Workgroup-cooperative copy skeleton (see selftest.hip copy_wg_kernel): block rows x output style x workgroup count."""
import ctypes, torch
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
L.tsqr_selftest_copy_wg.restype = ctypes.c_float
L.tsqr_selftest_copy_wg.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t] + [ctypes.c_int] * 4
m, n = 1 << 20, 64
for pad in (0, 1056):
    ld = m + pad
    a = torch.rand(n, ld, device='cuda'); q = torch.zeros(n, ld, device='cuda')
    for rows in (128, 256):
        for lin in (0, 1):
            res = []
            for nwg in (256, 512, 768, 1024, 2048):
                q.zero_()
                ms = L.tsqr_selftest_copy_wg(q.data_ptr(), a.data_ptr(), ld, m, rows, lin, nwg, 20)
                ok = torch.equal(a[:, :m], q[:, :m])
                res.append('%4d: %6.1f us %.2f TB/s%s' % (nwg, ms * 1e3, 8 * m * n / ms / 1e9, '' if ok else ' WRONG'))
            print('pad %5d rows %3d out_linear %d | ' % (pad, rows, lin) + ' | '.join(res), flush=True)
    del a, q
