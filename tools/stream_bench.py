"""One-directional limits of the workgroup streaming skeleton (512 B contiguous per column and instruction, interleaved blocks):
copy with nontemporal stores, load only, store only -- at the headline size (A may live in the Infinity Cache) and at 8x that."""
import ctypes, torch
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
L.tsqr_selftest_stream_wg.restype = ctypes.c_float
L.tsqr_selftest_stream_wg.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t] + [ctypes.c_int] * 3
n = 64
for lm in (20, 23):
    m = 1 << lm
    a = torch.rand(n, m, device='cuda'); q = torch.zeros(n, m, device='cuda')
    for mode, name, nbytes in ((0, 'copy (nt stores)', 8), (1, 'load only', 4), (2, 'store only (nt)', 4)):
        res = []
        for nwg in (512, 1024, 2048):
            ms = L.tsqr_selftest_stream_wg(q.data_ptr(), a.data_ptr(), m, m, mode, nwg, 10)
            res.append('%4d WGs %7.1f us %.2f TB/s' % (nwg, ms * 1e3, nbytes * m * n / ms / 1e9))
        print('2^%d rows %-18s | ' % (lm, name) + ' | '.join(res), flush=True)
    del a, q
