import sys, ctypes, torch
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
L.tsqr_selftest_read_cold.restype = ctypes.c_float
L.tsqr_selftest_read_cold.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
m, n = 1 << 20, 64
a = torch.rand(n, m, device='cuda'); q = torch.empty(n, m, device='cuda')
big = torch.empty(256 << 20, device='cuda')
for mode in (0, 1, 2):
    for waves in (1024, 2048, 4096):
        cold = L.tsqr_selftest_read_cold(q.data_ptr(), a.data_ptr(), m, mode, waves, big.data_ptr(), big.numel(), 6)
        warm = L.tsqr_selftest_read_cold(q.data_ptr(), a.data_ptr(), m, mode, waves, None, 0, 6)
        print('mode %d waves %5d: cold %.1f us (%.2f TB/s)   warm %.1f us (%.2f TB/s)' % (mode, waves, cold * 1e3, 4 * m * n / cold / 1e9, warm * 1e3, 4 * m * n / warm / 1e9))
