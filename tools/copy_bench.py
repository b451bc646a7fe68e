import sys, ctypes, torch
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
L.tsqr_selftest_copy_time.restype = ctypes.c_float
L.tsqr_selftest_copy_time.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int]
m, n = 1 << 20, 64
a = torch.rand(n, m, device='cuda'); q = torch.empty(n, m, device='cuda')
for mode in (0, 1):
    for waves in (1024, 2048, 4096, 8192, 16384):
        ms = L.tsqr_selftest_copy_time(q.data_ptr(), a.data_ptr(), m, mode, waves, 20)
        print('mode %d waves %5d: %.1f us  %.2f TB/s' % (mode, waves, ms * 1e3, 2 * 4 * m * n / ms / 1e9))
    assert torch.equal(a, q)
L.tsqr_selftest_read_time.restype = ctypes.c_float
L.tsqr_selftest_read_time.argtypes = L.tsqr_selftest_copy_time.argtypes
for mode in (0, 1):
    for waves in (1024, 2048, 3072, 4096, 8192, 16384):
        ms = L.tsqr_selftest_read_time(q.data_ptr(), a.data_ptr(), m, mode, waves, 20)
        print('READ mode %d waves %5d: %.1f us  %.2f TB/s' % (mode, waves, ms * 1e3, 4 * m * n / ms / 1e9))
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
q.copy_(a); t0.record()
for _ in range(20): q.copy_(a)
t1.record(); torch.cuda.synchronize()
print('torch copy_: %.1f us' % (t0.elapsed_time(t1) / 20 * 1e3))

# cold-cache read: evict the Infinity Cache (256 MiB) with a 1 GiB fill before every timed launch
big = torch.empty(256 << 20, device='cuda')
def cold(fn, reps=8):
    tot = 0.0
    for _ in range(reps):
        big.fill_(1.0); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3
for mode in (0, 1):
    for waves in (2048, 4096, 8192):
        us = cold(lambda: L.tsqr_selftest_read_time(q.data_ptr(), a.data_ptr(), m, mode, waves, 1))
        print('COLD READ mode %d waves %5d: %.1f us (2 launches incl. warm-up) ' % (mode, waves, us))
us = cold(lambda: q.copy_(a)); print('COLD torch copy: %.1f us' % us)
us = cold(lambda: a.sum()); print('COLD torch sum (read-only): %.1f us' % us)
