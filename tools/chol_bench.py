"""chol_kernel timing at several n (back-to-back launches): the n x n step between the Gram pass and the apply pass."""
import ctypes, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_chol as tc
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
L.tsqr_selftest_chol.restype = ctypes.c_float
L.tsqr_selftest_chol.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_double, ctypes.c_int]
g, _ = tc.spd(64, 3.0, 1)
for n in (4, 16, 32, 48, 64):
    r = tc.run((L, torch), g[:n, :n], n, level=2, reps=50)
    print('n=%d: %.1f us  status %d' % (n, r[5] * 1e3, r[2]))
