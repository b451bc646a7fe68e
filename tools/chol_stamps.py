"""Cycle stamps inside chol_wg (build libtsqr_selftest.so with EXTRA=-DTSQR_CHOL_DBG): per group and wave, s_memtime at
0 loop top, 1 before the barrier, 2 after the barrier, 3 after the G updates, 4 after the M updates."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_gpu_chol as tc
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
L.tsqr_selftest_chol_mfma.restype = ctypes.c_float
L.tsqr_selftest_chol_mfma.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_double, ctypes.c_double, ctypes.c_int]
g, _ = tc.spd(64, 3.0, 1)
tc.run((L, torch), g, 64, 1, level=2, reps=3)
buf = (ctypes.c_longlong * (4 * 16 * 8))()
L.tsqr_selftest_chol_stamps(buf)
s = np.array(buf).reshape(4, 16, 8)
t0 = s[0, 0, 0]
for w in (0, 1, 2, 3):
    print('wave', w)
    for g_ in range(16):
        print('  g=%2d' % g_, ' '.join('%7d' % (s[w, g_, k] - t0) for k in range(5)))
