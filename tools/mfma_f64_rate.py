"""Achievable fp64 MFMA rate (v_mfma_f64_16x16x4_f64, ten independent accumulators per wave, registers only)."""
import ctypes, torch
out = torch.zeros(4, dtype=torch.float64, device='cuda')
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
L.tsqr_selftest_mfma_f64_rate.restype = ctypes.c_float
L.tsqr_selftest_mfma_f64_rate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
for wgs in (256, 512, 1024):
    iters = 2000
    ms = L.tsqr_selftest_mfma_f64_rate(out.data_ptr(), wgs, iters)
    flops = wgs * 4 * iters * 10 * 2 * 16 * 16 * 4
    print('%4d workgroups (x4 waves): %.3f ms  %.1f TFLOP/s fp64' % (wgs, ms, flops / ms / 1e9))
