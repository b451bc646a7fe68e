#!/usr/bin/env python3
"""Do vector instructions issue in the shadow of MFMAs on this chip?  (selftest kernel issue_overlap_kernel: ten independent
v_mfma_f32_16x16x32_bf16 and thirty independent vector instructions per iteration; shader cycles per iteration of one wave)"""
import ctypes, os, time, torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tsqr_gpu_amd", "csrc", "libtsqr_selftest.so"))
L.tsqr_selftest_issue_overlap.restype = ctypes.c_int
L.tsqr_selftest_issue_overlap.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
out = torch.zeros(4, device="cuda"); cyc = torch.zeros(2, dtype=torch.int64, device="cuda")
iters = 20000
for wgs, what in ((1, "one workgroup: one wave per SIMD of one CU"), (2, "two workgroups"), (512, "512 workgroups: two waves per SIMD on every CU"), (1024, "1024 workgroups: four waves per SIMD")):
    for mode, name in ((0, "10 MFMA"), (1, "30 vector"), (2, "10 MFMA + 30 vector, interleaved 1:3"), (3, "10 MFMA, then 30 vector")):
        L.tsqr_selftest_issue_overlap(out.data_ptr(), cyc.data_ptr(), mode, wgs, 200)
        t0 = time.perf_counter()
        assert L.tsqr_selftest_issue_overlap(out.data_ptr(), cyc.data_ptr(), mode, wgs, iters) == 0
        dt = time.perf_counter() - t0
        c = int(cyc[0].item())
        print("%-52s %-40s %8.1f shader cycles / iteration (wave 0), kernel %.2f ms => %.2f GHz" % (what, name, c / iters, dt * 1e3, c / dt / 1e9), flush=True)
