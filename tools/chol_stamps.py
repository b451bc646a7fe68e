#!/usr/bin/env python3
"""In-kernel time stamps of the Cholesky step (libtsqr_selftest.so is built with -DTSQR_CHOL_STAMPS): per-group breakdown in shader
cycles for the first four groups of chol16_kernel (the stamp table has four wave rows).  usage: chol_stamps.py [n]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_chol as tc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = ctypes.CDLL(os.path.join(tc.ROOT, "tsqr_gpu_amd", "csrc", "libtsqr_selftest.so"))
L.tsqr_selftest_chol_stamps.restype = ctypes.c_int
L.tsqr_selftest_chol_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
g, _ = tc.spd(n, 3.0, 1)
nt = (n + 15) // 16
gs = torch.from_numpy(tc.pack_tiles(g, n, 1)).cuda()
r = torch.zeros(n * n, device="cuda"); z = torch.zeros(256 * nt * nt, device="cuda"); st = torch.zeros(4, dtype=torch.int32, device="cuda")
out = torch.zeros(4 * 160, dtype=torch.int64, device="cuda")
rc = L.tsqr_selftest_chol_stamps(out.data_ptr(), r.data_ptr(), n, z.data_ptr(), st.data_ptr(), gs.data_ptr(), n, nt, 2, float(1 << 20))
assert rc == 0, rc
s = out.cpu().numpy().reshape(4, 160)
clk = (s[0, 3] - s[0, 0]) / max(1, (s[0, 4] - s[0, 5])) * 100.0      # MHz: shader cycles per 100 MHz tick
print("chol16_kernel n=%d: kernel body %d cycles (wave 0), clock ~%.0f MHz => %.2f us" % (n, s[0, 3] - s[0, 0], clk, (s[0, 3] - s[0, 0]) / clk))
print("prologue (start -> first group) %d   elimination %d   epilogue (verdict, images out) %d" % (s[0, 1] - s[0, 0], s[0, 2] - s[0, 1], s[0, 3] - s[0, 2]))
print("group (the stamp table keeps waves 0-3, i.e. the owners of groups 0-3): barrier wait of the owner / of wave (g + 2) % 4 | work after the barrier, owner / that wave | period (wave 0)")
for gi in range(min(4, (n + 3) // 4)):
    oth = (gi + 2) % 4
    b = 8 + 8 * gi
    o, x, w0 = s[gi], s[oth], s[0]
    print("g%02d owner w%d: bar %5d %5d | after %5d %5d | period %5d" % (
        gi, gi, o[b + 2] - o[b + 1], x[b + 2] - x[b + 1], o[b + 3] - o[b + 2], x[b + 3] - x[b + 2], w0[b + 9] - w0[b + 1]))

print("next owner's path (wave g+1 in iteration g): barrier release -> update done -> section arithmetic done -> rows published")
for gi in range(min(3, (n + 3) // 4 - 1)):
    o, b = s[gi + 1], 8 + 8 * gi
    print("g%02d -> owner w%d: update %5d | section %5d | publish %5d | total %5d" % (gi, gi + 1, o[b + 4] - o[b + 2], o[b + 5] - o[b + 4], o[b + 7] - o[b + 5], o[b + 7] - o[b + 2]))
