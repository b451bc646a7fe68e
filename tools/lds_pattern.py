#!/usr/bin/env python3
"""ds_read_b128 bank behaviour: lane (c, q) reads 16 bytes at dword A c + B q, one launch per (A, B).  Run under
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS; tools/lds_pattern.py prints the launch order."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tsqr_gpu_amd", "csrc", "libtsqr_selftest.so"))
out = torch.zeros(4, device="cuda")
PATS = [(4, 64), (36, 4), (36, 8), (36, 16), (4, 1152), (68, 4), (20, 4), (40, 4), (44, 4), (52, 4), (12, 4), (36, 68), (4, 16)]
for a, b in PATS:
    lib.tsqr_selftest_lds_pattern(ctypes.c_void_p(out.data_ptr()), a, b, 4096)
print("patterns (A, B) in launch order:", PATS)
