"""Launch-path costs on this box: dependent tiny kernels, the 4-byte status copy, stream sync vs spinning on a pinned flag."""
import ctypes, torch
torch.zeros(1, device='cuda')
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
L.tsqr_selftest_launch_cost.restype = ctypes.c_double
L.tsqr_selftest_launch_cost.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4
d = torch.zeros(16, dtype=torch.int32, device='cuda'); h = torch.zeros(16, dtype=torch.int32).pin_memory()
for nk in (1, 2, 3, 5):
    for (cp, spin, name) in ((0, 0, 'sync'), (1, 0, 'copy+sync'), (0, 1, 'pinned-flag spin'), (0, 2, 'streamquery spin'), (1, 2, 'copy+query spin')):
        L.tsqr_selftest_launch_cost(d.data_ptr(), h.data_ptr(), nk, cp, spin, 50)
        us = L.tsqr_selftest_launch_cost(d.data_ptr(), h.data_ptr(), nk, cp, spin, 500)
        print('%d kernels, %-17s: %6.1f us per iteration' % (nk, name, us), flush=True)
