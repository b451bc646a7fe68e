"""Real streaming kernels next to the skeleton passes (same A, alternating): which side loses the time?"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tsqr_gpu_amd import blockqr as bq
S = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
m, n = 1 << 20, 64
mode = bq.compute_mode.fp32_tc_cor
a = torch.rand(n, m, device='cuda') * 2 - 1
q = torch.empty(n, m, device='cuda'); r = torch.zeros(n, n, device='cuda')
bf = bq.buffer(mode, False); bf.allocate(m, n)
assert bq.qr(q, m, r, n, a, m, m, n, bf) == 0
L = bq.lib()
st = torch.cuda.current_stream().cuda_stream
g = torch.empty(2560, dtype=torch.float64, device='cuda')
def real_apply():
    assert L.tsqr_mi_apply_z_f32(int(mode), q.data_ptr(), m, a.data_ptr(), m, m, n, bf.dwq.data_ptr(), st) == 0
def real_gram():
    assert L.tsqr_mi_gram_f32(2, g.data_ptr(), a.data_ptr(), m, m, n, bf.dwq.data_ptr(), bf.dwr.data_ptr(), st) == 0
def skel(modev, back):
    def f():
        modes = (ctypes.c_int * 1)(modev); dirs = (ctypes.c_int * 1)(back); ntl = (ctypes.c_int * 1)(0); out = (ctypes.c_float * 1)()
        S.tsqr_selftest_seq(ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(m), ctypes.c_size_t(m), 768, 1, modes, dirs, ntl, -1, out)
    return f
# reps = -1 in tsqr_selftest_seq: the loop runs reps + 2 = 1 time and skips the accumulation -> one plain launch (+ a device sync)
def timeit(fns, reps=12):
    k = len(fns)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
    acc = [0.0] * k
    for it in range(reps + 2):
        for i, f in enumerate(fns):
            ev[i][0].record(); f(); ev[i][1].record()
        torch.cuda.synchronize()
        if it >= 2:
            for i in range(k): acc[i] += ev[i][0].elapsed_time(ev[i][1]) * 1e3 / reps
    return acc
for name, fns in (("real gram+reduce | real apply", [real_gram, real_apply]),
                  ("real gram+reduce | skel copy bwd", [real_gram, skel(0, 1)]),
                  ("skel load fwd | real apply", [skel(1, 0), real_apply]),
                  ("skel load(c,q) | real apply", [skel(3, 0), real_apply]),
                  ("real apply alone", [real_apply]),
                  ("real gram alone", [real_gram])):
    t = timeit(fns)
    print("%-34s: %s   sum %.1f" % (name, "  ".join("%.1f" % x for x in t), sum(t)), flush=True)
