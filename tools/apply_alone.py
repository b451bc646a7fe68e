"""apply kernel in isolation (staged C ABI, Z left in the work buffer by a previous qr call) vs inside the pipeline."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tsqr_gpu_amd import blockqr as bq
m, n = 1 << 20, 64
mode = bq.compute_mode.fp32_tc_cor
a = torch.rand(n, m, device='cuda') * 2 - 1
q = torch.empty(n, m, device='cuda'); r = torch.zeros(n, n, device='cuda')
bf = bq.buffer(mode, False); bf.allocate(m, n)
assert bq.qr(q, m, r, n, a, m, m, n, bf) == 0
L = bq.lib()
st = torch.cuda.current_stream().cuda_stream
def apply_only():
    assert L.tsqr_mi_apply_z_f32(int(mode), q.data_ptr(), m, a.data_ptr(), m, m, n, bf.dwq.data_ptr(), st) == 0
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('apply alone, back to back      : %.1f us' % timeit(apply_only))
g = torch.empty(2560, dtype=torch.float64, device='cuda')
def gram_only():
    assert L.tsqr_mi_gram_f32(2, g.data_ptr(), a.data_ptr(), m, m, n, bf.dwq.data_ptr(), bf.dwr.data_ptr(), st) == 0
print('gram (+reduce) alone           : %.1f us' % timeit(gram_only))
def both():
    gram_only(); apply_only()
print('gram (+reduce) then apply      : %.1f us' % timeit(both))
