"""Per-call wall-clock distribution of the blocking call (headline size): looks for sporadic stalls."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tsqr_gpu_amd import blockqr as bq
m, n = 1 << 20, 64
mode = bq.compute_mode[sys.argv[1]] if len(sys.argv) > 1 else bq.compute_mode.fp32_tc_cor
reorth = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
N = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
a = torch.rand(n, m, device='cuda') * 2 - 1
q = torch.empty(n, m, device='cuda'); r = torch.zeros(n, n, device='cuda')
bf = bq.buffer(mode, reorth); bf.allocate(m, n)
call = bq.bind(q, m, r, n, a, m, m, n, bf)
for _ in range(10): call()
ts = []
for _ in range(N):
    t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
ts_sorted = sorted(ts)
print("%s reorth %d: median %.1f us  p99 %.1f us  max %.1f us  >1ms: %d of %d  positions %s" % (
    mode.name, reorth, ts_sorted[N // 2] * 1e6, ts_sorted[int(N * 0.99)] * 1e6, ts_sorted[-1] * 1e6,
    sum(t > 1e-3 for t in ts), N, [i for i, t in enumerate(ts) if t > 1e-3][:10]))
