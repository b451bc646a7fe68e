#!/usr/bin/env python3
"""Several 64-column panels (n > 128): speed (the reference's protocol, src/test.cu:257-343: one warm-up, C blocking calls) and accuracy of
mtk::qr::qr on the wide shapes of the reference's own sweep (src/main.cu:89-113: 4096 x 1024, 32768 x 1024) and some taller ones.
TSQR_MI_LIB=<other build> for a same-box A/B (tools/r04_ab.sh style).  usage: wide_speed.py [label]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsqr_gpu_amd import blockqr as bq, harness
label = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("TSQR_MI_LIB", "in-tree")
shapes = [(4096, 1024), (32768, 1024), (65536, 320), (1 << 18, 512), (1 << 20, 192), (1 << 20, 256)]
print("# %s (library version %d)" % (label, bq.lib().tsqr_mi_version()))
for mode in (bq.compute_mode.fp32_tc_cor, bq.compute_mode.fp32_notc):
    for reorth in (False, True):
        if reorth and mode != bq.compute_mode.fp32_tc_cor:
            continue
        for (m, n) in shapes:
            g = torch.Generator(device="cuda"); g.manual_seed(1)
            a = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
            a0 = a.clone()
            q = torch.empty(n, m, device="cuda"); r = torch.zeros(n, n, device="cuda")
            bf = bq.buffer(mode, reorth); bf.allocate(m, n)
            assert bq.qr(q, m, r, n, a, m, m, n, bf) == 0
            res = harness.residual(q, r, a0, m, n); orth = harness.orthogonality_fro(q, m, n)
            import hashlib
            sig = hashlib.sha1(q.cpu().numpy().tobytes() + r.cpu().numpy().tobytes()).hexdigest()[:12]
            ts = []
            for _ in range(6):
                a.copy_(a0); torch.cuda.synchronize()
                t0 = time.perf_counter()
                bq.qr(q, m, r, n, a, m, m, n, bf)
                ts.append(time.perf_counter() - t0)
            el = sorted(ts)[len(ts) // 2]
            f = 4.0 * m * n * n - 4.0 / 3.0 * n ** 3
            print("%-12s reorth %d  %8d x %4d  %9.1f us  %7.1f TFLOP/s (F_QR)  residual %.2e  ||QtQ-I||_F %.2e  engine %s  sha1(Q|R) %s" % (
                mode.name, int(reorth), m, n, el * 1e6, f / el * 1e-12, res, orth, bq.last_engine(), sig), flush=True)
            del a, a0, q, r, bf
