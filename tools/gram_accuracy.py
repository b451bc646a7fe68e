#!/usr/bin/env python3
"""Accuracy of the bf16-split Gram engine on inputs that would show a biased or too short accumulation: orthogonality and residual
of 64-column factorisations at 2^20 .. 2^24 rows for zero-mean, same-sign and column-scaled inputs (the sweep that showed what fp32
wave totals cost: profiles/r03_experiment_log.md).  usage: gram_accuracy.py [log2 m ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsqr_gpu_amd import blockqr as bq, harness
n = 64
for lm in ([int(x) for x in sys.argv[1:]] or [20, 22, 23, 24]):
    m = 1 << lm
    g = torch.Generator(device="cuda"); g.manual_seed(lm)
    for name, make in (("U(-1,1)", lambda: torch.rand(n, m, generator=g, device="cuda") * 2 - 1),
                       ("U(0,1)", lambda: torch.rand(n, m, generator=g, device="cuda")),
                       ("U(0,1)+0.25", lambda: torch.rand(n, m, generator=g, device="cuda") + 0.25),
                       ("N(0,1)", lambda: torch.randn(n, m, generator=g, device="cuda")),
                       ("|N(0,1)| * 2^col", lambda: torch.randn(n, m, generator=g, device="cuda").abs_() * (2.0 ** torch.arange(n, device="cuda", dtype=torch.float32))[:, None])):
        a = make()
        st, q, r = harness.qr(a.clone(), m, n, bq.compute_mode.fp32_tc_cor, False)
        print("2^%d x %d %-18s engine %d  orth %.3e  residual %.3e" % (lm, n, name, bq.last_engine(),
              harness.orthogonality_fro(q, m, n), harness.residual(q, r, a, m, n)), flush=True)
        del a, q, r
