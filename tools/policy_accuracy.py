"""Orthogonality of the three R-factor engines on inputs of growing column correlation (which level should accept what)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tsqr_gpu_amd import blockqr as bq, harness

def run(a, m, n, pol):
    bq.set_policy(pol)
    st, q, r = harness.qr(a.clone(), m, n, bq.compute_mode.fp32_tc_cor, False)
    return harness.orthogonality_fro(q, m, n), bq.last_engine()

g = torch.Generator(device="cuda"); g.manual_seed(1)
cases = []
for (m, n) in [(1 << 20, 64), (1 << 15, 64), (17, 17), (300, 64)]:
    cases.append(("U(-1,1) %dx%d" % (m, n), m, n, torch.rand(n, m, generator=g, device="cuda") * 2 - 1))
    cases.append(("U(0,1) %dx%d" % (m, n), m, n, torch.rand(n, m, generator=g, device="cuda")))
    cases.append(("U(0,1)+10 %dx%d" % (m, n), m, n, torch.rand(n, m, generator=g, device="cuda") + 10))
for c in (1e1, 1e2, 1e3):
    cases.append(("latms cond %g 32768x64" % c, 1 << 15, 64, harness.get_rand_matrix_with_cond_number(1 << 15, 64, c, seed=2)))
for name, m, n, a in cases:
    cond = harness.get_cond(a, m, n) if m <= (1 << 15) else float("nan")
    out = []
    for pol, pn in ((3, "bf16"), (2, "fp64"), (1, "hh"), (0, "auto")):
        o, e = run(a, m, n, pol)
        out.append("%s %.2e(e%d)" % (pn, o, e))
    print("%-28s cond %9.3g | %s" % (name, cond, " | ".join(out)), flush=True)
bq.set_policy(0)
