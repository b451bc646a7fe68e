"""Extreme input scales: products that overflow / underflow fp32 must end in a level that works (fp64 Gram), never in NaN."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tsqr_gpu_amd import blockqr as bq, harness
def run(m=1 << 16, n=64, verbose=True):
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    base = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
    bad = 0
    for scale in (1e-30, 1e-22, 1e-16, 1e-14, 1e-13, 1e-12, 1.0, 1e12, 1e18, 1e19, 1e25):
        for mode in (bq.compute_mode.fp32_tc_cor, bq.compute_mode.fp32_notc):
            for reorth in (False, True):
                a = (base.double() * scale).float()
                st, q, r = harness.qr(a.clone(), m, n, mode, reorth)
                orth = harness.orthogonality_fro(q, m, n)
                res = harness.residual(q, r, a, m, n)
                ok = st == 0 and orth < 5e-6 and res < 2e-6 and torch.isfinite(r).all().item()
                bad += (not ok)
                if verbose: print("%s scale %-7g %-12s reorth %d engine %d orth %.2e res %.2e" % ("OK " if ok else "BAD", scale, mode.name, reorth, bq.last_engine(), orth, res), flush=True)
    return bad


if __name__ == "__main__":
    bad = run()
    print("bad:", bad)
    sys.exit(1 if bad else 0)
