"""Robustness probes: square matrices, wildly scaled columns / rows, constant columns, non-finite input (must return, not hang)."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tsqr_gpu_amd import blockqr as bq, harness


def check(name, a, m, n, mode, reorth, orth_tol, res_tol=2e-6, verbose=True):
    st, q, r = harness.qr(a.clone(), m, n, mode, reorth)
    orth = harness.orthogonality_fro(q, m, n)
    res = harness.residual(q, r, a, m, n)
    ok = st == 0 and orth < orth_tol and res < res_tol
    if verbose:
        print("%s %-34s %5dx%-4d %-12s reorth %d engine %d orth %.2e res %.2e" % ("OK " if ok else "BAD", name, m, n, mode.name, reorth, bq.last_engine(), orth, res), flush=True)
    return 0 if ok else 1


def run(verbose=True):
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    bad = 0
    modes = (bq.compute_mode.fp32_tc_cor, bq.compute_mode.fp32_notc)
    for n in (1, 7, 16, 64, 65, 128, 200):                     # square matrices (m == n)
        a = torch.rand(n, n, generator=g, device="cuda") * 2 - 1
        cond = harness.get_cond(a, n, n)
        for mode in modes:
            bad += check("square", a, n, n, mode, True, max(2e-5, 1e-7 * cond), verbose=verbose)
    m = 1 << 15
    for n in (48, 64, 100):
        base = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
        cs = torch.logspace(-8, 8, n, device="cuda")[torch.randperm(n, generator=g, device="cuda")]
        rs = torch.logspace(0, -6, m, device="cuda")
        for mode in modes:
            bad += check("columns scaled 1e-8..1e8", base * cs[:, None], m, n, mode, False, 5e-6, verbose=verbose)
            bad += check("rows scaled 1..1e-6", base * rs[None, :], m, n, mode, True, 5e-6, verbose=verbose)
        ones = base.clone(); ones[0, :] = 1.0; ones[n // 2, :] = -3.0          # two constant (parallel) columns: rank deficient
        for mode in modes:
            st, q, r = harness.qr(ones.clone(), m, n, mode, True)
            res = harness.residual(q, r, ones, m, n)
            orth = harness.orthogonality_fro(q, m, n)
            # exactly dependent columns: the residual stays at rounding level, R shows the deficiency, and all columns of Q but the
            # dependent one are orthonormal (that one is left un-normalised: ||Q^T Q - I||_F = 1)
            ok = st == 0 and math.isfinite(res) and res < 2e-6 and orth < 1.01
            bad += (not ok)
            if verbose:
                print("%s %-34s %5dx%-4d %-12s engine %d res %.2e orth %.3f" % ("OK " if ok else "BAD", "two parallel constant columns", m, n, mode.name, bq.last_engine(), res, orth), flush=True)
    # in place (q aliases a) on inputs that make the first level(s) reject: the speculative launches must leave A intact until a
    # level is accepted (device-side skip), otherwise the retry would factor garbage
    for cond in (1e0, 1e4, 1e8):
        mm, nn = 1 << 15, 64
        a0 = harness.get_rand_matrix_with_cond_number(mm, nn, cond, seed=11) if cond > 1 else (torch.rand(nn, mm, generator=g, device="cuda") * 2 - 1)
        for mode in modes:
            for reorth in (False, True):
                buf = a0.clone()
                r = torch.zeros(nn, nn, device="cuda")
                bf = bq.buffer(mode, reorth); bf.allocate(mm, nn)
                st = bq.qr(buf, mm, r, nn, buf, mm, mm, nn, bf)
                res = harness.residual(buf, r, a0, mm, nn)
                orth = harness.orthogonality_fro(buf, mm, nn)
                ok = st == 0 and res < 2e-6 and (orth < 2e-5 if (reorth or cond < 10) else orth < 1e-6 * cond * 50)
                bad += (not ok)
                if verbose:
                    print("%s in place cond %-6g %-12s reorth %d engine %d orth %.2e res %.2e" % ("OK " if ok else "BAD", cond, mode.name, reorth, bq.last_engine(), orth, res), flush=True)
    for (mm, nn) in ((20000, 100), (3000, 200)):                 # in place with several panels (a is the workspace of the coupling anyway)
        a0 = torch.rand(nn, mm, generator=g, device="cuda") * 2 - 1
        for mode in modes:
            for reorth in (False, True):
                buf = a0.clone()
                r = torch.zeros(nn, nn, device="cuda")
                bf = bq.buffer(mode, reorth); bf.allocate(mm, nn)
                st = bq.qr(buf, mm, r, nn, buf, mm, mm, nn, bf)
                res = harness.residual(buf, r, a0, mm, nn)
                orth = harness.orthogonality_fro(buf, mm, nn)
                ok = st == 0 and res < 2e-6 and orth < 2e-5
                bad += (not ok)
                if verbose:
                    print("%s in place %dx%d %-12s reorth %d engine %d orth %.2e res %.2e" % ("OK " if ok else "BAD", mm, nn, mode.name, reorth, bq.last_engine(), orth, res), flush=True)
    # one buffer allocated for the largest problem serves smaller ones
    bf = bq.buffer(bq.compute_mode.fp32_tc_cor, True); bf.allocate(1 << 16, 128)
    for (mm, nn) in ((1 << 16, 128), (5000, 64), (100, 7), (1 << 16, 64)):
        a0 = torch.rand(nn, mm, generator=g, device="cuda") * 2 - 1
        q = torch.empty(nn, mm, device="cuda"); r = torch.zeros(nn, nn, device="cuda")
        st = bq.qr(q, mm, r, nn, a0.clone(), mm, mm, nn, bf)
        res = harness.residual(q, r, a0, mm, nn); orth = harness.orthogonality_fro(q, mm, nn)
        ok = st == 0 and res < 2e-6 and orth < 2e-5
        bad += (not ok)
        if verbose:
            print("%s shared buffer %dx%d orth %.2e res %.2e" % ("OK " if ok else "BAD", mm, nn, orth, res), flush=True)
    nanm = torch.rand(64, 4096, generator=g, device="cuda"); nanm[3, 100] = float("nan"); nanm[10, 7] = float("inf")
    for mode in modes:                                           # non-finite input: must come back (state 0, non-finite output), not hang
        st, q, r = harness.qr(nanm.clone(), 4096, 64, mode, False)
        ok = st == 0
        bad += (not ok)
        if verbose:
            print("%s non-finite input returns, engine %d, finite Q entries: %d of %d" % ("OK " if ok else "BAD", bq.last_engine(), int(torch.isfinite(q).sum()), q.numel()), flush=True)
    return bad


if __name__ == "__main__":
    b = run()
    print("bad:", b)
    sys.exit(1 if b else 0)
