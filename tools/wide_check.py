#!/usr/bin/env python3
"""GPU check of the one-panel path for 64 < n <= 128 (tsqr_wide.hip) against fp64 numpy and against the 64-column panel path
(policy 5), plus per-call time of both.  usage: wide_check.py [--big]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tsqr_gpu_amd import blockqr as bq


def run(m, n, mode, reorth, policy, a_np=None, steps=1, seed=0):
    if a_np is None:
        g = torch.Generator(device="cuda"); g.manual_seed(seed)
        d_keep = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
    else:
        d_keep = torch.from_numpy(np.ascontiguousarray(a_np.T)).cuda()
    d_a = d_keep.clone()
    d_q = torch.empty(n, m, device="cuda"); d_r = torch.full((n, n), 7.0, device="cuda")
    bf = bq.buffer(bq.compute_mode[mode], reorth); bf.allocate(m, n)
    bq.set_policy(policy)
    assert bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf) == 0
    eng = bq.last_engine()
    torch.cuda.synchronize()
    ms = 0.0
    if steps > 1:
        tot = 0.0
        for _ in range(steps):
            d_a.copy_(d_keep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf)          # blocking
            tot += time.perf_counter() - t0
        ms = tot / steps * 1e3
    bq.set_policy(0)
    a = d_keep.double(); q = d_q.double(); r = d_r.double()          # stored transposed: (n, m), (n, n)
    res = (torch.linalg.norm(r @ q - a) / torch.linalg.norm(a)).item()        # tensors hold the transposes: (Q R)^T = R^T Q^T
    orth = torch.linalg.norm(q @ q.T - torch.eye(n, device="cuda", dtype=torch.float64)).item()
    low = torch.tril(d_r.T, -1).abs().max().item() if n > 1 else 0.0          # d_r.T is R (row i, col j)
    return dict(res=res, orth=orth, low=low, eng=eng, ms=ms, r=d_r.T.cpu().numpy().astype(np.float64))


def main():
    big = "--big" in sys.argv
    ok = True
    cases = [(4096, 128), (5000, 100), (3000, 65), (70001, 113), (1 << 16, 128), (200, 128), (129, 128)]
    for (m, n) in cases:
        for mode in ("fp32_tc_cor", "fp32_notc", "fp32_tc_nocor"):
            for reorth in (False, True):
                w = run(m, n, mode, reorth, 0)
                p = run(m, n, mode, reorth, 5)
                rw, rp = w["r"], p["r"]
                sw = np.sign(np.diag(rw)); sp = np.sign(np.diag(rp))
                dr = np.abs(sw[:, None] * rw - sp[:, None] * rp).max() / np.abs(rp).max()
                tol_o = 5e-6 if mode != "fp32_tc_nocor" else 5e-3
                tol_r = 5e-7 if mode != "fp32_tc_nocor" else 2e-3
                if w["eng"] != 5:                                # rejected (few rows / ill conditioned): the panel path ran, bit for bit
                    good = w["res"] < tol_r and w["low"] == 0.0 and dr == 0.0
                else:
                    good = w["res"] < tol_r and w["orth"] < tol_o and w["low"] == 0.0 and dr < (2e-5 if mode != "fp32_tc_nocor" else 1e-3)
                ok &= good
                print(f"{m}x{n} {mode} reorth={int(reorth)} wide: res {w['res']:.2e} orth {w['orth']:.2e} low {w['low']:.1e} eng {w['eng']} | "
                      f"panels: res {p['res']:.2e} orth {p['orth']:.2e} eng {p['eng']} | dR {dr:.2e} {'ok' if good else 'FAIL'}", flush=True)
    # ill-conditioned: the verdict must reject and the panel path take over
    rng = np.random.default_rng(1)
    m, n = 20000, 128
    u, _ = np.linalg.qr(rng.standard_normal((m, n))); v, _ = np.linalg.qr(rng.standard_normal((n, n)))
    for cond in (1e2, 1e4, 1e7):
        a = ((u * np.geomspace(1.0, 1.0 / cond, n)) @ v.T).astype(np.float32)
        for reorth in (False, True):
            w = run(m, n, "fp32_tc_cor", reorth, 0, a_np=a)
            p = run(m, n, "fp32_tc_cor", reorth, 5, a_np=a)
            print(f"cond {cond:.0e} reorth={int(reorth)} auto: res {w['res']:.2e} orth {w['orth']:.2e} eng {w['eng']} | panels: res {p['res']:.2e} orth {p['orth']:.2e} eng {p['eng']}", flush=True)
    if big:
        for mode in ("fp32_tc_cor", "fp32_notc"):
            for reorth in (False, True):
                w = run(1 << 20, 128, mode, reorth, 0, steps=10)
                p = run(1 << 20, 128, mode, reorth, 5, steps=10)
                print(f"2^20x128 {mode} reorth={int(reorth)}: wide {w['ms']:.3f} ms orth {w['orth']:.2e} res {w['res']:.2e} | panels {p['ms']:.3f} ms orth {p['orth']:.2e}", flush=True)
    print("ALL OK" if ok else "SOME FAILED")


main()
