#!/usr/bin/env python3
"""Small driver for rocprofv3: a few mtk::qr::qr calls of one workload.
usage: prof_run.py [mode] [steps] [--m M] [--n N] [--policy P] [--reorth] [--cond C] [--lda LD]
Defaults: the headline workload (2^20 x 64, fp32_tc_cor, auto policy).  --cond: latms-style matrix with the
reference's test_cond.cu singular-value draw (harness.get_rand_matrix_with_cond_number)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsqr_gpu_amd import blockqr as bq

ap = argparse.ArgumentParser()
ap.add_argument("mode", nargs="?", default="fp32_tc_cor")
ap.add_argument("steps", nargs="?", type=int, default=5)
ap.add_argument("--m", type=int, default=1 << 20)
ap.add_argument("--n", type=int, default=64)
ap.add_argument("--policy", type=int, default=0)
ap.add_argument("--reorth", action="store_true")
ap.add_argument("--cond", type=float, default=0.0)
ap.add_argument("--lda", type=int, default=0, help="leading dimension of A and Q (default m)")
args = ap.parse_args()
m, n = args.m, args.n
mode = bq.compute_mode[args.mode]
if args.cond > 0:
    # latms-style matrix with the reference's singular-value draw (src/test_cond.cu:31-50), built on the CPU so that the profile holds
    # only the factorisation's own kernels (harness.get_rand_matrix_with_cond_number orthogonalises its factors with the engine itself)
    import numpy as np
    rng = np.random.default_rng(0)
    s = np.empty(n); s[0] = 1.0 / np.sqrt(args.cond); s[-1] = 1.0
    s[1:-1] = 1.0 + (np.sqrt(args.cond) - 1.0) * rng.random(n - 2)
    s = np.sort(s)[::-1]
    u, _ = np.linalg.qr(rng.standard_normal((m, n)))
    v, _ = np.linalg.qr(rng.standard_normal((n, n)))
    d_a = torch.from_numpy(np.ascontiguousarray(((u * s) @ v.T).T.astype(np.float32))).cuda()
    del u
else:
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    d_a = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
ld = args.lda if args.lda > m else m
if ld > m:
    pad = torch.zeros(n, ld, device="cuda"); pad[:, :m] = d_a; d_a = pad
d_keep = d_a.clone()
d_q = torch.empty(n, ld, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
bf = bq.buffer(mode, args.reorth); bf.allocate(m, n)
bq.set_policy(args.policy)
for _ in range(args.steps):
    if n > 64:
        d_a.copy_(d_keep)                            # a is overwritten for n > 64
    assert bq.qr(d_q, ld, d_r, n, d_a, ld, m, n, bf) == 0
torch.cuda.synchronize()
print("done engine", bq.ENGINE_NAMES.get(bq.last_engine()))
