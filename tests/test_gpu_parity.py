"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the oracle.

Floating-point path -> tolerance-based parity (north_star: match the reference's fp32_notc / fp32_tc_cor outputs
within a stated ||A-QR||/||A|| and ||Q^T Q - I|| tolerance on identical inputs):
  * residual  ||A-QR||_F/||A||_F  <= 5e-7 (the oracle itself reaches 1e-6 / 2.5e-6)
  * ||Q^T Q - I||_F               <= 5e-6 for cond(A) < 10 (oracle: 3e-6 / 1e-5); scaled by cond(A) otherwise
  * sign-normalised R and Q agree with the oracle's to 2e-5 * cond(A) (relative to max|R|, absolute for Q)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RES_TOL = 5e-7
ORTH_TOL = 5e-6
PAR_TOL = 2e-5


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def run_gpu(bq, torch, a, mode, reorth, lda_pad=0, ldq_pad=0):
    m, n = a.shape
    lda, ldq = m + lda_pad, m + ldq_pad
    buf = np.zeros((n, lda), np.float32)
    buf[:, :m] = a.T
    d_a = torch.from_numpy(buf).cuda()
    d_q = torch.full((n, ldq), float("nan"), dtype=torch.float32, device="cuda")
    d_r = torch.zeros(n, n, dtype=torch.float32, device="cuda")      # caller pre-zeros R (reference src/test.cu:129)
    bf = bq.buffer(mode, reorth)
    bf.allocate(m, n)
    st = bq.qr(d_q, ldq, d_r, n, d_a, lda, m, n, bf)
    q_full = d_q.cpu().numpy()
    if ldq_pad:
        assert np.isnan(q_full[:, m:]).all(), "wrote outside the m x n block of Q"
    return st, q_full[:, :m].T.copy(), d_r.cpu().numpy().T.copy()


CASES = [(128, 16), (64, 16), (33, 16), (32, 16), (20, 7), (17, 17), (100, 7), (200, 40), (1000, 64), (4096, 64),
         (9211, 51), (4097, 33), (65536, 64), (4096, 128), (9000, 100), (5000, 200)]


@pytest.mark.parametrize("m,n", CASES)
@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor"])
def test_parity_with_oracle(bq, oracle, torch_cuda, m, n, mode):
    md = bq.compute_mode[mode]
    a = oracle.uniform_matrix(m, n, seed=11)
    cond = np.linalg.cond(a.astype(np.float64))
    st, q, r = run_gpu(bq, torch_cuda, a, md, False, lda_pad=(5 if m % 2 else 0), ldq_pad=3)
    assert st == bq.success_factorization
    assert np.isfinite(q).all() and np.isfinite(r).all()
    assert np.abs(np.tril(r, -1)).max() == 0.0                      # exact zeros below the diagonal
    assert oracle.residual(a, q, r) < RES_TOL
    assert oracle.orthogonality_fro(q) < ORTH_TOL * max(1.0, cond / 10)
    st_o, q_o, r_o = oracle.qr(a, int(md), False)
    assert st_o == 0
    qn, rn = oracle.sign_normalise(q, r)
    qon, ron = oracle.sign_normalise(q_o, np.triu(r_o))
    scale = max(1.0, cond / 10)
    assert np.abs(rn - ron).max() / np.abs(ron).max() < PAR_TOL * scale
    assert np.abs(qn - qon).max() < PAR_TOL * scale


@pytest.mark.parametrize("m", [128, 256, 384, 640, 128 * 17, 128 * 512, 128 * 513, 128 * 1025, 128 * 1538])
def test_block_pattern_gram_kernel_against_the_chunk_kernel(bq, oracle, torch_cuda, m):
    """Full 64-column matrices with m % 128 == 0 and 16-byte aligned columns take gram_blk_kernel (block-pattern loads, LDS staging),
    everything else gram_bf16_kernel: the same matrix through both (lda = m and lda = m + 4 against lda = m + 1, which is not
    aligned) must give the same R up to the summation order of the fp64 totals, for every loop shape of the new kernel -- fewer
    blocks than workgroups, one / two / three blocks per workgroup, an odd block count in the last round."""
    md = bq.compute_mode.fp32_tc_cor
    a = oracle.uniform_matrix(m, 64, seed=m)
    out = {}
    bq.set_policy(bq.POLICY_GRAM_BF16)                              # (a 128 x 64 matrix is too short for the auto policy's bf16 level)
    try:
        for pad in (0, 4, 1):
            st, q, r = run_gpu(bq, torch_cuda, a, md, False, lda_pad=pad, ldq_pad=3)
            assert st == bq.success_factorization and bq.last_engine() == 3, (pad, st, bq.last_engine())
            assert oracle.residual(a, q, r) < RES_TOL and oracle.orthogonality_fro(q) < ORTH_TOL * (1 if m >= 1024 else 10)
            out[pad] = r
    finally:
        bq.set_policy(bq.POLICY_AUTO)
    scale = np.abs(out[1]).max()
    for pad in (0, 4):
        assert np.abs(out[pad] - out[1]).max() <= 2e-6 * scale, (pad, np.abs(out[pad] - out[1]).max() / scale)
    assert np.array_equal(out[0], out[4])                           # same kernel, same order: bit-identical


@pytest.mark.parametrize("m,n", [(9211, 51), (4096, 128), (9000, 100), (5000, 200)])
@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor"])
def test_parity_with_oracle_reorth(bq, oracle, torch_cuda, m, n, mode):
    """Reorthogonalize = true against the oracle's BCGS2 (reference src/blockqr.cu:180-390), element-wise on the sign-normalised
    factors, for single-panel and multi-panel (n > 64) shapes."""
    md = bq.compute_mode[mode]
    a = oracle.uniform_matrix(m, n, seed=12)
    st, q, r = run_gpu(bq, torch_cuda, a, md, True, lda_pad=(5 if m % 2 else 0), ldq_pad=3)
    assert st == bq.success_factorization
    assert np.abs(np.tril(r, -1)).max() == 0.0
    assert oracle.residual(a, q, r) < RES_TOL
    assert oracle.orthogonality_fro(q) < ORTH_TOL                    # O(eps) whatever cond(A) is
    st_o, q_o, r_o = oracle.qr(a, int(md), True)
    assert st_o == 0
    qn, rn = oracle.sign_normalise(q, r)
    qon, ron = oracle.sign_normalise(q_o, np.triu(r_o))
    scale = max(1.0, np.linalg.cond(a.astype(np.float64)) / 10)
    assert np.abs(rn - ron).max() / np.abs(ron).max() < PAR_TOL * scale
    assert np.abs(qn - qon).max() < PAR_TOL * scale
    q2, r2 = np.linalg.qr(a.astype(np.float64))                       # and against fp64 LAPACK
    _, r2n = oracle.sign_normalise(q2, r2)
    assert np.abs(rn - r2n).max() / np.abs(r2n).max() < 5e-6 * scale


@pytest.mark.parametrize("n", [16, 7])
@pytest.mark.parametrize("cond", [1e2, 1e4, 1e6])
@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor"])
def test_small_n_ill_conditioned_without_reorth(bq, oracle, torch_cuda, n, cond, mode):
    """n <= 16, Reorthogonalize = false: the reference's tsqr16 forms Q from its Householder tree and stays O(eps) orthogonal at
    any conditioning (src/tsqr.cu:1064-1310).  Q = A * inverse(R) alone would lose cond * eps here; the engine notices the scaled
    conditioning of the accepted sweep and runs its second sweep by itself -- parity with the oracle at O(eps)."""
    a = oracle.matrix_with_cond(1 << 13, n, cond, seed=3)
    md = bq.compute_mode[mode]
    st, q, r = run_gpu(bq, torch_cuda, a, md, False)
    assert st == 0
    assert oracle.residual(a, q, r) < 2e-6
    orth = oracle.orthogonality_fro(q)
    _, q_o, r_o = oracle.qr(a, int(md), False)
    assert orth < 5e-6 and orth < 3 * max(oracle.orthogonality_fro(q_o), 2e-6)
    assert np.abs(np.tril(r, -1)).max() == 0.0


@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor", "fp32_tc_nocor"])
def test_golden_fixtures(bq, oracle, torch_cuda, mode):
    import json, os
    G = os.path.join(os.path.dirname(__file__), "golden")
    for case in json.load(open(os.path.join(G, "golden.json")))["cases"]:
        data = np.load(os.path.join(G, case["file"]))
        a = oracle.uniform_matrix(case["m"], case["n"], seed=case["seed"])
        st, q, r = run_gpu(bq, torch_cuda, a, bq.compute_mode[mode], False)
        assert st == 0
        absr = np.abs(r)
        assert np.allclose(absr, data["absr_lapack64"], rtol=0, atol=5e-6 * absr.max())     # (R of every mode comes from the corrected Gram path)
        # the oracle's |R| of the same mode: fp32 modes within 2e-5 of the largest entry; fp32_tc_nocor within the reference
        # arithmetic's own distance from the exact factor (its fp16 H and products put its R 1e-3 .. 1e-2 away)
        far = np.abs(data["absr_" + mode] - data["absr_lapack64"]).max()
        assert np.allclose(absr, data["absr_" + mode], rtol=0, atol=(2e-5 * absr.max() if mode != "fp32_tc_nocor" else far + 5e-6 * absr.max()))
        assert oracle.residual(a, q, r) < case["residual_max"][mode]
        assert oracle.orthogonality_fro(q) < case["orth_fro_max"][mode]


@pytest.mark.parametrize("cond", [1e2, 1e4, 2.0 ** 15, 1e8])
@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor"])
def test_ill_conditioned_reorth(bq, oracle, torch_cuda, cond, mode):
    """latms-style inputs (reference src/test_cond.cu:20-76); Reorthogonalize=true must give O(eps) orthogonality,
    never worse than the oracle's BCGS2; without it the loss may grow like cond*eps but the residual stays small."""
    a = oracle.matrix_with_cond(1 << 14, 64, cond, seed=5)
    md = bq.compute_mode[mode]
    st, q, r = run_gpu(bq, torch_cuda, a, md, True)
    assert st == 0
    assert oracle.residual(a, q, r) < 2e-6
    orth = oracle.orthogonality_fro(q)
    assert orth < 1e-5
    _, q_o, r_o = oracle.qr(a, int(md), True)
    assert orth < 3 * max(oracle.orthogonality_fro(q_o), 2e-6)
    # the factors themselves against the oracle's (not only the metrics).  R: every entry within 2e-5 of the LARGEST entry -- an
    # absolute bound, which is what both algorithms deliver for the rows that belong to tiny singular values (beyond cond ~1e7 the
    # fp32 matrix is numerically rank deficient and those rows are determined to eps * |A| only).  Q: column j of either result is
    # determined to (its algorithm's error level) * |r_00 / r_jj|; the columns for which that is below 1e-3 are compared entry-wise
    # within base * max(1, |r_00 / r_jj| / 10), base = 2e-5 (fp32_notc) / 5e-5 (fp32_tc_cor: the reference's fp16-split arithmetic,
    # which the oracle restates, is the less accurate side: 8e-6 against 3e-6 in ||Q^T Q - I||).
    qn, rn = oracle.sign_normalise(q, r)
    qon, ron = oracle.sign_normalise(q_o, np.triu(r_o))
    assert np.abs(rn - ron).max() <= 2e-5 * np.abs(ron).max()
    tol = (2e-5 if mode == "fp32_notc" else 5e-5) * np.maximum(1.0, np.abs(ron[0, 0] / np.diag(ron)) / 10.0)
    good = tol < 1e-3
    assert good.sum() >= (32 if cond <= 1e4 else 2)
    assert np.all(np.abs(qn - qon).max(axis=0)[good] <= tol[good])
    st, q0, r0 = run_gpu(bq, torch_cuda, a, md, False)
    assert oracle.residual(a, q0, r0) < 2e-6
    # single sweep: loss of orthogonality ~ cond * eps32 up to cond ~ 1e6 (the Gram levels); beyond that the shifted Cholesky QR
    # two-step takes over and the loss is that of ITS second sweep on Q1 (cond(Q1) <~ 1e5): measured 2e-4 .. 6e-4 at cond 1e8
    assert oracle.orthogonality_fro(q0) < (max(1e-5, 1e-6 * cond) if cond < 1e7 else 5e-3)
    _, q_o0, _ = oracle.qr(a, int(md), False)
    if cond <= 2.0 ** 15:       # the reference's own sweep range (src/main.cu:104-111): never worse than the oracle
        assert oracle.orthogonality_fro(q0) < 3 * max(oracle.orthogonality_fro(q_o0), 2e-6)


def test_error_codes_on_gpu(bq, torch_cuda):
    torch = torch_cuda
    t = torch.zeros(64, device="cuda")
    bf = bq.buffer(bq.compute_mode.fp32_tc_cor, False)
    bf.allocate(8, 4)
    assert bq.qr(t, 4, t, 8, t, 4, 4, 8, bf) == bq.error_invalid_matrix_size
    assert bq.qr(t, 1, t, 1, t, 1, 0, 0, bf) == bq.error_invalid_matrix_size
    assert bq.qr(t, 8, t, 4, t, 8, 8, 4, bf, mode=bq.compute_mode.tf32_tc_cor) == bq.error_unsupported_mode
    with pytest.raises(RuntimeError):
        bf.allocate(8, 4)                                            # reference src/blockqr.hpp:77-79


def test_call_loop_entry_point(bq, oracle, torch_cuda):
    """tsqr_mi_qr_f32_loop (the reference's speed loop, src/test.cu:299-309, on the C side of the ABI): count back-to-back blocking
    calls give bit for bit what one call gives, the first non-zero state is returned, count = 0 does nothing."""
    torch = torch_cuda
    m, n = 20000, 64
    a = oracle.uniform_matrix(m, n, seed=3)
    d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    bf = bq.buffer(bq.compute_mode.fp32_tc_cor, False); bf.allocate(m, n)
    q1 = torch.empty(n, m, device="cuda"); r1 = torch.zeros(n, n, device="cuda")
    assert bq.qr(q1, m, r1, n, d_a, m, m, n, bf) == 0
    q3 = torch.full((n, m), float("nan"), device="cuda"); r3 = torch.zeros(n, n, device="cuda")
    loop = bq.bind_loop(q3, m, r3, n, d_a, m, m, n, bf)
    assert loop(0) == 0 and bool(torch.isnan(q3).all())
    assert loop(3) == 0
    torch.cuda.synchronize()
    assert torch.equal(q1, q3) and torch.equal(r1, r3)
    bad = bq.bind_loop(q3, 4, r3, 8, d_a, 4, 4, 8, bf)                  # n > m: every call of the loop returns 1, the loop stops at once
    assert bad(5) == bq.error_invalid_matrix_size


def test_input_not_clobbered_for_single_panel(bq, oracle, torch_cuda):
    torch = torch_cuda
    m, n = 3000, 64
    a = oracle.uniform_matrix(m, n, seed=2)
    d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    d_q = torch.empty(n, m, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
    bf = bq.buffer(bq.compute_mode.fp32_tc_cor, False); bf.allocate(m, n)
    bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf)
    assert np.array_equal(d_a.cpu().numpy().T, a)


def test_inplace_q_aliases_a(bq, oracle, torch_cuda):
    # the reference's BCGS2 calls tsqr16 with q == a (src/blockqr.cu:297-307); the engine supports the same aliasing
    torch = torch_cuda
    m, n = 5000, 48
    a = oracle.uniform_matrix(m, n, seed=4)
    d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    d_r = torch.zeros(n, n, device="cuda")
    bf = bq.buffer(bq.compute_mode.fp32_notc, False); bf.allocate(m, n)
    assert bq.qr(d_a, m, d_r, n, d_a, m, m, n, bf) == 0
    q = d_a.cpu().numpy().T; r = d_r.cpu().numpy().T
    assert oracle.residual(a, q, r) < RES_TOL and oracle.orthogonality_fro(q) < ORTH_TOL


@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor"])
def test_full_size_properties(bq, oracle, torch_cuda, mode):
    """BASELINE.json C2 at full size (2^20 x 64): size-independent properties evaluated on the device in fp64 --
    orthogonality, residual, R upper triangular, idempotence (qr(Q) gives R2 = +-I)."""
    torch = torch_cuda
    m, n = 1 << 20, 64
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    d_a = (torch.rand(n, m, generator=g, device="cuda", dtype=torch.float32) * 2 - 1)
    d_q = torch.empty(n, m, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
    bf = bq.buffer(bq.compute_mode[mode], False); bf.allocate(m, n)
    assert bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf) == 0
    q64 = d_q.double()                                               # (n, m) = Q^T
    gram = q64 @ q64.T - torch.eye(n, device="cuda", dtype=torch.float64)
    assert gram.norm().item() < 1e-5                                 # north_star: ||Q^T Q - I||_F < 1e-5
    r64 = d_r.double().T.contiguous()                                # R (n x n)
    resid = (r64.T @ q64 - d_a.double()).norm().item() / d_a.double().norm().item()
    assert resid < RES_TOL
    assert torch.tril(d_r.T, -1).abs().max().item() == 0.0
    d_r2 = torch.zeros(n, n, device="cuda"); d_q2 = torch.empty(n, m, device="cuda")
    assert bq.qr(d_q2, m, d_r2, n, d_q, m, m, n, bf) == 0
    assert (d_r2.T.abs() - torch.eye(n, device="cuda")).abs().max().item() < 5e-6


def test_scaling_linearity(bq, oracle, torch_cuda):
    # qr(s*A) = (Q, s*R) for a power of two s: exact in every engine (bf16 split keeps the fp32 exponent range)
    m, n = 6000, 64
    a = oracle.uniform_matrix(m, n, seed=9)
    for mode in (bq.compute_mode.fp32_notc, bq.compute_mode.fp32_tc_cor):
        _, q1, r1 = run_gpu(bq, torch_cuda, a, mode, False)
        for s in (2.0 ** -20, 2.0 ** 12):
            _, q2, r2 = run_gpu(bq, torch_cuda, (a * s).astype(np.float32), mode, False)
            assert np.array_equal(r2, (r1 * s).astype(np.float32))
            assert np.array_equal(q2, q1)


def test_cpp_sample_through_header(bq, torch_cuda):
    """The reference README's sample, written against include/tsqr/blockqr.hpp (prebuilt by the CPU check)."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "cpp", "sample_blockqr")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "SAMPLE OK" in out.stdout, out.stdout + out.stderr
    # the lower-level entry mtk::tsqr::tsqr16 / mtk::tsqr::buffer (reference src/tsqr.hpp:49-140) through include/tsqr/tsqr.hpp
    t16 = os.path.join(os.path.dirname(exe), "sample_tsqr16")
    if not os.path.exists(t16):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s", "sample_tsqr16"])
    out = subprocess.run([t16], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "TSQR16 SAMPLE OK" in out.stdout, out.stdout + out.stderr
    # the reference's speed protocol from C++ (no Python in the loop): schema and sanity only, the numbers live in profiles/
    spd = os.path.join(os.path.dirname(exe), "speed_blockqr")
    if not os.path.exists(spd):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s", "speed_blockqr"])
    out = subprocess.run([spd, "65536", "64", "2"], capture_output=True, text=True, timeout=300)
    lines = out.stdout.strip().split("\n")
    assert out.returncode == 0 and len(lines) == 6 and lines[1].startswith("65536,64,1,float,fp32_tc_cor,0,"), out.stdout + out.stderr
    assert lines[2].startswith("65536,64,1,float,fp32_tc_cor/two_in_flight,0,")      # the same calls through qr_submit / qr_finish
    assert lines[3].startswith("65536,64,1,float,fp32_tc_cor/qr_batch_4_rotating_matrices,0,")   # four different matrices through mtk::qr::qr_batch
    assert all(float(l.split(",")[6]) > 0 for l in lines[1:])


@pytest.mark.parametrize("policy", ["householder", "gram_f64", "gram_bf16"])
@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor"])
@pytest.mark.parametrize("m,n", [(9211, 51), (4096, 64), (100, 7), (5000, 130)])
def test_all_r_engines(bq, oracle, torch_cuda, policy, mode, m, n):
    """The three R-factor engines (Householder TSQR, fp64 Gram, bf16-split Gram) agree within tolerance in both modes."""
    a = oracle.uniform_matrix(m, n, seed=21)
    pol = {"householder": bq.POLICY_HOUSEHOLDER, "gram_f64": bq.POLICY_GRAM_F64, "gram_bf16": bq.POLICY_GRAM_BF16}[policy]
    bq.set_policy(pol)
    try:
        st, q, r = run_gpu(bq, torch_cuda, a, bq.compute_mode[mode], False)
        eng = bq.last_engine()
    finally:
        bq.set_policy(bq.POLICY_AUTO)
    assert st == 0 and eng == {"householder": 0, "gram_f64": 1, "gram_bf16": 3}[policy]
    assert oracle.residual(a, q, r) < RES_TOL and oracle.orthogonality_fro(q) < ORTH_TOL
    q2, r2 = np.linalg.qr(a.astype(np.float64))
    qn, rn = oracle.sign_normalise(q, r)
    q2n, r2n = oracle.sign_normalise(q2, r2)
    assert np.abs(rn - r2n).max() / np.abs(r2n).max() < 5e-6
    assert np.abs(qn - q2n).max() < 5e-6
    # and against the restatement of the reference's own Householder TSQR / block Gram-Schmidt (oracle/ref_tsqr.c)
    _, q_o, r_o = oracle.qr(a, int(bq.compute_mode[mode]), False)
    qon, ron = oracle.sign_normalise(q_o, np.triu(r_o))
    assert np.abs(rn - ron).max() / np.abs(ron).max() < PAR_TOL
    assert np.abs(qn - qon).max() < PAR_TOL


def test_two_host_threads_two_streams(bq, oracle, torch_cuda):
    """The boundary is re-entrant like the reference's (src/blockqr.cu:394-433 keeps no state): two host threads factor different
    matrices at the same time, each with its own buffer and stream, many times over; both must match the oracle every time."""
    import threading
    torch = torch_cuda
    shapes = [(60000, 64, "fp32_tc_cor", False), (41111, 51, "fp32_notc", True)]
    results, errors = {}, []

    def work(idx):
        try:
            m, n, mode, reorth = shapes[idx]
            md = bq.compute_mode[mode]
            a = oracle.uniform_matrix(m, n, seed=100 + idx)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
                d_q = torch.empty(n, m, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
                bf = bq.buffer(md, reorth); bf.allocate(m, n)
                stream.synchronize()
                for _ in range(40):
                    assert bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf, stream=stream) == 0
                    assert bq.last_engine() == 3             # per-thread diagnostics
                stream.synchronize()
                results[idx] = (a, d_q.cpu().numpy().T.copy(), d_r.cpu().numpy().T.copy(), md, reorth)
        except Exception as e:                               # surface failures of the worker threads
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for idx in range(2):
        a, q, r, md, reorth = results[idx]
        assert oracle.residual(a, q, r) < RES_TOL and oracle.orthogonality_fro(q) < ORTH_TOL
        _, q_o, r_o = oracle.qr(a, int(md), reorth)
        qn, rn = oracle.sign_normalise(q, r)
        qon, ron = oracle.sign_normalise(q_o, np.triu(r_o))
        assert np.abs(rn - ron).max() / np.abs(ron).max() < PAR_TOL and np.abs(qn - qon).max() < PAR_TOL


def test_auto_policy_escalation(bq, oracle, torch_cuda):
    """auto: bf16-split Gram for nearly orthogonal columns (any mode), fp64 Gram for moderate cond, shifted Cholesky QR when
    the plain Cholesky factorisation is rejected (rank deficient, cond ~1e8); policy 4 skips the bf16-split level."""
    a = oracle.uniform_matrix(4096, 64, seed=3)
    run_gpu(bq, torch_cuda, a, bq.compute_mode.fp32_tc_cor, False)
    assert bq.last_engine() == 3
    run_gpu(bq, torch_cuda, a, bq.compute_mode.fp32_notc, False)
    assert bq.last_engine() == 3                                   # same R-factor levels for every mode
    bq.set_policy(bq.POLICY_AUTO_NO_BF16) if hasattr(bq, "POLICY_AUTO_NO_BF16") else bq.set_policy(4)
    try:
        run_gpu(bq, torch_cuda, a, bq.compute_mode.fp32_notc, False)
        assert bq.last_engine() == 1                               # policy 4: auto without the bf16-split level
    finally:
        bq.set_policy(bq.POLICY_AUTO)
    # a matrix whose Gram matrix is numerically singular in fp64 must end in the Householder engine
    sing = oracle.uniform_matrix(8192, 64, seed=8)
    sing[:, 63] = sing[:, 0] * 0.5 + sing[:, 1] * 0.25                # exact linear dependence (rank 63)
    for md in (bq.compute_mode.fp32_tc_cor, bq.compute_mode.fp32_notc):
        st, q, r = run_gpu(bq, torch_cuda, sing, md, False)
        assert st == 0 and bq.last_engine() == 4 and oracle.residual(sing, q, r) < 2e-6      # shifted Cholesky QR
        assert oracle.orthogonality_fro(q) < 1e-4                  # a full orthonormal basis even for the dependent column
        assert abs(r[63, 63]) < 1e-4 * abs(r[0, 0])                # the rank deficiency shows in R
    mid = oracle.matrix_with_cond(1 << 14, 64, 1e3, seed=5)
    for reorth in (False, True):
        st, q, r = run_gpu(bq, torch_cuda, mid, bq.compute_mode.fp32_tc_cor, reorth)
        assert st == 0 and bq.last_engine() in (1, 3)
        assert oracle.residual(mid, q, r) < 2e-6
        assert oracle.orthogonality_fro(q) < (1e-5 if reorth else 1e-2)
    bad = oracle.matrix_with_cond(1 << 14, 64, 1e8, seed=5)
    for reorth in (False, True):
        st, q, r = run_gpu(bq, torch_cuda, bad, bq.compute_mode.fp32_tc_cor, reorth)
        assert st == 0 and bq.last_engine() in (1, 2, 4)           # never the bf16-split level
        assert np.isfinite(q).all() and oracle.residual(bad, q, r) < 2e-6
        if reorth:
            assert oracle.orthogonality_fro(q) < 1e-5


@pytest.mark.parametrize("mode,policy,want_engine", [("fp32_notc", 0, 3), ("fp32_tc_cor", 0, 3),
                                                     ("fp32_notc", 1, 0), ("fp32_tc_cor", 1, 0)])
@pytest.mark.parametrize("reorth", [False, True])
def test_dist_driver_single_rank(bq, oracle, torch_cuda, mode, policy, want_engine, reorth):
    """The row-partitioned driver on one rank through its torch.distributed-callback transport (the collectives degenerate to
    copies): Gram / Cholesky / apply by default, fold / all-gather / fold / apply when the Householder engine is forced."""
    torch = torch_cuda
    from tsqr_gpu_amd import dist as tdist
    m, n = 20000, 64
    a = oracle.uniform_matrix(m, n, seed=13)
    d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    d_q = torch.empty(n, m, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
    eng = tdist.RowPartitionedQR(bq.compute_mode[mode], m, n, comm="callbacks")
    bq.set_policy(policy)
    try:
        st = eng.qr(d_q, m, d_r, d_a, m, reorthogonalize=reorth)
    finally:
        bq.set_policy(bq.POLICY_AUTO)
    torch.cuda.synchronize()
    assert st == 0 and eng.last_engine == want_engine
    q = d_q.cpu().numpy().T; r = d_r.cpu().numpy().T
    assert oracle.residual(a, q, r) < RES_TOL and oracle.orthogonality_fro(q) < ORTH_TOL
    assert np.abs(np.tril(r, -1)).max() == 0.0


@pytest.mark.parametrize("m,n", [(128, 16), (20, 7), (4096, 64), (9211, 51), (65536, 64), (9000, 100)])
@pytest.mark.parametrize("reorth", [False, True])
def test_fp32_tc_nocor_mode(bq, oracle, torch_cuda, m, n, reorth):
    """fp32_tc_nocor (reference src/tcqr32x16.cu:186-226, 498-614, src/tsqr.cu:206-266, 724-788: fp16 matrix-core products, no
    correction terms): fp16-level accuracy by construction.  Compared with (1) the oracle's restatement of that mode on the same
    input -- the engine must be at least as accurate as the reference's arithmetic in both metrics and agree with its R and Q
    within the sum of the two half-precision error levels -- and (2) fp64 LAPACK with the tolerance of one fp16 rounding per operand
    (2^-11 = 4.9e-4): residual < 1e-3, ||Q^T Q - I||_F < 4e-3 * sqrt(n / 64 + 1), sign-normalised R within 1e-3.  The lower bound on
    the residual says the uncorrected half-precision product really ran (fp32_tc_cor would give 5e-8); the R factor itself keeps
    fp32_tc_cor accuracy for one sweep of one panel (it is computed the same way)."""
    a = oracle.uniform_matrix(m, n, seed=21)
    st, q, r = run_gpu(bq, torch_cuda, a, bq.compute_mode.fp32_tc_nocor, reorth, lda_pad=3, ldq_pad=1)
    assert st == bq.success_factorization
    assert np.isfinite(q).all() and np.isfinite(r).all() and np.abs(np.tril(r, -1)).max() == 0.0
    res, orth = oracle.residual(a, q, r), oracle.orthogonality_fro(q)
    assert 1e-6 < res < 1e-3, res                       # really the uncorrected half-precision product, not fp32_tc_cor
    assert orth < 4e-3 * np.sqrt(n / 64 + 1), orth
    q2, r2 = np.linalg.qr(a.astype(np.float64))
    qn, rn = oracle.sign_normalise(q, r)
    q2n, r2n = oracle.sign_normalise(q2, r2)
    tol = 1e-3 if (reorth or n > 64) else 5e-6           # one sweep, one panel: R comes from the corrected Gram path alone
    assert np.abs(rn - r2n).max() / np.abs(r2n).max() < tol
    # the reference's own arithmetic for this mode (oracle/ref_tsqr.c, REF_FP32_TC_NOCOR) on the same input
    sto, qo, ro_ = oracle.qr(a, oracle.FP32_TC_NOCOR, reorth)
    assert sto == 0
    res_o, orth_o = oracle.residual(a, qo, ro_), oracle.orthogonality_fro(qo)
    assert 1e-4 < res_o < 2e-2 and orth_o < 5e-2, (res_o, orth_o)      # the restatement is at fp16 level too (and not broken)
    assert res <= res_o and orth <= orth_o, (res, res_o, orth, orth_o)  # never worse than the reference's mode
    qon, ron = oracle.sign_normalise(qo, np.triu(ro_))
    dr_o = np.abs(ron - r2n).max() / np.abs(r2n).max()                  # how far the reference's R is from the exact one ...
    assert np.abs(rn - ron).max() / np.abs(ron).max() < dr_o + tol      # ... bounds how far ours may be from the reference's
    dq_o = np.abs(qon - q2n).max()
    assert np.abs(qn - qon).max() < dq_o + 2e-3 * np.abs(q2n).max() + 1e-4
