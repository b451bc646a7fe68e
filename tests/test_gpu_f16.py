"""GPU tests of the fp16 I/O modes (mtk::qr::qr<fp16_notc | fp16_tc_nocor, Reorth>: io type half, reference src/tsqr.hpp:38-39,
instantiated at src/blockqr.cu:437-449), through the C ABI entry tsqr_mi_qr_f16.

The reference computes these modes IN half (restated in oracle/ref_tsqr.c: REF_FP16_NOTC, REF_FP16_TC_NOCOR); this engine computes
in fp32 with fp16 at the boundary.  Parity is therefore stated from both sides: never less accurate than the reference's own
arithmetic for the mode, R and Q within the sum of the two error levels of it -- and the fp16 outputs are the roundings of what the
reference's fp32 algorithm gives on the same (fp16-valued) data:
  * residual ||A - QR||_F / ||A||_F <= 1e-3 and ||Q^T Q - I||_F <= 5e-3 max(1, n / 100) with Q, R read back as fp16 (the rounding of the outputs:
    relative 2^-11 per entry),
  * every entry of Q and R within one fp16 rounding of the fp32 pipeline's result on the same input,
  * sign-normalised R and Q agree with the oracle's fp32_notc factorisation of the same data to 2e-3 (relative to max|R|)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RES_TOL, ORTH_TOL, PAR_TOL = 1e-3, 5e-3, 2e-3


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def run_f16(bq, torch, a16, mode, reorth, lda_pad=0, ldq_pad=0):
    """a16: (m, n) float16 array -> state, Q (m, n) float16, R (n, n) float16 through mtk::qr::qr of an fp16 mode"""
    m, n = a16.shape
    lda, ldq = m + lda_pad, m + ldq_pad
    buf = np.zeros((n, lda), np.float16)
    buf[:, :m] = a16.T
    d_a = torch.from_numpy(buf).cuda()
    keep = d_a.clone()
    d_q = torch.full((n, ldq), float("nan"), dtype=torch.float16, device="cuda")
    d_r = torch.full((n, n), float("nan"), dtype=torch.float16, device="cuda")
    bf = bq.buffer(mode, reorth)
    bf.allocate(m, n)
    st = bq.qr(d_q, ldq, d_r, n, d_a, lda, m, n, bf)
    assert torch.equal(keep.view(torch.int16), d_a.view(torch.int16)), "the fp16 entry must not modify A"
    q_full = d_q.cpu().numpy()
    if ldq_pad:
        assert np.isnan(q_full[:, m:]).all(), "wrote outside the m x n block of Q"
    return st, q_full[:, :m].T.copy(), d_r.cpu().numpy().T.copy()


def run_f32(bq, torch, a32, mode, reorth):
    m, n = a32.shape
    d_a = torch.from_numpy(np.ascontiguousarray(a32.T)).cuda()
    d_q = torch.empty(n, m, dtype=torch.float32, device="cuda")
    d_r = torch.zeros(n, n, dtype=torch.float32, device="cuda")
    bf = bq.buffer(mode, reorth)
    bf.allocate(m, n)
    st = bq.qr(d_q, m, d_r, n, d_a, m, m, n, bf)
    return st, d_q.cpu().numpy().T.copy(), d_r.cpu().numpy().T.copy()


CASES = [(128, 16), (33, 16), (20, 7), (1000, 64), (4096, 64), (9211, 51), (65536, 64), (4096, 128), (5000, 200)]


@pytest.mark.parametrize("m,n", CASES)
@pytest.mark.parametrize("mode", ["fp16_notc", "fp16_tc_nocor"])
def test_fp16_modes(bq, oracle, torch_cuda, m, n, mode):
    md = bq.compute_mode[mode]
    a16 = oracle.uniform_matrix(m, n, seed=5).astype(np.float16)
    a64 = a16.astype(np.float64)
    st, q, r = run_f16(bq, torch_cuda, a16, md, False, lda_pad=(5 if m % 2 else 0), ldq_pad=(3 if m % 2 else 8))
    assert st == bq.success_factorization
    assert q.dtype == np.float16 and r.dtype == np.float16
    assert np.isfinite(q).all() and np.isfinite(r).all()
    assert np.abs(np.tril(r.astype(np.float32), -1)).max() == 0.0           # exact zeros below the diagonal
    q64, r64 = q.astype(np.float64), r.astype(np.float64)
    assert np.linalg.norm(q64 @ r64 - a64) / np.linalg.norm(a64) < RES_TOL
    assert np.linalg.norm(q64.T @ q64 - np.eye(n)) < ORTH_TOL * max(1.0, n / 100)        # (n^2 entries, each a sum of roundings of Q)
    # the outputs are the fp16 roundings of the fp32 pipeline's result on the same data (same apply engine: fp32_tc_cor / fp32_tc_nocor)
    st32, q32, r32 = run_f32(bq, torch_cuda, a16.astype(np.float32), bq.compute_mode["fp32_tc_cor" if mode == "fp16_notc" else "fp32_tc_nocor"], False)
    assert st32 == 0
    ulp = 2.0 ** -10                                                         # (one unit in the last place of a value below 1; relative otherwise)
    assert np.abs(q.astype(np.float32) - q32).max() <= ulp * max(1.0, np.abs(q32).max())
    assert np.abs(r.astype(np.float32) - r32).max() <= ulp * np.abs(r32).max()
    # and agree with the reference's fp32 algorithm (oracle restatement, fp32_notc) on the same data to fp16 rounding
    st_o, q_o, r_o = oracle.qr(a16.astype(np.float32), int(bq.compute_mode.fp32_notc), False)
    assert st_o == 0
    qn, rn = oracle.sign_normalise(q.astype(np.float32), r.astype(np.float32))
    qon, ron = oracle.sign_normalise(q_o, np.triu(r_o))
    assert np.abs(rn - ron).max() / np.abs(ron).max() < PAR_TOL
    assert np.abs(qn - qon).max() < PAR_TOL * max(1.0, np.linalg.cond(a64) / 10)
    # the reference's OWN arithmetic for this mode (half everywhere: oracle REF_FP16_NOTC / REF_FP16_TC_NOCOR) on the same input:
    # the engine is never less accurate in either metric, and its R / Q lie within that restatement's distance from the exact factors
    st_h, q_h, r_h = oracle.qr(a16.astype(np.float32), int(md), False)
    assert st_h == 0
    res = np.linalg.norm(q64 @ r64 - a64) / np.linalg.norm(a64)
    orth = np.linalg.norm(q64.T @ q64 - np.eye(n))
    res_h, orth_h = oracle.residual(a16.astype(np.float32), q_h, r_h), oracle.orthogonality_fro(q_h)
    assert 5e-4 < res_h < 2e-2 and orth_h < 8e-2, (res_h, orth_h)            # the restatement is at half-arithmetic level (and not broken)
    assert res <= res_h and orth <= orth_h, (res, res_h, orth, orth_h)
    q2, r2 = np.linalg.qr(a64)
    q2n, r2n = oracle.sign_normalise(q2, r2)
    qhn, rhn = oracle.sign_normalise(q_h, np.triu(r_h))
    dr_h = np.abs(rhn - r2n).max() / np.abs(r2n).max()
    assert np.abs(rn - rhn).max() / np.abs(rhn).max() < dr_h + PAR_TOL
    assert np.abs(qn - qhn).max() < np.abs(qhn - q2n).max() + PAR_TOL * max(1.0, np.abs(q2n).max())


@pytest.mark.parametrize("mode", ["fp16_notc", "fp16_tc_nocor"])
def test_fp16_golden_fixtures(bq, oracle, torch_cuda, mode):
    """tests/golden: |R| of the oracle's half-arithmetic restatement and of fp64 LAPACK for the committed seeds"""
    import json, os
    G = os.path.join(os.path.dirname(__file__), "golden")
    for case in json.load(open(os.path.join(G, "golden.json")))["cases"]:
        data = np.load(os.path.join(G, case["file"]))
        a16 = oracle.uniform_matrix(case["m"], case["n"], seed=case["seed"]).astype(np.float16)
        st, q, r = run_f16(bq, torch_cuda, a16, bq.compute_mode[mode], False)
        assert st == 0
        absr = np.abs(r.astype(np.float32))
        assert np.allclose(absr, data["absr_lapack64"], rtol=0, atol=1.5e-3 * absr.max())       # (fp16 rounding of the input and of R)
        far = np.abs(data["absr_" + mode] - data["absr_lapack64"]).max()
        assert np.allclose(absr, data["absr_" + mode], rtol=0, atol=far + 1.5e-3 * absr.max())
        a64, q64, r64 = a16.astype(np.float64), q.astype(np.float64), r.astype(np.float64)
        assert np.linalg.norm(q64 @ r64 - a64) / np.linalg.norm(a64) < case["residual_max"][mode]
        assert np.linalg.norm(q64.T @ q64 - np.eye(case["n"])) < case["orth_fro_max"][mode]


@pytest.mark.parametrize("mode", ["fp16_notc", "fp16_tc_nocor"])
def test_fp16_modes_reorthogonalised_ill_conditioned(bq, oracle, torch_cuda, mode):
    """cond 1e3 (what fp16 data can carry), reorth = true: orthogonality at fp16-rounding level"""
    rng = np.random.default_rng(3)
    m, n = 20000, 64
    u, _ = np.linalg.qr(rng.standard_normal((m, n)))
    v, _ = np.linalg.qr(rng.standard_normal((n, n)))
    a16 = (u * np.logspace(0, -3, n)) @ v.T
    a16 = (a16 / np.abs(a16).max()).astype(np.float16)
    a64 = a16.astype(np.float64)
    st, q, r = run_f16(bq, torch_cuda, a16, bq.compute_mode[mode], True)
    assert st == 0 and np.isfinite(q).all() and np.isfinite(r).all()
    q64, r64 = q.astype(np.float64), r.astype(np.float64)
    # (fp16_tc_nocor: inverse(R) enters the apply pass rounded to fp16, one product, no correction -- its entries span the
    # conditioning, so the residual carries a few roundings more than with the error-corrected engine of fp16_notc)
    assert np.linalg.norm(q64 @ r64 - a64) / np.linalg.norm(a64) < (3 * RES_TOL if mode == "fp16_tc_nocor" else RES_TOL)
    assert np.linalg.norm(q64.T @ q64 - np.eye(n)) < ORTH_TOL


def test_fp16_r_beyond_the_half_range_is_infinite_like_a_half_typed_r(bq, oracle, torch_cuda):
    """column norms above 65504 cannot be stored in a half-typed R (the reference's io type): they come back as infinities, Q is unaffected"""
    m, n = 65536, 16
    a16 = (oracle.uniform_matrix(m, n, seed=9) * 1000.0).astype(np.float16)      # ||a_j|| ~ 1000 sqrt(m / 3) = 1.5e5
    # (fp16_notc: the error-corrected apply engine.  The single-product engine of the *_tc_nocor modes rounds inverse(R) to fp16 without a
    # scale factor: with column norms of 1.5e5 its entries fall into the fp16 subnormal range and Q loses accuracy -- the range
    # limit of fp16 operands)
    st, q, r = run_f16(bq, torch_cuda, a16, bq.compute_mode.fp16_notc, False)
    assert st == 0
    assert np.isinf(np.diag(r.astype(np.float32))).all()
    q64 = q.astype(np.float64)
    assert np.isfinite(q).all() and np.linalg.norm(q64.T @ q64 - np.eye(n)) < ORTH_TOL


def test_fp16_entry_errors(bq, torch_cuda):
    torch = torch_cuda
    m, n = 256, 16
    h = lambda *s: torch.zeros(*s, dtype=torch.float16, device="cuda")
    bf = bq.buffer(bq.compute_mode.fp16_notc, False)
    bf.allocate(m, n)
    assert bq.qr(h(m, n), n, h(m, m), m, h(m, n), n, n, m, bf) == bq.error_invalid_matrix_size          # n > m (reference src/blockqr.cu:409-411)
    with pytest.raises(TypeError):                                                                        # fp32 tensors into an fp16 mode
        bq.qr(torch.zeros(n, m, device="cuda"), m, torch.zeros(n, n, device="cuda"), n, torch.zeros(n, m, device="cuda"), m, m, n, bf)
    bf32 = bq.buffer(bq.compute_mode.fp32_tc_cor, False)
    bf32.allocate(m, n)
    with pytest.raises(RuntimeError):                                                                     # work space of an fp32 mode is too small
        bq.qr(h(n, m), m, h(n, n), n, h(n, m), m, m, n, bf32, mode=bq.compute_mode.fp16_notc)
    L = bq.lib()
    rc = L.tsqr_mi_qr_f16(int(bq.compute_mode.fp32_tc_cor), 0, h(n, m).data_ptr(), m, h(n, n).data_ptr(), n, h(n, m).data_ptr(), m, m, n,
                          bf.dwq.data_ptr(), bf.dwr.data_ptr(), 0, bf.dl.data_ptr(), bf.hl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == bq.error_unsupported_mode                                                                # the fp16 entry takes the two fp16 modes only
    assert L.tsqr_mi_working_q_size_f16(m, n) >= L.tsqr_mi_working_q_size(m, n) + 2 * m * n + n * n


def test_cpp_sample_runs_the_fp16_modes(bq, torch_cuda):
    """tests/cpp/sample_blockqr.cpp instantiates mtk::qr::qr<fp16_tc_nocor, false> and <fp16_notc, true> on half buffers through
    include/tsqr/blockqr.hpp (get_io_type<mode>::type = half_t) next to the fp32 modes"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "tests", "cpp"), "-s", "sample_blockqr"])
    out = subprocess.check_output([os.path.join(root, "tests", "cpp", "sample_blockqr")], text=True)
    lines = [l for l in out.splitlines() if l.startswith("mode=")]
    assert "SAMPLE OK" in out and {l.split()[0] for l in lines} >= {"mode=0", "mode=1", "mode=2", "mode=3"}, out


@pytest.mark.parametrize("mode", ["fp16_tc_nocor", "fp16_notc"])
@pytest.mark.parametrize("m,n,kind", [(1 << 17, 64, "uniform"), (20000, 48, "uniform"), (30000, 64, "cond1e5"), (4096, 12, "uniform")])
def test_loop_entry_streams_the_native_path(bq, oracle, torch_cuda, m, n, kind, mode):
    """tsqr_mi_qr_f16_loop: calls of the native path go out as a stream, two in flight (completion word of call i raised by the Gram
    kernel of call i + 1); a rejected matrix (cond 1e5) and n <= 16 fall back to blocking calls -- the same halves as five blocking calls."""
    torch = torch_cuda
    md = bq.compute_mode[mode]
    a = oracle.uniform_matrix(m, n, seed=3) if kind == "uniform" else oracle.matrix_with_cond(m, n, 1e5, seed=3).astype(np.float32)
    d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda().half()
    res = []
    for depth in (1, 3):
        d_q = torch.full((n, m), float("nan"), dtype=torch.float16, device="cuda")
        d_r = torch.zeros(n, n, dtype=torch.float16, device="cuda")
        bf = bq.buffer(md, False); bf.allocate(m, n)
        bq.set_loop_depth(depth)
        try:
            assert bq.bind_loop(d_q, m, d_r, n, d_a, m, m, n, bf)(5) == 0
            eng = bq.last_engine()
        finally:
            bq.set_loop_depth(3)
        res.append((d_q.cpu().numpy().copy(), d_r.cpu().numpy().copy(), eng))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
    assert np.isfinite(res[1][0].astype(np.float32)).all()


@pytest.mark.parametrize("mode", ["fp16_tc_nocor", "fp16_notc"])
@pytest.mark.parametrize("m,n,k,bad", [(1 << 15, 64, 5, ()), (1 << 15, 64, 5, (2,)), (1 << 15, 64, 4, (0, 3)), (9216, 48, 4, ()), (4096, 128, 3, ())])
def test_batch_of_half_typed_matrices(bq, oracle, torch_cuda, m, n, k, bad, mode):
    """tsqr_mi_qr_f16_batch / mtk::qr::qr_batch for the fp16 I/O modes: K different matrices through one call at loop depths 1 / 2 / 3
    (chained for 2^k x 64, two in flight for other native shapes, blocking for the conversion path) -- bit for bit the K blocking calls,
    also with matrices the bf16-split level rejects in the stream (they take the conversion path and its whole ladder)."""
    torch = torch_cuda
    md = bq.compute_mode[mode]
    mats = [(oracle.matrix_with_cond(m, n, 1e6, seed=70 + i) if i in bad else oracle.uniform_matrix(m, n, seed=70 + i)).astype(np.float16) for i in range(k)]
    want = [run_f16(bq, torch, a, md, False) for a in mats]
    assert all(w[0] == 0 for w in want)
    for depth in (1, 2, 3):
        d_a = [torch.from_numpy(np.ascontiguousarray(a.T)).cuda() for a in mats]
        d_q = [torch.full((n, m), float("nan"), dtype=torch.float16, device="cuda") for _ in mats]
        d_r = [torch.full((n, n), float("nan"), dtype=torch.float16, device="cuda") for _ in mats]
        bf = bq.buffer(md, False); bf.allocate(m, n)
        bq.set_loop_depth(depth)
        try:
            st, states = bq.qr_batch(d_q, m, d_r, n, d_a, m, m, n, bf)
        finally:
            bq.set_loop_depth(3)
        assert st == 0 and states == [0] * k
        for i in range(k):
            assert np.array_equal(d_q[i].cpu().numpy().T.view(np.int16), want[i][1].view(np.int16)), (depth, i)
            assert np.array_equal(d_r[i].cpu().numpy().T.view(np.int16), want[i][2].view(np.int16)), (depth, i)
            assert np.array_equal(d_a[i].cpu().numpy().T, mats[i])               # A untouched
    with pytest.raises(TypeError):                                               # float32 tensors for a half-typed mode
        bf = bq.buffer(md, False); bf.allocate(64, 16)
        x = torch.zeros(16, 64, device="cuda")
        bq.qr_batch([x], 64, [x], 16, [x], 64, 64, 16, bf)
