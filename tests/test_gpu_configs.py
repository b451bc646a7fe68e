"""BASELINE.json configs C1-C5 at their full sizes, checked through size-independent properties evaluated on the device in
fp64 (tsqr_mi_validate_f32): residual, ||Q^T Q - I||_F, R upper triangular.  C2 lives in test_gpu_parity.py::test_full_size_properties;
C4's 8-GPU partitioning is covered by the gloo tests (CPU) and by the row-partitioned driver on one rank here."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import torch
    from tsqr_gpu_amd import blockqr as bq, harness
    from oracle import ref_oracle as oracle
    assert torch.cuda.is_available()
    return torch, bq, harness, oracle


def test_c1_128x16_against_host_lapack(env):
    """C1: M=128, N=16 -- host LAPACK sgeqrf/sorgqr is the reference (seeds the residual / orthogonality plumbing)."""
    torch, bq, harness, oracle = env
    from scipy.linalg import lapack
    m, n = 128, 16
    a = oracle.uniform_matrix(m, n, seed=0)
    qr_, tau, _, info = lapack.sgeqrf(np.asfortranarray(a))
    assert info == 0
    q_l, _, info = lapack.sorgqr(qr_[:, :n].copy(order="F"), tau)
    r_l = np.triu(qr_[:n, :n])
    ql, rl = oracle.sign_normalise(q_l, r_l)
    assert oracle.residual(a, q_l, r_l) < 5e-7 and oracle.orthogonality_fro(q_l) < 2e-6      # the plumbing itself
    for mode in (bq.compute_mode.fp32_notc, bq.compute_mode.fp32_tc_cor):
        d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
        st, d_q, d_r = harness.qr(d_a, m, n, mode, False)
        assert st == 0
        q, r = d_q.cpu().numpy().T, d_r.cpu().numpy().T
        qn, rn = oracle.sign_normalise(q, r)
        assert np.abs(rn - rl).max() / np.abs(rl).max() < 5e-6 and np.abs(qn - ql).max() < 5e-6
        assert harness.residual(d_q, d_r, d_a, m, n) < 5e-7 and harness.orthogonality_fro(d_q, m, n) < 2e-6


@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor"])
def test_c2_2pow20_x_64_against_the_oracle_at_full_size(env, mode):
    """C2 (the headline config, M=2^20, N=64) against the reference restatement run on the SAME full-size matrix (about 15 s of
    host time per mode): sign-normalised R within the parity band, Q compared on a sample of rows, metrics of both printed side
    by side.  (Parity unpinned: bands, not bits -- DESIGN.md section 7.)"""
    torch, bq, harness, oracle = env
    m, n = 1 << 20, 64
    md = bq.compute_mode[mode]
    a = oracle.uniform_matrix(m, n, seed=2)
    d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
    st, d_q, d_r = harness.qr(d_a, m, n, md, False)
    assert st == 0 and bq.last_engine() == 3
    orth = harness.orthogonality_fro(d_q, m, n); res = harness.residual(d_q, d_r, d_a, m, n)
    assert orth < 1e-5 and res < 5e-7
    st_o, q_o, r_o = oracle.qr(a, int(md), False)
    assert st_o == 0
    r = d_r.cpu().numpy().T.astype(np.float64)
    ro = np.triu(r_o).astype(np.float64)
    s = np.sign(np.diag(r)); so = np.sign(np.diag(ro))
    rn, ron = s[:, None] * r, so[:, None] * ro
    dr = np.abs(rn - ron).max() / np.abs(ron).max()
    rows = np.linspace(0, m - 1, 4096).astype(np.int64)
    q_s = d_q[:, torch.from_numpy(rows).cuda()].cpu().numpy().T.astype(np.float64) * s[None, :]
    dq = np.abs(q_s - q_o[rows].astype(np.float64) * so[None, :]).max()
    print("C2 %s: GPU orth %.2e res %.2e | oracle orth %.2e res %.2e | |dR|/|R| %.2e |dQ| %.2e" % (
        mode, orth, res, oracle.orthogonality_fro(q_o), oracle.residual(a, q_o, ro.astype(np.float32)), dr, dq))
    assert dr < 2e-5 and dq < 2e-5


@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor"])
def test_c3_2pow20_x_128(env, mode):
    """C3: M=2^20, N=128 (auto policy: all 128 columns as one Cholesky-QR panel; tests/test_gpu_wide.py also runs the 64-column
    panel path -- block modified Gram-Schmidt on the matrix cores -- on the same matrix)."""
    torch, bq, harness, oracle = env
    m, n = 1 << 20, 128
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    d_a = torch.rand(n, m, generator=g, device="cuda", dtype=torch.float32) * 2 - 1
    d_a0 = d_a.clone()                                                 # a is clobbered for n > 64
    st, d_q, d_r = harness.qr(d_a, m, n, bq.compute_mode[mode], False)
    assert st == 0
    assert harness.orthogonality_fro(d_q, m, n) < 1e-5
    assert harness.residual(d_q, d_r, d_a0, m, n) < 5e-7
    assert torch.tril(d_r.T, -1).abs().max().item() == 0.0


def test_c4_2pow23_x_64_row_partitioned_driver_one_rank(env):
    """C4's global shape (2^23 x 64) through the row-partitioned driver (one rank: no collective, same kernels and staging)."""
    torch, bq, harness, oracle = env
    from tsqr_gpu_amd import dist as tdist
    m, n = 1 << 23, 64
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    d_a = torch.rand(n, m, generator=g, device="cuda", dtype=torch.float32) * 2 - 1
    d_q = torch.empty(n, m, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
    eng = tdist.RowPartitionedQR(bq.compute_mode.fp32_tc_cor, m, n)
    assert eng.qr(d_q, m, d_r, d_a, m) == 0
    torch.cuda.synchronize()
    assert eng.last_engine == 3
    assert harness.orthogonality_fro(d_q, m, n) < 1e-5
    assert harness.residual(d_q, d_r, d_a, m, n) < 5e-7


@pytest.mark.parametrize("spectrum", ["test_cond", "geometric"])
@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor"])
def test_c5_latms_cond_1e8_reorth(env, mode, spectrum):
    """C5: latms-generated 2^20 x 64 with cond 1e8 (numerically rank-deficient in fp32), Reorthogonalize = true.  Singular values as
    the reference draws them (src/test_cond.cu:31-50: one 1/sqrt(cond), one 1, the rest U(1, sqrt(cond)) -- SURVEY 8d's default)
    and the geometric spectrum (SURVEY 8d's second case)."""
    torch, bq, harness, oracle = env
    m, n = 1 << 20, 64
    if spectrum == "test_cond":
        d_a = harness.get_rand_matrix_with_cond_number(m, n, 1e8, seed=5)
    else:
        s = torch.logspace(0, -8, n, dtype=torch.float64)
        d_a = harness.latms(m, n, n, s, seed=5)
    st, d_q, d_r = harness.qr(d_a, m, n, bq.compute_mode[mode], True)
    assert st == 0
    assert bq.last_engine() in (1, 2, 4)                               # never the bf16-split level on such input
    assert harness.orthogonality_fro(d_q, m, n) < 1e-5                 # O(eps) after the second sweep
    assert harness.residual(d_q, d_r, d_a, m, n) < 2e-6
    assert torch.tril(d_r.T, -1).abs().max().item() == 0.0


@pytest.mark.parametrize("mode,policy,want", [("fp32_tc_cor", 0, 3), ("fp32_notc", 0, 1), ("fp32_tc_cor", 1, 0)])
@pytest.mark.parametrize("reorth", [0, 1])
def test_cpp_dist_entry_point_over_rccl_single_rank(env, mode, policy, want, reorth):
    """tsqr_mi_qr_f32_dist_fn (the one-call driver over an ncclComm_t) on a one-rank RCCL communicator created through ctypes; the
    ncclAllReduce / ncclAllGather entry points handed to the C side come from the SAME library handle as the communicator (whichever
    librccl copy that name resolves to).  Exercises the speculative apply and the Householder stack."""
    import ctypes
    torch, bq, harness, oracle = env
    try:
        rccl = ctypes.CDLL("librccl.so")
    except OSError:
        pytest.skip("librccl.so not loadable by name")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_byte * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        m, n = 50000, 64
        a = oracle.uniform_matrix(m, n, seed=17)
        d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
        d_q = torch.empty(n, m, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
        wq = torch.empty(bq.lib().tsqr_mi_working_q_size_dist(m, n, 1), device="cuda")
        wr = torch.empty(bq.lib().tsqr_mi_working_r_size_dist(m, n, 1), device="cuda")
        gather = torch.empty(n * n, device="cuda")
        bq.set_policy(policy)
        try:
            st = bq.lib().tsqr_mi_qr_f32_dist_fn(int(bq.compute_mode[mode]), reorth, d_q.data_ptr(), m, d_r.data_ptr(), n,
                                                 d_a.data_ptr(), m, m, n, wq.data_ptr(), wr.data_ptr(), gather.data_ptr(),
                                                 comm, ctypes.cast(rccl.ncclAllReduce, ctypes.c_void_p),
                                                 ctypes.cast(rccl.ncclAllGather, ctypes.c_void_p), 1, torch.cuda.current_stream().cuda_stream)
        finally:
            bq.set_policy(bq.POLICY_AUTO)
        torch.cuda.synchronize()
        assert st == 0, bq.last_error()
        assert harness.orthogonality_fro(d_q, m, n) < 5e-6 and harness.residual(d_q, d_r, d_a, m, n) < 5e-7
        assert torch.tril(d_r.T, -1).abs().max().item() == 0.0
        q2, r2 = np.linalg.qr(a.astype(np.float64))
        _, rn = oracle.sign_normalise(d_q.cpu().numpy().T, d_r.cpu().numpy().T)
        _, r2n = oracle.sign_normalise(q2, r2)
        assert np.abs(rn - r2n).max() / np.abs(r2n).max() < 5e-6
        del want
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_cpp_caller_that_links_rccl(env):
    """tests/cpp/sample_dist_rccl.cpp links -lrccl and passes its own ncclComm_t to tsqr_mi_qr_f32_dist: the library must take
    ncclAllReduce / ncclAllGather from the global symbol scope (the caller's copy), for the Gram and the Householder exchange."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "sample_dist_rccl")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s", "sample_dist_rccl"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "DIST SAMPLE OK" in out.stdout, out.stdout + out.stderr
    assert "policy=0 engine=3" in out.stdout and "policy=1 engine=0" in out.stdout


def test_dist_entry_without_entry_points_fails_cleanly(env):
    """no communicator / no entry points: an error code and a message, never a crash or a silently loaded second RCCL"""
    torch, bq, harness, oracle = env
    t = torch.zeros(64 * 64, device="cuda")
    st = bq.lib().tsqr_mi_qr_f32_dist_fn(3, 0, t.data_ptr(), 64, t.data_ptr(), 64, t.data_ptr(), 64, 64, 64, t.data_ptr(), t.data_ptr(),
                                         t.data_ptr(), None, None, None, 1, torch.cuda.current_stream().cuda_stream)
    assert st == 2 and "ncclComm_t" in bq.last_error()


@pytest.mark.parametrize("mode", ["fp32_notc", "fp32_tc_cor"])
def test_row_partitioned_driver_ill_conditioned_shifted_path(env, mode):
    """cond 1e8 through the row-partitioned driver (one rank): both Gram levels reject, the shifted-Cholesky step + one plain
    fp64 sweep finish the first sweep (engine 4), the reorthogonalisation sweep brings ||Q^T Q - I||_F to O(eps)."""
    torch, bq, harness, oracle = env
    from tsqr_gpu_amd import dist as tdist
    m, n = 1 << 17, 64
    s = torch.logspace(0, -8, n, dtype=torch.float64)
    d_a = harness.latms(m, n, n, s, seed=9)
    d_q = torch.empty(n, m, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
    eng = tdist.RowPartitionedQR(bq.compute_mode[mode], m, n)
    assert eng.qr(d_q, m, d_r, d_a, m, reorthogonalize=True) == 0
    torch.cuda.synchronize()
    assert eng.last_engine == 4
    assert harness.orthogonality_fro(d_q, m, n) < 1e-5 and harness.residual(d_q, d_r, d_a, m, n) < 2e-6
    assert torch.tril(d_r.T, -1).abs().max().item() == 0.0


def test_row_partitioned_driver_exactly_dependent_columns(env):
    """Two constant (parallel) columns through the row-partitioned driver: bounded result, residual at rounding level."""
    torch, bq, harness, oracle = env
    from tsqr_gpu_amd import dist as tdist
    m, n = 1 << 15, 64
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    d_a = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
    d_a[0, :] = 1.0; d_a[n // 2, :] = -3.0
    d_q = torch.empty(n, m, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
    eng = tdist.RowPartitionedQR(bq.compute_mode.fp32_tc_cor, m, n)
    assert eng.qr(d_q, m, d_r, d_a, m, reorthogonalize=True) == 0
    torch.cuda.synchronize()
    assert eng.last_engine == 4
    assert harness.residual(d_q, d_r, d_a, m, n) < 2e-6 and harness.orthogonality_fro(d_q, m, n) < 1.01


@pytest.mark.parametrize("m", [1 << 18, 1 << 19, 3 << 18, 65 << 14, (1 << 20) - 16384, (1 << 20) + 64])
@pytest.mark.parametrize("mode", ["fp32_tc_cor", "fp32_notc"])
def test_64_columns_block_shares_of_the_two_passes(env, m, mode):
    """Row counts around the conditions of the uneven block shares (Gram pass: a CU's blocks 20 : 12 between its two workgroups; apply
    pass: 69 : 59 between the CUs of an even and an odd XCD, 18 / 17 / 15 / 14 inside a CU -- only for grids that fill the chip and
    block counts that divide evenly): every block is computed exactly once whatever the split -- residual and orthogonality at
    rounding level, Q finite everywhere, nothing written beyond row m."""
    torch, bq, harness, oracle = env
    n = 64
    g = torch.Generator(device="cuda"); g.manual_seed(m & 0xFFFF)
    ld = m + 64
    d_a = torch.zeros(n, ld, device="cuda")
    d_a[:, :m] = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
    d_q = torch.full((n, ld), float("nan"), device="cuda")
    d_r = torch.zeros(n, n, device="cuda")
    md = bq.compute_mode[mode]
    bf = bq.buffer(md, False); bf.allocate(m, n)
    assert bq.qr(d_q, ld, d_r, n, d_a, ld, m, n, bf) == 0 and bq.last_engine() == 3
    assert torch.isnan(d_q[:, m:]).all() and torch.isfinite(d_q[:, :m]).all()
    q = d_q[:, :m].contiguous(); a = d_a[:, :m].contiguous()
    assert harness.orthogonality_fro(q, m, n) < 5e-6 and harness.residual(q, d_r, a, m, n) < 5e-7
    # and the loop entry (stream of calls) returns the same bits
    d_q2 = torch.full((n, ld), float("nan"), device="cuda"); d_r2 = torch.zeros(n, n, device="cuda")
    assert bq.bind_loop(d_q2, ld, d_r2, n, d_a, ld, m, n, bf)(4) == 0
    assert torch.equal(d_q2[:, :m], d_q[:, :m]) and torch.equal(d_r2, d_r)
