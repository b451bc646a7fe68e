"""Harness layer: on-device validation metrics, latms / condition-number generator and the CSV drivers,
checked against the CPU oracle's metric definitions.  Reference: src/validation.cu, src/latms.cu, src/test.cu, src/test_cond.cu."""
import io
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from tsqr_gpu_amd import blockqr as bq, harness
    from oracle import ref_oracle as oracle
    assert torch.cuda.is_available()
    return torch, bq, harness, oracle


def _cm(torch, x):
    """numpy (m, n) matrix -> device tensor holding it column-major."""
    return torch.from_numpy(np.ascontiguousarray(x.T)).cuda()


@pytest.mark.parametrize("m,n", [(9211, 51), (5000, 130), (4096, 64), (777, 16)])
def test_validation_metrics_match_cpu_definitions(env, m, n):
    torch, bq, harness, oracle = env
    a = oracle.uniform_matrix(m, n, seed=11)
    d_a = _cm(torch, a)
    d_a0 = d_a.clone()
    st, d_q, d_r = harness.qr(d_a, m, n, bq.compute_mode.fp32_tc_cor, False)
    assert st == 0
    q = d_q.cpu().numpy().T.astype(np.float64)
    r = d_r.cpu().numpy().T.astype(np.float64)
    e = q.T @ q - np.eye(n)
    fro2 = float((e * e).sum())
    diag2 = float((np.diag(e) ** 2).sum())
    assert harness.check_orthogonality16(d_q, m, n) == pytest.approx(math.sqrt(fro2 / n), rel=1e-6)
    dg, nd = harness.check_orthogonality16_each(d_q, m, n)
    assert dg == pytest.approx(math.sqrt(diag2), rel=1e-6)
    assert nd == pytest.approx(math.sqrt(fro2 - diag2), rel=1e-6)
    assert harness.orthogonality_fro(d_q, m, n) == pytest.approx(math.sqrt(fro2), rel=1e-6)
    res = harness.residual(d_q, d_r, d_a0, m, n)
    ref = math.sqrt(((q @ r - a.astype(np.float64)) ** 2).sum() / (a.astype(np.float64) ** 2).sum())
    assert res == pytest.approx(ref, rel=1e-6)
    if n % 16 == 0:
        sub = harness.check_submatrix_orthogonality(d_q, m, n).numpy()
        nb = n // 16
        want = np.sqrt((e.reshape(nb, 16, nb, 16) ** 2).sum(axis=(1, 3)) / 16)
        np.testing.assert_allclose(sub, want, rtol=1e-6, atol=1e-12)


def test_validation_with_leading_dimension(env):
    torch, bq, harness, oracle = env
    m, n, ld = 3000, 40, 3333
    a = oracle.uniform_matrix(m, n, seed=5)
    buf = torch.zeros(n, ld, dtype=torch.float32, device="cuda")
    buf[:, :m] = _cm(torch, a)
    buf[:, m:] = 7.0                                  # padding must not enter the sums
    q64 = a.astype(np.float64)
    e = q64.T @ q64 - np.eye(n)
    assert harness.orthogonality_fro(buf, m, n, ldq=ld) == pytest.approx(math.sqrt((e * e).sum()), rel=1e-9)


def test_validate_rejects_bad_sizes(env):
    torch, bq, harness, oracle = env
    q = torch.zeros(16, 16, device="cuda")
    with pytest.raises(RuntimeError):
        harness.orthogonality_fro(q, 4, 16)           # n > m


@pytest.mark.parametrize("cond", [1e2, 1e4, 1e6])
def test_cond_generator_hits_target(env, cond):
    torch, bq, harness, oracle = env
    m, n = 4096, 64
    a = harness.get_rand_matrix_with_cond_number(m, n, cond, seed=3)
    got = harness.get_cond(a, m, n)
    assert 0.9 * cond <= got <= 1.1 * cond
    sv = torch.linalg.svdvals(a.double()).cpu().numpy()
    assert 0.9 * math.sqrt(cond) <= sv.max() <= math.sqrt(cond) * (1 + 1e-3)   # sigma in [1/sqrt(c), sqrt(c)], src/test_cond.cu:30-47
    assert sv.min() == pytest.approx(1 / math.sqrt(cond), rel=5e-2)


def test_latms_rank_and_spectrum(env):
    torch, bq, harness, oracle = env
    m, n, rank = 2048, 48, 20
    s = np.linspace(3.0, 1.0, rank)
    a = harness.latms(m, n, rank, s, seed=1)
    sv = torch.linalg.svdvals(a.double()).cpu().numpy()
    np.testing.assert_allclose(sv[:rank], s, rtol=1e-4)
    assert sv[rank:].max() < 1e-5


def test_csv_drivers_schema_and_values(env):
    torch, bq, harness, oracle = env
    out = io.StringIO()
    rows = harness.accuracy([(4096, 64, 1.0), (2000, 100, 1.0)], C=2, mode=bq.compute_mode.fp32_tc_cor, reorth=False, out=out)
    lines = out.getvalue().strip().split("\n")
    assert lines[0] == "m,n,rand_range,type,compute_mode,reorthogonalization,residual,residual_variance,orthogonality,orthogonality_variance"
    assert len(lines) == 3 and lines[1].startswith("4096,64,1,float,fp32_tc_cor,0,")
    for (_, n, rm, rv, om, ov) in rows:
        assert rm < 5e-6 and om < 1e-5 / math.sqrt(n) * 4
    out = io.StringIO()
    rows = harness.speed([(1 << 16, 64, 1.0)], C=4, out=out)
    lines = out.getvalue().strip().split("\n")
    assert lines[0] == "m,n,rand_range,type,compute_mode,reorthogonalization,elapsed_time,tflops,working_memory_size"
    f = lines[1].split(",")
    assert len(f) == 9 and float(f[6]) > 0 and float(f[7]) > 0 and int(f[8]) > 0
    out = io.StringIO()
    rows = harness.accuracy_cond([(4096, 64, 1e4)], C=2, mode=bq.compute_mode.fp32_tc_cor, reorth=True, out=out)
    lines = out.getvalue().strip().split("\n")
    assert lines[0] == "m,n,cond,type,compute_mode,reorthogonalization,residual,residual_deviation,orthogonality,orthogonality_deviation"
    assert lines[1].startswith("4096,64,10000,float,fp32_tc_cor,1,")
    assert rows[0][3] < 5e-6 and rows[0][5] < 2e-6         # BCGS2 restores orthogonality at cond 1e4


def test_csv_drivers_fp16_modes(env):
    """the half-typed lines of the reference's sweep (src/main.cu:15-16, 38-39, 65-66: type column `half`)"""
    torch, bq, harness, oracle = env
    out = io.StringIO()
    rows = harness.accuracy([(4096, 64, 1.0)], C=2, mode=bq.compute_mode.fp16_tc_nocor, reorth=False, out=out)
    rows += harness.accuracy([(2000, 100, 1.0)], C=2, mode=bq.compute_mode.fp16_notc, reorth=True, out=out, head=False)
    lines = out.getvalue().strip().split("\n")
    assert lines[1].startswith("4096,64,1,half,fp16_tc_nocor,0,") and lines[2].startswith("2000,100,1,half,fp16_notc,1,")
    for (_, n, rm, rv, om, ov) in rows:
        assert rm < 1e-3 and om < 5e-3 / math.sqrt(n)       # the fp16 rounding of Q and R
    out = io.StringIO()
    harness.speed([(1 << 16, 64, 1.0)], C=4, mode=bq.compute_mode.fp16_tc_nocor, out=out)
    f = out.getvalue().strip().split("\n")[1].split(",")
    assert f[3] == "half" and f[4] == "fp16_tc_nocor" and float(f[6]) > 0 and int(f[8]) > 0
    out = io.StringIO()
    rows = harness.accuracy_cond([(4096, 64, 1e2)], C=2, mode=bq.compute_mode.fp16_notc, reorth=True, out=out)
    assert out.getvalue().strip().split("\n")[1].startswith("4096,64,100,half,fp16_notc,1,")
    assert rows[0][3] < 1e-3 and rows[0][5] < 5e-3 / 8


def test_rocsolver_comparison_columns(env):
    """The vendor-library lines of the reference's sweep (src/test.cu:366-593, cuSOLVER there, rocSOLVER here)."""
    torch, bq, harness, oracle = env
    out = io.StringIO()
    rows = harness.rocsolver_accuracy([(4096, 64, 1.0)], C=2, dtype=torch.float32, out=out)
    rows += harness.rocsolver_accuracy([(4096, 64, 1.0)], C=2, dtype=torch.float64, out=out, head=False)
    lines = out.getvalue().strip().split("\n")
    assert lines[1].startswith("4096,64,1,float,rocsolver,0,") and lines[2].startswith("4096,64,1,double,rocsolver,0,")
    assert rows[0][2] < 2e-6 and rows[0][4] < 2e-6            # fp32 Householder
    assert rows[1][2] < 1e-14 and rows[1][4] < 1e-14          # fp64
    out = io.StringIO()
    rows = harness.rocsolver_speed([(1 << 16, 64, 1.0)], C=2, out=out)
    assert out.getvalue().split("\n")[1].startswith("65536,64,1,float,rocsolver,0,") and rows[0][2] > 0
