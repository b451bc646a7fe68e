"""GPU tests of chol16_kernel (tsqr_gpu_amd/csrc/tsqr_kernels.hip), the n x n step between the Gram pass and the apply pass:
R = chol(G), Z = inverse(R), accept / reject verdict -- the role of the reference's root tile QR (src/tsqr.cu:1164-1172).
Checked against numpy fp64 Cholesky and inverse for both accumulator layouts the Gram kernels produce, ragged n, the shifted
variant and the reject paths."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def st():
    import torch
    assert torch.cuda.is_available()
    so = os.path.join(ROOT, "tsqr_gpu_amd", "csrc", "libtsqr_selftest.so")
    L = ctypes.CDLL(so)
    L.tsqr_selftest_chol.restype = ctypes.c_float
    L.tsqr_selftest_chol.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + \
        [ctypes.c_int] * 3 + [ctypes.c_double, ctypes.c_int]
    return L, torch


def pack_tiles(g, n, f32_layout):
    """n x n symmetric matrix -> summed Gram tiles in the (tile, reg, lane) accumulator order of the Gram kernels; diagonal tiles
    carry both triangles like an MFMA accumulator does; entries beyond n are zero (the kernels pad with zero columns)."""
    nt = (n + 15) // 16
    gp = np.zeros((16 * nt, 16 * nt))
    gp[:n, :n] = g
    out = []
    lanes = np.arange(64)
    for ti in range(nt):
        for tj in range(ti, nt):
            for reg in range(4):
                rows = (4 * (lanes >> 4) + reg) if f32_layout else ((lanes >> 4) + 4 * reg)
                out.append(gp[16 * ti + rows, 16 * tj + (lanes & 15)])
    return np.concatenate(out)


def run(st, g, n, level=1, rows=1 << 20, ldr=None, reps=0):
    """level 2 packs the tiles in the f32 accumulator layout, levels 1 / 3 in the f64 layout (what each Gram kernel writes)"""
    f32_layout = 1 if level == 2 else 0
    L, torch = st
    nt = (n + 15) // 16
    NP = 16 * nt
    ldr = ldr or n
    gs = torch.from_numpy(pack_tiles(g, n, f32_layout)).cuda()
    r = torch.full((n * ldr,), float("nan"), device="cuda")
    z = torch.full((NP * NP,), float("nan"), device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    ms = L.tsqr_selftest_chol(r.data_ptr(), ldr, z.data_ptr(), status.data_ptr(), gs.data_ptr(), n, nt, level, float(rows), reps)
    torch.cuda.synchronize()
    assert ms >= 0
    R = r.cpu().numpy().reshape(n, ldr)[:, :n].T.astype(np.float64)      # column-major -> R[row, col]
    Z = z.cpu().numpy().reshape(NP, NP).T.astype(np.float64)
    s = status.cpu().numpy().view(np.uint32)
    return R, Z, int(s[0]), float(s[1:2].view(np.float32)[0]), float(s[2:3].view(np.float32)[0]), ms


def spd(n, cond, seed):
    rng = np.random.default_rng(seed)
    m = max(4 * n, 256)
    a = rng.standard_normal((m, n))
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    sv = np.geomspace(1.0, 1.0 / cond, n)
    a = (u * sv) @ vt
    a *= rng.uniform(0.5, 2.0, n)                                        # column scaling: S must not care
    return a.T @ a, a


@pytest.mark.parametrize("n", [64, 51, 48, 33, 17, 16, 5, 1])
@pytest.mark.parametrize("level", [1, 2])
def test_matches_numpy_cholesky(st, n, level):
    g, _ = spd(n, 3.0 if level == 2 else 30.0, n)
    R, Z, status, ratio, scond, _ = run(st, g, n, level=level, ldr=n + 3)
    ref = np.linalg.cholesky(g).T
    assert status == 0
    assert np.all(np.tril(R, -1) == 0.0)                                  # exact zeros below the diagonal
    scale = np.abs(ref).max()
    assert np.abs(R - ref).max() <= 2e-7 * scale                          # fp64 arithmetic, fp32 output rounding
    zi = np.linalg.inv(ref)
    assert np.abs(Z[:n, :n] - zi).max() <= 3e-7 * np.abs(zi).max()
    assert np.all(Z[n:, :] == 0.0) and np.all(Z[:, n:] == 0.0) and np.all(np.tril(Z, -1) == 0.0)
    assert np.abs(Z[:n, :n] @ R.astype(np.float32).astype(np.float64) - np.eye(n)).max() < 5e-6
    d = np.sqrt(np.diag(g))
    s_ref = np.sum((d[:, None] * zi) ** 2) / n
    assert abs(scond - s_ref) <= 1e-4 * s_ref
    piv = np.diag(ref) ** 2 / np.diag(g)
    assert abs(ratio - piv.min()) <= 1e-5 * piv.min()


def test_diag_tiles_use_upper_triangle(st):
    """the bf16-split Gram matrix can differ by an ulp between (i,j) and (j,i) of a diagonal tile: only the upper triangle counts"""
    n = 32
    g, _ = spd(n, 10.0, 3)
    g_bad = g.copy()
    il = np.tril_indices(n, -1)
    g_bad[il] *= 1.0 + 1e-3                                               # spoil the strict lower triangle
    # pack_tiles reads diagonal tiles from g_bad (both triangles), off-diagonal tiles from its upper part only
    R, _, status, _, _, _ = run(st, g_bad, n, level=2)
    assert status == 0
    ref = np.linalg.cholesky(g).T
    assert np.abs(R - ref).max() <= 2e-7 * np.abs(ref).max()


def test_verdicts(st):
    n = 64
    g, _ = spd(n, 3.0, 1)
    assert run(st, g, n, level=2)[2] == 0                                 # well conditioned: the bf16 level accepts
    g2, _ = spd(n, 1e4, 2)
    assert run(st, g2, n, level=2)[2] == 1                                # cond 1e4: S far beyond the bound of the bf16 level
    assert run(st, g2, n, level=1)[2] == 0                                # the fp64 level takes it
    g3 = g.copy(); g3[:, 7] = g3[:, 6]; g3[7, :] = g3[6, :]               # exactly dependent columns -> zero pivot
    assert run(st, g3, n, level=1)[2] == 1
    R, Z, status, _, _, _ = run(st, g3, n, level=3)
    assert status == 0 and np.all(np.isfinite(R)) and np.all(np.isfinite(Z))
    g4 = g.copy(); g4[3, 3] = np.nan
    assert run(st, g4, n, level=1)[2] == 1 and run(st, g4, n, level=3)[2] == 1
    g5 = g * 1e-60                                                        # column norms in the fp32 denormal product range
    assert run(st, g5, n, level=2, rows=1 << 20)[2] == 1 and run(st, g5, n, level=1)[2] == 0


def test_shift_value(st):
    n, rows = 48, 1 << 18
    g, _ = spd(n, 5.0, 9)
    coef = 11 * 2.0 ** -53
    R, _, status, _, _, _ = run(st, g, n, level=3, rows=rows)
    s = coef * (rows * n + n * (n + 1)) * np.trace(g)
    ref = np.linalg.cholesky(g + s * np.eye(n)).T
    assert status == 0 and np.abs(R - ref).max() <= 2e-7 * np.abs(ref).max()


def test_timing_report(st):
    g, _ = spd(64, 3.0, 1)
    for n in (16, 32, 48, 64):
        ms = run(st, g[:n, :n], n, level=2, reps=50)[5]
        print("%s n=%d: %.2f us per launch (back to back; diagnostic build with time stamps)" % ("chol16_kernel", n, ms * 1e3))
    assert ms < 0.05


@pytest.mark.parametrize("m,kind", [(1 << 16, "uniform"), (1 << 20, "uniform"), (8192, "uniform"), (1 << 16, "cond30"), (1 << 16, "dependent")])
def test_chained_launch_against_the_launches_it_merges(st, m, kind):
    """gram_blk_chain_kernel (tsqr_mi_qr_f32_loop's chained schedule): its chain role -- 160 reduction workgroups, the last adder factors
    on four waves -- against gram_reduce1_kernel + chol16_kernel on the same partials, its Gram role against gram_blk_kernel; twice in
    a row (the ticket is re-armed by the last adder)."""
    L, torch = st
    L.tsqr_selftest_chain.restype = ctypes.c_int
    L.tsqr_selftest_chain.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    a = torch.rand(64, m, generator=g, device="cuda") * 2 - 1
    if kind == "cond30":                                                  # rejected by the bf16-split level: same verdict words
        a = a * torch.logspace(0, -1.5, 64, device="cuda")[:, None] + a[:1] * 3.0
    elif kind == "dependent":
        a[7] = a[3]
    nparts = min(m // 128, 512)
    r3 = torch.full((3, 64, 64), float("nan"), device="cuda")
    z3 = torch.full((3, 64, 64), float("nan"), device="cuda")
    st3 = torch.full((3, 16), 77, dtype=torch.int32, device="cuda")
    scratch = torch.zeros(2 * nparts * 2560 + 2 * 2568 + 2, dtype=torch.float64, device="cuda")
    eq = ctypes.c_int(-1)
    rc = L.tsqr_selftest_chain(a.data_ptr(), m, m, nparts, r3.data_ptr(), z3.data_ptr(), st3.data_ptr(), scratch.data_ptr(), ctypes.byref(eq))
    assert rc == 0 and eq.value == 1                                      # Gram role: the partials of gram_blk_kernel, bit for bit
    r3, z3, st3 = r3.cpu().numpy(), z3.cpu().numpy(), st3.cpu().numpy()
    assert np.array_equal(st3[0, :1], st3[1, :1]) and np.array_equal(st3[1, :3], st3[2, :3])
    assert np.array_equal(r3[1], r3[2], equal_nan=True) and np.array_equal(z3[1], z3[2], equal_nan=True)
    if kind != "dependent":
        # same reduction order, same elimination order: four waves and sixteen agree to the last bit on finite data
        assert np.array_equal(r3[0], r3[1]) and np.array_equal(z3[0], z3[1])
        assert np.array_equal(st3[0, :2], st3[1, :2])
        assert abs(float(st3[0, 2:3].view(np.float32)[0]) - float(st3[1, 2:3].view(np.float32)[0])) <= 1e-5 * abs(float(st3[0, 2:3].view(np.float32)[0]))
    assert st3[0, 0] == (0 if kind == "uniform" else 1)


@pytest.mark.parametrize("m,kind", [(1 << 16, "uniform"), (1 << 20, "uniform"), (64 * 300, "uniform"), (1 << 16, "correlated")])
def test_wide_chained_launch_against_the_launches_it_merges(st, m, kind):
    """gram_wide_chain_kernel (a stream of 128-column calls): its chain role -- the two-block factorisation on four waves of workgroup 0,
    chol_wide4_body -- against chol_wide_kernel on the same summed tiles (R, the 128 x 128 Z, the verdict), its Gram role against
    gram_wide_kernel (partials bit for bit)."""
    L, torch = st
    L.tsqr_selftest_wide_chain.restype = ctypes.c_int
    L.tsqr_selftest_wide_chain.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    a = torch.rand(128, m, generator=g, device="cuda") * 2 - 1
    if kind == "correlated":                                              # rejected: the verdict words agree as well
        a = a * 0.05 + a[:1]
    nwg = min(m // 64, 256)
    r3 = torch.full((3, 128, 128), float("nan"), device="cuda")
    zw3 = torch.full((3, 128, 128), float("nan"), device="cuda")
    st3 = torch.full((3, 16), 77, dtype=torch.int32, device="cuda")
    scratch = torch.zeros(2 * nwg * 36 * 256 + 36 * 256 + 16 + (3 * 2 * 4096) // 2 + 64, dtype=torch.float64, device="cuda")
    eq = ctypes.c_int(-1)
    rc = L.tsqr_selftest_wide_chain(a.data_ptr(), m, m, nwg, r3.data_ptr(), zw3.data_ptr(), st3.data_ptr(), scratch.data_ptr(), ctypes.byref(eq))
    assert rc == 0 and eq.value == 1
    r3, zw3, st3 = r3.cpu().numpy(), zw3.cpu().numpy(), st3.cpu().numpy()
    assert st3[0, 0] == st3[1, 0] == (0 if kind == "uniform" else 1)
    if kind == "uniform":
        assert np.array_equal(r3[0], r3[1]) and np.array_equal(zw3[0], zw3[1])
        assert st3[0, 1] == st3[1, 1]                                     # smallest pivot ratio: the same bits
        s0, s1 = float(st3[0, 2:3].view(np.float32)[0]), float(st3[1, 2:3].view(np.float32)[0])
        assert abs(s0 - s1) <= 1e-5 * abs(s0)                             # (S is added up in another order)
