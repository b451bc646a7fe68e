"""ctypes loader for oracle/libref_oracle.so -- TEST INFRASTRUCTURE ONLY.

Parity unpinned (see the header of ref_tsqr.c): the reference holds no golden
vectors; this restatement is pinned by formula checks and LAPACK agreement in
tests/test_oracle.py.  Importers allowed: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libref_oracle.so")

# position in mtk::qr::compute_mode (reference src/blockqr.hpp:12-23)
FP32_NOTC = 2
FP32_TC_COR = 3
FP32_TC_NOCOR = 4
FP16_NOTC = 0          # half-typed modes: pass / receive float32 arrays holding fp16 values (qr() rounds its input)
FP16_TC_NOCOR = 1


def build(force=False):
    src = os.path.join(_HERE, "ref_tsqr.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libref_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        sz = ctypes.c_size_t
        fp = ctypes.POINTER(ctypes.c_float)
        for name, args in [("ref_get_batch_size_log2", [sz]), ("ref_get_batch_size", [sz]),
                           ("ref_working_q_size", [sz, sz]), ("ref_working_r_size", [sz, sz]),
                           ("ref_working_l_size", [sz])]:
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = args
        L.ref_qr_f32.restype = ctypes.c_int
        L.ref_qr_f32.argtypes = [ctypes.c_int, ctypes.c_int, fp, sz, fp, sz, fp, sz, sz, sz]
        L.ref_h16.restype = ctypes.c_float
        L.ref_h16.argtypes = [ctypes.c_float]
        _lib = L
    return _lib


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def qr(a, mode=FP32_TC_COR, reorth=False):
    """Run the restated mtk::qr::qr on a column-major copy of `a` (m x n float32).

    Returns (state, Q (m x n), R (n x n)).  `a` itself is not modified (the reference
    clobbers its input for n > 16; the clobbered copy is discarded here).
    """
    a = np.asarray(a, dtype=np.float32)
    if int(mode) in (FP16_NOTC, FP16_TC_NOCOR):         # io type half (src/tsqr.hpp:38-39): the input enters as fp16 values
        a = a.astype(np.float16).astype(np.float32)
    m, n = a.shape
    af = np.asfortranarray(a).copy(order="F")
    q = np.zeros((max(m, 1), max(n, 1)), dtype=np.float32, order="F")
    r = np.zeros((max(n, 1), max(n, 1)), dtype=np.float32, order="F")   # caller pre-zeros R (test.cu:129)
    st = lib().ref_qr_f32(int(mode), int(bool(reorth)), _fptr(q), max(m, 1), _fptr(r), max(n, 1),
                          _fptr(af), max(m, 1), m, n)
    return st, q[:m, :n], r[:n, :n]


# ---- metric definitions, exactly as the reference computes them ------------------------------
def residual(a, q, r):
    """sqrt(sum((QR-A)^2) / sum(A^2))  -- reference src/test.cu:147-165 (evaluated in fp64 here)."""
    a64 = np.asarray(a, dtype=np.float64)
    d = np.asarray(q, dtype=np.float64) @ np.asarray(r, dtype=np.float64) - a64
    return float(np.sqrt((d * d).sum() / (a64 * a64).sum()))


def orthogonality_fro(q):
    """||Q^T Q - I||_F in fp64 (the quantity BASELINE.json names)."""
    q64 = np.asarray(q, dtype=np.float64)
    g = q64.T @ q64 - np.eye(q64.shape[1])
    return float(np.sqrt((g * g).sum()))


def orthogonality_ref(q):
    """sqrt(||Q^T Q - I||_F^2 / n) -- reference src/validation.cu:43-80 (check_orthogonality16)."""
    n = np.asarray(q).shape[1]
    return orthogonality_fro(q) / np.sqrt(n)


def sign_normalise(q, r):
    """Flip column signs so diag(R) >= 0 (QR is unique up to these signs for full-rank A;
    the reference itself compares |entries|, src/test_compare.hpp:241,251)."""
    s = np.where(np.diag(r) < 0, -1.0, 1.0).astype(r.dtype)
    return q * s[None, :], r * s[:, None]


# ---- synthetic inputs ---------------------------------------------------------------------------
def uniform_matrix(m, n, seed=0, rand_range=1.0):
    """U(-r, r) column-major fill -- same distribution as src/test.cu:283-287 / test/library_link.cu:28-39
    (std::mt19937 stream itself is not reproduced; numpy's MT19937 with the given seed is used)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    return np.asfortranarray(rng.uniform(-rand_range, rand_range, size=(n, m)).T.astype(np.float32))


def latms(m, n, s, seed=0):
    """A = orth(randn(m x rank)) diag(s) orth(randn(n x rank))^T -- src/latms.cu:8-121."""
    rng = np.random.Generator(np.random.MT19937(seed))
    rank = len(s)
    u, _ = np.linalg.qr(rng.standard_normal((m, rank)))
    v, _ = np.linalg.qr(rng.standard_normal((n, rank)))
    return np.asfortranarray(((u * np.asarray(s, dtype=np.float64)[None, :]) @ v.T).astype(np.float32))


def cond_singular_values(n, cond, seed=0):
    """Singular values as src/test_cond.cu:31-50 draws them: one 1/sqrt(c), one 1, n-2 uniform in (1, sqrt(c)),
    sorted descending."""
    rng = np.random.Generator(np.random.MT19937(seed))
    s = np.empty(n, dtype=np.float64)
    s[0] = 1.0 / np.sqrt(cond)
    s[-1] = 1.0
    if n > 2:
        s[1:-1] = rng.uniform(1.0, np.sqrt(cond), size=n - 2)
    return np.sort(s)[::-1]


def matrix_with_cond(m, n, cond, seed=0, geometric=False):
    """src/test_cond.cu:20-76 (redraw until measured cond >= 0.9 target) or geometric s_i = cond^(-i/(n-1))."""
    if geometric:
        s = cond ** (-np.arange(n) / max(n - 1, 1))
        return latms(m, n, s, seed)
    for t in range(64):
        s = cond_singular_values(n, cond, seed + 1000 * t)
        if s[0] / s[-1] >= 0.9 * cond:
            break
    return latms(m, n, s, seed)
