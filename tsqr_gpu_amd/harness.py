"""Accuracy / speed harness mirroring the reference's L3 driver layer (rows f1-f3).

Same names, protocols, metric definitions and CSV schemas as the reference so that its plot scripts
(scripts/*/mk_*.py) read the output unchanged:

  validation   check_orthogonality16, check_orthogonality16_each, check_submatrix_orthogonality   src/validation.cu:43-181
  latms        latms, get_cond                                                                      src/latms.cu:8-170
  test_qr      accuracy, speed, accuracy_cond, get_rand_matrix_with_cond_number                     src/test.cu:81-343, src/test_cond.cu:20-248

Metrics are evaluated on the device in fp64 (tsqr_mi_validate_f32); inputs are generated on the device with torch's
generators (seeded -> reproducible, unlike the reference's std::random_device).  Harness code, not the hot path.
"""
import ctypes
import math
import sys
import time

import torch

from . import blockqr as bq


# ---- validation.cu --------------------------------------------------------------------------------------------------
def _validate(q, ldq, m, n, r=None, ldr=0, a=None, lda=0, want_gram=False):
    scratch = torch.empty(n * n + 8, dtype=torch.float64, device=q.device)
    out = (ctypes.c_double * 5)()
    gram = (ctypes.c_double * (n * n))() if want_gram else None
    st = bq.lib().tsqr_mi_validate_f32(q.data_ptr(), ldq, 0 if r is None else r.data_ptr(), ldr,
                                       0 if a is None else a.data_ptr(), lda, m, n, scratch.data_ptr(), out,
                                       gram, torch.cuda.current_stream().cuda_stream)
    if st != 0:
        raise RuntimeError("tsqr_mi_validate_f32 -> %d %s" % (st, bq.last_error()))
    return list(out), gram


def check_orthogonality16(q, m, n, ldq=None):
    """sqrt(||Q^T Q - I||_F^2 / n) in fp64 -- reference src/validation.cu:43-80."""
    out, _ = _validate(q, ldq or m, m, n)
    return math.sqrt(out[0] / n)


def check_orthogonality16_each(q, m, n, ldq=None):
    """(diag, non_diag) = sqrt of the diagonal / off-diagonal parts of ||Q^T Q - I||_F^2 -- src/validation.cu:86-127."""
    out, _ = _validate(q, ldq or m, m, n)
    return math.sqrt(out[1]), math.sqrt(out[2])


def check_submatrix_orthogonality(q, m, n, ldq=None):
    """sqrt(sum over each 16 x 16 block of (Q^T Q - I)^2 / 16) as an (n/16) x (n/16) map -- src/validation.cu:133-181."""
    _, gram = _validate(q, ldq or m, m, n, want_gram=True)
    g = torch.tensor(list(gram), dtype=torch.float64).reshape(n, n).T - torch.eye(n, dtype=torch.float64)
    nb = n // 16
    blk = g[: nb * 16, : nb * 16].reshape(nb, 16, nb, 16)
    return torch.sqrt((blk * blk).sum(dim=(1, 3)) / 16)


def orthogonality_fro(q, m, n, ldq=None):
    """||Q^T Q - I||_F, the un-normalised quantity BASELINE.json names."""
    out, _ = _validate(q, ldq or m, m, n)
    return math.sqrt(out[0])


def residual(q, r, a, m, n, ldq=None, ldr=None, lda=None):
    """sqrt(sum((QR-A)^2) / sum(A^2)) -- reference src/test.cu:147-165 (fp64 accumulation here)."""
    out, _ = _validate(q, ldq or m, m, n, r, ldr or n, a, lda or m)
    return math.sqrt(out[3] / out[4])


# ---- engine wrapper -------------------------------------------------------------------------------------------------
def io_dtype(mode):
    """element type of q, r, a for a compute mode (reference src/tsqr.hpp:36-39): half for the two fp16 I/O modes, float otherwise"""
    return torch.float16 if bq.compute_mode(mode) in bq.FP16_MODES else torch.float32


def io_name(mode):
    return "half" if bq.compute_mode(mode) in bq.FP16_MODES else "float"


def qr(a, m, n, mode, reorth, bf=None):
    """(state, Q, R) of the column-major m x n matrix held by tensor `a` ((n, m) row-major, of the mode's io type); `a` may be
    clobbered for n > 64 (fp32 modes)."""
    q = torch.empty(n, m, dtype=io_dtype(mode), device=a.device)
    r = torch.zeros(n, n, dtype=io_dtype(mode), device=a.device)         # caller pre-zeros R (src/test.cu:129)
    if bf is None:
        bf = bq.buffer(mode, reorth, device=a.device)
        bf.allocate(m, n)
    st = bq.qr(q, m, r, n, a, m, m, n, bf, mode=mode, reorthogonalize=reorth)
    return st, q, r


# ---- latms.cu -------------------------------------------------------------------------------------------------------
def latms(m, n, rank, s, seed=0, device="cuda"):
    """A = orth(randn(m x rank)) diag(s) orth(randn(n x rank))^T -- src/latms.cu:8-121.  The Gaussian factors are
    orthogonalised by this engine itself (fp32_notc, Reorthogonalize=true), which doubles as a self-test."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.randn(rank, m, generator=g, device=device, dtype=torch.float32)      # column-major m x rank
    v = torch.randn(rank, n, generator=g, device=device, dtype=torch.float32)      # column-major n x rank
    _, uq, _ = qr(u, m, rank, bq.compute_mode.fp32_notc, True)
    assert 1 <= rank <= min(m, n)
    _, vq, _ = qr(v, n, rank, bq.compute_mode.fp32_notc, True)
    sd = torch.as_tensor(s, dtype=torch.float64, device=device)
    # column-major A (m x n) stored as (n, m): A^T = V diag(s) U^T
    at = (vq.double().T * sd[None, :]) @ uq.double()
    return at.float().contiguous()


def get_cond(a, m, n):
    """sigma_max / sigma_min of the fp32 matrix (src/latms.cu:128-170 uses cusolver gesvd)."""
    sv = torch.linalg.svdvals(a.double())
    return float(sv.max() / sv.min())


def get_rand_matrix_with_cond_number(m, n, cond_number, seed=0, device="cuda"):
    """src/test_cond.cu:20-76: singular values {1/sqrt(c), 1, n-2 values uniform in (1, sqrt(c))}, sorted descending,
    redrawn until the measured condition number reaches 0.9 * target."""
    assert cond_number >= 1.0 and m >= n
    gen = torch.Generator()
    gen.manual_seed(seed)
    a = None
    for _ in range(32):
        s = torch.empty(n, dtype=torch.float64)
        s[0] = 1.0 / math.sqrt(cond_number)
        s[-1] = 1.0
        if n > 2:
            s[1:-1] = 1.0 + (math.sqrt(cond_number) - 1.0) * torch.rand(n - 2, generator=gen, dtype=torch.float64)
        s = torch.sort(s, descending=True).values
        a = latms(m, n, n, s, seed=seed, device=device)
        if get_cond(a, m, n) / cond_number >= 0.9:
            break
    return a


# ---- test.cu / test_cond.cu -----------------------------------------------------------------------------------------
ACCURACY_HEAD = "m,n,rand_range,type,compute_mode,reorthogonalization,residual,residual_variance,orthogonality,orthogonality_variance"
SPEED_HEAD = "m,n,rand_range,type,compute_mode,reorthogonalization,elapsed_time,tflops,working_memory_size"
ACCURACY_COND_HEAD = "m,n,cond,type,compute_mode,reorthogonalization,residual,residual_deviation,orthogonality,orthogonality_deviation"


def _mean_var(xs):
    mu = sum(xs) / len(xs)
    return mu, sum((x - mu) ** 2 for x in xs) / len(xs)


def reference_flop_formula(m, n):
    """The executed-flop count of the reference's explicit-H algorithm (src/test.cu:311-326), kept only so that the
    'tflops' CSV column stays comparable with the reference's own output (divided by 1024^4 there)."""
    batch = bq.get_batch_size(m)

    def qc(mm, nn):
        return 2 * nn * (mm * mm * nn + mm * mm * mm)

    total = 0
    for i in range((n + 15) // 16):
        ln = min(16, n - 16 * i)
        total += batch * qc(m // batch, ln) + (batch - 1) * qc(2 * ln, ln) + (batch - 1) * 4 * ln ** 3 + 4 * ln * ln * m
        total += 2 * 2 * 16 * 16 * i * m
    return total


def accuracy(matrix_config_list, C=16, mode=bq.compute_mode.fp32_tc_cor, reorth=False, out=sys.stdout, seed=0, head=True):
    """src/test.cu:81-234: per (m, n, rand_range): C matrices U(-r, r), mean / variance of residual and orthogonality."""
    if head:
        print(ACCURACY_HEAD, file=out)
    rows = []
    for (m, n, rr) in matrix_config_list:
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        bf = bq.buffer(mode, reorth)
        bf.allocate(m, n)
        res, orth = [], []
        for _ in range(C):
            a = ((torch.rand(n, m, generator=gen, device="cuda", dtype=torch.float32) * 2 - 1) * rr).to(io_dtype(mode))
            a0 = a.float() if a.dtype != torch.float32 else a.clone()    # the engine may clobber a (n > 64); metrics are taken in fp64 from fp32 copies
            st, q, r = qr(a, m, n, mode, reorth, bf)
            assert st == 0
            res.append(residual(q.float(), r.float(), a0, m, n))
            orth.append(check_orthogonality16(q.float(), m, n))
        (rm, rv), (om, ov) = _mean_var(res), _mean_var(orth)
        line = "%d,%d,%g,%s,%s,%d,%e,%e,%e,%e" % (m, n, rr, io_name(mode), bq.compute_mode(mode).name, int(reorth), rm, rv, om, ov)
        print(line, file=out, flush=True)
        rows.append((m, n, rm, rv, om, ov))
    return rows


def speed(matrix_config_list, C=16, mode=bq.compute_mode.fp32_tc_cor, reorth=False, out=sys.stdout, seed=0, head=True):
    """src/test.cu:257-343: one warm-up call, then wall clock over C blocking calls; 'tflops' uses the reference's own
    executed-flop formula / 1024^4 (not the algorithmic F_QR that bench.py reports)."""
    if head:
        print(SPEED_HEAD, file=out)
    rows = []
    for (m, n, rr) in matrix_config_list:
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        a = ((torch.rand(n, m, generator=gen, device="cuda", dtype=torch.float32) * 2 - 1) * rr).to(io_dtype(mode))
        q = torch.empty(n, m, dtype=io_dtype(mode), device="cuda")
        r = torch.zeros(n, n, dtype=io_dtype(mode), device="cuda")
        bf = bq.buffer(mode, reorth)
        bf.allocate(m, n)
        bq.qr(q, m, r, n, a, m, m, n, bf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(C):
            bq.qr(q, m, r, n, a, m, m, n, bf)                        # like the reference: later calls see a clobbered a for n > 64
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / C
        tf = reference_flop_formula(m, n) / el / 1024.0 ** 4
        print("%d,%d,%g,%s,%s,%d,%e,%e,%d" % (m, n, rr, io_name(mode), bq.compute_mode(mode).name, int(reorth), el, tf,
                                                  bf.get_device_memory_size()), file=out, flush=True)
        rows.append((m, n, el, tf))
    return rows


def accuracy_cond(matrix_config_list, C=8, mode=bq.compute_mode.fp32_tc_cor, reorth=False, out=sys.stdout, seed=0, head=True):
    """src/test_cond.cu:129-248: (m, n, cond) with latms-generated inputs; 'deviation' columns hold the variance like the
    reference does (its column names say deviation)."""
    if head:
        print(ACCURACY_COND_HEAD, file=out)
    rows = []
    for (m, n, cond) in matrix_config_list:
        bf = bq.buffer(mode, reorth)
        bf.allocate(m, n)
        res, orth = [], []
        for c in range(C):
            a = get_rand_matrix_with_cond_number(m, n, float(cond), seed=seed + c).to(io_dtype(mode))
            a0 = a.float() if a.dtype != torch.float32 else a.clone()
            st, q, r = qr(a, m, n, mode, reorth, bf)
            assert st == 0
            res.append(residual(q.float(), r.float(), a0, m, n))
            orth.append(check_orthogonality16(q.float(), m, n))
        (rm, rv), (om, ov) = _mean_var(res), _mean_var(orth)
        print("%d,%d,%g,%s,%s,%d,%e,%e,%e,%e" % (m, n, cond, io_name(mode), bq.compute_mode(mode).name, int(reorth), rm, rv, om, ov),
              file=out, flush=True)
        rows.append((m, n, cond, rm, rv, om, ov))
    return rows


# ---- vendor-library comparison: rocSOLVER geqrf + orgqr in the place cuSOLVER has in the reference (src/test.cu:366-593) ----
class _RocSolver:
    """ctypes binding of the four rocSOLVER entry points the comparison needs (harness only, not the hot path)."""
    _inst = None

    def __init__(self):
        self.rb = ctypes.CDLL("librocblas.so", mode=ctypes.RTLD_GLOBAL)
        self.rs = ctypes.CDLL("librocsolver.so")
        self.handle = ctypes.c_void_p()
        if self.rb.rocblas_create_handle(ctypes.byref(self.handle)) != 0:
            raise RuntimeError("rocblas_create_handle failed")
        vp, ci = ctypes.c_void_p, ctypes.c_int
        for name in ("rocsolver_sgeqrf", "rocsolver_dgeqrf"):
            getattr(self.rs, name).argtypes = [vp, ci, ci, vp, ci, vp]
        for name in ("rocsolver_sorgqr", "rocsolver_dorgqr"):
            getattr(self.rs, name).argtypes = [vp, ci, ci, ci, vp, ci, vp]
        self.rb.rocblas_set_stream.argtypes = [vp, vp]

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def qr(self, a, m, n):
        """a: (n, m) tensor = column-major m x n, float32 or float64; returns (Q as (n, m) tensor, R as (n, n) column-major)."""
        self.rb.rocblas_set_stream(self.handle, torch.cuda.current_stream().cuda_stream)
        w = a.clone()
        tau = torch.empty(n, dtype=a.dtype, device=a.device)
        p = "s" if a.dtype == torch.float32 else "d"
        st = getattr(self.rs, "rocsolver_%sgeqrf" % p)(self.handle, m, n, w.data_ptr(), m, tau.data_ptr())
        if st != 0:
            raise RuntimeError("rocsolver geqrf -> %d" % st)
        r = torch.triu(w[:, :n].T).T.contiguous()            # column-major R: entry (i, j) at r[j, i], i <= j
        st = getattr(self.rs, "rocsolver_%sorgqr" % p)(self.handle, m, n, n, w.data_ptr(), m, tau.data_ptr())
        if st != 0:
            raise RuntimeError("rocsolver orgqr -> %d" % st)
        return w, r


def _metrics_torch(q, r, a, m, n):
    """residual and the reference's orthogonality metric in fp64 with torch (vendor comparison only; any dtype)."""
    q64, r64, a64 = q.double(), r.double(), a.double()
    res = torch.sqrt(((r64 @ q64) - a64).pow(2).sum() / a64.pow(2).sum()).item()       # (QR)^T = R^T Q^T in this storage
    e = q64 @ q64.T - torch.eye(n, dtype=torch.float64, device=q.device)
    return res, math.sqrt(e.pow(2).sum().item() / n)


def rocsolver_accuracy(matrix_config_list, C=16, dtype=torch.float32, out=sys.stdout, seed=0, head=True):
    """src/test.cu:366-491 (cusolver_accuracy<T>) with rocSOLVER; compute_mode column says 'rocsolver'."""
    if head:
        print(ACCURACY_HEAD, file=out)
    rows = []
    rs = _RocSolver.get()
    for (m, n, rr) in matrix_config_list:
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        res, orth = [], []
        for _ in range(C):
            a = ((torch.rand(n, m, generator=gen, device="cuda", dtype=torch.float32) * 2 - 1) * rr).to(dtype)
            q, r = rs.qr(a, m, n)
            x, y = _metrics_torch(q, r, a, m, n)
            res.append(x)
            orth.append(y)
        (rm, rv), (om, ov) = _mean_var(res), _mean_var(orth)
        print("%d,%d,%g,%s,rocsolver,0,%e,%e,%e,%e" % (m, n, rr, "float" if dtype == torch.float32 else "double", rm, rv, om, ov),
              file=out, flush=True)
        rows.append((m, n, rm, rv, om, ov))
    return rows


def rocsolver_speed(matrix_config_list, C=16, dtype=torch.float32, out=sys.stdout, seed=0, head=True):
    """src/test.cu:496-593 (cusolver_speed<T>): geqrf + orgqr, one warm-up, wall clock over C blocking repetitions."""
    if head:
        print(SPEED_HEAD, file=out)
    rows = []
    rs = _RocSolver.get()
    for (m, n, rr) in matrix_config_list:
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        a = ((torch.rand(n, m, generator=gen, device="cuda", dtype=torch.float32) * 2 - 1) * rr).to(dtype)
        rs.qr(a, m, n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(C):
            rs.qr(a, m, n)
            torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / C
        tf = reference_flop_formula(m, n) / el / 1024.0 ** 4
        print("%d,%d,%g,%s,rocsolver,0,%e,%e,%d" % (m, n, rr, "float" if dtype == torch.float32 else "double", el, tf, 0),
              file=out, flush=True)
        rows.append((m, n, el, tf))
    return rows
