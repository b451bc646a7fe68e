"""Row-partitioned TSQR across the GPUs of one node (new functionality -- the reference
has no multi-GPU path).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rank p holds the row block A_p.

Householder engine (fp32_notc; fallback of fp32_tc_cor):
  1. R_p = fold(A_p)                    local streaming Householder TSQR         (tsqr_mi_local_r_f32)
  2. all_gather(R_p)                    the ONLY exchange: n*n floats per rank (16 KiB at n = 64, latency-bound)
  3. R   = fold([R_0; ...; R_{P-1}])    every rank folds the same stack -> R is bitwise identical on all ranks
  4. Q_p = A_p * inverse(R)             local                                    (tsqr_mi_apply_rinv_f32)
Gram engine (fp32_tc_cor, same acceptance levels as the single-GPU path):
  1. G_p = A_p^T A_p                    local, on the matrix cores               (tsqr_mi_gram_f32)
  2. all_reduce(G_p)                    the ONLY exchange: <= 2560 doubles (20 KiB at n = 64)
  3. R = chol(G), Z = inverse(R)        every rank, identical input -> identical R (tsqr_mi_chol_f32); rejected -> next level
  4. Q_p = A_p * Z                      local                                    (tsqr_mi_apply_z_f32)
Reorthogonalize=true repeats the sweep on Q and sets R <- R2 * R.

The arithmetic is behind an `engine` object so that the exchange logic can be exercised on CPU with the gloo
backend and a test double (tests/test_dist_cpu.py); the product engine is HipEngine (C ABI, no CPU fallback).
"""
import torch
import torch.distributed as dist

from . import blockqr as bq


class HipEngine:
    """The product engine: staged C-ABI entry points of libtsqr_mi.so on the current HIP stream."""

    def __init__(self, mode, m_local, n, world_size, use_gram=None):
        assert n <= 64, "the row-partitioned path factors one 64-wide panel"
        self.mode = bq.compute_mode(mode)
        self.n = n
        self.m_local = m_local
        # same default policy as tsqr_mi_qr_f32: bf16-split Gram level, then fp64 Gram, shifted Cholesky QR, Householder TSQR
        self.use_gram = True if use_gram is None else bool(use_gram)
        self.gram_levels = (2, 1)
        self.last_engine = 0
        rows = max(m_local, world_size * n)
        self.wq = torch.empty(max(bq.get_working_q_size(rows, n), 1), dtype=torch.float32, device="cuda")
        self.wr = torch.empty(max(bq.get_working_r_size(rows, n), 1), dtype=torch.float32, device="cuda")
        self._g = torch.empty(bq.lib().tsqr_mi_gram_elems(n), dtype=torch.float64, device="cuda")

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def local_r(self, a, lda, m, r):
        """r (n x n, column-major in an (n, n) tensor) <- R factor of the m x n column-major block `a`."""
        st = bq.lib().tsqr_mi_local_r_f32(r.data_ptr(), self.n, a.data_ptr(), lda, m, self.n,
                                          self.wq.data_ptr(), self.wr.data_ptr(), self._stream())
        if st != 0:
            raise RuntimeError("tsqr_mi_local_r_f32 -> %d %s" % (st, bq.last_error()))

    def apply_rinv(self, q, ldq, a, lda, m, r):
        st = bq.lib().tsqr_mi_apply_rinv_f32(int(self.mode), q.data_ptr(), ldq, a.data_ptr(), lda, r.data_ptr(), self.n,
                                             m, self.n, self.wq.data_ptr(), self._stream())
        if st != 0:
            raise RuntimeError("tsqr_mi_apply_rinv_f32 -> %d %s" % (st, bq.last_error()))

    def rmul(self, r, r2):
        st = bq.lib().tsqr_mi_rmul_f32(r.data_ptr(), self.n, r2.data_ptr(), self.n, self.n, self.wq.data_ptr(), self._stream())
        if st != 0:
            raise RuntimeError("tsqr_mi_rmul_f32 -> %d %s" % (st, bq.last_error()))

    def gram(self, level, a, lda, m):
        """Gram tiles of the local block in MFMA-accumulator order (float64 tensor); level 2 = bf16-split, 1 = fp64."""
        g = self._g
        st = bq.lib().tsqr_mi_gram_f32(level, g.data_ptr(), a.data_ptr(), lda, m, self.n,
                                       self.wq.data_ptr(), self.wr.data_ptr(), self._stream())
        if st != 0:
            raise RuntimeError("tsqr_mi_gram_f32 -> %d %s" % (st, bq.last_error()))
        return g

    def chol(self, level, g, m, r):
        """r <- chol(G); inverse(R) stays in the work buffer for apply_z.  Returns 0 accepted / 1 rejected (blocking)."""
        import ctypes
        status = ctypes.c_uint(0)
        st = bq.lib().tsqr_mi_chol_f32(level, r.data_ptr(), self.n, g.data_ptr(), m, self.n, self.wq.data_ptr(),
                                       ctypes.byref(status), self._stream())
        if st != 0:
            raise RuntimeError("tsqr_mi_chol_f32 -> %d %s" % (st, bq.last_error()))
        return int(status.value)

    def chol_async(self, level, g, m, r):
        """Like chol() but without the stream sync: the verdict is fetched later with chol_status()."""
        st = bq.lib().tsqr_mi_chol_f32(level, r.data_ptr(), self.n, g.data_ptr(), m, self.n, self.wq.data_ptr(), None, self._stream())
        if st != 0:
            raise RuntimeError("tsqr_mi_chol_f32 -> %d %s" % (st, bq.last_error()))

    def chol_status(self, m):
        import ctypes
        status = ctypes.c_uint(0)
        st = bq.lib().tsqr_mi_chol_status(self.wq.data_ptr(), m, self.n, ctypes.byref(status), self._stream())
        if st != 0:
            raise RuntimeError("tsqr_mi_chol_status -> %d %s" % (st, bq.last_error()))
        return int(status.value)

    def chol_shifted(self, g, m, r):
        """r <- chol(G + s I) of the fp64 Gram tiles in g (level 3 of tsqr_mi_chol_f32); inverse(R) stays in the work buffer."""
        import ctypes
        status = ctypes.c_uint(0)
        st = bq.lib().tsqr_mi_chol_f32(3, r.data_ptr(), self.n, g.data_ptr(), m, self.n, self.wq.data_ptr(),
                                       ctypes.byref(status), self._stream())
        if st != 0:
            raise RuntimeError("tsqr_mi_chol_f32(shifted) -> %d %s" % (st, bq.last_error()))
        return int(status.value)

    def wait(self):
        """Block until everything enqueued on the current stream has completed (the single-GPU call is blocking too)."""
        st = bq.lib().tsqr_mi_stream_wait(self._stream())
        if st != 0:
            raise RuntimeError("tsqr_mi_stream_wait -> %d %s" % (st, bq.last_error()))

    def apply_z(self, q, ldq, a, lda, m):
        st = bq.lib().tsqr_mi_apply_z_f32(int(self.mode), q.data_ptr(), ldq, a.data_ptr(), lda, m, self.n,
                                          self.wq.data_ptr(), self._stream())
        if st != 0:
            raise RuntimeError("tsqr_mi_apply_z_f32 -> %d %s" % (st, bq.last_error()))

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device="cuda")


def qr_dist(q, ldq, r, a, lda, m_local, n, engine, reorthogonalize=False, group=None):
    """Row-partitioned QR.  a, q: column-major m_local x n blocks (tensors; q may alias a); r: (n, n) tensor receiving
    the column-major R (identical on every rank).  Collective over `group`.  Returns state_t."""
    if n == 0 or m_local == 0:
        return bq.error_invalid_matrix_size
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world * m_local < n:
        return bq.error_invalid_matrix_size
    src, ld_src = a, lda
    pending = False                                    # asynchronous work enqueued after the last blocking call?
    for sweep in range(2 if reorthogonalize else 1):
        r_new = None
        r_shift = None
        if getattr(engine, "use_gram", False):
            for level in getattr(engine, "gram_levels", (2, 1)):   # bf16-split Gram, then fp64 Gram; every rank takes the same decision
                g = engine.gram(level, src, ld_src, m_local)
                if world > 1:
                    dist.all_reduce(g, group=group)
                # first sweep: factor straight into the caller's r (a rejected level is simply overwritten by the next one)
                r_try = r if sweep == 0 else engine.empty(n, n)
                if hasattr(engine, "chol_async") and q.data_ptr() != src.data_ptr():
                    # the output does not alias the source: enqueue the apply speculatively behind the Cholesky and look at the
                    # verdict afterwards -- one synchronisation per sweep, no idle gap between the two kernels
                    engine.chol_async(level, g, m_local, r_try)
                    engine.apply_z(q, ldq, src, ld_src, m_local)
                    ok = engine.chol_status(m_local) == 0   # (blocking: enqueued behind the apply, so that one is complete too)
                    pending = False
                else:
                    ok = engine.chol(level, g, m_local, r_try) == 0
                    if ok:
                        engine.apply_z(q, ldq, src, ld_src, m_local)
                        pending = True
                if ok:
                    r_new = r_try
                    engine.last_engine = max(getattr(engine, "last_engine", 0), 3 if level == 2 else 1)
                    break
            if r_new is None and hasattr(engine, "chol_shifted"):
                # both Gram levels rejected (cond beyond ~1e6 or rank deficient): shifted Cholesky of the fp64 Gram matrix that is
                # still in g (already all-reduced), Q1 = src * inverse(R1); the rest of the sweep factors Q1 in place (one plain
                # fp64 Gram level, else the Householder engine) and R = R_second * R1
                r_shift = engine.empty(n, n)
                if engine.chol_shifted(g, m_local, r_shift) == 0:
                    engine.apply_z(q, ldq, src, ld_src, m_local)
                    pending = True
                    src, ld_src = q, ldq
                    g = engine.gram(1, q, ldq, m_local)
                    if world > 1:
                        dist.all_reduce(g, group=group)
                    r2 = engine.empty(n, n)
                    if engine.chol(1, g, m_local, r2) == 0 or engine.chol_shifted(g, m_local, r2) == 0:
                        # (second alternative: Q1 still numerically rank deficient -- exactly dependent columns -- a second shifted
                        # step keeps the result bounded, see panel_qr in csrc/tsqr_mi.hip)
                        engine.apply_z(q, ldq, q, ldq, m_local)
                        pending = True
                        r_new = r2
                        engine.last_engine = 4
                else:
                    r_shift = None
            if r_new is None:
                engine.last_engine = 2
        if r_new is None:                              # Householder TSQR engine (on Q1 after a shifted step)
            r_loc = engine.empty(n, n)
            engine.local_r(src, ld_src, m_local, r_loc)
            if world > 1:
                gathered = [engine.empty(n, n) for _ in range(world)]
                dist.all_gather(gathered, r_loc, group=group)
                # column-major (world*n) x n stack: an (n, world*n) row-major tensor whose row j is column j
                stack = torch.cat(gathered, dim=1).contiguous()
                r_new = engine.empty(n, n)
                engine.local_r(stack, world * n, world * n, r_new)
            else:
                r_new = r_loc
            engine.apply_rinv(q, ldq, src, ld_src, m_local, r_new)
            pending = True
        if r_shift is not None:
            engine.rmul(r_shift, r_new)                # R of this sweep = R_second * R1
            r_new = r_shift
            pending = True
        if sweep == 0:
            if r_new is not r:
                r.copy_(r_new)
                pending = True
        else:
            engine.rmul(r, r_new)
            pending = True
        src, ld_src = q, ldq
    if pending and hasattr(engine, "wait"):
        engine.wait()                                  # blocking like mtk::qr::qr: complete on return
    return bq.success_factorization
