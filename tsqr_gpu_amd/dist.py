"""Row-partitioned TSQR across the GPUs of one node (new functionality -- the reference
has no multi-GPU path).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Rank p holds the row block A_p:
  1. R_p = fold(A_p)                    local streaming Householder TSQR         (tsqr_mi_local_r_f32)
  2. all_gather(R_p)                    the ONLY exchange: n*n floats per rank (16 KiB at n = 64, latency-bound)
  3. R   = fold([R_0; ...; R_{P-1}])    every rank folds the same stack -> R is bitwise identical on all ranks
  4. Q_p = A_p * inverse(R)             local                                    (tsqr_mi_apply_rinv_f32)
Reorthogonalize=true repeats 1-4 on Q and sets R <- R2 * R.

The arithmetic is behind an `engine` object so that the exchange logic can be exercised on CPU with the gloo
backend and a test double (tests/test_dist_cpu.py); the product engine is HipEngine (C ABI, no CPU fallback).
"""
import torch
import torch.distributed as dist

from . import blockqr as bq


class HipEngine:
    """The product engine: staged C-ABI entry points of libtsqr_mi.so on the current HIP stream."""

    def __init__(self, mode, m_local, n, world_size):
        assert n <= 64, "the row-partitioned path factors one 64-wide panel"
        self.mode = bq.compute_mode(mode)
        self.n = n
        rows = max(m_local, world_size * n)
        self.wq = torch.empty(max(bq.get_working_q_size(rows, n), 1), dtype=torch.float32, device="cuda")
        self.wr = torch.empty(max(bq.get_working_r_size(rows, n), 1), dtype=torch.float32, device="cuda")

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
    for sweep in range(2 if reorthogonalize else 1):
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
        if sweep == 0:
            r.copy_(r_new)
        else:
            engine.rmul(r, r_new)
        src, ld_src = q, ldq
    return bq.success_factorization
