"""Test double for the per-rank executor of the row-partitioned driver (tsqr_gpu_amd/dist.py: HipBackend).

Same `qr_dist` signature, LAPACK (numpy) arithmetic on CPU tensors, and the exchange PROTOCOL of the C ladder
(tsqr_gpu_amd/csrc/tsqr_mi.hip: gram_g / householder_r / qr_core):
  * Gram engine: the all-reduced payload is [Gram entries ..., local row count]; thresholds come from the summed row count;
  * a rejected Gram level escalates on every rank alike; Householder engine: all-gather of the n x n local R factors in rank order;
  * Reorthogonalize = true: second sweep on Q, R <- R2 * R1.
Test infrastructure only: used by tests/test_dist_cpu.py (world-size-2 gloo on CPU) and by `bench.py --rehearse` (launcher /
collectives / JSON plumbing without a GPU -- its numbers say nothing about the product and are labelled so)."""
import numpy as np
import torch


class NumpyRowBackend:
    """Test double for dist.HipBackend: same qr_dist signature, LAPACK arithmetic, the C ladder's exchange protocol."""

    def __init__(self, n, coll, engine="gram", reject_levels=()):
        self.n, self.coll = n, coll
        self.engine = engine                        # "gram" (auto ladder) or "householder" (policy 1)
        self.reject_levels = set(reject_levels)     # Gram levels this double pretends to reject (exercises the escalation)
        self.last_engine = 0
        self.rows_seen = []                         # global row counts read from the all-reduced payloads

    @staticmethod
    def _cm(t, ld, m, n):          # column-major m x n view of a tensor
        return t.numpy().reshape(-1)[: ld * n].reshape(n, ld)[:, :m].T

    def _gram_level(self, am, level):
        n = self.n
        payload = torch.zeros(n * n + 1, dtype=torch.float64)
        payload[: n * n] = torch.from_numpy((am.T @ am).reshape(-1).copy())
        payload[n * n] = float(am.shape[0])                         # the row count travels with the tiles
        self.coll.allreduce_f64(payload)
        g = payload[: n * n].numpy().reshape(n, n)
        rows = float(payload[n * n])
        self.rows_seen.append(rows)
        if level in self.reject_levels:
            return None
        limit = min(128.0, max(4.0, 0.12 * np.sqrt(rows)))          # the bf16 level's S bound: a function of the GLOBAL row count
        try:
            r = np.linalg.cholesky(g).T
        except np.linalg.LinAlgError:
            return None
        z = np.linalg.inv(r)
        s = float(np.sum((np.sqrt(np.diag(g))[:, None] * z) ** 2) / n)
        if level == 2 and s > limit:
            return None
        return r

    def _householder(self, am):
        n = self.n
        rl = np.zeros((n, n))
        rr = np.linalg.qr(am, mode="r")
        rl[: rr.shape[0], :] = rr
        send = torch.from_numpy(np.ascontiguousarray(rl.T.astype(np.float32)).reshape(-1))     # column-major n x n
        recv = torch.zeros(self.coll.world * n * n, dtype=torch.float32)
        self.coll.allgather_f32(send, recv)
        stack = np.concatenate([recv[k * n * n:(k + 1) * n * n].numpy().reshape(n, n).T for k in range(self.coll.world)], axis=0)
        return np.linalg.qr(stack.astype(np.float64), mode="r")

    def _sweep(self, am):
        if self.engine == "gram":
            for level in (2, 1):
                r = self._gram_level(am, level)
                if r is not None:
                    self.last_engine = max(self.last_engine, 3 if level == 2 else 1)
                    return r
            self.last_engine = 2
        return self._householder(am)

    def qr_dist(self, q, ldq, r, ldr, a, lda, m_local, reorth):
        am = self._cm(a, lda, m_local, self.n).astype(np.float64)
        r1 = self._sweep(am)
        d = np.sign(np.diag(r1)); d[d == 0] = 1.0
        r1 = d[:, None] * r1
        qm = np.linalg.solve(r1.T, am.T).T
        if reorth:
            r2 = self._sweep(qm)
            d = np.sign(np.diag(r2)); d[d == 0] = 1.0
            r2 = d[:, None] * r2
            qm = np.linalg.solve(r2.T, qm.T).T
            r1 = r2 @ r1
        self._cm(q, ldq, m_local, self.n)[:] = qm.astype(np.float32)
        r.copy_(torch.from_numpy(np.ascontiguousarray(np.triu(r1).T.astype(np.float32))))
        return 0
