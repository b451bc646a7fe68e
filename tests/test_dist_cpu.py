"""world_size-2 gloo test of the row-partitioned TSQR driver (tsqr_gpu_amd/dist.py) on CPU.

The product engine is the HIP library (no CPU fallback); here the exchange logic -- row partitioning, all_gather of the
R factors, stacking order, R identical on all ranks, reorthogonalisation sweep -- is exercised with a numpy test double
standing in for the two local kernels."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class NumpyEngine:
    """Test double for dist.HipEngine: same interface, LAPACK arithmetic."""

    def __init__(self, n, use_gram=False, reject_levels=()):
        self.n = n
        self.use_gram = use_gram
        self.reject_levels = reject_levels          # Gram levels this double pretends to reject (exercises the escalation)
        self.last_engine = 0
        self._z = None

    def gram(self, level, a, lda, m):
        am = self._cm(a, lda, m, self.n).astype(np.float64)
        return torch.from_numpy((am.T @ am).reshape(-1).copy())

    def chol(self, level, g, m, r):
        if level in self.reject_levels:
            return 1
        if level == 1 and "1-once" in self.reject_levels and not getattr(self, "_l1_rejected", False):
            self._l1_rejected = True                # the fp64 level fails on A but works on Q1 (the shifted-Cholesky situation)
            return 1
        gm = g.numpy().reshape(self.n, self.n)
        rr = np.linalg.cholesky(gm).T
        r.copy_(torch.from_numpy(np.ascontiguousarray(rr.T.astype(np.float32))))
        self._z = np.linalg.inv(rr)
        return 0

    def chol_shifted(self, g, m, r):
        if "s" in self.reject_levels:               # pretend even the shifted factorisation fails (non-finite data on the GPU)
            return 1
        gm = g.numpy().reshape(self.n, self.n)
        rr = np.linalg.cholesky(gm + 1e-7 * np.trace(gm) * np.eye(self.n)).T
        r.copy_(torch.from_numpy(np.ascontiguousarray(rr.T.astype(np.float32))))
        self._z = np.linalg.inv(rr)
        self.shifted_calls = getattr(self, "shifted_calls", 0) + 1
        return 0

    def chol_async(self, level, g, m, r):
        self._pending = self.chol(level, g, m, r)
        if self._pending:                           # a rejected level still leaves *some* Z behind on the GPU; mimic that
            self._z = np.eye(self.n)

    def chol_status(self, m):
        return self._pending

    def apply_z(self, q, ldq, a, lda, m):
        am = self._cm(a, lda, m, self.n).astype(np.float64)
        self._cm(q, ldq, m, self.n)[:] = (am @ self._z).astype(np.float32)

    @staticmethod
    def _cm(t, ld, m, n):          # column-major m x n view of a tensor
        return t.numpy().reshape(-1)[: ld * n].reshape(n, ld)[:, :m].T

    def local_r(self, a, lda, m, r):
        am = self._cm(a, lda, m, self.n).astype(np.float64)
        rr = np.linalg.qr(am, mode="r")
        full = np.zeros((self.n, self.n))
        full[: rr.shape[0], :] = rr
        r.copy_(torch.from_numpy(np.ascontiguousarray(full.T.astype(np.float32))))

    def apply_rinv(self, q, ldq, a, lda, m, r):
        am = self._cm(a, lda, m, self.n).astype(np.float64)
        rm = r.numpy().T.astype(np.float64)
        qm = np.linalg.solve(rm.T, am.T).T
        self._cm(q, ldq, m, self.n)[:] = qm.astype(np.float32)

    def rmul(self, r, r2):
        r.copy_(torch.from_numpy(np.ascontiguousarray((r2.numpy().T.astype(np.float64) @ r.numpy().T.astype(np.float64)).T.astype(np.float32))))

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32)


def _worker(rank, world, port, m_local, n, reorth, out, use_gram=False, reject=()):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tsqr_gpu_amd import dist as tdist
    rng = np.random.Generator(np.random.MT19937(123))
    a_glob = rng.uniform(-1, 1, size=(world * m_local, n)).astype(np.float32)
    a_loc = a_glob[rank * m_local:(rank + 1) * m_local]
    a = torch.from_numpy(np.ascontiguousarray(a_loc.T))
    q = torch.zeros(n, m_local)
    r = torch.zeros(n, n)
    eng = NumpyEngine(n, use_gram, reject)
    st = tdist.qr_dist(q, m_local, r, a, m_local, m_local, n, eng, reorthogonalize=reorth)
    rs = [torch.zeros(n, n) for _ in range(world)]
    dist.all_gather(rs, r)
    qs = [torch.zeros(n, m_local) for _ in range(world)]
    dist.all_gather(qs, q)
    if rank == 0:
        qg = np.concatenate([t.numpy().T for t in qs], axis=0).astype(np.float64)
        rg = r.numpy().T.astype(np.float64)
        out.put({"engine": eng.last_engine, "st": st, "r_same": all(torch.equal(rs[0], t) for t in rs),
                 "res": float(np.linalg.norm(qg @ rg - a_glob) / np.linalg.norm(a_glob)),
                 "orth": float(np.linalg.norm(qg.T @ qg - np.eye(n))),
                 "lower": float(np.abs(np.tril(rg, -1)).max())})
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(m_local, n, reorth, use_gram=False, reject=()):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, m_local, n, reorth, out, use_gram, reject)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_rank_row_partitioned_tsqr():
    res = _run(500, 24, False)
    assert res["st"] == 0 and res["r_same"]
    assert res["res"] < 1e-6 and res["orth"] < 1e-5 and res["lower"] == 0.0


def test_two_rank_reorth_and_short_blocks():
    res = _run(40, 64, True)      # each rank's block has fewer rows than columns: only the global matrix is tall
    assert res["st"] == 0 and res["r_same"]
    assert res["res"] < 1e-6 and res["orth"] < 1e-5


def test_two_rank_gram_engine_and_escalation():
    res = _run(600, 32, False, use_gram=True)                       # Gram level 2 accepted: all-reduce of G
    assert res["st"] == 0 and res["r_same"] and res["engine"] == 3
    assert res["res"] < 1e-6 and res["orth"] < 1e-5 and res["lower"] == 0.0
    res = _run(600, 32, True, use_gram=True, reject=(2,))           # level 2 rejected on every rank -> fp64 level
    assert res["st"] == 0 and res["r_same"] and res["engine"] == 1 and res["orth"] < 1e-5
    res = _run(600, 32, False, use_gram=True, reject=(2, 1, "s"))   # nothing Gram-based works -> Householder all-gather path
    assert res["st"] == 0 and res["r_same"] and res["engine"] == 2 and res["orth"] < 1e-5 and res["res"] < 1e-6
    res = _run(600, 32, False, use_gram=True, reject=(2, 1))        # both rejected, also on Q1 -> two shifted steps
    assert res["st"] == 0 and res["r_same"] and res["engine"] == 4 and res["orth"] < 1e-4 and res["res"] < 1e-6
    res = _run(600, 32, False, use_gram=True, reject=(2, "1-once")) # both rejected on A, fp64 level fine on Q1 -> shifted Cholesky QR
    assert res["st"] == 0 and res["r_same"] and res["engine"] == 4 and res["orth"] < 1e-5 and res["res"] < 1e-6 and res["lower"] == 0.0
