"""world_size-2 gloo tests of the row-partitioned TSQR driver (tsqr_gpu_amd/dist.py) on CPU.

The product executes a rank's factorisation as ONE call into the HIP library (no CPU fallback, tsqr_mi_qr_f32_dist[_cb]); what can
run without a GPU is everything around that call: RowPartitionedQR, the TorchCollectives transport (real gloo collectives between two
processes) and the exchange PROTOCOL of the C ladder, restated here by a numpy double that keeps its rules:
  * Gram engine: the all-reduced payload is [Gram entries ..., local row count]; thresholds come from the summed row count, so
    ranks with DIFFERENT block heights take the same accept / reject decision;
  * a rejected Gram level escalates on every rank alike (the verdict is a function of the all-reduced payload only);
  * Householder engine: all-gather of the n x n local R factors in rank order, every rank folds the same stack -> identical R;
  * Reorthogonalize = true: second sweep on Q, R <- R2 * R1.
The same collectives + the real engine are covered on the GPU box by tests/test_gpu_dist.py (two processes, one GPU, gloo)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


from dist_double import NumpyRowBackend  # noqa: E402  (tests/ is on sys.path under pytest's rootdir conftest)


def _worker(rank, world, port, heights, n, reorth, out, engine, reject):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tsqr_gpu_amd import dist as tdist
    rng = np.random.Generator(np.random.MT19937(123))
    m_glob = sum(heights)
    a_glob = rng.uniform(-1, 1, size=(m_glob, n)).astype(np.float32)
    row0 = sum(heights[:rank]); m_local = heights[rank]
    a_loc = a_glob[row0:row0 + m_local]
    a = torch.from_numpy(np.ascontiguousarray(a_loc.T))
    q = torch.zeros(n, m_local)
    r = torch.zeros(n, n)
    backend = NumpyRowBackend(n, tdist.TorchCollectives(), engine, reject)
    drv = tdist.RowPartitionedQR(3, m_local, n, backend=backend)
    st = drv.qr(q, m_local, r, a, m_local, reorthogonalize=reorth)
    rs = [torch.zeros(n, n) for _ in range(world)]
    dist.all_gather(rs, r)
    mmax = max(heights)
    qpad = torch.zeros(n, mmax); qpad[:, :m_local] = q
    qs = [torch.zeros(n, mmax) for _ in range(world)]
    dist.all_gather(qs, qpad)
    if rank == 0:
        qg = np.concatenate([t.numpy().T[:heights[k]] for k, t in enumerate(qs)], axis=0).astype(np.float64)
        rg = r.numpy().T.astype(np.float64)
        out.put({"engine": drv.last_engine, "st": st, "r_same": all(torch.equal(rs[0], t) for t in rs), "world": drv.world,
                 "rows_seen": backend.rows_seen,
                 "res": float(np.linalg.norm(qg @ rg - a_glob) / np.linalg.norm(a_glob)),
                 "orth": float(np.linalg.norm(qg.T @ qg - np.eye(n))),
                 "lower": float(np.abs(np.tril(rg, -1)).max())})
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(heights, n, reorth, engine="gram", reject=()):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, heights, n, reorth, out, engine, reject)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_rank_householder_all_gather():
    res = _run((500, 500), 24, False, engine="householder")
    assert res["st"] == 0 and res["r_same"] and res["world"] == 2
    assert res["res"] < 1e-6 and res["orth"] < 1e-5 and res["lower"] == 0.0


def test_two_rank_reorth_and_short_blocks():
    res = _run((40, 40), 64, True, engine="householder")      # each block has fewer rows than columns: only the global matrix is tall
    assert res["st"] == 0 and res["r_same"]
    assert res["res"] < 1e-6 and res["orth"] < 1e-5


def test_two_rank_gram_engine_unequal_blocks_share_the_row_count():
    """ranks with different block heights: the verdict thresholds must come from the all-reduced row count, not from m_local"""
    res = _run((700, 300), 32, False)
    assert res["st"] == 0 and res["r_same"] and res["engine"] == 3
    assert res["rows_seen"] == [1000.0]                       # one exchange, global rows in the payload
    assert res["res"] < 1e-6 and res["orth"] < 1e-5 and res["lower"] == 0.0


def test_two_rank_escalation_is_identical_on_all_ranks():
    res = _run((600, 600), 32, True, reject=(2,))             # level 2 rejected on every rank -> fp64 level, both sweeps
    assert res["st"] == 0 and res["r_same"] and res["engine"] == 1 and res["orth"] < 1e-5
    assert res["rows_seen"] == [1200.0] * 4
    res = _run((600, 500), 32, False, reject=(2, 1))          # nothing Gram-based works -> Householder all-gather path
    assert res["st"] == 0 and res["r_same"] and res["engine"] == 2 and res["orth"] < 1e-5 and res["res"] < 1e-6


def _worker_heights_change(rank, world, port, n, out):
    """the empty-block check of a per-call m_local is collective.  Only rank 1's block shrinks on the second call; both
    ranks pass m_local to that call (rank 0 its unchanged height) -- every rank posts the check's all-reduce, nothing hangs, and the
    Gram all-reduce that follows pairs with the right collective."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tsqr_gpu_amd import dist as tdist
    rng = np.random.Generator(np.random.MT19937(7 + rank))
    m0 = 400
    backend = NumpyRowBackend(n, tdist.TorchCollectives())
    drv = tdist.RowPartitionedQR(3, m0, n, backend=backend)
    a_full = rng.uniform(-1, 1, size=(m0, n)).astype(np.float32)
    results = []
    for m_local in ((m0, m0), (m0, 150), (m0, 0))[0:3]:
        ml = m_local[rank]
        a = torch.from_numpy(np.ascontiguousarray(a_full[:ml].T)) if ml else torch.zeros(n, 1)
        q = torch.zeros(n, max(ml, 1)); r = torch.zeros(n, n)
        try:
            st = drv.qr(q, max(ml, 1), r, a, max(ml, 1), m_local=ml)
            results.append(("ok", st, float(backend.rows_seen[-1])))
        except ValueError:
            results.append(("empty", None, None))              # raised on EVERY rank when any rank's block is empty
    gathered = [None] * world
    dist.all_gather_object(gathered, results)
    if rank == 0:
        out.put(gathered)
    dist.destroy_process_group()


def test_per_call_block_height_is_a_collective_decision():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_heights_change, args=(r, 2, port, 16, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1]                                    # both ranks saw the same sequence of outcomes
    assert res[0][0] == ("ok", 0, 800.0) and res[0][1] == ("ok", 0, 550.0) and res[0][2] == ("empty", None, None)
