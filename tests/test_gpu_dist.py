"""Two (and four) processes, ONE GPU, gloo: the row-partitioned driver with the PRODUCT engine (tsqr_mi_qr_f32_dist_cb: the C ladder calling
back into torch.distributed for its two exchanges).  Checks the global factorisation against fp64 LAPACK and the reference
restatement (oracle), unequal block heights, the Householder all-gather branch (policy 1, incl. blocks shorter than the panel),
reorthogonalisation and the escalation of an ill-conditioned input -- i.e. that every rank takes the same branch and ends with
the same R.  (RCCL itself refuses two ranks on one device; the raw-communicator transport is exercised on one rank in
tests/test_gpu_configs.py and on the 8-GPU node by bench.py.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, heights, n, mode, reorth, policy, cond, loop, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tsqr_gpu_amd import blockqr as bq, dist as tdist
    rng = np.random.Generator(np.random.MT19937(7))
    m_glob = sum(heights)
    if cond > 1:
        u, _ = np.linalg.qr(rng.standard_normal((m_glob, n)))
        v, _ = np.linalg.qr(rng.standard_normal((n, n)))
        a_glob = ((u * np.geomspace(1.0, 1.0 / cond, n)) @ v.T).astype(np.float32)
    else:
        a_glob = rng.uniform(-1, 1, size=(m_glob, n)).astype(np.float32)
    row0 = sum(heights[:rank]); m_local = heights[rank]
    d_a = torch.from_numpy(np.ascontiguousarray(a_glob[row0:row0 + m_local].T)).cuda()
    keep = d_a.clone()
    d_q = torch.empty(n, m_local, device="cuda"); d_r = torch.zeros(n, n, device="cuda")
    bq.set_policy(policy)
    drv = tdist.RowPartitionedQR(bq.compute_mode[mode], m_local, n, comm="callbacks")
    if loop:                                                  # `loop` calls from the C loop (tsqr_mi_qr_f32_dist_cb_loop: two calls in flight)
        st = drv.bind_loop(d_q, m_local, d_r, d_a, m_local, reorthogonalize=reorth)(loop)
    else:
        st = drv.qr(d_q, m_local, d_r, d_a, m_local, reorthogonalize=reorth)
    torch.cuda.synchronize()
    a_untouched = bool(torch.equal(keep, d_a))
    r = d_r.cpu()
    rs = [torch.zeros(n, n) for _ in range(world)]
    dist.all_gather(rs, r)
    mmax = max(heights)
    qpad = torch.zeros(n, mmax); qpad[:, :m_local] = d_q.cpu()
    qs = [torch.zeros(n, mmax) for _ in range(world)]
    dist.all_gather(qs, qpad)
    eng = torch.tensor([drv.last_engine]); engs = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(engs, eng)
    if rank == 0:
        qg = np.concatenate([t.numpy().T[:heights[k]] for k, t in enumerate(qs)], axis=0).astype(np.float64)
        rg = r.numpy().T.astype(np.float64)
        out.put({"st": st, "transport": drv.transport, "engines": [int(e.item()) for e in engs], "a_untouched": a_untouched,
                 "r_same": all(torch.equal(rs[0], t) for t in rs), "q": qg, "r": rg, "a": a_glob})
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(heights, n, mode="fp32_tc_cor", reorth=False, policy=0, cond=1.0, loop=0, worker=None):
    import queue
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    world = len(heights)                                      # (at most 4 here: the GPU box allows 6 processes on its card)
    procs = [ctx.Process(target=_guarded, args=(worker or _worker, r, world, port, heights, n, mode, reorth, policy, cond, loop, out)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        # a rank that fails puts its exception text on the queue at once; a rank that dies without a word is noticed by polling
        res = None
        for _ in range(300):
            try:
                res = out.get(timeout=1.0)
                break
            except queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    raise RuntimeError("a rank exited with %s" % [p.exitcode for p in procs])
        assert res is not None, "no result within 300 s"
        assert "error" not in res, res.get("error")
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        return res
    finally:
        # never leave a rank behind on the card (blocked in a gloo collective its partner will not enter any more)
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
                p.join(timeout=10)


def _guarded(worker, rank, *args):
    out = args[-1]
    try:
        worker(rank, *args)
    except BaseException as e:                                # surfaces at once in the parent instead of after its timeout
        import traceback
        out.put({"error": "rank %d: %s\n%s" % (rank, e, traceback.format_exc())})
        raise


def _check(res, oracle, orth_tol=5e-6, r_tol=1e-5):
    a, q, r = res["a"], res["q"], res["r"]
    n = a.shape[1]
    assert res["st"] == 0 and res["r_same"] and res["a_untouched"]
    assert len(set(res["engines"])) == 1                       # every rank ended on the same engine
    assert np.abs(np.tril(r, -1)).max() == 0.0
    assert np.linalg.norm(q @ r - a) / np.linalg.norm(a) < 5e-7
    assert np.linalg.norm(q.T @ q - np.eye(n)) < orth_tol
    q2, r2 = np.linalg.qr(a.astype(np.float64))               # fp64 LAPACK, sign-normalised
    _, rn = oracle.sign_normalise(q, r)
    _, r2n = oracle.sign_normalise(q2, r2)
    assert np.abs(rn - r2n).max() / np.abs(r2n).max() < r_tol


@pytest.mark.parametrize("mode", ["fp32_tc_cor", "fp32_notc"])
def test_two_ranks_gram_engine_unequal_blocks(oracle, mode):
    res = _run((30000, 17777), 64, mode=mode)
    assert res["transport"] == "torch.distributed callbacks" and res["engines"][0] == 3
    _check(res, oracle)
    # the reference restatement on the same global matrix (parity unpinned: bands, not bits)
    _, qo, ro = oracle.qr(res["a"], oracle.FP32_TC_COR if mode == "fp32_tc_cor" else oracle.FP32_NOTC, False)
    _, rn = oracle.sign_normalise(res["q"], res["r"])
    _, ron = oracle.sign_normalise(qo, np.triu(ro))
    assert np.abs(rn - ron).max() / np.abs(ron).max() < 2e-5


@pytest.mark.parametrize("reorth", [False, True])
def test_two_ranks_householder_all_gather(oracle, reorth):
    """policy 1: local folds, all-gather of the two R factors, fold of the 128 x 64 stack on both ranks"""
    res = _run((20000, 12345), 64, reorth=reorth, policy=1)
    assert res["engines"][0] == 0
    _check(res, oracle)


def test_two_ranks_householder_blocks_shorter_than_the_panel(oracle):
    """m_local < n on both ranks: the gathered stack and its fold scratch need the *_size_dist work buffers"""
    res = _run((100, 60), 48, policy=1)
    _check(res, oracle)


@pytest.mark.parametrize("policy", [0, 1])
def test_four_ranks_unequal_blocks(oracle, policy):
    """four processes on one GPU: the Gram all-reduce (policy 0) and the Householder all-gather of four R factors restacked to a
    256 x 64 matrix (policy 1), block heights from 1 row to 30000"""
    res = _run((30000, 1, 7777, 12345), 64, policy=policy)
    assert res["engines"][0] == (3 if policy == 0 else 0)
    _check(res, oracle)


@pytest.mark.parametrize("n,policy,reorth", [(100, 0, False), (128, 0, True), (100, 1, False), (200, 0, False), (330, 0, True)])
def test_two_ranks_more_than_64_columns(oracle, n, policy, reorth):
    """n > 64 over two ranks: 64-column panels, every coupling coefficient block S = Qb^T Ap all-reduced like the Gram tiles (policy 0)
    or the panel factors all-gathered (policy 1); both ranks end with the same R and a globally orthogonal Q.  n = 200 / 330: four / six panels,
    right-looking (round 4): the whole block row of S -- every trailing panel, the last one ragged -- travels in ONE all-reduce per finished panel"""
    import numpy as np
    heights = (9000, 5001)
    res = _run(heights, n, reorth=reorth, policy=policy)
    a, q, r = res["a"], res["q"], res["r"]
    assert res["st"] == 0 and res["r_same"] and len(set(res["engines"])) == 1
    assert np.abs(np.tril(r, -1)).max() == 0.0
    assert np.linalg.norm(q @ r - a) / np.linalg.norm(a) < 5e-7
    assert np.linalg.norm(q.T @ q - np.eye(n)) < 5e-6
    q2, r2 = np.linalg.qr(a.astype(np.float64))
    _, rn = oracle.sign_normalise(q, r)
    _, r2n = oracle.sign_normalise(q2, r2)
    assert np.abs(rn - r2n).max() / np.abs(r2n).max() < 1e-5


def test_two_ranks_ill_conditioned_escalates_identically(oracle):
    """cond 1e9 (1e7..1e8 after rounding to fp32): the bf16 and fp64 Gram levels reject on both ranks (same all-reduced matrix, same
    thresholds), the shifted Cholesky QR two-step finishes the first sweep, the reorthogonalisation sweep brings Q back to O(eps)"""
    res = _run((40000, 25000), 64, reorth=True, cond=1e9)
    assert res["engines"][0] == 4
    a, q, r = res["a"], res["q"], res["r"]
    assert res["st"] == 0 and res["r_same"] and len(set(res["engines"])) == 1
    assert np.abs(np.tril(r, -1)).max() == 0.0
    assert np.linalg.norm(q @ r - a) / np.linalg.norm(a) < 2e-6
    assert np.linalg.norm(q.T @ q - np.eye(64)) < 1e-5


@pytest.mark.parametrize("heights,cond,loop", [((30000, 17777), 1.0, 3), ((30000, 1, 7777, 12345), 1.0, 4), ((40000, 25000), 1e9, 3),
                                               # blocks of 128 k rows on every rank: the chained schedule (Cholesky of call i inside the Gram launch of call i + 1)
                                               ((32768, 16384), 1.0, 4), ((16384, 8192, 32768, 128), 1.0, 3), ((32768, 32768), 1e9, 3),
                                               # one rank eligible, one not: the ranks agree on the plain stream
                                               ((32768, 17777), 1.0, 3)])
def test_stream_of_row_partitioned_calls(oracle, heights, cond, loop):
    """The C loop entry with two calls in flight on every rank: call i + 1 (Gram pass, all-reduce, Cholesky, apply) is enqueued before
    the verdict of call i is read; a rejected matrix (cond 1e9: shifted Cholesky QR) takes the ladder inside the loop on all ranks
    alike -- same engine on every rank, same R, the factorisation a single call returns (its properties: test_two_ranks_* above)."""
    res = _run(heights, 64, cond=cond, loop=loop)
    assert res["st"] == 0 and res["r_same"] and res["a_untouched"] and len(set(res["engines"])) == 1
    if cond == 1.0:
        assert res["engines"][0] == 3
        _check(res, oracle)
    else:
        a, q, r = res["a"], res["q"], res["r"]
        assert res["engines"][0] == 4
        assert np.abs(np.tril(r, -1)).max() == 0.0
        assert np.linalg.norm(q @ r - a) / np.linalg.norm(a) < 2e-6
        assert np.linalg.norm(q.T @ q - np.eye(64)) < 1e-2                 # (cond 1e7..1e8 after rounding, no reorthogonalisation)


def _batch_worker(rank, world, port, heights, n, mode, reorth, policy, cond, loop, out):
    """`loop` DIFFERENT global matrices (matrix number 1 is ill conditioned when cond > 1): every rank factors its blocks one call after
    the other, then all of them through one qr_batch call (tsqr_mi_qr_f32_dist_cb_batch) at loop depth 3 -- same bits on every rank."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tsqr_gpu_amd import blockqr as bq, dist as tdist
    rng = np.random.Generator(np.random.MT19937(11))
    m_glob = sum(heights)
    row0 = sum(heights[:rank]); m_local = heights[rank]
    mats = []
    for k in range(loop):
        a_glob = rng.uniform(-1, 1, size=(m_glob, n)).astype(np.float32)
        if cond > 1 and k == 1:
            u, _ = np.linalg.qr(rng.standard_normal((m_glob, n)))
            v, _ = np.linalg.qr(rng.standard_normal((n, n)))
            a_glob = ((u * np.geomspace(1.0, 1.0 / cond, n)) @ v.T).astype(np.float32)
        mats.append(a_glob)
    d_a = [torch.from_numpy(np.ascontiguousarray(a[row0:row0 + m_local].T)).cuda() for a in mats]
    drv = tdist.RowPartitionedQR(bq.compute_mode[mode], m_local, n, comm="callbacks")
    q1 = [torch.empty(n, m_local, device="cuda") for _ in mats]; r1 = [torch.zeros(n, n, device="cuda") for _ in mats]
    st1 = [drv.qr(q, m_local, r, a, m_local, reorthogonalize=reorth) for q, r, a in zip(q1, r1, d_a)]
    q2 = [torch.full((n, m_local), float("nan"), device="cuda") for _ in mats]; r2 = [torch.zeros(n, n, device="cuda") for _ in mats]
    st2, states = drv.qr_batch(q2, m_local, r2, d_a, m_local, reorthogonalize=reorth)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(q1, q2)) and all(torch.equal(a, b) for a, b in zip(r1, r2))
    flag = torch.tensor([1 if (same and st2 == 0 and states == [0] * loop and st1 == [0] * loop) else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    rs = [torch.zeros(n, n) for _ in range(world)]
    dist.all_gather(rs, r2[-1].cpu())
    if rank == 0:
        out.put({"ok": int(flag.item()), "r_same": all(torch.equal(rs[0], t) for t in rs), "engine": drv.last_engine})
    dist.destroy_process_group()


@pytest.mark.parametrize("heights,cond,k", [((32768, 16384), 1.0, 4), ((32768, 16384), 1e9, 5), ((30000, 17777), 1.0, 3), ((16384, 8192, 32768), 1e9, 4)])
def test_batch_of_row_partitioned_matrices(heights, cond, k):
    """tsqr_mi_qr_f32_dist_cb_batch: K different global matrices, every rank its row blocks, through one call per rank -- the chained
    row-partitioned schedule over distinct operands where every rank is eligible (blocks of 128 k rows), two in flight otherwise, a
    rejected matrix in the stream (all ranks take its ladder alike): bit for bit the K single calls, on every rank."""
    res = _run(heights, 64, cond=cond, loop=k, worker=_batch_worker)
    assert res["ok"] == 1 and res["r_same"]
