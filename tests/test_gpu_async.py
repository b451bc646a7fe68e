"""GPU tests of the stream-asynchronous entry (tsqr_mi_qr_f32_submit / _finish, include/tsqr_mi.h) and of the loop entry built on it.

A submitted call is the blocking call cut in two: the same kernels in the same order on the same stream, so every result below is
compared BIT FOR BIT with tsqr_mi_qr_f32 on the same input -- accepted matrices, matrices the conditioning check rejects (the
ladder then runs inside finish), the n <= 16 second sweep, the 128-column one-panel path, and calls without a speculative first
attempt (executed inside submit)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


class Problem:
    """device copies of one matrix, its outputs and (optionally its own) work buffer"""

    def __init__(self, bq, torch, a, mode, reorth=False, bf=None):
        self.m, self.n = a.shape
        self.a_host = np.ascontiguousarray(a.T)
        self.d_a = torch.from_numpy(self.a_host.copy()).cuda()
        self.d_q = torch.full((self.n, self.m), float("nan"), dtype=torch.float32, device="cuda")
        self.d_r = torch.zeros(self.n, self.n, dtype=torch.float32, device="cuda")
        self.mode, self.reorth = mode, reorth
        if bf is None:
            bf = bq.buffer(mode, reorth)
            bf.allocate(self.m, self.n)
        self.bf = bf

    def args(self):
        return (self.d_q, self.m, self.d_r, self.n, self.d_a, self.m, self.m, self.n, self.bf)

    def result(self):
        return self.d_q.cpu().numpy().copy(), self.d_r.cpu().numpy().copy()


def blocking(bq, torch, a, mode, reorth=False):
    p = Problem(bq, torch, a, mode, reorth)
    st = bq.qr(*p.args())
    return (st, bq.last_engine()) + p.result()


def same(res, want):
    return np.array_equal(res[0], want[2], equal_nan=True) and np.array_equal(res[1], want[3], equal_nan=True)


@pytest.mark.parametrize("mode", ["fp32_tc_cor", "fp32_notc", "fp32_tc_nocor"])
@pytest.mark.parametrize("m,n", [(20000, 64), (9211, 51), (65536, 64), (4096, 33)])
def test_two_calls_in_flight_share_a_buffer(bq, oracle, torch_cuda, m, n, mode):
    md = bq.compute_mode[mode]
    a1, a2 = oracle.uniform_matrix(m, n, seed=1), oracle.uniform_matrix(m, n, seed=2)
    w1, w2 = blocking(bq, torch_cuda, a1, md), blocking(bq, torch_cuda, a2, md)
    p1 = Problem(bq, torch_cuda, a1, md)
    p2 = Problem(bq, torch_cuda, a2, md, bf=p1.bf)                      # one work buffer for both calls (stream order)
    t1 = bq.submit(*p1.args())
    t2 = bq.submit(*p2.args())
    assert t1.pending == 1 and t2.pending == 1 and t1.slot != t2.slot
    assert bq.finish(t1) == 0 and bq.last_engine() == 3
    assert bq.finish(t2) == 0
    assert bq.finish(t2) == 0                                            # finishing twice returns the state again
    assert same(p1.result(), w1) and same(p2.result(), w2)
    assert np.array_equal(p1.d_a.cpu().numpy(), p1.a_host)               # A untouched


def test_rejected_matrix_between_two_accepted_ones(bq, oracle, torch_cuda):
    """The middle call is rejected by the bf16-split level: its apply pass skipped itself, finish runs the ladder for it (fp64 Gram
    level here) while the third call's attempt is already behind it in the stream."""
    md = bq.compute_mode.fp32_tc_cor
    m, n = 30000, 64
    mats = [oracle.uniform_matrix(m, n, seed=5), oracle.matrix_with_cond(m, n, 1e5, seed=6).astype(np.float32), oracle.uniform_matrix(m, n, seed=7)]
    want = [blocking(bq, torch_cuda, a, md) for a in mats]
    assert want[0][1] == 3 and want[1][1] != 3 and want[2][1] == 3
    ps = [Problem(bq, torch_cuda, mats[0], md)]
    ps += [Problem(bq, torch_cuda, a, md, bf=ps[0].bf) for a in mats[1:]]
    t0 = bq.submit(*ps[0].args())
    t1 = bq.submit(*ps[1].args())
    assert bq.finish(t0) == 0 and bq.last_engine() == 3
    t2 = bq.submit(*ps[2].args())
    assert bq.finish(t1) == 0 and bq.last_engine() == want[1][1] and t1.verdict == 1
    assert t2.pending == 2                                               # its verdict was read before the ladder reused the slots
    assert bq.finish(t2) == 0 and bq.last_engine() == 3
    for p, w in zip(ps, want):
        assert same(p.result(), w)
    assert oracle.orthogonality_fro(ps[1].result()[0].T.astype(np.float64)) < 1e-2    # (cond 1e5 without reorthogonalisation: ~cond * eps)


@pytest.mark.parametrize("cond", [1e7, 1e9, 1e12])
def test_every_rung_of_the_ladder(bq, oracle, torch_cuda, cond):
    md = bq.compute_mode.fp32_tc_cor
    a = oracle.matrix_with_cond(20000, 64, cond, seed=9, geometric=True).astype(np.float32)
    w = blocking(bq, torch_cuda, a, md)
    p = Problem(bq, torch_cuda, a, md)
    t = bq.submit(*p.args())
    assert bq.finish(t) == 0 and bq.last_engine() == w[1] and w[1] in (1, 2, 4) and t.verdict == 1
    assert same(p.result(), w)


def test_narrow_panel_second_sweep(bq, oracle, torch_cuda):
    """n <= 16: a call accepted with a scaled conditioning beyond 32 gets a second sweep -- inside finish.  (Geometric singular
    values 1 .. 1/45: S = 55; the bf16-split level turns it down for its smallest pivot ratio, the fp64 Gram level inside finish
    accepts it and the second sweep follows.)"""
    md = bq.compute_mode.fp32_notc
    a = oracle.matrix_with_cond(400000, 12, 45.0, seed=3, geometric=True).astype(np.float32)
    w = blocking(bq, torch_cuda, a, md)
    p = Problem(bq, torch_cuda, a, md)
    t = bq.submit(*p.args())
    assert bq.finish(t) == 0 and bq.last_engine() == w[1]
    assert same(p.result(), w)
    assert oracle.orthogonality_fro(p.result()[0].T.astype(np.float64)) < 5e-6


def test_one_panel_of_128_columns(bq, oracle, torch_cuda):
    md = bq.compute_mode.fp32_tc_cor
    m, n = 30000, 128
    a1, a2 = oracle.uniform_matrix(m, n, seed=1), oracle.matrix_with_cond(m, n, 1e7, seed=2).astype(np.float32)
    w1, w2 = blocking(bq, torch_cuda, a1, md), blocking(bq, torch_cuda, a2, md)
    assert w1[1] == 5 and w2[1] != 5
    p1 = Problem(bq, torch_cuda, a1, md)
    p2 = Problem(bq, torch_cuda, a2, md, bf=p1.bf)
    t1 = bq.submit(*p1.args())
    t2 = bq.submit(*p2.args())
    assert bq.finish(t1) == 0 and bq.last_engine() == 5
    assert bq.finish(t2) == 0 and bq.last_engine() == w2[1]            # rejected as one panel: 64-column panels inside finish
    assert same(p1.result(), w1) and same(p2.result(), w2)


@pytest.mark.parametrize("m,n,reorth", [(9000, 64, True), (5000, 200, False), (4096, 128, True)])
def test_calls_without_a_speculative_attempt_run_inside_submit(bq, oracle, torch_cuda, m, n, reorth):
    md = bq.compute_mode.fp32_tc_cor
    a = oracle.uniform_matrix(m, n, seed=4)
    w = blocking(bq, torch_cuda, a, md, reorth)
    p = Problem(bq, torch_cuda, a, md, reorth)
    t = bq.submit(*p.args())
    assert t.pending == 0 and t.state == 0
    assert bq.finish(t) == 0
    assert same(p.result(), w)


def test_third_submit_and_blocking_call_with_tickets_in_flight(bq, oracle, torch_cuda):
    md = bq.compute_mode.fp32_tc_cor
    m, n = 40000, 64
    mats = [oracle.uniform_matrix(m, n, seed=s) for s in (11, 12, 13, 14)]
    want = [blocking(bq, torch_cuda, a, md) for a in mats]
    ps = [Problem(bq, torch_cuda, a, md) for a in mats]                 # own buffers
    t0, t1 = bq.submit(*ps[0].args()), bq.submit(*ps[1].args())
    t2 = bq.submit(*ps[2].args())                                        # takes t0's slot: t0's verdict is read first
    assert t0.pending == 2 and t1.pending == 1 and t2.slot == t0.slot
    assert bq.qr(*ps[3].args()) == 0                                     # a blocking call reads the verdicts of everything in flight first
    assert t1.pending == 2 and t2.pending == 2
    assert [bq.finish(t) for t in (t0, t1, t2)] == [0, 0, 0]
    for p, w in zip(ps, want):
        assert same(p.result(), w)


def test_submit_error_paths(bq, torch_cuda):
    torch = torch_cuda
    bf = bq.buffer(bq.compute_mode.fp32_tc_cor, False)
    bf.allocate(64, 64)
    d = torch.zeros(64 * 64, dtype=torch.float32, device="cuda")
    t = bq.submit(d, 8, d, 16, d, 8, 8, 16, bf)                          # n > m (reference src/blockqr.cu:409-411)
    assert t.state == 1 and t.pending == 0 and bq.finish(t) == 1
    t = bq.submit(d, 8, d, 8, d, 8, 8, 8, bf, mode=bq.compute_mode.tf32_tc_cor)
    assert t.state == 2 and bq.finish(t) == 2
    with pytest.raises(TypeError):
        bq.submit(d, 8, d, 8, d, 8, 8, 8, bf, mode=bq.compute_mode.fp16_notc)


@pytest.mark.parametrize("m,n,kind", [(1 << 17, 64, "uniform"), (1 << 15, 64, "cond1e6"), (30000, 64, "cond1e6"), (1 << 17, 128, "uniform"), (50000, 128, "uniform"), (400000, 12, "geo45")])
def test_loop_entry_depths_agree(bq, oracle, torch_cuda, m, n, kind):
    """tsqr_mi_qr_f32_loop with two calls in flight against the same loop of blocking calls: same outputs, same engine."""
    md = bq.compute_mode.fp32_tc_cor
    if kind == "uniform":
        a = oracle.uniform_matrix(m, n, seed=8)
    elif kind == "geo45":                                                # (n <= 16, accepted with S > 32: second sweep in every call)
        a = oracle.matrix_with_cond(m, n, 45.0, seed=3, geometric=True).astype(np.float32)
    else:
        a = oracle.matrix_with_cond(m, n, float(kind[4:]), seed=8).astype(np.float32)
    res = []
    for depth in (1, 2, 3):                                              # (3: also the chained schedule for 2^k x 64)
        bq.set_loop_depth(depth)
        try:
            p = Problem(bq, torch_cuda, a, md)
            loop = bq.bind_loop(*p.args())
            assert loop(5) == 0
            res.append(p.result() + (bq.last_engine(),))
            if n <= 64:
                assert np.array_equal(p.d_a.cpu().numpy(), p.a_host)
        finally:
            bq.set_loop_depth(3)
    for k in (1, 2):
        assert np.array_equal(res[0][0], res[k][0]) and np.array_equal(res[0][1], res[k][1]) and res[0][2] == res[k][2]
    assert np.isfinite(res[0][0]).all()


@pytest.mark.parametrize("count", [1, 2, 3, 8])
@pytest.mark.parametrize("m,n", [(1 << 16, 64), (9211, 51), (20000, 128), (20000, 100), (1 << 16, 128), (64 * 700, 128)])
def test_loop_counts(bq, oracle, torch_cuda, m, n, count):
    """Inside the loop a call's completion word is raised by the first kernel of the call behind it (every Gram kernel form: block
    pattern, chunked, 128-column fast and general), the last call by a completion kernel of its own: any count ends complete."""
    md = bq.compute_mode.fp32_tc_cor
    a = oracle.uniform_matrix(m, n, seed=21)
    w = blocking(bq, torch_cuda, a, md)
    p = Problem(bq, torch_cuda, a, md)
    loop = bq.bind_loop(*p.args())
    assert loop(count) == 0 and bq.last_engine() == w[1]
    assert same(p.result(), w)
    p.d_q.fill_(float("nan"))
    assert loop(count) == 0                                              # and again on the same buffers (sequence numbers move on)
    assert same(p.result(), w)


@pytest.mark.parametrize("mode", ["fp32_tc_cor", "fp32_notc", "fp32_tc_nocor"])
@pytest.mark.parametrize("m,pad,count", [(128, 0, 3), (384, 4, 5), (4096, 0, 4), (65536, 8, 3), (1 << 20, 0, 6)])
def test_chained_schedule_shapes(bq, oracle, torch_cuda, m, pad, count, mode):
    """Full 64-column matrices of 128 k rows up to 2^20 take the chained schedule (the R-factor chain of call i inside the Gram launch
    of call i + 1): one block to 8192 blocks, padded leading dimensions, every apply engine -- the results of the blocking call."""
    torch = torch_cuda
    md = bq.compute_mode[mode]
    a = oracle.uniform_matrix(m, 64, seed=31)
    ld = m + pad
    buf = np.zeros((64, ld), np.float32); buf[:, :m] = a.T
    res = []
    for depth in (1, 3):
        d_a = torch.from_numpy(buf).cuda()
        d_q = torch.full((64, ld), float("nan"), dtype=torch.float32, device="cuda")
        d_r = torch.zeros(64, 64, dtype=torch.float32, device="cuda")
        bf = bq.buffer(md, False); bf.allocate(m, 64)
        bq.set_loop_depth(depth)
        try:
            assert bq.bind_loop(d_q, ld, d_r, 64, d_a, ld, m, 64, bf)(count) == 0 and bq.last_engine() == 3
        finally:
            bq.set_loop_depth(3)
        qh = d_q.cpu().numpy()
        assert np.isnan(qh[:, m:]).all() and np.array_equal(d_a.cpu().numpy(), buf)
        res.append((qh[:, :m].copy(), d_r.cpu().numpy().copy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    if mode != "fp32_tc_nocor":                                          # (uncorrected fp16 products: 1e-3 by construction)
        q = res[1][0].T.astype(np.float64)
        assert oracle.orthogonality_fro(q) < 5e-6 and oracle.residual(a, q, res[1][1].T.astype(np.float64)) < 5e-6


# ---- the batch entry (tsqr_mi_qr_f32_batch): many DIFFERENT matrices through the stream the loop entry issues ----
class Batch:
    """K problems of one shape sharing one work buffer; q_of[i] / a_of[i] let a test alias outputs onto inputs"""

    def __init__(self, bq, torch, mats, mode, inplace=False, ld_pad=0):
        self.m, self.n = mats[0].shape
        self.ld = self.m + ld_pad
        self.mode = mode
        self.bf = bq.buffer(mode, False)
        self.bf.allocate(self.m, self.n)
        self.hosts = []
        self.d_a, self.d_q, self.d_r = [], [], []
        for a in mats:
            buf = np.zeros((self.n, self.ld), np.float32); buf[:, :self.m] = a.T
            self.hosts.append(buf)
            self.d_a.append(torch.from_numpy(buf.copy()).cuda())
            self.d_q.append(self.d_a[-1] if inplace else torch.full((self.n, self.ld), float("nan"), dtype=torch.float32, device="cuda"))
            self.d_r.append(torch.zeros(self.n, self.n, dtype=torch.float32, device="cuda"))

    def run(self, bq):
        return bq.qr_batch(self.d_q, self.ld, self.d_r, self.n, self.d_a, self.ld, self.m, self.n, self.bf)

    def results(self):
        return [(q.cpu().numpy()[:, :self.m].copy(), r.cpu().numpy().copy()) for q, r in zip(self.d_q, self.d_r)]


def batch_at_depths(bq, torch, mats, mode, **kw):
    """the batch at loop depth 1 (blocking calls), 2 (two in flight), 3 (chained where it applies): results per depth + states"""
    out = []
    for depth in (1, 2, 3):
        bq.set_loop_depth(depth)
        try:
            b = Batch(bq, torch, mats, mode, **kw)
            st, states = b.run(bq)
            out.append((st, states, b.results(), b))
        finally:
            bq.set_loop_depth(3)
    return out


@pytest.mark.parametrize("mode", ["fp32_tc_cor", "fp32_notc"])
@pytest.mark.parametrize("m,n,k,pad", [(1 << 16, 64, 5, 0), (4096, 64, 3, 4), (1 << 18, 64, 4, 0), (9211, 51, 4, 0), (64 * 520, 128, 4, 0), (20000, 100, 3, 0)])
def test_batch_of_different_matrices_is_the_blocking_calls(bq, oracle, torch_cuda, m, n, k, pad, mode):
    """K different matrices: every schedule returns, bit for bit, what K blocking calls return (chained for 128 k x 64 and 64 k x 128)."""
    md = bq.compute_mode[mode]
    mats = [oracle.uniform_matrix(m, n, seed=40 + i) for i in range(k)]
    want = [blocking(bq, torch_cuda, a, md) for a in mats] if pad == 0 else None
    res = batch_at_depths(bq, torch_cuda, mats, md, ld_pad=pad)
    for st, states, rs, b in res:
        assert st == 0 and states == [0] * k
        for i in range(k):
            assert np.array_equal(rs[i][0], res[0][2][i][0]) and np.array_equal(rs[i][1], res[0][2][i][1])
            if want is not None:
                assert same(rs[i], want[i])
            assert np.array_equal(b.d_a[i].cpu().numpy(), b.hosts[i])        # A untouched (one panel / accepted one-panel path)
    # different matrices really gave different factors
    assert not np.array_equal(res[2][2][0][1], res[2][2][1][1])


@pytest.mark.parametrize("bad", [[2], [0], [4], [1, 2], [0, 1, 2, 3, 4]])
def test_batch_with_rejected_matrices(bq, oracle, torch_cuda, bad):
    """Matrices the bf16-split level rejects anywhere in a chained batch: they get their whole ladder, the accepted ones around them
    stand, the schedule goes on behind them -- every result is the blocking call's."""
    md = bq.compute_mode.fp32_tc_cor
    m, n, k = 1 << 15, 64, 5
    mats = [oracle.matrix_with_cond(m, n, 1e6, seed=60 + i).astype(np.float32) if i in bad else oracle.uniform_matrix(m, n, seed=60 + i) for i in range(k)]
    want = [blocking(bq, torch_cuda, a, md) for a in mats]
    for i in range(k):
        assert (want[i][1] != 3) == (i in bad)
    for st, states, rs, b in batch_at_depths(bq, torch_cuda, mats, md):
        assert st == 0 and states == [0] * k
        for i in range(k):
            assert same(rs[i], want[i]), (i, bad)


def test_batch_in_place(bq, oracle, torch_cuda):
    """q[i] == a[i]: allowed by the chained schedule (Q(i) is clear of A(i + 1)); an accepted call is never redone when a later
    one is rejected."""
    md = bq.compute_mode.fp32_tc_cor
    m, n, k = 1 << 15, 64, 6
    mats = [oracle.uniform_matrix(m, n, seed=80 + i) for i in range(k)]
    mats[2] = oracle.matrix_with_cond(m, n, 1e6, seed=82).astype(np.float32)
    want = [blocking(bq, torch_cuda, a, md) for a in mats]
    for st, states, rs, b in batch_at_depths(bq, torch_cuda, mats, md, inplace=True):
        assert st == 0 and states == [0] * k
        for i in range(k):
            assert same(rs[i], want[i]), i


def test_batch_output_feeding_the_next_input(bq, oracle, torch_cuda):
    """q[i] is a[i + 1]: call i + 1 must factor what call i wrote -- the blocking order.  The chained schedule would read A(i + 1)
    before Q(i) is there; the library sees the overlap and runs the batch in stream order."""
    torch = torch_cuda
    md = bq.compute_mode.fp32_tc_cor
    m, n, k = 1 << 14, 64, 4
    a0 = oracle.uniform_matrix(m, n, seed=90)
    res = []
    for depth in (1, 3):
        bufs = [torch.from_numpy(np.ascontiguousarray(a0.T)).cuda()] + [torch.zeros(n, m, dtype=torch.float32, device="cuda") for _ in range(k)]
        rs = [torch.zeros(n, n, dtype=torch.float32, device="cuda") for _ in range(k)]
        bf = bq.buffer(md, False); bf.allocate(m, n)
        bq.set_loop_depth(depth)
        try:
            st, states = bq.qr_batch(bufs[1:], m, rs, n, bufs[:-1], m, m, n, bf)       # a[i] = bufs[i], q[i] = bufs[i + 1] = a[i + 1]
        finally:
            bq.set_loop_depth(3)
        assert st == 0 and states == [0] * k
        res.append(([b.cpu().numpy() for b in bufs], [r.cpu().numpy() for r in rs]))
    for x, y in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
        assert np.array_equal(x, y)
    r1 = res[1][1][1].T                                                   # R of an orthonormal input: the identity up to signs
    assert np.abs(np.abs(np.diag(r1)) - 1).max() < 1e-5


@pytest.mark.parametrize("m,n", [(1 << 16, 64), (64 * 520, 128)])
def test_loop_entry_in_place_agrees_at_every_depth(bq, oracle, torch_cuda, m, n):
    """tsqr_mi_qr_f32_loop with q == a.  Call i + 1 of the blocking loop factors the Q that call i left in A; the chained
    schedule (Gram pass of call i + 1 before the apply pass of call i) must not be taken -- all depths give the blocking loop's result."""
    torch = torch_cuda
    md = bq.compute_mode.fp32_tc_cor
    a = oracle.uniform_matrix(m, n, seed=95)
    res = []
    for depth in (1, 2, 3):
        d_a = torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
        d_r = torch.zeros(n, n, dtype=torch.float32, device="cuda")
        bf = bq.buffer(md, False); bf.allocate(m, n)
        bq.set_loop_depth(depth)
        try:
            assert bq.bind_loop(d_a, m, d_r, n, d_a, m, m, n, bf)(4) == 0
        finally:
            bq.set_loop_depth(3)
        res.append((d_a.cpu().numpy(), d_r.cpu().numpy()))
    for k in (1, 2):
        assert np.array_equal(res[0][0], res[k][0]) and np.array_equal(res[0][1], res[k][1])
    q = res[2][0].T.astype(np.float64)
    assert oracle.orthogonality_fro(q) < 5e-6
    assert np.abs(np.abs(np.diag(res[2][1])) - 1).max() < 1e-5           # the last call factored an orthonormal matrix


def test_batch_edge_cases(bq, oracle, torch_cuda):
    torch = torch_cuda
    md = bq.compute_mode.fp32_tc_cor
    bf = bq.buffer(md, False); bf.allocate(4096, 64)
    assert bq.qr_batch([], 4096, [], 64, [], 4096, 4096, 64, bf) == (0, [])          # empty batch
    a = oracle.uniform_matrix(4096, 64, seed=3)
    w = blocking(bq, torch, a, md)
    for k in (1, 2):                                                               # below the chained schedule's minimum of three
        b = Batch(bq, torch, [a] * k, md)
        assert b.run(bq) == (0, [0] * k)
        assert all(same(r, w) for r in b.results())
    d = torch.zeros(64 * 64, dtype=torch.float32, device="cuda")
    st, states = bq.qr_batch([d, d, d], 8, [d, d, d], 16, [d, d, d], 8, 8, 16, bf)    # n > m: every call reports it
    assert st == 1 and states == [1, 1, 1]
    with pytest.raises(ValueError):
        bq.qr_batch([d], 8, [d, d], 8, [d], 8, 8, 8, bf)


@pytest.mark.parametrize("m,n,k", [(1 << 17, 64, 6), (64 * 520, 128, 4), (9211, 51, 5)])
def test_cpp_caller_of_qr_batch(bq, torch_cuda, m, n, k):
    """tests/cpp/sample_batch.cpp: a C++ caller's loop of mtk::qr::qr calls over K matrices against one mtk::qr::qr_batch call --
    every byte of every Q and R agrees (exit code 0)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "tests", "cpp"), "-s", "sample_batch"])
    out = subprocess.run([os.path.join(root, "tests", "cpp", "sample_batch"), str(m), str(n), str(k)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatching matrices: 0" in out.stdout
