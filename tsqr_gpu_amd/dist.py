"""Row-partitioned TSQR across the GPUs of one node (new functionality -- the reference has no
multi-GPU path).  One process per GPU; rank p holds the row block A_p (the blocks may have different heights).

The whole factorisation of a rank is ONE call into the C ABI (tsqr_mi_qr_f32_dist / tsqr_mi_qr_f32_dist_cb, include/tsqr_mi.h):
the same ladder as the single-GPU call, stream-ordered, with two exchange hooks

  Gram levels:         G_p = A_p^T A_p on the matrix cores -> all-reduce of the Gram tiles + the local row count (<= 2561
                       doubles, ~20 KiB) -> every rank factors the SAME matrix with thresholds from the SAME global row count
                       (identical accept / reject decisions on all ranks by construction) -> Q_p = A_p * inverse(R);
  Householder engine:  R_p = fold(A_p) -> all-gather of the n x n factors (16 KiB per rank at n = 64, the exchange the north
                       star names) -> every rank folds the same (P n) x n stack -> R bitwise identical -> Q_p = A_p * inverse(R).

Transport of the hooks:
  * RcclComm: a raw ncclComm_t created through ctypes on the librccl the process has already loaded (unique id from rank 0,
    broadcast over torch.distributed); the C code then calls ncclAllReduce / ncclAllGather itself on the caller's stream -- no
    Python and no host wait between the kernels of a call.  The addresses of those two entry points are taken from the SAME
    library handle that created the communicator and travel with it into the C call (tsqr_mi_qr_f32_dist_fn): a process can have
    two RCCL copies mapped (torch's bundled one and /opt/rocm's) and they must never meet.  This is what bench.py --gpus N uses.
  * TorchCollectives: callbacks into torch.distributed (any backend).  The two-process tests run the product engine with it over
    gloo on one GPU, and bench.py falls back to it when a raw communicator cannot be created.

`backend` is the object that executes the per-rank call: HipBackend (the product; no CPU fallback) or a test double with the
same `qr_dist` signature (tests/test_dist_cpu.py drives the exchange protocol on CPU over gloo with a numpy double).
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import blockqr as bq


class TorchCollectives:
    """The two exchange hooks on torch.distributed tensors (works with gloo and nccl process groups)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def allreduce_f64(self, t):
        if self.world > 1:
            dist.all_reduce(t, group=self.group)

    def allgather_f32(self, send, recv):
        """recv (world * send.numel()) <- concatenation of every rank's send, in rank order"""
        k = send.numel()
        if self.world > 1:
            dist.all_gather([recv[i * k:(i + 1) * k] for i in range(self.world)], send, group=self.group)
        else:
            recv[:k].copy_(send)


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


def _loaded_rccl_path():
    """Path of the librccl this process has mapped (torch brings its own copy); None if there is none yet."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "librccl" in line:
                    return line.split()[-1]
    except OSError:
        pass
    return None


class RcclComm:
    """Raw RCCL communicator for the C driver.  Collective over `group`: every rank must construct it at the same point."""

    def __init__(self, group=None):
        assert dist.is_initialized(), "RcclComm needs an initialised torch.distributed process group (to ship the unique id)"
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.comm = None
        path = _loaded_rccl_path()
        lib = None
        for cand in ([path] if path else []) + ["librccl.so.1", "librccl.so"]:
            try:
                lib = ctypes.CDLL(cand)
                break
            except OSError:
                continue
        ok = lib is not None and all(hasattr(lib, s) for s in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy",
                                                               "ncclAllReduce", "ncclAllGather"))
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)     # every rank must be able to, or nobody tries
        if int(flag.item()) == 0:
            raise RuntimeError("librccl not loadable on every rank")
        self._lib = lib
        uid = _NcclUniqueId()
        if self.rank == 0:
            rc = lib.ncclGetUniqueId(ctypes.byref(uid))
            if rc != 0:
                uid = _NcclUniqueId()                  # all zeros: ncclCommInitRank then fails on every rank alike
        buf = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone().to(dev)
        dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ctypes.memmove(ctypes.byref(uid), bytes(buf.cpu().numpy().tobytes()), 128)
        comm = ctypes.c_void_p()
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
        lib.ncclCommInitRank.restype = ctypes.c_int
        rc = lib.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank)
        good = (rc == 0 and bool(comm.value))
        # second agreement: a communicator that exists on some ranks only would hang the first exchange -- all or nobody
        flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            if good:
                lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
                lib.ncclCommDestroy(comm)
            raise RuntimeError("ncclCommInitRank failed on at least one rank (%d here)" % rc)
        self.comm = comm
        # the collectives of THIS library instance, as plain addresses for the C driver
        self.allreduce_fn = ctypes.cast(lib.ncclAllReduce, ctypes.c_void_p)
        self.allgather_fn = ctypes.cast(lib.ncclAllGather, ctypes.c_void_p)
        self.lib_path = getattr(lib, "_name", None)

    def destroy(self):
        if self.comm is not None:
            self._lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            self._lib.ncclCommDestroy(self.comm)
            self.comm = None


_ALLREDUCE_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
_ALLGATHER_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class HipBackend:
    """The product: one C-ABI call per factorisation (libtsqr_mi.so; raises if the library is missing -- no CPU fallback)."""

    def __init__(self, mode, m_local, n, world, comm=None, collectives=None):
        self.mode = bq.compute_mode(mode)
        self.n, self.m_local, self.world = n, m_local, world
        L = bq.lib()
        self.wq = torch.empty(max(L.tsqr_mi_working_q_size_dist(m_local, n, world), 1), dtype=torch.float32, device="cuda")
        self.wr = torch.empty(max(L.tsqr_mi_working_r_size_dist(m_local, n, world), 1), dtype=torch.float32, device="cuda")
        pc = min(n, 64)                                # (n > 64: 64-column panels; the all-gather is one panel's factors)
        self.gather = torch.empty(world * pc * pc, dtype=torch.float32, device="cuda")
        self.comm = comm                               # RcclComm or None
        self.coll = collectives or TorchCollectives()
        self._cb_error = None
        self._ar = _ALLREDUCE_CB(self._allreduce)
        self._ag = _ALLGATHER_CB(self._allgather)

    # ---- callbacks: device pointers inside our own buffers -> tensor views -> torch.distributed ----
    def _view(self, ptr, count, dtype):
        for t in (self.wq, self.wr, self.gather):
            base = t.data_ptr()
            if base <= ptr < base + 4 * t.numel():
                off = (ptr - base) // 4
                if dtype == torch.float64:
                    return t[off:off + 2 * count].view(torch.float64)
                return t[off:off + count]
        raise RuntimeError("collective on a pointer outside the engine's buffers")

    def _allreduce(self, user, buf, count, stream):
        try:
            self.coll.allreduce_f64(self._view(buf, count, torch.float64))
            return 0
        except Exception as e:                         # never let an exception cross the C frame
            self._cb_error = e
            return 1

    def _allgather(self, user, send, recv, count, stream):
        try:
            self.coll.allgather_f32(self._view(send, count, torch.float32), self._view(recv, self.world * count, torch.float32))
            return 0
        except Exception as e:
            self._cb_error = e
            return 1

    @property
    def last_engine(self):
        return bq.last_engine()                        # (per host thread: the engine of this thread's last call)

    def _check_block(self, m_local):
        """The work buffers were sized for the constructor's m_local; a taller block would make the kernels write past them
        (the C side cannot see the allocation).  Shorter blocks are fine."""
        L = bq.lib()
        if (L.tsqr_mi_working_q_size_dist(m_local, self.n, self.world) > self.wq.numel() or
                L.tsqr_mi_working_r_size_dist(m_local, self.n, self.world) > self.wr.numel()):
            raise ValueError("row block of %d rows needs larger work buffers than this engine allocated for %d rows: "
                             "construct RowPartitionedQR with the largest block height" % (m_local, self.m_local))

    def qr_dist(self, q, ldq, r, ldr, a, lda, m_local, reorth):
        self._check_block(m_local)
        st = torch.cuda.current_stream().cuda_stream   # the callbacks issue torch work on the current stream: it must be this one
        L = bq.lib()
        if self.comm is not None:
            rc = L.tsqr_mi_qr_f32_dist_fn(int(self.mode), int(reorth), q.data_ptr(), ldq, r.data_ptr(), ldr, a.data_ptr(), lda,
                                          m_local, self.n, self.wq.data_ptr(), self.wr.data_ptr(), self.gather.data_ptr(),
                                          self.comm.comm, self.comm.allreduce_fn, self.comm.allgather_fn, self.world, st)
        else:
            self._cb_error = None
            rc = L.tsqr_mi_qr_f32_dist_cb(int(self.mode), int(reorth), q.data_ptr(), ldq, r.data_ptr(), ldr, a.data_ptr(), lda,
                                          m_local, self.n, self.wq.data_ptr(), self.wr.data_ptr(), self.gather.data_ptr(),
                                          self._ar, self._ag, None, self.world, st)
            if self._cb_error is not None:
                raise self._cb_error
        if rc < 0:
            raise RuntimeError("tsqr_mi_qr_f32_dist failed: %s" % bq.last_error())
        return rc


    def bind_dist(self, q, ldq, r, ldr, a, lda, m_local, reorth, loop=False):
        """qr_dist with every argument marshalled once (as blockqr.bind): a zero-argument callable = one C-ABI call per invocation.
        loop = True: the callable takes a count k and issues k back-to-back calls from one C loop (every rank the same k)."""
        self._check_block(m_local)
        st = torch.cuda.current_stream()
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L = bq.lib()
        head = (ci(int(self.mode)), ci(int(reorth)), vp(q.data_ptr()), sz(ldq), vp(r.data_ptr()), sz(ldr), vp(a.data_ptr()), sz(lda),
                sz(m_local), sz(self.n), vp(self.wq.data_ptr()), vp(self.wr.data_ptr()), vp(self.gather.data_ptr()))
        if self.comm is not None:
            fn = L.tsqr_mi_qr_f32_dist_fn_loop if loop else L.tsqr_mi_qr_f32_dist_fn
            args = head + (self.comm.comm, self.comm.allreduce_fn, self.comm.allgather_fn, ci(self.world), vp(st.cuda_stream))
        else:
            fn = L.tsqr_mi_qr_f32_dist_cb_loop if loop else L.tsqr_mi_qr_f32_dist_cb
            args = head + (self._ar, self._ag, None, ci(self.world), vp(st.cuda_stream))

        def call(k=None):
            self._cb_error = None
            rc = fn(ci(int(k if k is not None else 1)), *args) if loop else fn(*args)
            if self._cb_error is not None:
                raise self._cb_error
            if rc < 0:
                raise RuntimeError("tsqr_mi_qr_f32_dist failed: %s" % bq.last_error())
            return rc
        call._keep = (q, r, a, st, self)
        return call


    def bind_batch(self, qs, ldq, rs, ldr, as_, lda, m_local, reorth):
        """`len(as_)` DIFFERENT row-partitioned matrices through one C call (tsqr_mi_qr_f32_dist_{fn,cb}_batch): returns a callable giving
        (first non-zero state, [states]).  Every rank binds the same number of matrices."""
        self._check_block(m_local)
        if not (len(qs) == len(rs) == len(as_)):
            raise ValueError("qr_batch: q, r and a must name the same number of matrices")
        st = torch.cuda.current_stream()
        count = len(as_)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        arr = lambda ts: (vp * max(count, 1))(*[t.data_ptr() for t in ts])
        pq, pr, pa = arr(qs), arr(rs), arr(as_)
        states = (ci * max(count, 1))()
        L = bq.lib()
        head = (ci(count), ci(int(self.mode)), ci(int(reorth)), pq, sz(ldq), pr, sz(ldr), pa, sz(lda), sz(m_local), sz(self.n),
                vp(self.wq.data_ptr()), vp(self.wr.data_ptr()), vp(self.gather.data_ptr()))
        if self.comm is not None:
            fn = L.tsqr_mi_qr_f32_dist_fn_batch
            args = head + (self.comm.comm, self.comm.allreduce_fn, self.comm.allgather_fn, ci(self.world), vp(st.cuda_stream), states)
        else:
            fn = L.tsqr_mi_qr_f32_dist_cb_batch
            args = head + (ctypes.cast(self._ar, vp), ctypes.cast(self._ag, vp), None, ci(self.world), vp(st.cuda_stream), states)

        def call():
            self._cb_error = None
            rc = fn(*args)
            if self._cb_error is not None:
                raise self._cb_error
            if rc < 0:
                raise RuntimeError("tsqr_mi_qr_f32_dist_batch failed: %s" % bq.last_error())
            return rc, list(states[:count])
        call._keep = (list(qs), list(rs), list(as_), st, self, pq, pr, pa, states)
        return call


class RowPartitionedQR:
    """mtk::qr::qr for a matrix whose rows are spread over the ranks of `group`.

    comm: "auto" (raw RCCL communicator if it can be created on every rank, else torch.distributed callbacks), "rccl",
    "callbacks", or an RcclComm instance.  backend: overrides the executor (tests)."""

    def __init__(self, mode, m_local, n, group=None, comm="auto", backend=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n, self.m_local = n, m_local
        self.transport = "none"
        # every rank must hold at least one row: a rank that bailed out alone would leave the others waiting in the first exchange
        # (include/tsqr_mi.h) -- so the check is collective and every rank raises together
        self._require_rows(m_local)
        if backend is not None:
            self.backend = backend
            self.transport = "test-double"
            return
        raw = None
        if isinstance(comm, RcclComm):
            raw = comm
        elif dist.is_initialized() and ((comm == "auto" and self.world > 1) or comm == "rccl"):
            try:
                raw = RcclComm(group)
            except Exception:
                if comm == "rccl":
                    raise
                raw = None
        self.transport = "rccl" if raw is not None else "torch.distributed callbacks"
        self.backend = HipBackend(mode, m_local, n, self.world, comm=raw, collectives=TorchCollectives(group))

    def _require_rows(self, m_local):
        """Collective: raises ValueError on EVERY rank when any rank's block is empty (or n == 0)."""
        least = min(int(m_local), int(self.n))
        if self.world > 1:
            on_gpu = dist.get_backend(self.group) == "nccl"
            t = torch.tensor([least], dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            least = int(t.item())
        if least < 1:
            raise ValueError("row-partitioned QR: every rank needs m_local >= 1 and n >= 1 (at least one rank passed an empty block)")

    def qr(self, q, ldq, r, a, lda, reorthogonalize=False, m_local=None):
        """q, a: column-major m_local x n blocks (tensors; q may alias a only with reorthogonalize... never for the first sweep);
        r: (n, n) tensor receiving the column-major R, identical on every rank.  Blocking, collective.  Returns state_t.
        m_local: this call's block height (not taller than the constructor's, which sized the work buffers).  Passing it is a
        COLLECTIVE decision: when any rank passes m_local to a call, every rank must pass it to that call (its own value; the
        constructor's is fine) -- the check that no rank holds an empty block is an all-reduce every rank has to post, whether or not
        its own height changed."""
        if m_local is not None:
            self._require_rows(m_local)
        m_local = self.m_local if m_local is None else m_local
        return self.backend.qr_dist(q, ldq, r, self.n, a, lda, m_local, bool(reorthogonalize))

    def bind(self, q, ldq, r, a, lda, reorthogonalize=False, m_local=None, loop=False):
        """The same call with its arguments marshalled once: returns a zero-argument callable (a C++ caller's loop body); loop = True:
        a callable taking a count k = k back-to-back calls from one C loop.  The tensors and the current stream must stay alive and
        unchanged while it is in use.  m_local: as in qr() -- passed by every rank or by none."""
        if m_local is not None:
            self._require_rows(m_local)
        m_local = self.m_local if m_local is None else m_local
        if hasattr(self.backend, "bind_dist"):
            return self.backend.bind_dist(q, ldq, r, self.n, a, lda, m_local, bool(reorthogonalize), loop=loop)
        if loop:
            def run(k=1):
                for _ in range(k):
                    st = self.qr(q, ldq, r, a, lda, reorthogonalize, m_local)
                    if st:
                        return st
                return 0
            return run
        return lambda: self.qr(q, ldq, r, a, lda, reorthogonalize, m_local)

    def bind_loop(self, q, ldq, r, a, lda, reorthogonalize=False, m_local=None):
        return self.bind(q, ldq, r, a, lda, reorthogonalize, m_local, loop=True)

    def qr_batch(self, qs, ldq, rs, as_, lda, reorthogonalize=False):
        """Several DIFFERENT row-partitioned matrices (this rank's block of each, the constructor's block height) through one C call, as the
        stream of calls the loop entry issues.  Collective: every rank passes the same number of matrices.  Returns (first non-zero state,
        [state of every call])."""
        if not hasattr(self.backend, "bind_batch"):
            states = [self.qr(q, ldq, r, a, lda, reorthogonalize) for q, r, a in zip(qs, rs, as_)]
            return next((s for s in states if s), 0), states
        return self.backend.bind_batch(qs, ldq, rs, self.n, as_, lda, self.m_local, bool(reorthogonalize))()

    @property
    def last_engine(self):
        return getattr(self.backend, "last_engine", 0)
