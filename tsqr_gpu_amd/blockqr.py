"""Python mirror of the reference's host interface for the tall-skinny QR path.

Same names and argument meaning as reference src/blockqr.hpp:
  compute_mode (:12-23), tsqr_colmun_size (:25), state_t codes (:27-29),
  get_working_{q,r,l}_size (:55-57), buffer (:59-140), qr (:142-175).

Everything here is plumbing over the C ABI of csrc/libtsqr_mi.so (include/tsqr_mi.h): the
arithmetic lives in hand-written HIP kernels.  There is NO CPU fallback: importing works without
the library (so sizes/enums can be inspected), but any call that needs it raises RuntimeError.
torch is used only to own device memory and streams.
"""
import ctypes
import enum
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TSQR_MI_LIB: another build of the library (same-box A/B of two builds, tools/r04_ab.sh); default: the in-tree one
LIB_PATH = os.environ.get("TSQR_MI_LIB") or os.path.join(_HERE, "csrc", "libtsqr_mi.so")


class compute_mode(enum.IntEnum):
    """mtk::qr::compute_mode, reference src/blockqr.hpp:12-23 (same order)."""
    fp16_notc = 0
    fp16_tc_nocor = 1
    fp32_notc = 2
    fp32_tc_cor = 3
    fp32_tc_nocor = 4
    mixed_tc_cor_emu = 5
    tf32_tc_cor = 6
    tf32_tc_cor_emu = 7
    tf32_tc_nocor = 8
    tf32_tc_nocor_emu = 9


tsqr_colmun_size = 16                 # reference src/blockqr.hpp:25 (name kept, typo included)
success_factorization = 0             # reference src/blockqr.hpp:28
error_invalid_matrix_size = 1         # reference src/blockqr.hpp:29
error_unsupported_mode = 2            # new: modes without a gfx950 implementation

C_ABI_SYMBOLS = [
    "tsqr_mi_version", "tsqr_mi_last_error",
    "tsqr_mi_working_q_size", "tsqr_mi_working_r_size", "tsqr_mi_working_l_size",
    "tsqr_mi_working_reorth_size", "tsqr_mi_batch_size_log2", "tsqr_mi_batch_size",
    "tsqr_mi_qr_f32", "tsqr_mi_local_r_f32", "tsqr_mi_apply_rinv_f32", "tsqr_mi_rmul_f32",
    "tsqr_mi_qr_f32_dist", "tsqr_mi_set_tuning", "tsqr_mi_profile_enable", "tsqr_mi_profile_read",
    "tsqr_mi_set_policy", "tsqr_mi_last_engine", "tsqr_mi_set_tuning2",
    "tsqr_mi_gram_elems", "tsqr_mi_gram_f32", "tsqr_mi_chol_f32", "tsqr_mi_chol_status", "tsqr_mi_stream_wait", "tsqr_mi_apply_z_f32", "tsqr_mi_validate_f32",
    "tsqr_mi_working_q_size_dist", "tsqr_mi_working_r_size_dist", "tsqr_mi_qr_f32_dist_cb",
    "tsqr_mi_qr_f32_loop", "tsqr_mi_qr_f32_dist_fn", "tsqr_mi_qr_f32_dist_fn_loop", "tsqr_mi_qr_f32_dist_cb_loop",
    "tsqr_mi_qr_f16", "tsqr_mi_qr_f16_loop", "tsqr_mi_working_q_size_f16", "tsqr_mi_working_r_size_f16",
    "tsqr_mi_qr_f32_submit", "tsqr_mi_qr_f32_finish", "tsqr_mi_set_loop_depth", "tsqr_mi_qr_f32_batch", "tsqr_mi_qr_f16_batch",
    "tsqr_mi_qr_f32_dist_fn_batch", "tsqr_mi_qr_f32_dist_cb_batch",
]


class Ticket(ctypes.Structure):
    """tsqr_mi_ticket (include/tsqr_mi.h): one call in flight between submit() and finish()."""
    _fields_ = [("state", ctypes.c_int), ("pending", ctypes.c_int), ("slot", ctypes.c_int), ("own_flag", ctypes.c_int), ("seq", ctypes.c_uint),
                ("verdict", ctypes.c_uint), ("scond", ctypes.c_float), ("mode", ctypes.c_int), ("reorth", ctypes.c_int),
                ("q", ctypes.c_void_p), ("r", ctypes.c_void_p), ("a", ctypes.c_void_p),
                ("ldq", ctypes.c_size_t), ("ldr", ctypes.c_size_t), ("lda", ctypes.c_size_t), ("m", ctypes.c_size_t), ("n", ctypes.c_size_t),
                ("wq", ctypes.c_void_p), ("wr", ctypes.c_void_p), ("stream", ctypes.c_void_p),
                ("h_wl", ctypes.c_void_p), ("words", ctypes.c_void_p), ("words_dev", ctypes.c_void_p)]

    def __del__(self):
        # The library holds the address of a ticket in flight, so one must not go away unfinished -- but finish belongs to the thread
        # that submitted it (include/tsqr_mi.h), and a finaliser runs wherever the collector happens to be, possibly at interpreter
        # shutdown: finish here only on the submitting thread of a live interpreter; otherwise say what went wrong.
        if self.pending != 1:
            return
        import sys
        import threading
        if _lib is not None and not sys.is_finalizing() and getattr(self, "_tid", None) == threading.get_ident():
            _lib.tsqr_mi_qr_f32_finish(ctypes.byref(self))
        else:
            import warnings
            warnings.warn("tsqr_gpu_amd: a Ticket in flight was dropped on another thread (or at shutdown) without finish(); "
                          "call blockqr.finish(ticket) on the submitting thread", ResourceWarning, stacklevel=2)


FP16_MODES = (compute_mode.fp16_notc, compute_mode.fp16_tc_nocor)     # io type half in the reference (src/tsqr.hpp:38-39)

_lib = None


def lib():
    """Load libtsqr_mi.so; raise loudly if it was not built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "tsqr_gpu_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tsqr_gpu_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # torch first: its bundled HIP runtime must be the one libtsqr_mi.so binds to (loading the library before torch
    # pulls the system libamdhip64 and the process ends up with two runtimes -- "no ROCm-capable device")
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    sz, vp, ci = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int
    L.tsqr_mi_version.restype = ci
    L.tsqr_mi_last_error.restype = ctypes.c_char_p
    for name, args in [("tsqr_mi_working_q_size", [sz, sz]), ("tsqr_mi_working_r_size", [sz, sz]),
                       ("tsqr_mi_working_l_size", [sz]), ("tsqr_mi_working_reorth_size", [sz]),
                       ("tsqr_mi_batch_size_log2", [sz]), ("tsqr_mi_batch_size", [sz])]:
        getattr(L, name).restype = sz
        getattr(L, name).argtypes = args
    L.tsqr_mi_qr_f32.restype = ci
    L.tsqr_mi_qr_f32.argtypes = [ci, ci, vp, sz, vp, sz, vp, sz, sz, sz, vp, vp, vp, vp, vp, vp]
    L.tsqr_mi_local_r_f32.restype = ci
    L.tsqr_mi_local_r_f32.argtypes = [vp, sz, vp, sz, sz, sz, vp, vp, vp]
    L.tsqr_mi_apply_rinv_f32.restype = ci
    L.tsqr_mi_apply_rinv_f32.argtypes = [ci, vp, sz, vp, sz, vp, sz, sz, sz, vp, vp]
    L.tsqr_mi_gram_elems.restype = sz
    L.tsqr_mi_gram_elems.argtypes = [sz]
    L.tsqr_mi_gram_f32.restype = ci
    L.tsqr_mi_gram_f32.argtypes = [ci, vp, vp, sz, sz, sz, vp, vp, vp]
    L.tsqr_mi_chol_f32.restype = ci
    L.tsqr_mi_chol_f32.argtypes = [ci, vp, sz, vp, sz, sz, vp, ctypes.POINTER(ctypes.c_uint), vp]
    L.tsqr_mi_chol_status.restype = ci
    L.tsqr_mi_chol_status.argtypes = [vp, sz, sz, ctypes.POINTER(ctypes.c_uint), vp]
    L.tsqr_mi_stream_wait.restype = ci
    L.tsqr_mi_stream_wait.argtypes = [vp]
    L.tsqr_mi_apply_z_f32.restype = ci
    L.tsqr_mi_apply_z_f32.argtypes = [ci, vp, sz, vp, sz, sz, sz, vp, vp]
    L.tsqr_mi_validate_f32.restype = ci
    L.tsqr_mi_validate_f32.argtypes = [vp, sz, vp, sz, vp, sz, sz, sz, vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), vp]
    L.tsqr_mi_rmul_f32.restype = ci
    L.tsqr_mi_rmul_f32.argtypes = [vp, sz, vp, sz, sz, vp, vp]
    L.tsqr_mi_qr_f32_dist.restype = ci
    L.tsqr_mi_qr_f32_dist.argtypes = [ci, ci, vp, sz, vp, sz, vp, sz, sz, sz, vp, vp, vp, vp, ci, vp]
    L.tsqr_mi_qr_f32_dist_cb.restype = ci
    L.tsqr_mi_qr_f32_dist_cb.argtypes = [ci, ci, vp, sz, vp, sz, vp, sz, sz, sz, vp, vp, vp, vp, vp, vp, ci, vp]
    L.tsqr_mi_qr_f32_loop.restype = ci
    L.tsqr_mi_qr_f32_loop.argtypes = [ci] + L.tsqr_mi_qr_f32.argtypes
    L.tsqr_mi_qr_f32_submit.restype = ci
    L.tsqr_mi_qr_f32_submit.argtypes = L.tsqr_mi_qr_f32.argtypes + [ctypes.POINTER(Ticket)]
    L.tsqr_mi_qr_f32_finish.restype = ci
    L.tsqr_mi_qr_f32_finish.argtypes = [ctypes.POINTER(Ticket)]
    L.tsqr_mi_qr_f32_batch.restype = ci
    L.tsqr_mi_qr_f32_batch.argtypes = [ci, ci, ci, vp, sz, vp, sz, vp, sz, sz, sz, vp, vp, vp, vp, vp, vp, vp]
    L.tsqr_mi_qr_f16_batch.restype = ci
    L.tsqr_mi_qr_f16_batch.argtypes = L.tsqr_mi_qr_f32_batch.argtypes
    L.tsqr_mi_set_loop_depth.restype = None
    L.tsqr_mi_set_loop_depth.argtypes = [ci]
    L.tsqr_mi_qr_f16.restype = ci
    L.tsqr_mi_qr_f16.argtypes = L.tsqr_mi_qr_f32.argtypes
    L.tsqr_mi_qr_f16_loop.restype = ci
    L.tsqr_mi_qr_f16_loop.argtypes = [ci] + L.tsqr_mi_qr_f32.argtypes
    for name in ("tsqr_mi_working_q_size_f16", "tsqr_mi_working_r_size_f16"):
        getattr(L, name).restype = sz
        getattr(L, name).argtypes = [sz, sz]
    L.tsqr_mi_qr_f32_dist_fn.restype = ci
    L.tsqr_mi_qr_f32_dist_fn.argtypes = [ci, ci, vp, sz, vp, sz, vp, sz, sz, sz, vp, vp, vp, vp, vp, vp, ci, vp]
    L.tsqr_mi_qr_f32_dist_fn_loop.restype = ci
    L.tsqr_mi_qr_f32_dist_fn_loop.argtypes = [ci] + L.tsqr_mi_qr_f32_dist_fn.argtypes
    L.tsqr_mi_qr_f32_dist_cb_loop.restype = ci
    L.tsqr_mi_qr_f32_dist_cb_loop.argtypes = [ci] + L.tsqr_mi_qr_f32_dist_cb.argtypes
    # (count, mode, reorth, q[], ldq, r[], ldr, a[], lda, m_local, n, wq, wr, gather, <transport: 3 x void*>, nranks, stream, states)
    L.tsqr_mi_qr_f32_dist_fn_batch.restype = ci
    L.tsqr_mi_qr_f32_dist_fn_batch.argtypes = [ci, ci, ci, vp, sz, vp, sz, vp, sz, sz, sz, vp, vp, vp, vp, vp, vp, ci, vp, vp]
    L.tsqr_mi_qr_f32_dist_cb_batch.restype = ci
    L.tsqr_mi_qr_f32_dist_cb_batch.argtypes = [ci, ci, ci, vp, sz, vp, sz, vp, sz, sz, sz, vp, vp, vp, vp, vp, vp, ci, vp, vp]
    for name in ("tsqr_mi_working_q_size_dist", "tsqr_mi_working_r_size_dist"):
        getattr(L, name).restype = sz
        getattr(L, name).argtypes = [sz, sz, ci]
    L.tsqr_mi_profile_enable.restype = None
    L.tsqr_mi_profile_enable.argtypes = [ci]
    L.tsqr_mi_profile_read.restype = ci
    L.tsqr_mi_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), ci]
    L.tsqr_mi_set_policy.restype = None
    L.tsqr_mi_set_policy.argtypes = [ci]
    L.tsqr_mi_last_engine.restype = ci
    L.tsqr_mi_set_tuning2.restype = None
    L.tsqr_mi_set_tuning2.argtypes = [ci, ci]
    L.tsqr_mi_set_tuning.restype = None
    L.tsqr_mi_set_tuning.argtypes = [ci, ci]
    _lib = L
    return L


def last_error():
    return lib().tsqr_mi_last_error().decode()


# ---- mtk::qr::get_working_*_size (reference src/blockqr.hpp:55-57) ----------------------------------
def get_working_q_size(m, n):
    return lib().tsqr_mi_working_q_size(m, n)


def get_working_r_size(m, n):
    return lib().tsqr_mi_working_r_size(m, n)


def get_working_l_size(m):
    return lib().tsqr_mi_working_l_size(m)


def get_batch_size_log2(m):
    """mtk::tsqr::get_batch_size_log2, reference src/tsqr.cu:39-41."""
    return lib().tsqr_mi_batch_size_log2(m)


def get_batch_size(m):
    return lib().tsqr_mi_batch_size(m)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class buffer:
    """mtk::qr::buffer<mode, Reorthogonalize>, reference src/blockqr.hpp:59-140.

    Fields dwq, dwr, dw_reorth_r, dl (device) and hl (pinned host) as in the reference;
    allocate() twice raises RuntimeError like the reference's std::runtime_error (:77-79).
    """

    def __init__(self, mode=compute_mode.fp32_tc_cor, reorthogonalize=False, device="cuda"):
        self.mode = compute_mode(mode)
        self.reorthogonalize = bool(reorthogonalize)
        self.device = device
        self.dwq = self.dwr = self.dw_reorth_r = self.dl = self.hl = None
        self.total_memory_size = 0

    def allocate(self, m, n):
        import torch
        if self.dwq is not None or self.dwr is not None or self.dl is not None or self.hl is not None:
            raise RuntimeError("The buffer has been already allocated")
        wq, wr, wl = get_working_q_size(m, n), get_working_r_size(m, n), get_working_l_size(m)
        if self.mode in FP16_MODES:                          # room for the widened A, Q and R of the fp16 entry (in 4-byte units)
            wq, wr = lib().tsqr_mi_working_q_size_f16(m, n), lib().tsqr_mi_working_r_size_f16(m, n)
        self.dwq = torch.empty(max(wq, 1), dtype=torch.float32, device=self.device)
        self.dwr = torch.empty(max(wr, 1), dtype=torch.float32, device=self.device)
        self.dl = torch.empty(max(wl, 1), dtype=torch.int32, device=self.device)
        self.hl = torch.empty(max(wl, 1), dtype=torch.int32).pin_memory()
        self.total_memory_size = 4 * (wq + wr + wl)
        if self.reorthogonalize:
            ro = lib().tsqr_mi_working_reorth_size(m)
            self.dw_reorth_r = torch.empty(max(ro, 1), dtype=torch.float32, device=self.device)
            self.total_memory_size += 4 * ro

    def destroy(self):
        self.dwq = self.dwr = self.dw_reorth_r = self.dl = self.hl = None

    def get_device_memory_size(self):
        return self.total_memory_size


def qr(q, ldq, r, ldr, a, lda, m, n, bf, stream=None, mode=None, reorthogonalize=None):
    """mtk::qr::qr<mode, Reorthogonalize>(q, ldq, r, ldr, a, lda, m, n, buffer, handle).

    q, r, a: float32 torch tensors on the GPU holding column-major data (any shape; only data_ptr is
    used) -- float16 tensors for the two fp16 I/O modes (io type half in the reference, src/tsqr.hpp:38-39; the buffer must have
    been allocated for such a mode).  `stream` (torch.cuda.Stream or None = current) takes the place of the cublasHandle_t, whose
    only role in the reference is to carry the stream (src/blockqr.cu:58-59).  Blocking, returns state_t.
    Runtime failures raise RuntimeError (the reference throws std::runtime_error from CUTF_CHECK_ERROR).
    """
    import torch
    mode = bf.mode if mode is None else compute_mode(mode)
    reorth = bf.reorthogonalize if reorthogonalize is None else bool(reorthogonalize)
    if stream is None:
        stream = torch.cuda.current_stream()
    if mode in FP16_MODES:
        for t in (q, r, a):
            if t is not None and t.dtype != torch.float16:
                raise TypeError("%s takes float16 tensors (io type half, reference src/tsqr.hpp:38-39)" % mode.name)
        if bf.mode not in FP16_MODES:
            raise RuntimeError("the buffer was allocated for %s: an fp16 mode needs the larger work space of its own allocate()" % bf.mode.name)
        fn, name = lib().tsqr_mi_qr_f16, "tsqr_mi_qr_f16"
    else:
        fn, name = lib().tsqr_mi_qr_f32, "tsqr_mi_qr_f32"
    st = fn(int(mode), int(reorth), _ptr(q), ldq, _ptr(r), ldr, _ptr(a), lda, m, n,
            _ptr(bf.dwq), _ptr(bf.dwr), _ptr(bf.dw_reorth_r), _ptr(bf.dl), _ptr(bf.hl), stream.cuda_stream)
    if st < 0:
        raise RuntimeError("%s failed: %s" % (name, last_error()))
    return st


def submit(q, ldq, r, ldr, a, lda, m, n, bf, stream=None, mode=None, reorthogonalize=None):
    """Stream-asynchronous qr() (tsqr_mi_qr_f32_submit): enqueues the call's first attempt and returns a ticket; finish(ticket) waits,
    completes the ladder for a rejected matrix and returns state_t.  Up to two calls of a thread are in flight; tickets are finished in
    submission order by the submitting thread.  fp32 I/O modes."""
    import torch
    mode = bf.mode if mode is None else compute_mode(mode)
    reorth = bf.reorthogonalize if reorthogonalize is None else bool(reorthogonalize)
    if mode in FP16_MODES:
        raise TypeError("submit() takes the fp32 I/O modes")
    if stream is None:
        stream = torch.cuda.current_stream()
    import threading
    t = Ticket()
    t._keep = (q, r, a, bf, stream)
    t._tid = threading.get_ident()
    st = lib().tsqr_mi_qr_f32_submit(int(mode), int(reorth), _ptr(q), ldq, _ptr(r), ldr, _ptr(a), lda, m, n,
                                     _ptr(bf.dwq), _ptr(bf.dwr), _ptr(bf.dw_reorth_r), _ptr(bf.dl), _ptr(bf.hl), stream.cuda_stream, ctypes.byref(t))
    if st < 0:
        raise RuntimeError("tsqr_mi_qr_f32_submit failed: %s" % last_error())
    return t


def finish(ticket):
    st = lib().tsqr_mi_qr_f32_finish(ctypes.byref(ticket))
    if st < 0:
        raise RuntimeError("tsqr_mi_qr_f32_finish failed: %s" % last_error())
    return st


def set_loop_depth(depth):
    """The loop entries (bind_loop): 1 = plain blocking calls, 2 = two calls in flight, 3 (default) = also the chained schedule for full
    64-column matrices of up to 2^20 rows (the R-factor chain of call i inside the Gram launch of call i + 1)."""
    lib().tsqr_mi_set_loop_depth(int(depth))


def bind(q, ldq, r, ldr, a, lda, m, n, bf, stream=None, mode=None, reorthogonalize=None):
    """qr() with every argument marshalled once: returns a zero-argument callable that issues exactly one C-ABI call
    (tsqr_mi_qr_f32) per invocation -- what a C++ caller's loop looks like, without per-call Python/ctypes conversions.
    The tensors, the buffer and the stream must stay alive and unchanged while the callable is in use."""
    import torch
    mode = bf.mode if mode is None else compute_mode(mode)
    reorth = bf.reorthogonalize if reorthogonalize is None else bool(reorthogonalize)
    if stream is None:
        stream = torch.cuda.current_stream()
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    fn = lib().tsqr_mi_qr_f32
    args = (ci(int(mode)), ci(int(reorth)), vp(_ptr(q)), sz(ldq), vp(_ptr(r)), sz(ldr), vp(_ptr(a)), sz(lda), sz(m), sz(n),
            vp(_ptr(bf.dwq)), vp(_ptr(bf.dwr)), vp(_ptr(bf.dw_reorth_r)), vp(_ptr(bf.dl)), vp(_ptr(bf.hl)), vp(stream.cuda_stream))
    keep = (q, r, a, bf, stream)

    def call():
        st = fn(*args)
        if st < 0:
            raise RuntimeError("tsqr_mi_qr_f32 failed: %s" % last_error())
        return st
    call._keep = keep
    return call


def bind_loop(q, ldq, r, ldr, a, lda, m, n, bf, stream=None, mode=None, reorthogonalize=None):
    """Like bind(), but the callable takes a count k and issues k calls from ONE C loop (tsqr_mi_qr_f32_loop): the reference's speed
    protocol (src/test.cu:299-309) without interpreter time between the calls.  The fp32 loop keeps two calls in flight (submit /
    finish) unless set_loop_depth(1)."""
    import torch
    mode = bf.mode if mode is None else compute_mode(mode)
    reorth = bf.reorthogonalize if reorthogonalize is None else bool(reorthogonalize)
    if stream is None:
        stream = torch.cuda.current_stream()
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    fn = lib().tsqr_mi_qr_f16_loop if mode in FP16_MODES else lib().tsqr_mi_qr_f32_loop      # (float16 tensors for the fp16 I/O modes)
    args = (ci(int(mode)), ci(int(reorth)), vp(_ptr(q)), sz(ldq), vp(_ptr(r)), sz(ldr), vp(_ptr(a)), sz(lda), sz(m), sz(n),
            vp(_ptr(bf.dwq)), vp(_ptr(bf.dwr)), vp(_ptr(bf.dw_reorth_r)), vp(_ptr(bf.dl)), vp(_ptr(bf.hl)), vp(stream.cuda_stream))
    keep = (q, r, a, bf, stream)

    def call(k=1):
        st = fn(ci(int(k)), *args)
        if st < 0:
            raise RuntimeError("tsqr_mi_qr_f32 failed: %s" % last_error())
        return st
    call._keep = keep
    return call


def bind_batch(qs, ldq, rs, ldr, as_, lda, m, n, bf, stream=None, mode=None, reorthogonalize=None):
    """mtk::qr::qr_batch (tsqr_mi_qr_f32_batch / tsqr_mi_qr_f16_batch): qs, rs, as_ are equally long sequences of float32 tensors
    (float16 for the two fp16 I/O modes), one (q, r, a) triple per matrix, all of the shape m x n with the shared leading dimensions.  Returns a callable: call() factors every matrix (blocking) and
    returns (first non-zero state, [state of every call]).  The tensors, the buffer and the stream must stay alive while it is in use."""
    import torch
    mode = bf.mode if mode is None else compute_mode(mode)
    reorth = bf.reorthogonalize if reorthogonalize is None else bool(reorthogonalize)
    if mode in FP16_MODES:
        if bf.mode not in FP16_MODES:
            raise RuntimeError("the buffer was allocated for %s: an fp16 mode needs the larger work space of its own allocate()" % bf.mode.name)
        for t in list(qs) + list(rs) + list(as_):
            if t.dtype != torch.float16:
                raise TypeError("%s takes float16 tensors (io type half, reference src/tsqr.hpp:38-39)" % mode.name)
    if not (len(qs) == len(rs) == len(as_)):
        raise ValueError("qr_batch: q, r and a must name the same number of matrices")
    if stream is None:
        stream = torch.cuda.current_stream()
    count = len(as_)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    arr = lambda ts: (vp * max(count, 1))(*[t.data_ptr() for t in ts])
    pq, pr, pa = arr(qs), arr(rs), arr(as_)
    states = (ci * max(count, 1))()
    fn = lib().tsqr_mi_qr_f16_batch if mode in FP16_MODES else lib().tsqr_mi_qr_f32_batch
    args = (ci(count), ci(int(mode)), ci(int(reorth)), pq, sz(ldq), pr, sz(ldr), pa, sz(lda), sz(m), sz(n),
            vp(_ptr(bf.dwq)), vp(_ptr(bf.dwr)), vp(_ptr(bf.dw_reorth_r)), vp(_ptr(bf.dl)), vp(_ptr(bf.hl)), vp(stream.cuda_stream), states)

    def call():
        st = fn(*args)
        if st < 0:
            raise RuntimeError("tsqr_mi_qr_batch failed: %s" % last_error())
        return st, list(states[:count])
    call._keep = (list(qs), list(rs), list(as_), bf, stream, pq, pr, pa, states)
    return call


def qr_batch(qs, ldq, rs, ldr, as_, lda, m, n, bf, stream=None, mode=None, reorthogonalize=None):
    """One-shot form of bind_batch(): returns (first non-zero state, [states])."""
    return bind_batch(qs, ldq, rs, ldr, as_, lda, m, n, bf, stream, mode, reorthogonalize)()


def set_tuning(level0_waves=0, tree_chunks_per_wave=0):
    lib().tsqr_mi_set_tuning(level0_waves, tree_chunks_per_wave)


KERNEL_CLASSES = ["fold_level0", "fold_tree", "trinv", "apply", "coupling", "other", "gram", "chol"]


def profile_enable(on=True):
    lib().tsqr_mi_profile_enable(int(bool(on)))


def profile_read():
    """{class: (milliseconds, launches)} accumulated since profile_enable(True)."""
    ms = (ctypes.c_double * 8)()
    cnt = (ctypes.c_long * 8)()
    k = lib().tsqr_mi_profile_read(ms, cnt, 8)
    return {KERNEL_CLASSES[i]: (ms[i], cnt[i]) for i in range(k)}


POLICY_AUTO, POLICY_HOUSEHOLDER, POLICY_GRAM_F64, POLICY_GRAM_BF16, POLICY_AUTO_NO_BF16 = 0, 1, 2, 3, 4
ENGINE_NAMES = {0: "householder_tsqr", 1: "gram_f64_cholesky", 2: "gram_rejected_then_householder", 3: "gram_bf16x3_cholesky",
                4: "gram_rejected_then_shifted_cholesky_qr2", 5: "gram_bf16x3_cholesky_one_panel_128"}


def set_policy(policy):
    """R-factor engine policy (include/tsqr_mi.h)."""
    lib().tsqr_mi_set_policy(int(policy))


def last_engine():
    return lib().tsqr_mi_last_engine()
