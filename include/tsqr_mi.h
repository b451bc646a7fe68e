/*
 * tsqr_mi.h -- C ABI of the MI355X-native tall-skinny QR engine (libtsqr_mi.so).
 *
 * This is the drop-in boundary for the reference's hot path
 *     mtk::qr::qr<compute_mode, Reorthogonalize>() / mtk::qr::buffer
 * (reference src/blockqr.hpp:59-175, src/blockqr.cu:394-433).  The C++ templates in
 * include/tsqr/blockqr.hpp forward to these entry points; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions (same as the reference): all matrices column-major; q, r, a and every work
 * buffer are DEVICE pointers owned by the caller (h_wl is pinned host memory); the call is
 * blocking -- it returns after the stream is idle (reference src/blockqr.cu:140); nothing
 * is allocated inside; `a` may be overwritten (it is for n > 64).
 * Re-entrant like the reference's entry point (src/blockqr.cu:394-433): a call keeps no state outside its arguments, so several
 * host threads may factor different matrices at the same time, each with its own buffers and stream (settings made with the
 * tsqr_mi_set_* calls are process-wide; tsqr_mi_last_error / tsqr_mi_last_engine / the event profile are per host thread).
 * h_wl: when it is pinned host memory of at least 8 words the engine writes its status words and completion flag there; a smaller
 * or pageable h_wl (e.g. sized with the reference's batch_size + 1 rule for m <= 128) is left untouched.
 */
#ifndef TSQR_MI_H
#define TSQR_MI_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mtk::qr::compute_mode, same names and order as reference src/blockqr.hpp:12-23 */
enum tsqr_mi_compute_mode {
	TSQR_MI_FP16_NOTC = 0,         /* supported through tsqr_mi_qr_f16: fp16 in / out, fp32-accurate arithmetic in between (the fp32_tc_cor engines) */
	TSQR_MI_FP16_TC_NOCOR = 1,     /* supported through tsqr_mi_qr_f16: fp16 in / out, the single-fp16-product apply engine (fp32_tc_nocor's) */
	TSQR_MI_FP32_NOTC = 2,         /* supported: Q = A*inverse(R) on exact fp32 MFMA (v_mfma_f32_16x16x4_f32) */
	TSQR_MI_FP32_TC_COR = 3,       /* supported: Q = A*inverse(R) on bf16 MFMA with 3-way split error correction (six products) */
	TSQR_MI_FP32_TC_NOCOR = 4,     /* supported: R as fp32_tc_cor; Q = A*inverse(R) on fp16 MFMA, one product, no correction */
	TSQR_MI_MIXED_TC_COR_EMU = 5,
	TSQR_MI_TF32_TC_COR = 6,       /* no xf32 MFMA on gfx950: unsupported */
	TSQR_MI_TF32_TC_COR_EMU = 7,
	TSQR_MI_TF32_TC_NOCOR = 8,
	TSQR_MI_TF32_TC_NOCOR_EMU = 9
};

/* mtk::qr::state_t codes, reference src/blockqr.hpp:27-29, plus one new code */
#define TSQR_MI_SUCCESS               0   /* success_factorization */
#define TSQR_MI_ERROR_INVALID_SIZE    1   /* error_invalid_matrix_size: n > m, m == 0 or n == 0 */
#define TSQR_MI_ERROR_UNSUPPORTED     2   /* new: compute_mode without a gfx950 implementation */
/* negative return values: -(hipError_t) of the failing runtime call */

int tsqr_mi_version(void);
const char* tsqr_mi_last_error(void);

/* replaces mtk::qr::get_working_{q,r,l}_size (reference src/blockqr.hpp:55-57, src/blockqr.cu:34-42).
 * Element counts (float / float / unsigned).  Never smaller than the reference's own formulas. */
size_t tsqr_mi_working_q_size(size_t m, size_t n);
size_t tsqr_mi_working_r_size(size_t m, size_t n);
size_t tsqr_mi_working_l_size(size_t m);
/* the reorthogonalisation scratch of mtk::qr::buffer::allocate (reference src/blockqr.hpp:91): 2*256 + 16 m floats */
size_t tsqr_mi_working_reorth_size(size_t m);
/* mtk::tsqr::get_batch_size_log2 / get_batch_size (reference src/tsqr.cu:39-44) */
size_t tsqr_mi_batch_size_log2(size_t m);
size_t tsqr_mi_batch_size(size_t m);

/*
 * replaces mtk::qr::qr<mode, Reorthogonalize>(q, ldq, r, ldr, a, lda, m, n, wq, wr, reorth_r, d_wl, h_wl, handle)
 * (reference src/blockqr.hpp:142-154).  `stream` is a hipStream_t (takes the place of the stream the
 * reference pulls out of its cublasHandle_t, src/blockqr.cu:58-59).  R: the full n x n upper triangle is
 * written and the strict lower triangle is set to exact zeros.
 * Orthogonality without reorthogonalisation (reorth = 0): one panel (n <= 64, or n <= 128 when the one-panel path accepts) loses
 * ||Q^T Q - I|| like cond(A) * eps32 at most; SEVERAL panels (n > 128, or 64 < n <= 128 on ill-conditioned input) are coupled by
 * block Gram-Schmidt as in the reference (src/blockqr.cu:45-178) and an ill-conditioned input loses orthogonality between the
 * panels -- the return code is still 0, exactly like the reference's.  Measured: 20000 x 128, cond 1e4: 5e-3 (the reference's
 * algorithm: 4.6); cond 1e7: 4.5 (reference: 15.8).  Pass reorth = 1 for O(eps32) at any conditioning up to ~1e8
 * (tests/test_gpu_wide.py, tests/test_gpu_parity.py state the bounds).
 */
int tsqr_mi_qr_f32(int mode, int reorth,
                   float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                   size_t m, size_t n,
                   void* wq, void* wr, float* reorth_w, unsigned* d_wl, unsigned* h_wl,
                   void* stream);
/*
 * Stream-asynchronous form of tsqr_mi_qr_f32.  The reference's mtk::qr::qr only ENQUEUES on the handle's stream and returns
 * (reference src/blockqr.cu:395-449: kernel launches and cuBLAS calls, no synchronisation), so a caller's loop keeps the GPU busy
 * back to back (its speed protocol, src/test.cu:299-309, is such a loop).  tsqr_mi_qr_f32 blocks, because its state includes the
 * verdict of the conditioning check; the pair below gives the reference's overlap back:
 *   tsqr_mi_qr_f32_submit  enqueues the call's first attempt (bf16-split Gram level: Gram pass, Cholesky + verdict, apply pass that
 *                          skips itself on rejection, completion word) and returns;
 *   tsqr_mi_qr_f32_finish  waits for that call, and when the verdict was "rejected" runs the rest of the ladder for it (blocking),
 *                          exactly as tsqr_mi_qr_f32 would have; returns the call's state.
 * Up to TWO calls of a host thread are in flight (their verdict words alternate between the two halves of h_wl); a third submit
 * first waits for the oldest one's verdict.  Calls that have no speculative first attempt (reorth, n > 128, policy != auto,
 * profiling on, h_wl not pinned) are executed inside submit, blocking; finish then just returns their state.
 * Rules: a ticket lives until it is finished, by the thread that submitted it, tickets are finished in submission order; the
 * arguments (A, Q, R, work buffers) stay valid and unmodified by the host until then.  Two calls in flight may share the work
 * buffers (everything runs in stream order) -- but not Q, R or (n > 64) A if the results of both are wanted.
 * Any other entry point of this library called by the thread meanwhile first waits for the verdicts of its tickets in flight.
 */
typedef struct tsqr_mi_ticket {
	int state;                      /* the call's state, valid after finish (or after a submit that ran the call itself) */
	int pending;                    /* 0 done, 1 first attempt in flight, 2 verdict read (ladder not yet run) */
	int slot;                       /* which half of the pinned words the attempt reports to */
	int own_flag;                   /* 1: a completion kernel of its own follows the attempt (always, for tsqr_mi_qr_f32_submit) */
	unsigned seq;                   /* sequence number its completion word will show */
	unsigned verdict;               /* 0 accepted / 1 rejected */
	float scond;                    /* scaled conditioning S the Cholesky kernel reported */
	int mode, reorth;
	float *q, *r, *a;
	size_t ldq, ldr, lda, m, n;
	void *wq, *wr, *stream;
	unsigned *h_wl, *words, *words_dev;   /* caller's pinned words; the pinned words actually used (host / device address) */
} tsqr_mi_ticket;
int tsqr_mi_qr_f32_submit(int mode, int reorth,
                          float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                          size_t m, size_t n,
                          void* wq, void* wr, float* reorth_w, unsigned* d_wl, unsigned* h_wl,
                          void* stream, tsqr_mi_ticket* ticket);
int tsqr_mi_qr_f32_finish(tsqr_mi_ticket* ticket);

/* `count` calls of tsqr_mi_qr_f32 with the same arguments -- the loop of the reference's speed protocol (reference
 * src/test.cu:299-309) on this side of the ABI, so that a host in an interpreted language times what a C++ caller's loop costs.
 * The loop knows that another call follows and keeps the stream fed (tsqr_mi_set_loop_depth selects how far it goes):
 *   depth 1  plain blocking calls, one host round trip between two of them (the latency of a call);
 *   depth 2  two calls in flight: call i + 1 is submitted before call i is finished (submit / finish above); inside the loop the
 *            completion word of call i is raised by the first kernel of call i + 1 instead of a kernel of its own;
 *   depth 3  (default) depth 2, and for full 64-column matrices of 128 k <= 2^20 rows (16-byte aligned, no reorth) the chained
 *            schedule: the R-factor chain of call i (reduction of the Gram partials, Cholesky, verdict -- one workgroup busy, the
 *            rest of the chip idle) runs INSIDE the launch that is the Gram pass of call i + 1, so a call costs its two streaming
 *            passes and nothing else.  Needs count >= 3; wr holds two sets of Gram partials for it.  Likewise for 128 columns (m a
 *            multiple of 64, at least 510 blocks): the two-block factorisation of call i rides in the Gram launch of call i + 1.
 * At every depth every call runs all of its kernels, every verdict is read, a rejected matrix gets its whole ladder, and Q and R
 * are bit for bit those of the blocking call (tests/test_gpu_async.py).  Returns the first non-zero state.
 * The chained schedules launch the Gram pass of call i + 1 before the apply pass of call i; they are taken only when that is the
 * blocking order, i.e. when Q and R do not overlap A (a loop that factors in place, q == a, runs at depth 2: call i + 1 of the
 * blocking loop factors the Q that call i left in A, and so does the stream). */
int tsqr_mi_qr_f32_loop(int count, int mode, int reorth,
                        float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                        size_t m, size_t n,
                        void* wq, void* wr, float* reorth_w, unsigned* d_wl, unsigned* h_wl,
                        void* stream);
void tsqr_mi_set_loop_depth(int depth);   /* 1, 2 or 3 (default); applies to every *_loop and *_batch entry of the process */

/* `count` DIFFERENT matrices of one shape: the caller of mtk::qr::qr with many matrices to factor (the reference's README.md:52-87
 * call inside the caller's loop over its matrices).  q, r, a are HOST arrays of `count` device pointers: call i factors a[i] (m x n,
 * lda) into q[i] (ldq) and r[i] (ldr); the leading dimensions, the shape, the mode and the work buffers are shared.  The calls are
 * issued as the stream tsqr_mi_qr_f32_loop issues (tsqr_mi_set_loop_depth): at depth 3 full 64-column matrices of 128 k <= 2^20 rows
 * (and 128-column matrices of 64 k >= 32640 rows) take the chained schedule -- the R-factor chain of matrix i inside the Gram launch of
 * matrix i + 1 -- as long as no output of call i is an input of call i + 1 (q[i], r[i] clear of a[i + 1]; q[i] == a[i], in place, is
 * fine); everything else, and everything at depth 2, runs two calls in flight in stream order; depth 1 is a loop of blocking calls.
 * Whatever the schedule, every matrix gets the blocking call's result bit for bit: a matrix the conditioning check rejects mid-batch
 * gets its whole ladder (the chained schedule ends at it and a fresh one starts behind it), the accepted ones around it stand.
 * states (optional, `count` ints): the state of every call.  Returns the first non-zero state (0: all factored); a negative value
 * (runtime failure) ends the batch at once.  Blocking: complete on return. */
int tsqr_mi_qr_f32_batch(int count, int mode, int reorth,
                         float* const* q, size_t ldq, float* const* r, size_t ldr, float* const* a, size_t lda,
                         size_t m, size_t n,
                         void* wq, void* wr, float* reorth_w, unsigned* d_wl, unsigned* h_wl,
                         void* stream, int* states);

/*
 * The fp16 I/O modes: replaces mtk::qr::qr<fp16_notc | fp16_tc_nocor, Reorthogonalize> (reference src/blockqr.cu:437-449; io type
 * half, src/tsqr.hpp:38-39).  q, r, a are IEEE binary16 (`half` / `_Float16`), column-major like the fp32 entry; everything else
 * as tsqr_mi_qr_f32.  Arithmetic is this engine's fp32 pipeline with fp16 at the boundary, so the result is the fp16 rounding of an
 * fp32-accurate factorisation (the reference computes these modes IN half): fp16_notc forms Q with the error-corrected bf16x3
 * products of fp32_tc_cor, fp16_tc_nocor with one fp16 product (inverse(R) rounded to fp16, no correction).  One panel (n <= 64),
 * no reorthogonalisation, 16-byte aligned columns (base pointers, lda and ldq multiples of 8) and a matrix the bf16-split level
 * accepts take the NATIVE path: the Gram pass uses the halves of A as MFMA operands (exact products), the apply pass reads and
 * writes halves -- half the bytes of the fp32 call.  Everything else is widened into the work buffer, factored by tsqr_mi_qr_f32
 * and narrowed on the way out (two conversion passes).  A is not modified.  R entries beyond the fp16 range (column norms > 65504) become infinities, as a half-typed R does in the reference.
 * Work space: wq of tsqr_mi_working_q_size_f16(m, n) FLOATS (4-byte units: the fp32 call's own space + the widened A, Q and R),
 * wr of tsqr_mi_working_r_size_f16(m, n) floats, d_wl / h_wl as for the fp32 entry.  n <= m, any n.
 */
size_t tsqr_mi_working_q_size_f16(size_t m, size_t n);
size_t tsqr_mi_working_r_size_f16(size_t m, size_t n);
int tsqr_mi_qr_f16(int mode, int reorth,
                   void* q, size_t ldq, void* r, size_t ldr, const void* a, size_t lda,
                   size_t m, size_t n,
                   void* wq, void* wr, void* reorth_w, unsigned* d_wl, unsigned* h_wl,
                   void* stream);
/* `count` calls (the reference's speed loop, src/test.cu:299-309) as tsqr_mi_qr_f32_loop: calls the native path takes (16 < n <= 64, no reorth,
 * aligned halves) are issued as a stream -- two in flight (loop depth >= 2), for n = 64 and count >= 3 also the chained schedule (depth 3) --,
 * everything else as blocking calls; the same halves either way */
int tsqr_mi_qr_f16_loop(int count, int mode, int reorth,
                        void* q, size_t ldq, void* r, size_t ldr, const void* a, size_t lda,
                        size_t m, size_t n,
                        void* wq, void* wr, void* reorth_w, unsigned* d_wl, unsigned* h_wl,
                        void* stream);

/* `count` DIFFERENT half-typed matrices of one shape: tsqr_mi_qr_f32_batch for the fp16 I/O modes (q, r, a: host arrays of device pointers to halves).
 * Calls the native path takes (16 < n <= 64, no reorth, aligned halves, accepted by the bf16-split level) are issued as a stream -- two in flight, for
 * n = 64 the chained schedule while q[i], r[i] are clear of a[i + 1] and a matrix is small enough for two to share the Infinity Cache --, everything else
 * as blocking calls; a matrix rejected mid-stream gets its whole ladder (conversion path), the accepted ones around it stand.  The same halves as
 * `count` calls of tsqr_mi_qr_f16.  states (optional): the state of every call.  Returns the first non-zero state. */
int tsqr_mi_qr_f16_batch(int count, int mode, int reorth,
                         void* const* q, size_t ldq, void* const* r, size_t ldr, const void* const* a, size_t lda,
                         size_t m, size_t n,
                         void* wq, void* wr, void* reorth_w, unsigned* d_wl, unsigned* h_wl,
                         void* stream, int* states);

/*
 * Staged entry points used by the row-partitioned multi-GPU path: every rank
 * factors its row block, the n x n R factors are all-gathered (RCCL), every rank folds the stack,
 * and forms its rows of Q.  n <= 64.
 */
/* R (n x n, ldr) of the local block a (m x n); m may be < n (stack of R factors is fine too). */
int tsqr_mi_local_r_f32(float* r, size_t ldr, const float* a, size_t lda, size_t m, size_t n,
                        void* wq, void* wr, void* stream);
/* q (m x n) = a (m x n) * inverse(r); q may alias a. */
int tsqr_mi_apply_rinv_f32(int mode, float* q, size_t ldq, const float* a, size_t lda,
                           const float* r, size_t ldr, size_t m, size_t n,
                           void* wq, void* stream);
/* Staged Gram engine for the row-partitioned path.  level 2 = bf16-split Gram matrix, 1 = fp64 Gram matrix.
 *   tsqr_mi_gram_f32: gsum (tsqr_mi_gram_elems(n) doubles, MFMA-accumulator order) = Gram tiles of the local block;
 *                     the caller sums gsum over the ranks (all-reduce) before
 *   tsqr_mi_chol_f32: R = chol(G) into r, inverse(R) into the work buffer, *status_out = 0 accepted / 1 rejected
 *                     (blocking); m must be the same value as in the other staged calls (it fixes the work-buffer layout)
 *   tsqr_mi_apply_z_f32: q = a * inverse(R) with the inverse left in wq by tsqr_mi_chol_f32. */
size_t tsqr_mi_gram_elems(size_t n);
int tsqr_mi_gram_f32(int level, double* gsum, const float* a, size_t lda, size_t m, size_t n, void* wq, void* wr, void* stream);
int tsqr_mi_chol_f32(int level, float* r, size_t ldr, const double* gsum, size_t m, size_t n, void* wq, unsigned* status_out, void* stream);
/* level 3 = shifted Cholesky (G + s I, s = 11 (m n + n (n+1)) 2^-53 trace(G)) of an fp64 Gram matrix: always accepted for finite
 * input; the caller must follow with one more plain sweep on the resulting Q and multiply the R factors (tsqr_mi_rmul_f32).
 * status_out == NULL above: asynchronous (no stream sync); the verdict is then read with tsqr_mi_chol_status (blocking) -- lets a
 * caller enqueue tsqr_mi_apply_z_f32 speculatively behind the Cholesky and pay one synchronisation per sweep. */
int tsqr_mi_chol_status(const void* wq, size_t m, size_t n, unsigned* status_out, void* stream);
/* blocks until everything enqueued on `stream` so far has completed (spin on a pinned completion word; stream sync as fallback) */
int tsqr_mi_stream_wait(void* stream);
int tsqr_mi_apply_z_f32(int mode, float* q, size_t ldq, const float* a, size_t lda, size_t m, size_t n, void* wq, void* stream);
/* r (n x n) <- r2 * r (upper triangular product, used after a reorthogonalisation sweep) */
int tsqr_mi_rmul_f32(float* r, size_t ldr, const float* r2, size_t ldr2, size_t n, void* wq, void* stream);

/* Row-partitioned TSQR (the reference has no multi-GPU path): one call per rank, every rank passes its own
 * row block (m_local may differ between ranks, 1 <= m_local on every rank -- a rank that returns 1 for an empty block would leave
 * the others waiting in their exchange; m_local < n is fine), r is bitwise identical on all ranks on return.  n <= 64 is one panel;
 * n > 64 runs 64-column panels whose coupling coefficients are all-reduced like the Gram tiles (a is then overwritten).  It is the SAME ladder as
 * tsqr_mi_qr_f32 with two exchange hooks switched on, all enqueued on `stream` with no host wait before the end of the call:
 *   Gram levels:        all-reduce (sum) of the Gram tiles + the local row count: <= 2561 doubles.  Every rank then factors the
 *                       same matrix with thresholds from the same (global) row count, so the accept / reject decisions agree on
 *                       all ranks by construction -- a NaN anywhere reaches everyone through the sum;
 *   Householder engine: all-gather of the n x n local R factors, every rank folds the same (nranks n) x n stack.
 * Work buffers: tsqr_mi_working_{q,r}_size_dist(m_local, n, nranks) elements; gather_buf: nranks * min(n, 64)^2 floats.
 * The communicator and the collectives that run on it must come from ONE RCCL instance (a process may have two mapped: torch
 * bundles a copy, /opt/rocm holds another), so the library never searches for librccl itself:
 * tsqr_mi_qr_f32_dist_fn: the caller passes its ncclComm_t (as void*) together with the addresses of ncclAllReduce and ncclAllGather
 *   taken from the library that created it (dlsym on that handle; Python: ctypes.cast(lib.ncclAllReduce, c_void_p)).
 * tsqr_mi_qr_f32_dist: for a caller that LINKS RCCL (C++): the two entry points are looked up in the global symbol scope
 *   (dlsym(RTLD_DEFAULT, ...)) -- the copy the caller's own ncclCommInitRank came from; TSQR_MI_ERROR_UNSUPPORTED when they are not there.
 * tsqr_mi_qr_f32_dist_cb: caller-supplied collectives (blocking or stream-ordered; in place sum / gather in rank order), e.g.
 *   torch.distributed over gloo -- what the multi-process tests use.
 * *_loop: `count` calls as a stream (see tsqr_mi_qr_f32_loop); every rank must pass the same count and run at the same loop depth -- the
 *   ranks take the same verdicts, hence the same path through the loop and the same order of collectives.  Depth 3, count >= 3: when
 *   EVERY rank holds a full 64-column block of 128 k <= 2^20 rows the Cholesky launch of call i rides in the Gram launch of call i + 1
 *   and allreduce(i + 1) is enqueued before apply(i).  The ranks agree on that without a collective of its own: the first all-reduce
 *   of the loop call (the Gram tiles of call 0) carries one more double, 1.0 from every eligible rank; nranks <= 255. */
size_t tsqr_mi_working_q_size_dist(size_t m_local, size_t n, int nranks);
size_t tsqr_mi_working_r_size_dist(size_t m_local, size_t n, int nranks);
int tsqr_mi_qr_f32_dist(int mode, int reorth,
                        float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                        size_t m_local, size_t n,
                        void* wq, void* wr, float* gather_buf /* nranks*n*n floats */,
                        void* nccl_comm, int nranks, void* stream);
int tsqr_mi_qr_f32_dist_fn(int mode, int reorth,
                           float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                           size_t m_local, size_t n,
                           void* wq, void* wr, float* gather_buf /* nranks*n*n floats */,
                           void* nccl_comm, void* nccl_allreduce_fn, void* nccl_allgather_fn, int nranks, void* stream);
int tsqr_mi_qr_f32_dist_fn_loop(int count, int mode, int reorth,
                                float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                                size_t m_local, size_t n,
                                void* wq, void* wr, float* gather_buf,
                                void* nccl_comm, void* nccl_allreduce_fn, void* nccl_allgather_fn, int nranks, void* stream);
/* *_batch: `count` DIFFERENT row-partitioned matrices of one shape (host arrays of device pointers; one block height per rank for all of them), issued as
 * the stream of calls of the *_loop entries; every rank passes the same count and operands of the same eligibility.  states (optional): per call. */
int tsqr_mi_qr_f32_dist_fn_batch(int count, int mode, int reorth,
                                 float* const* q, size_t ldq, float* const* r, size_t ldr, float* const* a, size_t lda,
                                 size_t m_local, size_t n,
                                 void* wq, void* wr, float* gather_buf,
                                 void* nccl_comm, void* nccl_allreduce_fn, void* nccl_allgather_fn, int nranks, void* stream, int* states);
typedef int (*tsqr_mi_allreduce_f64_cb)(void* user, double* buf /* device */, size_t count, void* stream);
typedef int (*tsqr_mi_allgather_f32_cb)(void* user, const float* send /* device */, float* recv /* device, nranks*count */, size_t count, void* stream);
int tsqr_mi_qr_f32_dist_cb(int mode, int reorth,
                           float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                           size_t m_local, size_t n,
                           void* wq, void* wr, float* gather_buf,
                           tsqr_mi_allreduce_f64_cb allreduce, tsqr_mi_allgather_f32_cb allgather, void* user, int nranks, void* stream);
int tsqr_mi_qr_f32_dist_cb_loop(int count, int mode, int reorth,
                                float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                                size_t m_local, size_t n,
                                void* wq, void* wr, float* gather_buf,
                                tsqr_mi_allreduce_f64_cb allreduce, tsqr_mi_allgather_f32_cb allgather, void* user, int nranks, void* stream);
int tsqr_mi_qr_f32_dist_cb_batch(int count, int mode, int reorth,
                                 float* const* q, size_t ldq, float* const* r, size_t ldr, float* const* a, size_t lda,
                                 size_t m_local, size_t n,
                                 void* wq, void* wr, float* gather_buf,
                                 tsqr_mi_allreduce_f64_cb allreduce, tsqr_mi_allgather_f32_cb allgather, void* user, int nranks, void* stream, int* states);

/* Harness support (reference src/validation.cu:43-127, src/test.cu:147-165): accuracy metrics evaluated on the device in fp64.
 * scratch: n*n + 8 doubles of device memory.  out_host[0..4] = ||Q^T Q - I||_F^2, its diagonal part, its off-diagonal part,
 * ||Q R - A||_F^2, ||A||_F^2 (the last two only when r and a are given).  gram_out_host: optional n*n doubles receiving Q^T Q. */
int tsqr_mi_validate_f32(const float* q, size_t ldq, const float* r, size_t ldr, const float* a, size_t lda,
                         size_t m, size_t n, double* scratch, double* out_host, double* gram_out_host, void* stream);

/* Optional timing of the engine's kernels with HIP events recorded on the caller's stream.  Classes:
 * 0 first fold level (streams A), 1 fold-tree levels, 2 triangular inverse, 3 apply (Q = A*inverse(R)),
 * 4 inter-panel coupling (n > 64), 5 other, 6 Gram matrix, 7 Gram reduction + Cholesky + inverse.  read() returns accumulated milliseconds and launch counts
 * since enable(1); call it after the blocking qr call(s). */
void tsqr_mi_profile_enable(int on);
int tsqr_mi_profile_read(double* ms, long* launches, int max_classes);

/* R-factor engine policy (Q is always formed by apply: Q = A * inverse(R) on the MFMA units).
 *   0 auto (default), every mode: Gram engine, R = chol(A^T A).  First the bf16x3-split MFMA Gram matrix (exact products, fp64
 *                     accumulation across K-steps, memory-bound), accepted when every Cholesky pivot keeps >= 2^-5 of its diagonal
 *                     entry and the scaled conditioning S = ||D inverse(R)||_F^2 / n stays below min(128, max(4, 0.12 sqrt(rows)));
 *                     else the fp64-MFMA Gram matrix, accepted down to a pivot ratio of 2^-40 (cond(A) up to ~1e6); else the
 *                     shifted Cholesky QR step on that fp64 Gram matrix + one plain fp64 sweep in place; Householder TSQR last.
 *                     The compute mode selects the MFMA engine of the apply pass (exact fp32 / bf16x3 split / single fp16 product).
 *   1 always Householder TSQR.   2 always fp64 Gram (no fallback).   3 always bf16-split Gram (no check; tests only).
 *   4 auto without the bf16-split level.
 *   5 auto, but 64 < n <= 128 goes straight to the 64-column panel path (policy 0 first tries all n columns as ONE Cholesky-QR
 *     panel at the bf16-split level: one pass for the 128 x 128 Gram matrix, one for Q; same acceptance rule; A stays intact when it
 *     is accepted, and when it is rejected the panel path runs on the untouched input).
 * tsqr_mi_last_engine(): 0 Householder, 1 fp64 Gram, 2 Gram rejected -> Householder, 3 bf16-split Gram,
 * 4 Gram rejected -> shifted Cholesky QR (shifted fp64 Cholesky + one plain fp64 sweep in place),
 * 5 bf16-split Gram over all n <= 128 columns at once (one panel). */
void tsqr_mi_set_policy(int policy);
int tsqr_mi_last_engine(void);

/* tuning knobs (0 = keep default): waves targeted by the first fold level, chunks folded per wave on tree levels */
void tsqr_mi_set_tuning(int level0_waves, int tree_chunks_per_wave);
/* waves of the Gram kernel / of the apply kernel (apply: persistent grid of apply_waves/4 workgroups; default = all that are resident
 * at once).  Work-buffer sizes follow the Gram setting: set it before allocating. */
void tsqr_mi_set_tuning2(int gram_waves, int apply_waves);

#ifdef __cplusplus
}
#endif
#endif /* TSQR_MI_H */
