// tsqr_mi.hip -- host orchestration and the extern "C" ABI of libtsqr_mi.so (declared in include/tsqr_mi.h).
//
// Host-side counterpart of the reference's block_qr_core / block_qr_reorthogonalization_core
// (reference src/blockqr.cu:45-178, 180-390) and tsqr16_geq32 (reference src/tsqr.cu:1064-1279),
// re-designed for MI355X:
//   * panel width 64 instead of 16: for n <= 64 there is no inter-panel coupling at all;
//   * R from the Gram matrix of the panel (bf16x3-split MFMA, then fp64 MFMA, then shifted Cholesky QR) with a streaming
//     Householder TSQR (fold_kernel + fold tree over the per-wave R factors: the R-stack reduction of the reference) as the
//     engine of last resort / on request;
//   * Q = A * inverse(R) on the MFMA units (apply_wg_kernel): "indirect TSQR".  Its loss of orthogonality grows like
//     cond(A)*eps, slower than the reference's 16-wide block Gram-Schmidt without reorthogonalisation; Reorthogonalize=true runs
//     a second sweep on Q (R <- R2*R), which restores ||Q^T Q - I|| to O(eps) as the reference's BCGS2 does;
//   * the same ladder serves a row-partitioned matrix (one rank per GPU): the only exchange is an all-reduce of the n x n Gram
//     tiles (+ the row count) or, for the Householder engine, an all-gather of the local R factors;
//   * no host synchronisation inside a sweep; the call is blocking like the reference's (src/blockqr.cu:140).
//
// Re-entrancy: like the reference's entry point (src/blockqr.cu:394-433) a call keeps no state outside its arguments: everything
// a call needs lives in a Ctx on the caller's stack, settings are process-wide atomics that a call snapshots once, per-device
// launch attributes are cached in lock-free tables, diagnostics (last error / last engine / event profile) are per host thread.
// Two host threads may factor different matrices with different buffers and streams at the same time.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>

#include "../../include/tsqr_mi.h"
#include "tsqr_kernels.hip"
#include "tsqr_wide.hip"
#include "validate.hip"

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// process-wide settings (atomics: written by the tsqr_mi_set_* calls, snapshotted once per call) and per-thread diagnostics
// ---------------------------------------------------------------------------------------------------------------------------
int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

struct Settings {
	std::atomic<int> policy{0};          // 0 auto, 1 Householder, 2 Gram without check/fallback
	std::atomic<int> gram_level{2};      // first Gram level tried: 2 bf16-split (then fp64), 1 fp64 only
	std::atomic<int> level0_waves{2048}, tree_cpw{4}, gram_waves{2048};
	std::atomic<int> wide{1};            // 64 < n <= 128: one Cholesky-QR panel of up to 128 columns first (policy 5 turns it off)
	std::atomic<int> apply_wgs{0};       // workgroups of the apply pass; 0: as many as are resident at once (tsqr_mi_set_tuning2)
	std::atomic<int> loop_depth{3};      // *_loop entries: 1 blocking calls, 2 two calls in flight, 3 also the chained schedule (tsqr_mi_set_loop_depth)
	// the two environment switches that are left (read once at load): the floor of the bf16-split level's bound on the scaled
	// conditioning S, and a diagnostic print of every Cholesky verdict
	const float bf16_scond_floor = (float)env_int("TSQR_MI_BF16_MAX_SCOND", 4);
	// chained schedules over DIFFERENT matrices: taken when one matrix is at most this many MiB.  The Gram pass of matrix i + 1 runs
	// between the Gram pass and the apply pass of matrix i; both fit the 256 MiB Infinity Cache only up to ~half of it each -- beyond
	// that the apply pass of matrix i finds its A evicted and streams it from HBM again (measured at 2^20 x 64, 256 MiB a matrix:
	// 0.181 ms per call chained against 0.163 in stream order, profiles/r04_experiment_log.md).  A loop over ONE matrix is not affected.
	const int chain_max_mib = env_int("TSQR_MI_CHAIN_MAX_MIB", 112);
	const int debug = env_int("TSQR_MI_DEBUG", 0);
};
Settings g_set;
std::atomic<unsigned> g_seq{0};                        // sequence numbers of the completion flags (any thread)

thread_local std::string t_last_error;
thread_local int t_last_engine = 0;   // 0 Householder TSQR, 1 fp64 Gram/Cholesky, 2 Gram broke down -> Householder, 3 bf16-split Gram, 4 shifted

// ---- optional per-kernel-class timing with HIP events on the caller's stream (bench.py's roofline leg); per host thread ----
enum { KC_FOLD0 = 0, KC_TREE = 1, KC_TRINV = 2, KC_APPLY = 3, KC_COUPLE = 4, KC_MISC = 5, KC_GRAM = 6, KC_CHOL = 7, KC_COUNT = 8 };
struct Prof {
	bool on = false;
	static constexpr int MAXEV = 4096;
	hipEvent_t ev[2 * MAXEV];
	int cls[MAXEV];
	int n = 0;
	bool created = false;
	double ms[KC_COUNT] = {};
	long launches[KC_COUNT] = {};
};
thread_local Prof t_prof;
struct ProfScope {                 // brackets one kernel launch (or a short launch group) with two events
	int idx = -1; hipStream_t st;
	ProfScope(int kc, hipStream_t s) : st(s) {
		if (t_prof.on && t_prof.n < Prof::MAXEV) {
			idx = t_prof.n++;
			t_prof.cls[idx] = kc;
			(void)hipEventRecord(t_prof.ev[2 * idx], st);
		}
	}
	~ProfScope() { if (idx >= 0) (void)hipEventRecord(t_prof.ev[2 * idx + 1], st); }
};
void prof_collect() {              // after the stream is idle
	for (int i = 0; i < t_prof.n; i++) {
		float t = 0.0f;
		if (hipEventElapsedTime(&t, t_prof.ev[2 * i], t_prof.ev[2 * i + 1]) == hipSuccess) {
			t_prof.ms[t_prof.cls[i]] += t;
			t_prof.launches[t_prof.cls[i]] += 1;
		}
	}
	t_prof.n = 0;
}

constexpr size_t PW = 64;          // panel width
constexpr int MAX_DEV = 64;

inline int fail(hipError_t e, const char* what) {
	t_last_error = std::string(what) + ": " + hipGetErrorString(e);
	return -(int)e;
}
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_, #expr); } while (0)

inline size_t cdiv(size_t a, size_t b) { return (a + b - 1) / b; }
inline size_t np_of(size_t n) { return 16 * cdiv(std::min(n, PW), 16); }
inline int cur_device() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < MAX_DEV) ? d : 0; }

// per-kernel, per-device launch attributes: hipFuncSetAttribute once per (kernel instance, device); lock-free (the call is idempotent)
struct DevOnce {
	std::atomic<unsigned long long> mask{0};
	bool need(int dev) const { return !(mask.load(std::memory_order_acquire) >> dev & 1ull); }
	void done(int dev) { mask.fetch_or(1ull << dev, std::memory_order_release); }
};

// ---- reference-compatible size rules (reference src/tsqr.cu:39-60, src/blockqr.cu:34-42) ----
size_t ref_bs_log2(size_t m) {
	const unsigned c = (unsigned)std::ceil(std::log2((float)m));
	return (size_t)(std::max(5u, c) - 5u);
}
size_t ref_bs(size_t m) { return (size_t)1 << ref_bs_log2(m); }
size_t ref_wq(size_t m, size_t n) { n = std::min<size_t>(16, n); return n * m + 2 * n * n * (ref_bs(m) - 1); }
size_t ref_wr(size_t m, size_t n) { n = std::min<size_t>(16, n); const size_t b = ref_bs(m); return n * n * b + n * n * b / 2; }

// ---- fold plan: level 0 over the matrix, then levels over the stacks of per-wave R factors ----
struct Plan {
	size_t NP;
	int nlevels;
	size_t rows[24];     // source rows of each level
	int nch[24], cpw[24], nw[24];
	size_t stack_a;      // floats needed in wr (levels 0, 2, 4, ... write here)
	size_t stack_b;      // floats needed in wq (levels 1, 3, ... write here)
};

Plan make_plan(size_t m, size_t n) {
	Plan p{};
	p.NP = np_of(n);
	const int level0_waves = g_set.level0_waves.load(), tree_cpw = g_set.tree_cpw.load();
	size_t rows = m;
	int lv = 0;
	// a wave turns cpw*64 source rows into NP rows: cpw*64 >= 2*NP keeps every level shrinking
	const size_t cpw_min = std::max<size_t>(1, cdiv(2 * p.NP, 64));
	for (;;) {
		const size_t nch = cdiv(rows, 64);
		// tree levels over 64-row triangular blocks are binary: the first block is copied into R (FoldArgs::tri_init), one fold per level
		size_t cpw = (lv == 0) ? std::max(cpw_min, cdiv(nch, (size_t)level0_waves))
		                       : (p.NP == 64 ? (size_t)2 : std::max(cpw_min, (size_t)tree_cpw));
		size_t nw = cdiv(nch, cpw);
		if (nw <= 1 || nw * p.NP >= rows) { nw = 1; cpw = nch; }
		p.rows[lv] = rows; p.nch[lv] = (int)nch; p.cpw[lv] = (int)cpw; p.nw[lv] = (int)nw;
		if (nw > 1) {
			const size_t sz = nw * p.NP * p.NP;
			if (lv % 2 == 0) p.stack_a = std::max(p.stack_a, sz); else p.stack_b = std::max(p.stack_b, sz);
		}
		lv++;
		if (nw == 1) break;
		rows = nw * p.NP;
	}
	p.nlevels = lv;
	return p;
}

// Gram engine geometry: waves / workgroups of the Gram kernels and the size of their per-workgroup partials (in floats)
struct GramPlan { int nch, cpw, nwaves, nblocks, ntri; size_t part_floats; };
GramPlan gram_plan(size_t m, size_t n) {
	GramPlan g{};
	const size_t NT = np_of(n) / 16;
	g.nch = (int)cdiv(m, 64);
	g.cpw = (int)std::max<size_t>(1, cdiv((size_t)g.nch, (size_t)g_set.gram_waves.load()));
	g.nwaves = (int)cdiv((size_t)g.nch, (size_t)g.cpw);
	g.nblocks = (g.nwaves + 3) / 4;
	g.ntri = (int)(NT * (NT + 1) / 2);
	g.part_floats = (size_t)g.nblocks * g.ntri * 256 * 2;
	return g;
}

// Coupling partials (cross_kernel: 16 tiles x 256 doubles per workgroup).  A launch covers as many trailing panels as the work space holds
// workgroup partials for -- every panel with the full workgroup count of one panel pair, so the grouping of the sums (and with it every bit
// of S) is that of rounds 1-3: up to CROSS_SLOT_CAP slots (32 MiB), never fewer than one panel's.
constexpr size_t CROSS_SLOT_CAP = 1024;
size_t cross_slots(size_t m, size_t n) {
	if (n <= PW) return 0;
	const size_t nb = (size_t)gram_plan(m, PW).nblocks, ntr = cdiv(n, PW) - 1;
	return std::max(nb, std::min(ntr * nb, CROSS_SLOT_CAP));
}
// extra work space of the one-panel path for 64 < n <= 128 (offsets in floats from WqLayout::wide; the doubles first, 16-byte aligned):
// [summed tiles 36*256 + row count (+pad)] | [Z 128 x 128 fp32][Z22 fp32 4096]
constexpr size_t WIDE_G_DOUBLES = 36 * 256 + 8;
constexpr size_t WIDE_OFF_ZW = 2 * WIDE_G_DOUBLES, WIDE_OFF_ZF2 = WIDE_OFF_ZW + 128 * 128, WIDE_FLOATS = WIDE_OFF_ZF2 + 4096;
constexpr int WIDE_MAX_WGS = 255;                       // gram_wide_kernel: one eight-wave workgroup per CU -- on 255 of the 256 CUs: in a stream of calls the
                                                        // last CU holds the factorisation of the call before (gram_wide_chain_kernel), and the partials must
                                                        // be those of the blocking call
inline size_t wide_part_floats(size_t m) { return (std::min<size_t>((m + 63) / 64, WIDE_MAX_WGS) + 1) * 36 * 256 * 2; }

// layout of wq (floats): [stack_b][Z: 4096][S: 4096][R1 copy: n*n][R2: n*n][r3, r4, r5, r6: 4096 each][summed tiles + row count][status]
constexpr size_t GSUM_DOUBLES = 16 * 256 + 8;          // 16 tiles (coupling) or 10 (Gram) + the row-count word of a row-partitioned run
struct WqLayout { size_t z, s, r1, r2, r3, r4, r5, r6, gsum, status, wide, smulti, gmulti, total; };
WqLayout wq_layout(size_t m, size_t n) {
	const Plan p = make_plan(m, n);
	WqLayout L{};
	size_t o = p.stack_b;
	o = (o + 63) & ~(size_t)63;
	L.z = o; o += 4096;
	L.s = o; o += 4096;
	L.r1 = o; o += n * n;
	L.r2 = o; o += n * n;
	L.r3 = o; o += 4096;                                 // panel-local R1, R2 of the shifted-Cholesky two-step (<= 64 x 64 each)
	L.r4 = o; o += 4096;
	L.r5 = o; o += 4096;                                 // third sweep's factor and R2 R1 of the shifted CholeskyQR3 of a reorthogonalised call (qr_core)
	L.r6 = o; o += 4096;
	o = (o + 63) & ~(size_t)63;
	L.gsum = o; o += 2 * GSUM_DOUBLES;
	L.status = o; o += 64;
	L.wide = o;
	if (n > PW) o += WIDE_FLOATS;                        // the one-panel path for 64 < n <= 128 (sweep_wide); n > 128: 128-column blocks of the panel loop (sweep)
	// several 64-column panels (round 4, right-looking coupling): the operands -S_j of the update and the summed coupling tiles of ALL trailing
	// panels of a finished panel (one slice of 4096 floats / CROSS_GSTRIDE doubles per trailing panel)
	o = (o + 63) & ~(size_t)63;
	L.smulti = o; L.gmulti = o;
	if (n > PW) {
		const size_t ntr = cdiv(n, PW) - 1;
		o += ntr * 4096;
		L.gmulti = o; o += ntr * 2 * (size_t)tsqrmi::CROSS_GSTRIDE;
	}
	L.total = o;
	return L;
}

// ---------------------------------------------------------------------------------------------------------------------------
// exchange hooks of a row-partitioned call (one rank per GPU).  Either RCCL entry points (resolved with dlopen) on the caller's
// ncclComm_t and stream, or caller-supplied callbacks (tests: torch.distributed over gloo on one GPU).
// ---------------------------------------------------------------------------------------------------------------------------
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
struct Comm {
	int nranks = 1;
	void* nccl = nullptr;                                // ncclComm_t
	nccl_allreduce_t nccl_allreduce = nullptr;
	nccl_allgather_t nccl_allgather = nullptr;
	tsqr_mi_allreduce_f64_cb cb_allreduce = nullptr;
	tsqr_mi_allgather_f32_cb cb_allgather = nullptr;
	void* cb_user = nullptr;
	float* gather_buf = nullptr;                         // nranks * n * n floats (Householder engine only)
	bool active() const { return nccl != nullptr || cb_allreduce != nullptr; }
	int allreduce_f64(double* buf, size_t count, hipStream_t st) const {
		if (nccl_allreduce) return nccl_allreduce(buf, buf, count, 8 /* ncclFloat64 */, 0 /* ncclSum */, nccl, st) == 0 ? 0 : -1;
		if (cb_allreduce) return cb_allreduce(cb_user, buf, count, st) == 0 ? 0 : -1;
		return 0;
	}
	int allgather_f32(const float* send, float* recv, size_t count, hipStream_t st) const {
		if (nccl_allgather) return nccl_allgather(send, recv, count, 7 /* ncclFloat32 */, nccl, st) == 0 ? 0 : -1;
		if (cb_allgather) return cb_allgather(cb_user, send, recv, count, st) == 0 ? 0 : -1;
		return -1;
	}
};
// ncclAllReduce / ncclAllGather of a caller that LINKS RCCL itself (C++): the symbols are then in the global scope and this lookup
// returns the very copy the caller created its communicator with.  No soname search and no dlopen: a second RCCL instance (torch
// bundles one, /opt/rocm has another) must never meet a communicator of the first.  Callers that loaded RCCL privately (Python
// ctypes: RTLD_LOCAL) pass the entry points of that very handle instead (tsqr_mi_qr_f32_dist_fn).
void* rccl_symbol(const char* name) { return dlsym(RTLD_DEFAULT, name); }

// ---------------------------------------------------------------------------------------------------------------------------
// per-call context
// ---------------------------------------------------------------------------------------------------------------------------
struct HostSig { unsigned* host = nullptr; unsigned* dev = nullptr; };
struct Ctx {
	hipStream_t st = nullptr;
	int dev = 0;
	float* wq = nullptr; float* wr = nullptr;
	WqLayout L{};
	size_t cross_slots = 0;                              // workgroup partials of the coupling launches the work buffer wr holds (cross_slots())
	HostSig hsig;                                        // pinned words the device can write: status words [0..2] / [4..6], completion flag [3]
	int policy = 0, gram_level = 2;
	int min_level = 2;                                   // lowest R-factor engine level used (2 bf16 Gram, 1 fp64 Gram, 0 Householder)
	bool used_shift = false, used_householder = false;
	bool wide = true;                                    // 64 < n <= 128: try the one-panel path first
	int slot = 0, prev_slot = -1;                        // status slot of the sweep being enqueued / of the sweep it depends on (-1: none)
	double* gramq_part = nullptr;                        // non-null: apply launches write per-workgroup Gram partials of their output there
	int gramq_cap = 0, gramq_nparts = 0;
	bool gramq_ready = false;                            // the next bf16-level Gram request can skip its pass (partials are in place)
	Comm comm;
	bool fold_cor = false;                               // Householder engine: block reflectors on the error-corrected bf16x3 MFMA (fp32_tc_cor)
	bool resume_accepted = false;                        // tsqr_mi_qr_f32_finish: the first attempt ran already and was accepted with
	float resume_scond = 0.0f;                           // this scaled conditioning (only the n <= 16 second sweep is left to do)
	int start_level = -1;                                // tsqr_mi_qr_f32_finish: the ladder resumes at this level (the ones above were rejected)
	unsigned* announce_word = nullptr;                   // completion word of the call in front of this one, raised by this call's first
	unsigned announce_seq = 0;                           // Gram kernel (consumed by the launch that carries it)
	bool q_for_next_sweep = false;                       // apply launches write Q for a sweep that reads it back at once: plain stores, ascending block order (ApplyArgs)
	int chol_relax = 0;                                  // the next bf16-level Cholesky launches use the relaxed rule (CholArgs::relax: another sweep follows)
	int chol_retry_shift = 0;                            // ... and factor a rejected matrix again at once, shifted (CholArgs::retry_shift)
	double rows_global = 0.0;                            // host's view of the global row count (the device thresholds of a row-partitioned
	                                                     // call use the all-reduced count instead)
	unsigned* status_dev(int s) const { return reinterpret_cast<unsigned*>(wq + L.status) + 16 * s; }
	double* gsum() const { return reinterpret_cast<double*>(wq + L.gsum); }
};

// library-owned pinned words, one set per host thread (callers whose h_wl is not pinned or too small; staged API)
struct OwnPinned {
	unsigned* host = nullptr; unsigned* dev = nullptr;
	~OwnPinned() { if (host) (void)hipHostFree(host); }
	bool get() {
		if (host) return dev != nullptr;
		if (hipHostMalloc(reinterpret_cast<void**>(&host), 64, hipHostMallocDefault) != hipSuccess) { host = nullptr; (void)hipGetLastError(); return false; }
		if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0) != hipSuccess) { dev = nullptr; (void)hipGetLastError(); return false; }
		return true;
	}
};
thread_local OwnPinned t_own;

// Is h_wl pinned host memory the device can write, and does it hold the eight words the engine uses?  mtk::qr::buffer allocates it
// with hipHostMalloc, but a caller that sized it with the REFERENCE's get_working_l_size (batch_size + 1 words: 2..5 for m <= 128)
// must not be written past its end -- then (and for pageable memory) the thread's own pinned words are used.
// (the answer for the last h_wl of this host thread is remembered: the query sits in front of the first launch of every call, and a
// caller's loop passes the same buffer every time.  A buffer freed and re-allocated at the same address as PAGEABLE memory between two
// calls would be mistaken for the pinned one; mtk::qr::buffer always allocates hl pinned.)
struct HostSigCache { unsigned* h_wl = nullptr; unsigned* dev = nullptr; };
thread_local HostSigCache t_hsig_cache;
void resolve_host_sig(Ctx& c, unsigned* h_wl, size_t m) {
	c.hsig = HostSig{};
	if (h_wl && ref_bs(m) + 1 >= 8) {
		if (t_hsig_cache.h_wl == h_wl) { c.hsig.host = h_wl; c.hsig.dev = t_hsig_cache.dev; return; }
		hipPointerAttribute_t at{};
		if (hipPointerGetAttributes(&at, h_wl) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) {
			c.hsig.host = h_wl; c.hsig.dev = reinterpret_cast<unsigned*>(at.devicePointer);
			t_hsig_cache.h_wl = h_wl; t_hsig_cache.dev = c.hsig.dev;
			return;
		}
		(void)hipGetLastError();
	}
	if (t_own.get()) { c.hsig.host = t_own.host; c.hsig.dev = t_own.dev; }
}

// End of a call on the fast path: a one-thread kernel behind the last kernel raises word 3 of the pinned words; the host spins on
// it (about 5 us cheaper than hipStreamSynchronize, tools/launch_cost.py, and free of its sporadic OS wake-up stalls) and polls
// the stream now and then so that a failed launch cannot hang the caller.  Returns 1 when the flag path is not available.
int signal_and_wait(Ctx& c) {
	if (!c.hsig.dev || t_prof.on) return 1;
	unsigned seq = ++g_seq;
	if (seq == 0) seq = ++g_seq;
	volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(c.hsig.host) + 3;
	*flag = 0;
	hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, c.hsig.dev + 3, seq);
	if (hipGetLastError() != hipSuccess) return 1;
	for (;;) {
		for (int i = 0; i < 20000; i++) {
			if (*flag == seq) return 0;
			__builtin_ia32_pause();
		}
		const hipError_t e = hipStreamQuery(c.st);
		if (e == hipSuccess) return 0;
		if (e != hipErrorNotReady) HIPCHK(e);
	}
}
int wait_done(Ctx& c) {
	const int w = signal_and_wait(c);
	if (w < 0) return w;
	if (w == 1) HIPCHK(hipStreamSynchronize(c.st));
	return 0;
}

// ---- calls in flight (tsqr_mi_qr_f32_submit / _finish): at most two per host thread, one per half of the pinned words (status
// words 4 * slot .. 4 * slot + 2, completion word 4 * slot + 3).  "Latching" a ticket = waiting for its completion word and copying
// the verdict out of the pinned words, after which the slot (and the stream behind it) is free for anything else. ----
thread_local tsqr_mi_ticket* t_pending[2] = {nullptr, nullptr};
thread_local unsigned t_submits = 0;
int ticket_latch(tsqr_mi_ticket* t) {
	if (!t || t->pending != 1) return 0;
	volatile unsigned* w = reinterpret_cast<volatile unsigned*>(t->words) + 4 * t->slot;
	hipStream_t st = reinterpret_cast<hipStream_t>(t->stream);
	for (bool done = false; !done;) {
		for (int i = 0; i < 20000 && !done; i++) {
			done = (w[3] == t->seq);
			if (!done) __builtin_ia32_pause();
		}
		if (!done) {                                     // (a failed launch must not hang the caller: look at the stream now and then)
			const hipError_t e = hipStreamQuery(st);
			if (e == hipSuccess) done = true;
			else if (e != hipErrorNotReady) { t_pending[t->slot] = nullptr; t->pending = 0; t->state = -1; HIPCHK(e); }
		}
	}
	t->verdict = w[0];
	const unsigned b = w[2];
	memcpy(&t->scond, &b, 4);
	t->pending = 2;
	if (t_pending[t->slot] == t) t_pending[t->slot] = nullptr;
	return 0;
}
int latch_all() {
	for (int s = 0; s < 2; s++)
		if (t_pending[s]) { const int rc = ticket_latch(t_pending[s]); if (rc) return rc; }
	return 0;
}

// read the status word of slot `s` (0 accepted / 1 rejected) after draining the stream (wait = false: the stream is known to be idle)
int read_status(Ctx& c, int s, unsigned* out, float* scond = nullptr, bool wait = true) {
	if (g_set.debug) {
		unsigned w3[3];
		HIPCHK(hipStreamSynchronize(c.st));
		HIPCHK(hipMemcpy(w3, c.status_dev(s), sizeof(w3), hipMemcpyDeviceToHost));
		float ratio, sc;
		memcpy(&ratio, &w3[1], 4); memcpy(&sc, &w3[2], 4);
		fprintf(stderr, "[tsqr_mi] chol status %u  min pivot ratio %.4g  scaled cond S %.4g\n", w3[0], ratio, sc);
	}
	if (c.hsig.dev) {                                    // the Cholesky kernel wrote the words to the pinned memory itself
		if (wait) {
			const int rc = wait_done(c);
			if (rc) return rc;
		}
		volatile unsigned* h = reinterpret_cast<volatile unsigned*>(c.hsig.host) + 4 * s;
		*out = h[0];
		if (scond) { const unsigned b = h[2]; memcpy(scond, &b, 4); }
		return 0;
	}
	unsigned w3[3];
	HIPCHK(hipStreamSynchronize(c.st));
	HIPCHK(hipMemcpy(w3, c.status_dev(s), sizeof(w3), hipMemcpyDeviceToHost));
	*out = w3[0];
	if (scond) memcpy(scond, &w3[2], 4);
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------------
template <int NT> int launch_fold(const tsqrmi::FoldArgs& a, hipStream_t st) {
	const int blocks = (a.nwaves + 3) / 4;
	if constexpr (NT == 4) {
		if (a.tri_init) {
			if (a.cor) hipLaunchKernelGGL((tsqrmi::fold_kernel<4, true, true>), dim3(blocks), dim3(256), 0, st, a);
			else hipLaunchKernelGGL((tsqrmi::fold_kernel<4, true, false>), dim3(blocks), dim3(256), 0, st, a);
			return 0;
		}
	}
	if (a.cor) hipLaunchKernelGGL((tsqrmi::fold_kernel<NT, false, true>), dim3(blocks), dim3(256), 0, st, a);
	else hipLaunchKernelGGL((tsqrmi::fold_kernel<NT, false, false>), dim3(blocks), dim3(256), 0, st, a);
	return 0;
}
int dispatch_fold(int NT, const tsqrmi::FoldArgs& a, hipStream_t st) {
	switch (NT) {
		case 1: return launch_fold<1>(a, st);
		case 2: return launch_fold<2>(a, st);
		case 3: return launch_fold<3>(a, st);
		default: return launch_fold<4>(a, st);
	}
}

// the R-stack reduction (fold_coop_kernel): nblocks upper-triangular 64 x 64 blocks of `stack` -> R, ping-ponging between `stack` and
// `other`; eight stacked blocks per workgroup and 64-step chain: 2048 -> 256 -> 32 -> 4 -> 1 in four launches / four dependent chains
int fold_coop(Ctx& c, float* r, size_t ldr, float* stack, size_t stack_ld, int nblocks, size_t n, float* other, bool cor) {
	constexpr int W = tsqrmi::FOLD_COOP_WAVES;
	constexpr int lds = tsqrmi::FOLD_COOP_LDS_FLOATS * (int)sizeof(float);
	static DevOnce attr[2];
	if (attr[cor].need(c.dev)) {
		HIPCHK(hipFuncSetAttribute(cor ? reinterpret_cast<const void*>(&tsqrmi::fold_coop_kernel<true>) : reinterpret_cast<const void*>(&tsqrmi::fold_coop_kernel<false>),
		                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
		attr[cor].done(c.dev);
	}
	float* cur = stack; size_t cur_ld = stack_ld;
	float* nxt = other;
	for (;;) {
		tsqrmi::FoldTreeArgs a{};
		a.src = cur; a.ld = cur_ld; a.nblocks = nblocks; a.n = (int)n;
		const int wgs = (nblocks + W - 1) / W;
		if (wgs == 1) { a.dst = r; a.dst_ld = ldr; a.rows_store = (int)n; a.cols_store = (int)n; }
		else { a.dst = nxt; a.dst_ld = (size_t)wgs * 64; a.rows_store = 64; a.cols_store = 64; }
		{
			ProfScope ps(KC_TREE, c.st);
			if (cor) hipLaunchKernelGGL(tsqrmi::fold_coop_kernel<true>, dim3(wgs), dim3(64 * W), lds, c.st, a);
			else hipLaunchKernelGGL(tsqrmi::fold_coop_kernel<false>, dim3(wgs), dim3(64 * W), lds, c.st, a);
		}
		HIPCHK(hipGetLastError());
		if (wgs == 1) break;
		float* t = cur; cur = nxt; nxt = t; cur_ld = (size_t)wgs * 64;
		nblocks = wgs;
	}
	return 0;
}

// R (n x n, ldr; full block written, zeros below the diagonal) of src (m x n), n <= 64: streaming Householder TSQR + fold tree.
// stack_a / stack_b: scratch for the R stacks of even / odd levels (Plan::stack_a / stack_b floats).
int fold_r(Ctx& c, float* r, size_t ldr, const float* src, size_t ld, size_t m, size_t n, float* stack_a, float* stack_b) {
	const Plan p = make_plan(m, n);
	const int NT = (int)(p.NP / 16);
	const bool cor = c.fold_cor;
	const float* cur = src; size_t cur_ld = ld;
	for (int lv = 0; lv < p.nlevels; lv++) {
		tsqrmi::FoldArgs a{};
		a.src = cur; a.ld = cur_ld; a.m = p.rows[lv];
		a.n = (int)n;                                    // stacks are NP wide, only the first n columns carry data
		a.nchunks = p.nch[lv]; a.cpw = p.cpw[lv]; a.nwaves = p.nw[lv];
		a.tri_init = (lv > 0 && p.NP == 64) ? 1 : 0;
		a.cor = cor ? 1 : 0;
		if (p.nw[lv] == 1) {
			a.dst = r; a.dst_ld = ldr; a.rows_store = (int)n; a.cols_store = (int)n;
		} else {
			float* stack = (lv % 2 == 0) ? stack_a : stack_b;
			a.dst = stack; a.dst_ld = (size_t)p.nw[lv] * p.NP; a.rows_store = (int)p.NP; a.cols_store = (int)p.NP;
			cur = stack; cur_ld = a.dst_ld;
		}
		{
			ProfScope ps(lv == 0 ? KC_FOLD0 : KC_TREE, c.st);
			dispatch_fold(NT, a, c.st);
		}
		HIPCHK(hipGetLastError());
		if (lv == 0 && p.nw[0] > 1 && p.NP == 64) {
			// 64-column panels: the whole R-stack reduction by cooperative eight-block folds (fold_coop_kernel) instead of one launch per binary level
			return fold_coop(c, r, ldr, stack_a, (size_t)p.nw[0] * 64, p.nw[0], n, stack_b, cor);
		}
	}
	return 0;
}

template <int NT> void launch_gram(const tsqrmi::GramArgs& a, int nblocks, bool bf16, hipStream_t st) {
	if (bf16) hipLaunchKernelGGL(tsqrmi::gram_bf16_kernel<NT>, dim3(nblocks), dim3(256), 0, st, a);
	else hipLaunchKernelGGL(tsqrmi::gram_kernel<NT>, dim3(nblocks), dim3(256), 0, st, a);
}

// Gram matrix of src (m x n) in MFMA-accumulator order -> c.gsum() (ntri*256 doubles + the local row count behind them), summed
// over the ranks of a row-partitioned call.  bf16 = true: bf16x3-split MFMA (memory-bound, f32 C/D layout), false: fp64 MFMA.
// io_half: src holds halves (fp16 I/O modes, bf16 level only): gram_h_kernel takes them as MFMA operands directly.
int gram_g(Ctx& c, const float* src, size_t ld, size_t m, size_t n, bool bf16, bool io_half = false) {
	const GramPlan g = gram_plan(m, n);
	const int NT = (int)(np_of(n) / 16);
	tsqrmi::GramArgs a{};
	a.a = src; a.lda = ld; a.m = m; a.n = (int)n; a.nchunks = g.nch; a.cpw = g.cpw; a.nwaves = g.nwaves;
	a.part = reinterpret_cast<double*>(c.wr);
	a.skip_status = c.prev_slot >= 0 ? c.status_dev(c.prev_slot) : nullptr;
	if (!(bf16 && c.gramq_ready)) { a.announce = c.announce_word; a.announce_seq = c.announce_seq; c.announce_word = nullptr; }
	int nparts = g.nblocks;                              // workgroups that wrote a partial
	if (bf16 && c.gramq_ready) {                         // the previous sweep's apply kernel accumulated this very Gram matrix
		c.gramq_ready = false;
		nparts = c.gramq_nparts;
	} else if (io_half) {
		ProfScope ps(KC_GRAM, c.st);
		switch (NT) {
			case 1: hipLaunchKernelGGL(tsqrmi::gram_h_kernel<1>, dim3(g.nblocks), dim3(256), 0, c.st, a); break;
			case 2: hipLaunchKernelGGL(tsqrmi::gram_h_kernel<2>, dim3(g.nblocks), dim3(256), 0, c.st, a); break;
			case 3: hipLaunchKernelGGL(tsqrmi::gram_h_kernel<3>, dim3(g.nblocks), dim3(256), 0, c.st, a); break;
			default: hipLaunchKernelGGL(tsqrmi::gram_h_kernel<4>, dim3(g.nblocks), dim3(256), 0, c.st, a); break;
		}
	} else if (bf16 && n == 64 && m % 128 == 0 && m <= ((size_t)1 << 20) && ld % 4 == 0 && ld <= ((size_t)1 << 24) && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
		// full 64-column matrices that fit the 256 MiB Infinity Cache: block-pattern loads + LDS staging (gram_blk_kernel); everything
		// else: gram_bf16_kernel.  (Measured, kernel / call period under the profiler: 2^21 rows 93.8 / 322.6 vs 95.8 / 319.4 us,
		// 2^22 rows 190.0 / 621.2 vs 192.0 / 621.0, 2^23 rows 421 / 1310 vs 381 / 1266 -- beyond the cache the chunk kernel is as good
		// or better; at 2^20 rows the call is 3-5 us faster with the block kernel, profiles/r03_experiment_log.md.)
		static DevOnce attr;
		if (attr.need(c.dev)) {
			HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_blk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GB_LDS_BYTES));
			attr.done(c.dev);
		}
		a.nchunks = (int)(m / 128);
		nparts = std::min(a.nchunks, g.nblocks);
		ProfScope ps(KC_GRAM, c.st);
		hipLaunchKernelGGL(tsqrmi::gram_blk_kernel, dim3(nparts), dim3(256), tsqrmi::GB_LDS_BYTES, c.st, a);
	} else {
		ProfScope ps(KC_GRAM, c.st);
		switch (NT) {
			case 1: launch_gram<1>(a, g.nblocks, bf16, c.st); break;
			case 2: launch_gram<2>(a, g.nblocks, bf16, c.st); break;
			case 3: launch_gram<3>(a, g.nblocks, bf16, c.st); break;
			default: launch_gram<4>(a, g.nblocks, bf16, c.st); break;
		}
	}
	HIPCHK(hipGetLastError());
	const int nelem = g.ntri * 256;
	{
		ProfScope ps(KC_CHOL, c.st);
		hipLaunchKernelGGL(tsqrmi::gram_reduce1_kernel, dim3((nelem + 15) / 16), dim3(256), 0, c.st, c.gsum(), a.part, nparts, nelem, (double)m,
		                   nullptr, (size_t)0, nullptr, 0);
	}
	HIPCHK(hipGetLastError());
	if (c.comm.active()) {
		// the ONLY exchange of the Gram engine: <= 2560 doubles + the row count.  Every rank then factors the same matrix with the
		// same thresholds, so all ranks take the same accept / reject decisions by construction (a NaN on one rank reaches all).
		ProfScope ps(KC_MISC, c.st);
		if (c.comm.allreduce_f64(c.gsum(), (size_t)nelem + 1, c.st)) { t_last_error = "all-reduce of the Gram tiles failed"; return -1; }
	}
	return 0;
}

// R = chol(G) (n x n, ldr), Z = inverse(R) (NP x NP in wq[L.z]), status words -> slot c.slot.  level: 2 bf16, 1 fp64, 3 shifted fp64.
int chol_from_g(Ctx& c, float* r, size_t ldr, size_t n, int level) {
	tsqrmi::CholArgs a{};
	a.r = r; a.ldr = ldr; a.z = c.wq + c.L.z;
	a.status = c.status_dev(c.slot);
	a.host_status = c.hsig.dev ? c.hsig.dev + 4 * c.slot : nullptr;
	a.gsum = c.gsum();
	a.prev_status = c.prev_slot >= 0 ? c.status_dev(c.prev_slot) : nullptr;
	const int NT = (int)(np_of(n) / 16);
	a.rows_dev = c.comm.active() ? c.gsum() + (size_t)(NT * (NT + 1) / 2) * 256 : nullptr;
	a.rows = c.rows_global;
	a.shift_coef = (level == 3) ? 11.0 * 1.1102230246251565e-16 : 0.0;
	a.n = (int)n; a.NT = NT; a.level = level; a.scond_floor = g_set.bf16_scond_floor;
	if (level == 2) { a.relax = c.chol_relax; a.retry_shift = c.chol_retry_shift; }
	{
		ProfScope ps(KC_CHOL, c.st);
		hipLaunchKernelGGL(tsqrmi::chol16_kernel, dim3(1), dim3(1024), 0, c.st, a);
	}
	HIPCHK(hipGetLastError());
	return 0;
}

// apply_wg_kernel launcher: args.nchunks = row blocks of ROWS, args.nwaves = workgroups (persistent grid)
template <int E, int NT, bool UPD, int ROWS, bool GRAMQ> constexpr auto apply_wg_entry() {
	if constexpr (GRAMQ) return &tsqrmi::apply_wg_gramq_kernel<E, NT, UPD, ROWS>;
	else return &tsqrmi::apply_wg_kernel<E, NT, UPD, ROWS>;
}
template <int E, int NT, bool UPD, int ROWS, bool GRAMQ = false> int launch_apply_wg(Ctx& c, tsqrmi::ApplyArgs a) {
	constexpr auto kernel = apply_wg_entry<E, NT, UPD, ROWS, GRAMQ>();
	constexpr int NP = 16 * NT, KT = (NP + 31) / 32;
	constexpr int NB = (!UPD && NT == 4) ? 6 : KT * NT;  // operand blocks of Z kept in LDS (apply_wg_kernel: COMPACT)
	size_t lds = sizeof(float) * NP * (ROWS + 4) +
	             (E == 0 ? sizeof(float) * NP * (NP + 16) : (size_t)(E == 2 ? 1 : 3) * NB * 512 * 2);
	if (GRAMQ) lds = std::max(lds, sizeof(double) * 2 * (NT * (NT + 1) / 2) * 256);   // the final workgroup reduction aliases the block
	static DevOnce attr;                                 // (per template instance)
	static std::atomic<int> per_cu_cache[MAX_DEV];
	if (attr.need(c.dev)) {
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kernel), 256, lds) != hipSuccess || nb < 1) {
			(void)hipGetLastError(); nb = 2;
		}
		// persistent grid: as many workgroups as are resident on the 256 CUs at once (LDS / register bound: 2 or 3 per CU);
		// measured: the fp32-MFMA engine is slower with three per CU (132 vs 112 us)
		per_cu_cache[c.dev].store(std::min(nb, ROWS == 64 ? 4 : (E == 0 ? 2 : 3)));
		attr.done(c.dev);
	}
	const size_t nblk = cdiv(a.m, (size_t)ROWS);
	a.nchunks = (int)nblk;
	const int wgs = g_set.apply_wgs.load();
	size_t want = wgs > 0 ? (size_t)wgs : (size_t)256 * per_cu_cache[c.dev].load();
	unsigned gy = 1;
	if constexpr (UPD) {
		if (a.multi_cols > 0) { gy = (unsigned)cdiv((size_t)a.multi_cols, PW); want = std::max<size_t>(1, want / gy); }   // (the resident workgroups split over the trailing panels)
	}
	a.nwaves = (int)std::min<size_t>(nblk, want);
	a.cpw = 0;
	if constexpr (GRAMQ) {
		a.nwaves = std::min(a.nwaves, c.gramq_cap);      // one partial per workgroup: never more than the buffer holds
		a.gpart = c.gramq_part;
		c.gramq_nparts = a.nwaves;
	}
	if constexpr (!UPD && !GRAMQ && ROWS == 64) {
		// Four workgroups per CU on the whole chip: uneven shares by XCD parity and by dispatch round (apply_wg_body; round 3, on every box
		// looked at THEN: the pass ends at 78 us instead of 84).  Round 4: on five boxes in a row the same shares COST 5-6 % of the call
		// (blocking 194.9-198.9 us against 186.1-188.1 with equal shares, chained 170.3-173.2 against 159.4-161.1; three interleaved runs,
		// profiles/r04_experiment_log.md) -- the XCD asymmetry they answer is a property of the box's state, not of the chip.  So the
		// choice is MEASURED once per process and device (and per side of the cache size), on the first large out-of-place pass: the two
		// candidates (shares by XCD parity and round / equal) run the pass itself in turn under HIP events, four times each (Q = A Z is the
		// same whoever computes a block: the last run leaves the result), ~0.7 ms once; the uneven shares are taken only when they are 1 %
		// faster.  TSQR_MI_APPLY_SHARES=0 / 1 / 2 pins equal / both / by round only.
		if (a.nwaves == 1024 && per_cu_cache[c.dev].load() == 4) {
			static const int pinned = env_int("TSQR_MI_APPLY_SHARES", -1);
			static std::atomic<int> chosen[MAX_DEV][2];      // 0 not measured yet, 1 both, 2 rounds only, 3 equal; [1]: a matrix beyond the Infinity Cache
			auto set_mode = [](tsqrmi::ApplyArgs& x, int mode) {
				x.share[0] = x.share[1] = x.share[2] = x.share[3] = 0; x.even_share = 0;
				if (mode == 1 || mode == 2) { x.share[0] = 18; x.share[1] = 17; x.share[2] = 15; x.share[3] = 14; x.even_share = (mode == 1) ? 69 : 64; }
			};
			const double bytes = (double)a.m * (double)a.n * sizeof(float);
			const int cls = bytes > 256.0 * 1048576.0 ? 1 : 0;
			int mode = pinned == 0 ? 3 : (pinned == 1 ? 1 : (pinned == 2 ? 2 : chosen[c.dev][cls].load()));
			const bool big = bytes >= 128.0 * 1048576.0;
			if (mode == 0 && big && reinterpret_cast<const void*>(a.q) != reinterpret_cast<const void*>(a.a) && !t_prof.on) {
				constexpr int REPS = 4;                          // interleaved: uneven, equal, uneven, equal, ...
				hipEvent_t ev[2 * REPS + 1];
				bool ok = true;
				for (auto& e : ev) ok = ok && hipEventCreate(&e) == hipSuccess;
				if (ok) {
					for (int i = 0; i < 2 * REPS; i++) {
						tsqrmi::ApplyArgs x = a;
						set_mode(x, (i & 1) ? 3 : 1);
						(void)hipEventRecord(ev[i], c.st);
						hipLaunchKernelGGL(kernel, dim3(a.nwaves, gy), dim3(256), lds, c.st, x);
					}
					(void)hipEventRecord(ev[2 * REPS], c.st);
					ok = hipEventSynchronize(ev[2 * REPS]) == hipSuccess;
					float t[2] = {0.f, 0.f};
					for (int i = 0; i < 2 * REPS && ok; i++) { float ms = 0.f; ok = hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess; t[i & 1] += ms; }
					// (a speculative pass that skipped itself -- rejected Gram matrix -- measures nothing: ask again next time)
					if (ok && t[0] > 0.01f * REPS && t[1] > 0.01f * REPS)
						chosen[c.dev][cls].store(t[0] < 0.99f * t[1] ? 1 : 3);   // equal shares unless the uneven ones are (1 %) faster: 3 % on the boxes they were tuned on, 5-10 % SLOWER elsewhere
				}
				for (auto& e : ev) (void)hipEventDestroy(e);
				(void)hipGetLastError();
				return 0;                                    // (the pass has run)
			}
			set_mode(a, mode == 0 ? 3 : mode);
		}
	}
	hipLaunchKernelGGL(kernel, dim3(a.nwaves, gy), dim3(256), lds, c.st, a);
	return 0;
}
template <int E, int NT, bool UPD> int launch_apply_any(Ctx& c, const tsqrmi::ApplyArgs& a) {
	if constexpr (!UPD && E != 0) {                      // (the fp32-MFMA engine's fused variant spills and loses: 0.29 vs 0.22 ms per apply)
		if (c.gramq_part && c.gramq_cap > 0) return launch_apply_wg<E, NT, UPD, 128, true>(c, a);
	}
	// block height per engine (measured, profiles/r02_experiment_log.md): the bf16x3 engine streams best with 64-row blocks, four
	// workgroups per CU and two blocks in flight (88 vs 92 us at 2^20 x 64); the exact-fp32 and fp16 engines and the coupling
	// update (UPD) use 128-row blocks
	if constexpr (!UPD && E == 1) return launch_apply_wg<E, NT, UPD, 64>(c, a);
	else return launch_apply_wg<E, NT, UPD, 128>(c, a);
}
// fp16 I/O modes: the plain product with halves at both ends -- bf16x3 engine 1 (fp16_notc) or single-fp16-product engine 2
// (fp16_tc_nocor); 64- / 128-row blocks as in the fp32 call, two blocks in flight
template <int E, int NT> int launch_apply_h(Ctx& c, tsqrmi::ApplyArgs a) {
	constexpr int ROWS = (E == 1) ? 64 : 128;           // (measured: the bf16x3 engine at 128-row blocks 90 us against 59 at 64)
	constexpr auto kernel = &tsqrmi::apply_wg_h_kernel<E, NT, ROWS>;
	constexpr int NP = 16 * NT, KT = (NP + 31) / 32;
	constexpr int NB = (NT == 4) ? 6 : KT * NT;
	const size_t lds = sizeof(float) * NP * (ROWS + 4) + (E == 0 ? sizeof(float) * NP * (NP + 16) : (size_t)(E == 2 ? 1 : 3) * NB * 512 * 2);
	static DevOnce attr;
	static std::atomic<int> per_cu_cache[MAX_DEV];
	if (attr.need(c.dev)) {
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kernel), 256, lds) != hipSuccess || nb < 1) {
			(void)hipGetLastError(); nb = 2;
		}
		per_cu_cache[c.dev].store(std::min(nb, ROWS == 64 ? 4 : 3));
		attr.done(c.dev);
	}
	const size_t nblk = cdiv(a.m, (size_t)ROWS);
	a.nchunks = (int)nblk;
	a.nwaves = (int)std::min<size_t>(nblk, (size_t)256 * per_cu_cache[c.dev].load());
	a.cpw = 0;
	hipLaunchKernelGGL(kernel, dim3(a.nwaves), dim3(256), lds, c.st, a);
	return 0;
}
template <int E> int dispatch_apply_h(Ctx& c, int NT, const tsqrmi::ApplyArgs& a) {
	switch (NT) {
		case 1: return launch_apply_h<E, 1>(c, a);
		case 2: return launch_apply_h<E, 2>(c, a);
		case 3: return launch_apply_h<E, 3>(c, a);
		default: return launch_apply_h<E, 4>(c, a);
	}
}
template <int E> int dispatch_apply_nt(Ctx& c, int NT, const tsqrmi::ApplyArgs& a) {
	switch (NT) {
		case 1: return launch_apply_any<E, 1, false>(c, a);
		case 2: return launch_apply_any<E, 2, false>(c, a);
		case 3: return launch_apply_any<E, 3, false>(c, a);
		default: return launch_apply_any<E, 4, false>(c, a);
	}
}

// q = a * inverse(r); n <= 64; Z in wq[L.z] (computed here from r unless z_ready)
// io_half: a and q hold halves (fp16 I/O modes; engines 1 and 2)
int apply_rinv(Ctx& c, int engine, float* q, size_t ldq, const float* a, size_t lda, const float* r, size_t ldr,
               size_t m, size_t n, bool z_ready = false, const unsigned* skip_status = nullptr, bool io_half = false,
               void* r16 = nullptr, size_t ldr16 = 0) {
	const size_t NP = np_of(n);
	const int NT = (int)(NP / 16);
	float* z_buf = c.wq + c.L.z;
	if (!z_ready) {
		ProfScope ps(KC_TRINV, c.st);
		hipLaunchKernelGGL(tsqrmi::trinv_kernel, dim3(1), dim3(256), 0, c.st, z_buf, r, ldr, (int)n, (int)NP);
	}
	HIPCHK(hipGetLastError());
	tsqrmi::ApplyArgs aa{};
	aa.a = a; aa.lda = lda; aa.q = q; aa.ldq = ldq; aa.m = m; aa.n = (int)n; aa.z = z_buf; aa.skip_status = skip_status;
	aa.r32 = r; aa.r16 = r16; aa.ldr16 = ldr16;         // (io_half with r16: r is then a packed n x n factor, ld n)
	aa.plain_q = aa.forward = c.q_for_next_sweep ? 1 : 0;
	int rc;
	{
		ProfScope ps(KC_APPLY, c.st);
		if (io_half) rc = (engine == 1) ? dispatch_apply_h<1>(c, NT, aa) : dispatch_apply_h<2>(c, NT, aa);
		else rc = (engine == 0) ? dispatch_apply_nt<0>(c, NT, aa) : (engine == 1 ? dispatch_apply_nt<1>(c, NT, aa) : dispatch_apply_nt<2>(c, NT, aa));
	}
	if (rc) return rc;
	HIPCHK(hipGetLastError());
	return 0;
}

int engine_of(int mode) {
	if (mode == TSQR_MI_FP32_NOTC) return 0;
	if (mode == TSQR_MI_FP32_TC_COR) return 1;
	if (mode == TSQR_MI_FP32_TC_NOCOR) return 2;         // R factor as fp32_tc_cor, Q = A * inverse(R) with fp16 operands and no correction
	return -1;
}

// r <- r2 * r1 (upper triangular n x n, fp64 accumulation; r may not alias r1 / r2)
void launch_rmul(float* r, size_t ldr, const float* r2, size_t ldr2, const float* r1, size_t ldr1, size_t n, hipStream_t st) {
	if (n <= 64) {
		hipLaunchKernelGGL(tsqrmi::rmul64_kernel, dim3(1), dim3(1024), 0, st, r, ldr, r2, ldr2, r1, ldr1, (int)n, (const float*)nullptr, (size_t)0);
		return;
	}
	const unsigned gb = (unsigned)std::min<size_t>(1024, cdiv(n * n, 256));
	hipLaunchKernelGGL(tsqrmi::rmul_kernel, dim3(gb), dim3(256), 0, st, r, ldr, r2, ldr2, r1, ldr1, (int)n);
}
// r <- r3 * r2 * r1 (n <= 64: one launch, the inner product stays in LDS as fp64; otherwise through tmp, n x n packed)
void launch_rmul3(float* r, size_t ldr, const float* r3, const float* r2, const float* r1, float* tmp, size_t n, hipStream_t st) {
	if (n <= 64) {
		hipLaunchKernelGGL(tsqrmi::rmul64_kernel, dim3(1), dim3(1024), 0, st, r, ldr, r2, n, r1, n, (int)n, r3, n);
		return;
	}
	launch_rmul(tmp, n, r2, n, r1, n, n, st);
	launch_rmul(r, ldr, r3, n, tmp, n, n, st);
}

// Householder TSQR R factor of one <= 64-column panel.  Row-partitioned: every rank folds its block, the n x n factors are
// all-gathered (the exchange the north star names: RCCL all-gather of the local R factors) and every rank folds the same
// (P n) x n stack, so R is bitwise identical everywhere.
int householder_r(Ctx& c, float* rpp, size_t ldr, const float* ap, size_t lda, size_t m, size_t cc) {
	if (!c.comm.active())
		return fold_r(c, rpp, ldr, ap, lda, m, cc, c.wr, c.wq);
	if (!c.comm.gather_buf) { t_last_error = "row-partitioned Householder engine needs the gather buffer"; return -1; }
	const int P = c.comm.nranks;
	float* rl = c.wq + c.L.r4;                           // local R, packed (ld = cc)
	int rc = fold_r(c, rl, cc, ap, lda, m, cc, c.wr, c.wq);
	if (rc) return rc;
	{
		ProfScope ps(KC_MISC, c.st);
		if (c.comm.allgather_f32(rl, c.comm.gather_buf, cc * cc, c.st)) { t_last_error = "all-gather of the local R factors failed"; return -1; }
	}
	// gather_buf is [rank][column][row]; restack into wr as a column-major (P cc) x cc matrix, fold scratch behind it (both sized
	// by tsqr_mi_working_r_size_dist)
	float* stack = c.wr;
	for (int k = 0; k < P; k++)
		hipLaunchKernelGGL(tsqrmi::copy2d_kernel, dim3(16), dim3(256), 0, c.st,
		                   stack + (size_t)k * cc, (size_t)P * cc, c.comm.gather_buf + (size_t)k * cc * cc, cc, (int)cc, (int)cc);
	HIPCHK(hipGetLastError());
	const size_t stack_floats = ((size_t)P * cc * cc + 63) & ~(size_t)63;
	return fold_r(c, rpp, ldr, stack, (size_t)P * cc, (size_t)P * cc, cc, c.wr + stack_floats, c.wq);
}

constexpr int R_SHIFT_DIRECT = 9;
// R factor and Q of one <= 64-column panel.  r_engine: first Gram level (2 / 1), 0 = Householder, R_SHIFT_DIRECT = the caller has
// just seen the fp64 Gram level reject this very panel (its Gram matrix is still in the work buffer): shifted-Cholesky step at once.
// check_now: read each verdict immediately (one wait) and escalate on rejection; otherwise enqueue speculatively.
int panel_qr(Ctx& c, int engine, int r_engine, bool check_now, float* qp, size_t ldq, float* rpp, size_t ldr, const float* ap, size_t lda,
             size_t m, size_t cc) {
	int rc;
	const bool direct_shift = (r_engine == R_SHIFT_DIRECT);
	if (direct_shift) r_engine = 0;
	for (int e = r_engine; e >= 1; e--) {                // 2: bf16-split Gram, 1: fp64 Gram; with check_now a rejected level escalates
		rc = gram_g(c, ap, lda, m, cc, e == 2);
		if (rc) return rc;
		rc = chol_from_g(c, rpp, ldr, cc, e);
		if (rc) return rc;
		bool ok = true;
		if (check_now) {
			unsigned status = 0;
			rc = read_status(c, c.slot, &status);
			if (rc) return rc;
			ok = (status == 0);
		}
		if (ok) {
			c.min_level = std::min(c.min_level, e);
			// speculative (unchecked) launch under the auto policy: the kernel itself skips the pass when the level was rejected
			const unsigned* skip = (!check_now && c.policy == 0) ? c.status_dev(c.slot) : nullptr;
			return apply_rinv(c, engine, qp, ldq, ap, lda, rpp, ldr, m, cc, /*z_ready=*/true, skip);
		}
	}
	if ((direct_shift || (r_engine >= 1 && check_now)) && c.policy == 0) {
		// Both Gram levels rejected the panel (cond beyond ~1e6, or rank deficient).  Shifted Cholesky QR: the fp64 Gram matrix is
		// still in the work buffer; R1 = chol(G + s I) always exists, Q1 = A inverse(R1) has cond(Q1) <~ 1e5, and one unshifted fp64
		// sweep on Q1 in place finishes the panel: A = Q (R2 R1).  About 2x faster than the Householder fold below and, after that
		// second step, at least as orthogonal as its single indirect sweep.
		float* r1 = c.wq + c.L.r3; float* r2 = c.wq + c.L.r4;
		rc = chol_from_g(c, r1, cc, cc, 3);
		if (rc) return rc;
		unsigned status = 0;
		rc = read_status(c, c.slot, &status);
		if (rc) return rc;
		if (status == 0) {
			rc = apply_rinv(c, engine, qp, ldq, ap, lda, r1, cc, m, cc, /*z_ready=*/true);
			if (rc) return rc;
			rc = gram_g(c, qp, ldq, m, cc, /*bf16=*/false);
			if (rc) return rc;
			rc = chol_from_g(c, r2, cc, cc, 1);
			if (rc) return rc;
			rc = read_status(c, c.slot, &status);
			if (rc) return rc;
			if (status == 0) {
				rc = apply_rinv(c, engine, qp, ldq, qp, ldq, r2, cc, m, cc, /*z_ready=*/true);
			} else {
				// Q1 is still numerically rank deficient: the input has an (almost) exactly dependent column whose rounding residue is
				// itself dependent (e.g. two constant columns).  No triangular solve can make an orthonormal column out of that; a
				// second SHIFTED step keeps everything bounded instead -- the other columns come out orthonormal, the residual stays
				// at rounding level, R shows the deficiency as a tiny diagonal entry and that one column of Q is left un-normalised.
				rc = chol_from_g(c, r2, cc, cc, 3);
				if (rc) return rc;
				rc = read_status(c, c.slot, &status);
				if (rc) return rc;
				if (status == 0) {
					rc = apply_rinv(c, engine, qp, ldq, qp, ldq, r2, cc, m, cc, /*z_ready=*/true);
				} else {                                     // non-finite data: last resort, the Householder engine on Q1
					c.used_householder = true;
					rc = householder_r(c, r2, cc, qp, ldq, m, cc);
					if (rc) return rc;
					rc = apply_rinv(c, engine, qp, ldq, qp, ldq, r2, cc, m, cc);
				}
			}
			if (rc) return rc;
			launch_rmul(rpp, ldr, r2, cc, r1, cc, cc, c.st);
			HIPCHK(hipGetLastError());
			c.min_level = 0;
			c.used_shift = true;
			return 0;
		}
	}
	c.min_level = 0;
	c.used_householder = true;
	rc = householder_r(c, rpp, ldr, ap, lda, m, cc);
	if (rc) return rc;
	return apply_rinv(c, engine, qp, ldq, ap, lda, rpp, ldr, m, cc);
}

int sweep_wide(Ctx& c, int engine, float* q, size_t ldq, float* r, size_t ldr, const float* a, size_t lda, size_t m, size_t n);

// one sweep of 64-wide-panel block QR:  (q, r) <- qr(a);  a is overwritten for n > 64; q may alias a.
// Panels are coupled by block MODIFIED Gram-Schmidt on the matrix cores (the role of the reference's cuBLAS GEMMs, src/blockqr.cu:92-116),
// RIGHT-LOOKING since round 4: as soon as panel p is factored, ONE launch each forms S = Qp^T [A_{p+1} ... A_last] (cross_kernel, a grid row
// per trailing panel), reduces it (writing R's block row and the operands -S_j) and updates every trailing panel (A_j <- A_j - Qp S_j).
// The same operations on the same operands as the left-looking order of rounds 1-3 (for every trailing panel the updates arrive in the
// same sequence, every S_j is summed in the same grouping: bit for bit the same factors), but one update launch per PANEL instead of one
// per panel pair, and the S launches grouped as far as the work space goes: at small row counts the launches finally have the chip's
// worth of workgroups (same-box A/B of the two builds, profiles/r04_experiment_log.md: 4096 x 1024 3.36 -> 1.07 ms, 32768 x 1024 3.87 -> 1.89 ms).
int sweep(Ctx& c, int engine, int r_engine, bool check_now, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda, size_t m, size_t n) {
	const size_t npanels = cdiv(n, PW);
	// 128-COLUMN BLOCKS (round 4): while the bf16-split level accepts them, two neighbouring panels are factored at once by the one-panel
	// path of 64 < n <= 128 (sweep_wide: ONE Gram pass over 128 columns, the two-block Cholesky factor, ONE apply pass -- speculative, A
	// untouched when the verdict over both blocks rejects) instead of panel + coupling + panel: a third of the passes over those columns
	// and one factorisation launch instead of two + a coupling.  Their two 64-column halves then couple with everything behind the block,
	// one after the other, as any two finished panels.  A rejected block (and everything after it in this sweep) takes the 64-column way.
	bool try_wide = c.wide && c.policy == 0 && r_engine == 2 && check_now && !c.comm.active() && n > 2 * PW;
	for (size_t pi = 0; pi < npanels;) {
		const size_t P = pi * PW, cc = std::min(PW, n - P);
		size_t blockw = cc;                              // columns factored in this step
		bool wide_blk = false;
		if (try_wide && n - P > PW) {
			const size_t w = std::min(2 * PW, n - P);
			int rcw = sweep_wide(c, engine, q + P * ldq, ldq, r + P * ldr + P, ldr, a + P * lda, lda, m, w);
			if (rcw) return rcw;
			unsigned stw = 1u;
			rcw = read_status(c, c.slot, &stw);
			if (rcw) return rcw;
			if (stw == 0) { wide_blk = true; blockw = w; c.min_level = std::min(c.min_level, 2); }
			else try_wide = false;
		}
		if (!wide_blk) {
			const int rc = panel_qr(c, engine, r_engine, check_now, q + P * ldq, ldq, r + P * ldr + P, ldr, a + P * lda, lda, m, cc);
			if (rc) return rc;
		}
		pi += cdiv(blockw, PW);
		const size_t T0 = P + blockw;                    // first trailing column (blockw is a multiple of 64 here: only the last block may be ragged)
		if (T0 >= n) break;
		for (size_t X0 = P; X0 < T0; X0 += PW) {         // the finished 64-column panels of this step, one after the other
		const size_t Pc = X0;
		const size_t ntc = n - T0, ntr = cdiv(ntc, PW);
		ProfScope ps(KC_COUPLE, c.st);
		// S_j = Qp^T A_j for every trailing panel j (bf16x3 MFMA products, fp64 sums): launches over groups of trailing panels, as many as
		// the work space holds workgroup partials for (cross_slots: every panel keeps the workgroup count -- and so the grouping of its
		// sums -- of a single panel pair; at small row counts that is all of them in one launch)
		const GramPlan g = gram_plan(m, PW);
		double* gm = reinterpret_cast<double*>(c.wq + c.L.gmulti);
		float* sm = c.wq + c.L.smulti;
		float* rblk = r + T0 * ldr + Pc;                 // R(Pc : Pc + 64, T0 : n)
		const size_t grp = std::max<size_t>(1, c.cross_slots / (size_t)g.nblocks);
		for (size_t j0 = 0; j0 < ntr; j0 += grp) {
			const unsigned gy = (unsigned)std::min(grp, ntr - j0);
			const int cols = (int)(ntc - PW * j0);       // columns from trailing panel j0 on
			tsqrmi::CrossArgs ca{};
			ca.x = q + Pc * ldq; ca.ldx = ldq; ca.y = a + (T0 + PW * j0) * lda; ca.ldy = lda; ca.m = m; ca.ny = std::min((int)PW, cols); ca.ny_total = cols;
			ca.nchunks = g.nch; ca.cpw = g.cpw; ca.nwaves = g.nwaves; ca.part = reinterpret_cast<double*>(c.wr);
			hipLaunchKernelGGL(tsqrmi::cross_kernel, dim3(g.nblocks, gy), dim3(256), 0, c.st, ca);
			// one GPU: the reduction writes -S_j (operand of the update) and the block row of R itself; row-partitioned: the sums only
			const bool fin = !c.comm.active();
			hipLaunchKernelGGL(tsqrmi::cross_reduce_multi_kernel, dim3(256, gy), dim3(256), 0, c.st, gm + j0 * (size_t)tsqrmi::CROSS_GSTRIDE, ca.part, g.nblocks,
			                   (double)m, fin ? rblk + j0 * PW * ldr : (float*)nullptr, ldr, fin ? sm + j0 * 4096 : (float*)nullptr, cols);
			HIPCHK(hipGetLastError());
		}
		if (c.comm.active()) {
			// ONE all-reduce for the whole block row (every rank holds the same S afterwards: the same R, the same update)
			if (c.comm.allreduce_f64(gm, ntr * (size_t)tsqrmi::CROSS_GSTRIDE, c.st)) { t_last_error = "all-reduce of the coupling tiles failed"; return -1; }
			hipLaunchKernelGGL(tsqrmi::cross_finish_kernel, dim3(16, (unsigned)ntr), dim3(256), 0, c.st, rblk, ldr, sm, gm, 0, (int)ntc);
			HIPCHK(hipGetLastError());
		}
		tsqrmi::ApplyArgs ua{};
		ua.a = q + Pc * ldq; ua.lda = ldq; ua.q = a + T0 * lda; ua.ldq = lda; ua.m = m; ua.n = (int)PW; ua.z = sm;
		ua.n_out = (int)std::min(PW, ntc); ua.multi_cols = (int)ntc;
		const int rc2 = (engine == 0) ? launch_apply_any<0, 4, true>(c, ua)
		                              : (engine == 1 ? launch_apply_any<1, 4, true>(c, ua) : launch_apply_any<2, 4, true>(c, ua));
		if (rc2) return rc2;
		HIPCHK(hipGetLastError());
		}
	}
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// 64 < n <= 128 as ONE Cholesky-QR panel (tsqr_wide.hip): Gram tiles of all n columns in one pass, the 128 x 128 Cholesky factor in
// two 64 x 64 blocks, Q = A * inverse(R) in one pass.  Everything is enqueued speculatively: the verdict over both blocks lands in
// slot c.slot (and its pinned alias), the apply pass skips itself on rejection, A is untouched (q == a is allowed: every workgroup
// has its block in LDS before it writes).  r receives the full n x n factor.
// ---------------------------------------------------------------------------------------------------------------------------
template <int E> int launch_apply_wide(Ctx& c, tsqrmi::ApplyArgs a) {
	// bf16x3 / fp16 engines: eight waves on 128-row blocks, one workgroup per CU; fp32-MFMA engine: four waves on 64-row blocks with
	// the compact triangular Z (40 KiB), two workgroups per CU
	constexpr bool F32 = (E == 0);
	constexpr int ROWS = F32 ? 64 : 128, THREADS = F32 ? 256 : 512, PER_CU = F32 ? 2 : 1;
	constexpr size_t lds = sizeof(float) * 128 * (ROWS + 4) + (F32 ? sizeof(float) * 10240 : (size_t)(E == 2 ? 1 : 3) * 20 * 512 * 2);
	const void* kernel;
	if constexpr (F32) kernel = reinterpret_cast<const void*>(&tsqrmi::apply_wide_f32_kernel);
	else kernel = reinterpret_cast<const void*>(&tsqrmi::apply_wide_kernel<E>);
	static DevOnce attr;
	if (attr.need(c.dev)) {
		HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		attr.done(c.dev);
	}
	const size_t nblk = cdiv(a.m, (size_t)ROWS);
	a.nchunks = (int)nblk;
	a.nwaves = (int)std::min<size_t>(nblk, (size_t)256 * PER_CU);
	if constexpr (F32) hipLaunchKernelGGL(tsqrmi::apply_wide_f32_kernel, dim3(a.nwaves), dim3(THREADS), lds, c.st, a);
	else hipLaunchKernelGGL(tsqrmi::apply_wide_kernel<E>, dim3(a.nwaves), dim3(THREADS), lds, c.st, a);
	return 0;
}
int sweep_wide(Ctx& c, int engine, float* q, size_t ldq, float* r, size_t ldr, const float* a, size_t lda, size_t m, size_t n) {
	float* w = c.wq + c.L.wide;
	double* gsum = reinterpret_cast<double*>(w);
	float* zw = w + WIDE_OFF_ZW;
	float* zf2 = w + WIDE_OFF_ZF2;
	unsigned* st1 = c.status_dev(2); unsigned* st2 = c.status_dev(3);
	// full 64-row blocks of a 128-column matrix go to the fast form of the Gram kernel, whatever is left (ragged last rows, or the
	// whole matrix when n < 128) to the general form; both write per-workgroup partials, one after the other
	const size_t nfull = (n == 2 * PW && lda <= ((size_t)1 << 23)) ? m / 64 : 0, nrest = cdiv(m, 64) - nfull;   // (fast form: 32-bit buffer offsets)
	const int wgs_fast = (int)std::min<size_t>(nfull, WIDE_MAX_WGS), wgs_rest = (int)std::min<size_t>(nrest, WIDE_MAX_WGS);
	const int wgs = wgs_fast + wgs_rest;
	{
		ProfScope ps(KC_GRAM, c.st);
		tsqrmi::GramWideArgs ga{};
		ga.a = a; ga.lda = lda; ga.m = m; ga.n = (int)n; ga.part = reinterpret_cast<double*>(c.wr);
		static DevOnce attr;
		if (attr.need(c.dev)) {
			HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_wide_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GW_LDS_BYTES));
			HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_wide_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GW_LDS_BYTES));
			attr.done(c.dev);
		}
		ga.announce = c.announce_word; ga.announce_seq = c.announce_seq; c.announce_word = nullptr;   // (the first launch carries it)
		if (wgs_fast) {
			ga.blk0 = 0; ga.nblk = (int)nfull;
			hipLaunchKernelGGL((tsqrmi::gram_wide_kernel<true>), dim3(wgs_fast), dim3(512), tsqrmi::GW_LDS_BYTES, c.st, ga);
			ga.announce = nullptr;
		}
		if (wgs_rest) {
			ga.blk0 = (int)nfull; ga.nblk = (int)nrest; ga.part += (size_t)wgs_fast * 36 * 256;
			hipLaunchKernelGGL((tsqrmi::gram_wide_kernel<false>), dim3(wgs_rest), dim3(512), tsqrmi::GW_LDS_BYTES, c.st, ga);
		}
	}
	HIPCHK(hipGetLastError());
	{
		ProfScope ps(KC_CHOL, c.st);
		const int nelem = 36 * 256;
		hipLaunchKernelGGL(tsqrmi::gram_reduce1_kernel, dim3(nelem / 16), dim3(256), 0, c.st, gsum, reinterpret_cast<const double*>(c.wr), wgs, nelem, (double)m,
		                   nullptr, (size_t)0, nullptr, 0);
		// chol(G11) -> Schur complement -> chol(G22') -> Z12 + verdict: one workgroup, one launch (chol_wide_kernel)
		tsqrmi::CholWideArgs wa{};
		wa.gsum = gsum; wa.r = r; wa.ldr = ldr; wa.n = (int)n; wa.zf1 = c.wq + c.L.z; wa.zf2 = zf2; wa.zw = zw;
		wa.st1 = st1; wa.st2 = st2; wa.status = c.status_dev(c.slot);
		wa.host_status = c.hsig.dev ? c.hsig.dev + 4 * c.slot : nullptr;
		wa.prev_status = c.prev_slot >= 0 ? c.status_dev(c.prev_slot) : nullptr;
		wa.rows = (double)m; wa.scond_floor = g_set.bf16_scond_floor;
		hipLaunchKernelGGL(tsqrmi::chol_wide_kernel, dim3(1), dim3(1024), 0, c.st, wa);
	}
	HIPCHK(hipGetLastError());
	tsqrmi::ApplyArgs aa{};
	aa.a = a; aa.lda = lda; aa.q = q; aa.ldq = ldq; aa.m = m; aa.n = (int)n; aa.z = zw; aa.skip_status = c.status_dev(c.slot);
	int rc;
	{
		ProfScope ps(KC_APPLY, c.st);
		rc = (engine == 0) ? launch_apply_wide<0>(c, aa) : (engine == 1 ? launch_apply_wide<1>(c, aa) : launch_apply_wide<2>(c, aa));
	}
	if (rc) return rc;
	HIPCHK(hipGetLastError());
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// the whole factorisation (single GPU: comm inactive; row-partitioned: this rank's block)
// ---------------------------------------------------------------------------------------------------------------------------
int qr_core(Ctx& c, int engine, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda, size_t m, size_t n) {
	const WqLayout& L = c.L;
	c.fold_cor = (engine == 1);
	// auto policy: every mode starts at the bf16-split Gram level (exact products, fp64 accumulation across K-steps: more accurate
	// than any plain fp32 evaluation of A^T A, accepted only for well-conditioned panels), then the fp64 Gram level, the shifted
	// Cholesky QR step and the Householder fold; the mode selects the MFMA engine of the apply pass.  Policy 4 skips the bf16 level.
	const bool use_gram = (c.policy == 2) || (c.policy == 0);
	const bool may_fall_back = use_gram && c.policy == 0;
	// single panel, single sweep: run speculatively and look at the status at the final wait (A is untouched for n <= 64);
	// otherwise (several panels or a second sweep consuming Q) verify each panel right away.
	const bool deferred = may_fall_back && n <= PW && !reorth;
	const bool check_now = may_fall_back && !deferred;
	const unsigned gb = (unsigned)std::min<size_t>(1024, cdiv(n * n, 256));
	c.min_level = 2;
	c.used_shift = c.used_householder = false;
	float scond1 = 0.0f;                                 // scaled conditioning S reported by the accepted first sweep (single panel)

	// R-factor engine levels: 2 bf16-split Gram (memory-bound), 1 fp64 Gram, 0 Householder TSQR.  Deferred mode runs a level
	// speculatively and steps down when chol16_kernel rejected it; with check_now panel_qr escalates per panel by itself.
	const int first_level = !use_gram ? 0 : c.gram_level;
	bool wide_done = false;
	if (c.wide && n > PW && n <= 2 * PW && may_fall_back && first_level == 2 && !c.comm.active() && !c.resume_accepted) {
		// one Cholesky-QR panel over all n columns; rejected (ill conditioned) -> the 64-column panel path below, A is still intact
		float* r1 = c.wq + L.r1; float* r2 = c.wq + L.r2;
		unsigned st = 1u;
		int rc = sweep_wide(c, engine, q, ldq, reorth ? r1 : r, reorth ? n : ldr, a, lda, m, n);
		if (rc) return rc;
		rc = read_status(c, c.slot, &st);
		if (rc) return rc;
		if (st == 0 && reorth) {
			rc = sweep_wide(c, engine, q, ldq, r2, n, q, ldq, m, n);
			if (rc) return rc;
			rc = read_status(c, c.slot, &st);
			if (rc) return rc;
			if (st != 0) {                               // (Q1 is well conditioned: not expected) second sweep on the panel path
				rc = sweep(c, engine, first_level, /*check_now=*/true, q, ldq, r2, n, q, ldq, m, n);
				if (rc) return rc;
				hipLaunchKernelGGL(tsqrmi::zero_lower_kernel, dim3(gb), dim3(256), 0, c.st, r2, n, (int)n);
			}
			launch_rmul(r, ldr, r2, n, r1, n, n, c.st);
			HIPCHK(hipGetLastError());
			rc = wait_done(c);
			if (rc) return rc;
			st = 0;
		}
		wide_done = (st == 0);
	}
	if (c.resume_accepted) scond1 = c.resume_scond;
	for (int level = (c.start_level >= 0 ? std::min(c.start_level, first_level) : first_level); level >= 0 && !wide_done && !c.resume_accepted; level--) {
		int rc;
		if (!reorth) {
			// level 0 reached in the speculative (deferred) mode: both Gram levels were rejected and the fp64 Gram matrix of A is
			// still in the work buffer: panel_qr takes the shifted-Cholesky path on it before the Householder fold
			const bool retry_checked = (level == 0 && deferred && first_level >= 1);
			rc = sweep(c, engine, retry_checked ? R_SHIFT_DIRECT : level, check_now || retry_checked, q, ldq, r, ldr, a, lda, m, n);
			if (rc) return rc;
			if (n > PW) hipLaunchKernelGGL(tsqrmi::zero_lower_kernel, dim3(gb), dim3(256), 0, c.st, r, ldr, (int)n);
		} else {
			// two sweeps: A = Q1 R1, then Q1 = Q R2 in place, R = R2 * R1 (the reference's BCGS2 plays this role).  R1 and R2 live in
			// the work buffer (packed, ld n); every sweep writes their upper triangles in full and rmul_kernel reads nothing else,
			// so neither needs zero-filling, and the product is written straight into the caller's r (zeros below the diagonal)
			float* r1 = c.wq + L.r1; float* r2 = c.wq + L.r2;
			// single panel: the first sweep's (last) apply launch accumulates Q^T Q while the block is in LDS, so that the second
			// sweep's bf16-level Gram pass over Q is not needed
			const bool fuse = n <= PW && c.policy == 0 && level == 2;
			if (n <= PW && c.policy == 0 && level == first_level && level == 2 && !t_prof.on && c.hsig.dev) {
				// Round 4: CholeskyQR2 / shifted CholeskyQR3 on the bf16-split Gram matrix, decided on the device, no pass wasted.
				//  sweep 1  Gram pass of A -> Cholesky under the RELAXED rule (another sweep follows: Q1 must come out well conditioned, not
				//           orthonormal) -- and when even that rule rejects, the same launch factors G + s I at once (shifted Cholesky QR,
				//           Fukaya et al. 2020).  s = c trace(G), c = 8 * 2^-23 / sqrt(rows): four times the Frobenius bound of the bf16-split Gram
				//           matrix's own error (products good to 2^-23, errors averaging over the rows: |dG_ij| ~ 2^-22 / sqrt(rows) sqrt(g_ii g_jj),
				//           measured as 8e-6 S / sqrt(rows) in Q^T Q), so the fp64 Gram pass of A (108 us) and its rejected Cholesky are not needed.
				//           -> Q1 = A inverse(R1), Gram tiles of Q1 from the same launch; Q1 leaves with plain stores in ASCENDING block order, so
				//           that sweep 2 (descending) starts with the half of Q1 the Infinity Cache still holds (ApplyArgs::forward).
				//  The host reads sweep 1's verdict word WHILE the apply pass of sweep 1 runs, then enqueues
				//  accepted plain   : sweep 2 under the strict rule -> Q, R = R2 R1 (CholeskyQR2: what rounds 1-3 did for such input);
				//  accepted shifted : sweep 2 under the relaxed rule (cond(Q1) ~ 1 / sqrt(c): 1e3 .. 4e3) with Gram tiles of Q2 from its
				//                     apply launch, sweep 3 under the strict rule -> Q, R = R3 R2 R1 (shifted CholeskyQR3);
				//  rejected         : (non-finite input, or columns near the fp32 denormal range) the checked ladder below, from scratch.
				// Every launch behind a rejected Cholesky skips itself; A is never written.
				constexpr unsigned PENDING = 0xffffffffu;
				volatile unsigned* hw = reinterpret_cast<volatile unsigned*>(c.hsig.host);
				float* r3 = c.wq + L.r5; float* r4 = c.wq + L.r6;    // (L.r3 / L.r4 belong to panel_qr's own shifted two-step)
				c.gramq_part = reinterpret_cast<double*>(c.wr); c.gramq_cap = gram_plan(m, n).nblocks; c.gramq_nparts = 0;
				c.slot = 0; c.prev_slot = -1;
				hw[0] = PENDING;
				c.chol_relax = 1; c.chol_retry_shift = 1;
				c.q_for_next_sweep = (double)ldq * (double)n * sizeof(float) <= 300.0e6;      // (a Q the Infinity Cache can hold half of)
				rc = sweep(c, engine, 2, /*check_now=*/false, q, ldq, r1, n, a, lda, m, n);
				c.q_for_next_sweep = false;
				c.chol_relax = 0; c.chol_retry_shift = 0;
				const bool have_gramq = c.gramq_nparts > 0;
				if (rc) { c.gramq_part = nullptr; c.gramq_cap = 0; return rc; }
				unsigned v0 = PENDING;
				for (;;) {                                       // sweep 1's verdict (its apply pass is running meanwhile)
					for (int i = 0; i < 20000 && v0 == PENDING; i++) { v0 = hw[0]; if (v0 == PENDING) __builtin_ia32_pause(); }
					if (v0 != PENDING) break;
					const hipError_t e = hipStreamQuery(c.st);
					if (e == hipSuccess) { v0 = hw[0]; if (v0 == PENDING) v0 = 1u; break; }
					if (e != hipErrorNotReady) { c.gramq_part = nullptr; c.gramq_cap = 0; HIPCHK(e); }
				}
				unsigned s1 = 1u, s2 = 1u;
				if (v0 == 0u) {
					c.gramq_part = nullptr; c.gramq_cap = 0;
					c.gramq_ready = have_gramq;
					c.slot = 1; c.prev_slot = 0;
					rc = sweep(c, engine, 2, /*check_now=*/false, q, ldq, r2, n, q, ldq, m, n);
					c.gramq_ready = false;
					c.slot = 0; c.prev_slot = -1;
					if (rc) return rc;
					launch_rmul(r, ldr, r2, n, r1, n, n, c.st);
					HIPCHK(hipGetLastError());
					rc = read_status(c, 1, &s1);
					if (rc) return rc;
					if (s1 == 0) break;                          // both sweeps accepted: done (min_level was set by panel_qr)
					// the first sweep stands (Q holds Q1, r1 is valid); only the second one must be redone, checked, one level down
					c.min_level = 2;
					rc = sweep(c, engine, 1, /*check_now=*/true, q, ldq, r2, n, q, ldq, m, n);
					if (rc) return rc;
					c.min_level = std::min(c.min_level, 2);
					launch_rmul(r, ldr, r2, n, r1, n, n, c.st);
					HIPCHK(hipGetLastError());
					rc = wait_done(c);
					if (rc) return rc;
					break;
				}
				if (v0 == 2u) {
					// shifted: cond(Q1) ~ sqrt(c n / 3) cond(A).  Sweep 2 first tries the bf16-split Gram matrix of Q1 -- free, its tiles came out
					// of sweep 1's apply launch -- under the relaxed rule (large row counts: c is small enough for cond(A) up to ~1e8); when
					// that is rejected (Q still holds Q1: the apply pass skipped itself) it takes the fp64 Gram matrix of Q1 in a pass of its
					// own.  Sweeps 2 and 3 are enqueued together, speculatively; sweep 3 (strict rule) reuses slot 0, whose first verdict
					// has been read.
					bool done3 = false;
					for (int att = have_gramq ? 0 : 1; att < 2; att++) {
						c.gramq_part = reinterpret_cast<double*>(c.wr); c.gramq_cap = gram_plan(m, n).nblocks;
						c.gramq_ready = (att == 0);
						c.slot = 1; c.prev_slot = -1;            // (sweep 1 is known to be accepted)
						c.chol_relax = 1;
						rc = sweep(c, engine, att == 0 ? 2 : 1, /*check_now=*/false, q, ldq, r2, n, q, ldq, m, n);
						c.chol_relax = 0; c.gramq_ready = false;
						c.gramq_part = nullptr; c.gramq_cap = 0;
						if (!rc) {
							c.gramq_ready = have_gramq;          // (the same apply variant ran: fused for every engine but the fp32-MFMA one)
							c.slot = 0; c.prev_slot = 1;
							rc = sweep(c, engine, 2, /*check_now=*/false, q, ldq, r3, n, q, ldq, m, n);
							c.gramq_ready = false;
						}
						c.slot = 0; c.prev_slot = -1;
						if (rc) return rc;
						launch_rmul3(r, ldr, r3, r2, r1, r4, n, c.st);
						HIPCHK(hipGetLastError());
						rc = read_status(c, 0, &s2);
						if (rc) return rc;
						rc = read_status(c, 1, &s1, nullptr, /*wait=*/false);
						if (rc) return rc;
						if (s1 == 0 && s2 == 0) { done3 = true; break; }
						if (s1 == 0) break;                      // sweep 3 alone was rejected: Q holds Q2 (checked sweep below)
					}
					if (!done3) {
						// Q1 is numerically rank deficient (e.g. exactly dependent columns), or sweep 3 found Q2 short of the strict rule.
						// Q holds Q1 (sweep 2 rejected: its apply pass skipped itself) or Q2 -- A may be gone (q may alias a), so the rest is
						// done on Q in place by checked sweeps, which escalate per panel (fp64 Gram -> shifted -> Householder)
						if (s1 != 0) {
							rc = sweep(c, engine, 1, /*check_now=*/true, q, ldq, r2, n, q, ldq, m, n);
							if (rc) return rc;
						}
						rc = sweep(c, engine, 2, /*check_now=*/true, q, ldq, r3, n, q, ldq, m, n);
						if (rc) return rc;
						launch_rmul3(r, ldr, r3, r2, r1, r4, n, c.st);
						HIPCHK(hipGetLastError());
						rc = wait_done(c);
						if (rc) return rc;
					}
					c.min_level = 0; c.used_shift = true;
					break;
				}
				c.gramq_part = nullptr; c.gramq_cap = 0;
				rc = wait_done(c);                               // rejected (non-finite input, columns near the denormal range): everything enqueued skipped itself
				if (rc) return rc;
				c.min_level = 2;
				level = 1;                                       // the checked ladder, from the fp64 Gram level, on the untouched A
			} else if (n <= PW && c.policy == 0 && level == first_level && !t_prof.on) {
				// Optimistic attempt: both sweeps, the R product and the completion flag are enqueued without looking at a verdict.
				// Device-side chain: apply 1 skips when Cholesky 1 rejected; Cholesky 2 then reports "rejected" at once; apply 2 (in
				// place) skips when Cholesky 2 rejected -- so A stays intact and Q holds Q1 or garbage, never a half-applied state.
				if (fuse) { c.gramq_part = reinterpret_cast<double*>(c.wr); c.gramq_cap = gram_plan(m, n).nblocks; c.gramq_nparts = 0; }
				c.slot = 0; c.prev_slot = -1;
				rc = sweep(c, engine, level, /*check_now=*/false, q, ldq, r1, n, a, lda, m, n);
				c.gramq_part = nullptr; c.gramq_cap = 0;
				if (!rc) {
					c.gramq_ready = fuse && c.gramq_nparts > 0;
					c.slot = 1; c.prev_slot = 0;
					rc = sweep(c, engine, level, /*check_now=*/false, q, ldq, r2, n, q, ldq, m, n);
					c.gramq_ready = false;
				}
				c.slot = 0; c.prev_slot = -1;
				if (rc) return rc;
				launch_rmul(r, ldr, r2, n, r1, n, n, c.st);
				HIPCHK(hipGetLastError());
				unsigned s0 = 1u, s1 = 1u;
				rc = read_status(c, 0, &s0);
				if (rc) return rc;
				rc = read_status(c, 1, &s1, nullptr, /*wait=*/false);
				if (rc) return rc;
				if (s0 == 0 && s1 == 0) break;               // both sweeps accepted: done (min_level was set by panel_qr)
				c.min_level = 2;
				if (s0 == 0) {
					// the first sweep stands (Q holds Q1, r1 is valid); only the second one must be redone, now checked and below
					// the level that was just rejected
					rc = sweep(c, engine, level - 1, /*check_now=*/true, q, ldq, r2, n, q, ldq, m, n);
					if (rc) return rc;
					c.min_level = std::min(c.min_level, level);  // (first sweep ran at `level`)
					launch_rmul(r, ldr, r2, n, r1, n, n, c.st);
					HIPCHK(hipGetLastError());
					rc = wait_done(c);
					if (rc) return rc;
					break;
				}
				level = std::max(level - 1, 0);              // first sweep rejected at this level: checked path from the next one
			}
			const bool fuse2 = n <= PW && c.policy == 0 && first_level == 2;
			if (fuse2) { c.gramq_part = reinterpret_cast<double*>(c.wr); c.gramq_cap = gram_plan(m, n).nblocks; c.gramq_nparts = 0; }
			rc = sweep(c, engine, level, check_now, q, ldq, r1, n, a, lda, m, n);
			c.gramq_part = nullptr; c.gramq_cap = 0;
			if (rc) return rc;
			c.gramq_ready = fuse2 && c.gramq_nparts > 0;     // the second sweep always starts at the first level again (Q1 is well conditioned)
			rc = sweep(c, engine, first_level, check_now, q, ldq, r2, n, q, ldq, m, n);
			c.gramq_ready = false;
			if (rc) return rc;
			launch_rmul(r, ldr, r2, n, r1, n, n, c.st);
		}
		HIPCHK(hipGetLastError());
		if (level > 0 && deferred) {
			unsigned status = 0;
			rc = read_status(c, 0, &status, &scond1);
			if (rc) return rc;
			if (status != 0) { c.min_level = 2; continue; }       // rejected: step down and redo
		} else {
			rc = wait_done(c);                           // completion flag in the pinned words, or a plain stream sync
			if (rc) return rc;
		}
		break;
	}
	// n <= 16 without reorthogonalisation: the reference's tsqr16 builds Q from its Householder tree and stays O(eps) orthogonal at
	// any conditioning, while Q = A * inverse(R) loses orthogonality like cond * eps.  When the accepted sweep reports a scaled
	// conditioning beyond 32 (measured loss ~ 3.6e-7 * sqrt(S)), a second sweep on Q in place restores O(eps): R <- R2 * R.
	if (!reorth && deferred && n <= 16 && c.min_level >= 1 && !c.used_shift && !c.used_householder && scond1 > 32.0f) {
		float* r1 = c.wq + L.r1; float* r2 = c.wq + L.r2;
		hipLaunchKernelGGL(tsqrmi::copy2d_kernel, dim3(gb), dim3(256), 0, c.st, r1, n, r, ldr, (int)n, (int)n);
		HIPCHK(hipGetLastError());
		int rc = sweep(c, engine, first_level, /*check_now=*/true, q, ldq, r2, n, q, ldq, m, n);
		if (rc) return rc;
		launch_rmul(r, ldr, r2, n, r1, n, n, c.st);
		HIPCHK(hipGetLastError());
		rc = wait_done(c);
		if (rc) return rc;
	}
	t_last_engine = !use_gram ? 0 : (c.used_householder ? 2 : (c.used_shift ? 4 : (c.min_level == 2 ? 3 : (c.min_level == 1 ? 1 : 2))));
	if (wide_done) t_last_engine = 5;                    // all n <= 128 columns as one Cholesky-QR panel (bf16-split Gram level)
	prof_collect();
	return TSQR_MI_SUCCESS;
}

void init_ctx(Ctx& c, void* wq, void* wr, size_t m_layout, size_t n, void* stream, bool keep_in_flight = false) {
	// every entry point starts here: calls of this thread still in flight (submit without finish) have their verdicts read first --
	// what follows may reuse their pinned words and status slots
	if (!keep_in_flight) (void)latch_all();
	c.st = reinterpret_cast<hipStream_t>(stream);
	c.dev = cur_device();
	c.wq = reinterpret_cast<float*>(wq);
	c.wr = reinterpret_cast<float*>(wr);
	c.L = wq_layout(m_layout, n);
	c.cross_slots = cross_slots(m_layout, n);
	c.policy = g_set.policy.load();
	c.gram_level = g_set.gram_level.load();
	c.wide = g_set.wide.load() != 0;
}

size_t working_r_need(size_t m, size_t n) {
	size_t need = 0;
	for (size_t P = 0; P < n; P += PW) {
		need = std::max(need, make_plan(m, std::min(PW, n - P)).stack_a);
		need = std::max(need, (n <= PW ? 2 : 1) * gram_plan(m, std::min(PW, n - P)).part_floats);   // (n <= 64: two sets, stream_of_calls_chained)
		if (n > PW) need = std::max(need, cross_slots(m, n) * 16 * 256 * 2);   // (coupling partials: sweep)
	}
	if (n > PW) need = std::max(need, wide_part_floats(m));   // (sweep_wide: the whole matrix for n <= 128, 128-column blocks of the panel loop beyond)
	return need;
}

int qr_dist_common(Ctx& c, int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda, size_t m_local, size_t n,
                   void* wq, void* wr, int nranks, void* stream) {
	// (n > 64: 64-column panels coupled by block Gram-Schmidt, every coupling coefficient block all-reduced like the Gram tiles; the
	// one-panel path for 64 < n <= 128 is a single-GPU path)
	if (m_local == 0 || n == 0 || nranks < 1) return TSQR_MI_ERROR_INVALID_SIZE;
	const int engine = engine_of(mode);
	if (engine < 0) { t_last_error = "compute_mode not implemented on gfx950"; return TSQR_MI_ERROR_UNSUPPORTED; }
	init_ctx(c, wq, wr, std::max(m_local, (size_t)nranks * std::min(n, PW)), n, stream);
	c.comm.nranks = nranks;
	c.rows_global = (double)m_local * (double)nranks;    // host-side estimate only: the device thresholds use the all-reduced count
	resolve_host_sig(c, nullptr, m_local);
	return qr_core(c, engine, reorth, q, ldq, r, ldr, a, lda, m_local, n);
}

}  // namespace

extern "C" {

int tsqr_mi_version(void) { return 410; }
const char* tsqr_mi_last_error(void) { return t_last_error.c_str(); }

size_t tsqr_mi_batch_size_log2(size_t m) { return ref_bs_log2(m); }
size_t tsqr_mi_batch_size(size_t m) { return ref_bs(m); }

size_t tsqr_mi_working_q_size(size_t m, size_t n) {
	if (m == 0 || n == 0) return 0;
	return std::max(ref_wq(m, n), wq_layout(m, n).total);
}
size_t tsqr_mi_working_r_size(size_t m, size_t n) {
	if (m == 0 || n == 0) return 0;
	return std::max(ref_wr(m, n), working_r_need(m, n));
}
size_t tsqr_mi_working_l_size(size_t m) { return m == 0 ? 0 : std::max<size_t>(ref_bs(m) + 1, 8); }   // 8 words: status words + completion flag
size_t tsqr_mi_working_reorth_size(size_t m) { return 16 * 16 * 2 + m * 16; }

// row-partitioned call: the work buffers must also hold the restacked (nranks n) x n matrix of gathered R factors and its fold
size_t tsqr_mi_working_q_size_dist(size_t m_local, size_t n, int nranks) {
	if (m_local == 0 || n == 0 || nranks < 1) return 0;
	return tsqr_mi_working_q_size(std::max(m_local, (size_t)nranks * std::min(n, PW)), n);
}
size_t tsqr_mi_working_r_size_dist(size_t m_local, size_t n, int nranks) {
	if (m_local == 0 || n == 0 || nranks < 1) return 0;
	const size_t pc = std::min(n, PW);                   // (the gathered stack is one panel's: nranks * pc rows of pc columns)
	const size_t stack = (((size_t)nranks * pc * pc + 63) & ~(size_t)63) + make_plan((size_t)nranks * pc, pc).stack_a;
	return std::max(tsqr_mi_working_r_size(std::max(m_local, (size_t)nranks * pc), n), std::max(working_r_need(m_local, n), stack));
}

void tsqr_mi_profile_enable(int on) {
	if (on && !t_prof.created) {
		for (int i = 0; i < 2 * Prof::MAXEV; i++) (void)hipEventCreate(&t_prof.ev[i]);
		t_prof.created = true;
	}
	t_prof.on = on != 0;
	t_prof.n = 0;
	for (int k = 0; k < KC_COUNT; k++) { t_prof.ms[k] = 0; t_prof.launches[k] = 0; }
}
int tsqr_mi_profile_read(double* ms, long* launches, int max_classes) {
	prof_collect();
	const int k = std::min(max_classes, (int)KC_COUNT);
	for (int i = 0; i < k; i++) { ms[i] = t_prof.ms[i]; launches[i] = t_prof.launches[i]; }
	return k;
}

void tsqr_mi_set_policy(int policy) {
	switch (policy) {
		case 0: g_set.policy = 0; g_set.gram_level = 2; g_set.wide = 1; break;    // auto
		case 5: g_set.policy = 0; g_set.gram_level = 2; g_set.wide = 0; break;    // auto, 64-column panels only (no one-panel path for n <= 128)
		case 1: g_set.policy = 1; g_set.gram_level = 2; break;    // always Householder TSQR
		case 2: g_set.policy = 2; g_set.gram_level = 1; break;    // always fp64 Gram (no fallback)
		case 3: g_set.policy = 2; g_set.gram_level = 2; break;    // always bf16-split Gram (no check, no fallback)
		case 4: g_set.policy = 0; g_set.gram_level = 1; break;    // auto without the bf16-split level
		default: break;
	}
}
int tsqr_mi_last_engine(void) { return t_last_engine; }

void tsqr_mi_set_tuning(int level0_waves, int tree_chunks_per_wave) {
	if (level0_waves > 0) g_set.level0_waves = level0_waves;
	if (tree_chunks_per_wave > 1) g_set.tree_cpw = tree_chunks_per_wave;
}
void tsqr_mi_set_tuning2(int gram_waves, int apply_waves) {
	if (gram_waves > 0) g_set.gram_waves = gram_waves;
	if (apply_waves > 0) g_set.apply_wgs = std::max(1, apply_waves / 4);   // the apply kernel is launched as a persistent grid of workgroups (4 waves each)
}

int tsqr_mi_qr_f32(int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                   size_t m, size_t n, void* wq_v, void* wr_v, float* reorth_w, unsigned* d_wl, unsigned* h_wl,
                   void* stream) {
	(void)reorth_w; (void)d_wl;
	if (n > m || m == 0 || n == 0) return TSQR_MI_ERROR_INVALID_SIZE;     // reference src/blockqr.cu:409-411
	const int engine = engine_of(mode);
	if (engine < 0) { t_last_error = "compute_mode not implemented on gfx950"; return TSQR_MI_ERROR_UNSUPPORTED; }
	Ctx c;
	init_ctx(c, wq_v, wr_v, m, n, stream);
	c.rows_global = (double)m;
	resolve_host_sig(c, h_wl, m);
	return qr_core(c, engine, reorth, q, ldq, r, ldr, a, lda, m, n);
}

// ---- submit / finish: the first attempt of a call (bf16-split Gram level, everything speculative) is enqueued and the host returns;
// finish reads the verdict and, for a rejected matrix, runs the rest of the ladder as the blocking call does (include/tsqr_mi.h) ----
// CallEnv: what a row-partitioned call adds to the arguments (its collectives); env.dist == false: the single-GPU call.
struct CallEnv { bool dist = false; Comm comm; int nranks = 1; };
static void env_ctx(Ctx& c, const CallEnv& env, void* wq_v, void* wr_v, size_t m, size_t n, unsigned* h_wl, void* stream, bool keep_in_flight) {
	if (env.dist) {                                      // (as qr_dist_common)
		c.comm = env.comm;
		init_ctx(c, wq_v, wr_v, std::max(m, (size_t)env.nranks * std::min(n, PW)), n, stream, keep_in_flight);
		c.comm.nranks = env.nranks;
		c.rows_global = (double)m * (double)env.nranks;
		resolve_host_sig(c, nullptr, m);
	} else {
		init_ctx(c, wq_v, wr_v, m, n, stream, keep_in_flight);
		c.rows_global = (double)m;
		resolve_host_sig(c, h_wl, m);
	}
}
// own_flag: a one-thread completion kernel behind the attempt (the public entry: always).  The loop entries leave it out for every call
// but the last and let the NEXT call's first kernel raise the word instead (`announce`: the ticket submitted just before this one).
static int submit_impl(const CallEnv& env, int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                       size_t m, size_t n, void* wq_v, void* wr_v, unsigned* h_wl, void* stream, tsqr_mi_ticket* t,
                       tsqr_mi_ticket* announce, bool own_flag) {
	if (!t) return TSQR_MI_ERROR_INVALID_SIZE;
	*t = tsqr_mi_ticket{};
	t->mode = mode; t->reorth = reorth; t->q = q; t->r = r; t->a = a; t->ldq = ldq; t->ldr = ldr; t->lda = lda; t->m = m; t->n = n;
	t->wq = wq_v; t->wr = wr_v; t->stream = stream; t->h_wl = h_wl;
	if ((!env.dist && n > m) || m == 0 || n == 0 || env.nranks < 1) return t->state = TSQR_MI_ERROR_INVALID_SIZE;
	const int engine = engine_of(mode);
	if (engine < 0) { t_last_error = "compute_mode not implemented on gfx950"; return t->state = TSQR_MI_ERROR_UNSUPPORTED; }
	Ctx c;
	env_ctx(c, env, wq_v, wr_v, m, n, h_wl, stream, /*keep_in_flight=*/true);
	const bool narrow = n <= PW, wide = c.wide && n > PW && n <= 2 * PW && !env.dist;
	if (!(c.policy == 0 && c.gram_level == 2 && !reorth && (narrow || wide) && c.hsig.dev && !t_prof.on && !g_set.debug)) {
		// no speculative first attempt for this call: run it here (finish returns its state)
		const int rc = latch_all();
		return t->state = rc ? rc : qr_core(c, engine, reorth, q, ldq, r, ldr, a, lda, m, n);
	}
	const int slot = (int)(t_submits++ & 1u);
	if (t_pending[slot]) { const int rc = ticket_latch(t_pending[slot]); if (rc) return t->state = rc; }
	c.fold_cor = (engine == 1);
	c.min_level = 2;
	c.slot = slot; c.prev_slot = -1;
	if (announce && announce->pending == 1 && !announce->own_flag) {
		if (announce->words == c.hsig.host && announce->stream == stream) {
			c.announce_word = c.hsig.dev + 4 * announce->slot + 3;
			c.announce_seq = announce->seq;
		} else {                                         // other pinned words or another stream: a completion kernel of its own after all
			hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(announce->stream),
			                   announce->words_dev + 4 * announce->slot + 3, announce->seq);
			(void)hipGetLastError();
		}
	}
	const int rc = narrow ? sweep(c, engine, 2, /*check_now=*/false, q, ldq, r, ldr, a, lda, m, n) : sweep_wide(c, engine, q, ldq, r, ldr, a, lda, m, n);
	if (rc) return t->state = rc;
	if (c.announce_word) {                               // (no kernel of the attempt carried the announcement: raise the word here)
		hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, c.announce_word, c.announce_seq);
		(void)hipGetLastError();
	}
	unsigned seq = ++g_seq;
	if (seq == 0) seq = ++g_seq;
	reinterpret_cast<volatile unsigned*>(c.hsig.host)[4 * slot + 3] = 0;
	if (own_flag) {
		hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, c.hsig.dev + 4 * slot + 3, seq);
		if (hipGetLastError() != hipSuccess) {           // the attempt is enqueued but cannot announce itself: drain and run the call
			HIPCHK(hipStreamSynchronize(c.st));
			return t->state = qr_core(c, engine, reorth, q, ldq, r, ldr, a, lda, m, n);
		}
	}
	t->slot = slot; t->seq = seq; t->words = c.hsig.host; t->words_dev = c.hsig.dev; t->own_flag = own_flag ? 1 : 0; t->pending = 1;
	t_pending[slot] = t;
	return TSQR_MI_SUCCESS;
}
int tsqr_mi_qr_f32_submit(int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                          size_t m, size_t n, void* wq_v, void* wr_v, float* reorth_w, unsigned* d_wl, unsigned* h_wl,
                          void* stream, tsqr_mi_ticket* t) {
	(void)reorth_w; (void)d_wl;
	return submit_impl(CallEnv{}, mode, reorth, q, ldq, r, ldr, a, lda, m, n, wq_v, wr_v, h_wl, stream, t, nullptr, /*own_flag=*/true);
}

static int finish_impl(const CallEnv& env, tsqr_mi_ticket* t) {
	if (!t) return TSQR_MI_ERROR_INVALID_SIZE;
	if (t->pending == 0) return t->state;
	if (t->pending == 1) { const int rc = ticket_latch(t); if (rc) { t->pending = 0; return t->state = rc; } }
	t->pending = 0;
	const bool wide = t->n > PW;
	if (t->verdict == 0 && !(t->n <= 16 && t->scond > 32.0f)) {
		t_last_engine = wide ? 5 : 3;                    // accepted at the bf16-split Gram level: Q and R are in place
		return t->state = TSQR_MI_SUCCESS;
	}
	// rejected (or the n <= 16 second sweep is due): the blocking path from here on; calls submitted after this one have run their
	// attempts by the time anything below executes (stream order) and init_ctx reads their verdicts before their slots are reused
	Ctx c;
	env_ctx(c, env, t->wq, t->wr, t->m, t->n, t->h_wl, t->stream, /*keep_in_flight=*/false);
	if (t->verdict == 0) { c.resume_accepted = true; c.resume_scond = t->scond; }
	else if (wide) c.wide = false;                       // the one-panel attempt was rejected: 64-column panels (A is intact)
	else c.start_level = 1;                              // the bf16-split level was rejected: the ladder resumes at the fp64 Gram level
	return t->state = qr_core(c, engine_of(t->mode), t->reorth, t->q, t->ldq, t->r, t->ldr, t->a, t->lda, t->m, t->n);
}
int tsqr_mi_qr_f32_finish(tsqr_mi_ticket* t) { return finish_impl(CallEnv{}, t); }

// ---- a stream of calls: `count` factorisations of ONE shape on one stream with one set of work buffers.  The loop entries pass the same
// (q, r, a) every time (the reference's speed protocol, src/test.cu:299-309), the batch entry one triple per call (a caller with many
// matrices, reference README.md:52-87 in a loop).  Mats hides the difference; Call carries what all calls of the stream share. ----
constexpr int NOT_MINE = -1000000;                     // "this schedule is not for this stream of calls; nothing enqueued" (not a hipError_t)
struct Mats {
	float* const* qs = nullptr; float* const* rs = nullptr; float* const* as = nullptr;   // host arrays of device pointers (batch entries) ...
	float* q0 = nullptr; float* r0 = nullptr; float* a0 = nullptr;                        // ... or one triple for every call (loop entries)
	int* states = nullptr;                               // optional: the state of every call
	bool same() const { return as == nullptr; }
	float* q(int i) const { return qs ? qs[i] : q0; }
	float* r(int i) const { return rs ? rs[i] : r0; }
	float* a(int i) const { return as ? as[i] : a0; }
	void state(int i, int st) const { if (states) states[i] = st; }
	Mats from(int i) const {
		Mats t = *this;
		if (!same()) { t.qs += i; t.rs += i; t.as += i; }
		if (states) t.states += i;
		return t;
	}
};
struct Call { int mode, reorth; size_t ldq, ldr, lda, m, n; void* wq; void* wr; unsigned* h_wl; void* stream; };

// The chained schedules launch the Gram pass of call i + 1 BEFORE the apply pass of call i, in the launch that writes R of call i.  That
// is the blocking order only if no output of call i is an input of call i + 1: Q(i) and R(i) must not overlap A(i + 1).  (A loop over one
// triple factored in place, q == a, fails this test: call i + 1 of the blocking loop factors the Q that call i left there.)
static bool chain_order_safe(const Mats& mt, int count, const Call& cl, size_t esz = sizeof(float)) {
	auto overlap = [](const void* p, size_t pb, const void* s, size_t sb) {
		const uintptr_t a0 = reinterpret_cast<uintptr_t>(p), b0 = reinterpret_cast<uintptr_t>(s);
		return a0 < b0 + sb && b0 < a0 + pb;
	};
	const size_t qb = ((cl.n - 1) * cl.ldq + cl.m) * esz, ab = ((cl.n - 1) * cl.lda + cl.m) * esz, rb = ((cl.n - 1) * cl.ldr + cl.n) * esz;
	const int pairs = mt.same() ? 1 : count - 1;
	for (int i = 0; i < pairs; i++) {
		const int j = mt.same() ? i : i + 1;
		if (overlap(mt.q(i), qb, mt.a(j), ab) || overlap(mt.r(i), rb, mt.a(j), ab)) return false;
	}
	return true;
}

// A batch of different matrices takes a chained schedule only while two matrices share the Infinity Cache (Settings::chain_max_mib)
static bool chain_fits_cache(const Mats& mt, const Call& cl) {
	return mt.same() || (double)cl.lda * (double)cl.n * sizeof(float) <= (double)g_set.chain_max_mib * 1048576.0;
}

// spin on a completion word of the pinned words (the stream is looked at now and then so that a failed launch cannot hang the caller)
static int wait_word(volatile unsigned* word, unsigned seq, hipStream_t st) {
	for (;;) {
		for (int k = 0; k < 20000; k++) {
			if (*word == seq) return 0;
			__builtin_ia32_pause();
		}
		const hipError_t e = hipStreamQuery(st);
		if (e == hipSuccess) return 0;
		if (e != hipErrorNotReady) HIPCHK(e);
	}
}

// one call of the stream as a plain blocking call with its whole ladder
static int blocking_one(const CallEnv& env, const Mats& mt, int i, const Call& cl) {
	if (!env.dist)
		return tsqr_mi_qr_f32(cl.mode, cl.reorth, mt.q(i), cl.ldq, mt.r(i), cl.ldr, mt.a(i), cl.lda, cl.m, cl.n, cl.wq, cl.wr, nullptr, nullptr, cl.h_wl, cl.stream);
	Ctx cc;
	cc.comm = env.comm;
	return qr_dist_common(cc, cl.mode, cl.reorth, mt.q(i), cl.ldq, mt.r(i), cl.ldr, mt.a(i), cl.lda, cl.m, cl.n, cl.wq, cl.wr, env.nranks, cl.stream);
}

// Call i of a chained stream was rejected by the bf16-split level: its apply pass skipped itself and A(i) is intact.  Everything of call
// i + 1 is enqueued behind it.  Drain the stream; call i gets its whole ladder as a blocking call; call i + 1 STANDS when it was accepted
// (it may have been factored in place -- an accepted call is never redone) and gets its ladder otherwise.  *done = calls dealt with; the
// caller goes on from there.  A loop over one triple would see every further attempt rejected as well: the rest of its count runs as
// blocking calls.  (Row-partitioned: the verdicts come from the all-reduced matrix -- every rank walks through here alike.)
static int rejected_tail(const CallEnv& env, const Mats& mt, int count, const Call& cl, int i, volatile unsigned* words, hipStream_t st, int* done) {
	HIPCHK(hipStreamSynchronize(st));
	prof_collect();
	const bool next = i + 1 < count, next_rejected = next && words[4 * ((i + 1) & 1)] != 0;     // (read before a blocking call reuses the words)
	int first = 0;
	if (mt.same()) {
		for (int k = i; k < count; k++) {
			const int s = blocking_one(env, mt, k, cl);
			if (s < 0) return s;
			mt.state(k, s);
			if (s && !first) first = s;
		}
		*done = count;
		return first;
	}
	int s = blocking_one(env, mt, i, cl);
	if (s < 0) return s;
	mt.state(i, s);
	first = s;
	*done = i + 1;
	if (next) {
		s = 0;
		if (next_rejected) { s = blocking_one(env, mt, i + 1, cl); if (s < 0) return s; }
		mt.state(i + 1, s);
		if (s && !first) first = s;
		*done = i + 2;
	}
	return first;
}

// The stream for full 64-column matrices of up to 2^20 rows (the shapes gram_blk_kernel takes): the R-factor chain of call i (reduction,
// Cholesky, verdict) runs inside the launch that is the Gram pass of call i + 1 (gram_blk_chain_kernel), so a call costs its two
// streaming passes and nothing else:
//     gram(0) | [chain(0) + gram(1)]  apply(0) | [chain(1) + gram(2)]  apply(1) | ... | reduce, Cholesky (launches of their own)  apply(last)
// Two sets of Gram partials (wr); verdict words alternate between the halves of the pinned words; the completion word of call i is raised
// by the first kernel behind apply(i).  Returns NOT_MINE when the shape / settings / operand order are not the ones this schedule is for
// (nothing enqueued).  A rejected verdict ends the schedule at that call (rejected_tail); *done tells the caller how far the stream got.
static int chained64(const Mats& mt, int count, const Call& cl, int* done) {
	*done = 0;
	const size_t m = cl.m, n = cl.n, lda = cl.lda, ldq = cl.ldq, ldr = cl.ldr;
	const int engine = engine_of(cl.mode);
	if (count < 3 || engine < 0 || cl.reorth || n != PW || m % 128 != 0 || m > ((size_t)1 << 20) || lda % 4 != 0 || lda > ((size_t)1 << 24) ||
	    lda < m || ldq < m || ldr < n)
		return NOT_MINE;
	for (int i = 0; i < (mt.same() ? 1 : count); i++)
		if ((reinterpret_cast<uintptr_t>(mt.a(i)) & 15) != 0) return NOT_MINE;
	if (!chain_order_safe(mt, count, cl) || !chain_fits_cache(mt, cl)) return NOT_MINE;
	Ctx c;
	init_ctx(c, cl.wq, cl.wr, m, n, cl.stream);
	c.rows_global = (double)m;
	resolve_host_sig(c, cl.h_wl, m);
	if (!(c.policy == 0 && c.gram_level == 2 && c.hsig.dev && !g_set.debug)) return NOT_MINE;
	c.fold_cor = (engine == 1);
	static DevOnce attr;
	if (attr.need(c.dev)) {
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_blk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GB_LDS_BYTES));
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_blk_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GB_LDS_BYTES));
		attr.done(c.dev);
	}
	const GramPlan g = gram_plan(m, n);
	const int nchunks = (int)(m / 128), nparts = std::min(nchunks, g.nblocks), nelem = 10 * 256, nred = nelem / 16;
	double* part[2] = {reinterpret_cast<double*>(c.wr), reinterpret_cast<double*>(c.wr) + (size_t)g.nblocks * nelem};
	unsigned* ticket = c.status_dev(0) + 8;              // (a word of the status area no kernel uses)
	HIPCHK(hipMemsetAsync(ticket, 0, sizeof(unsigned), c.st));
	volatile unsigned* words = reinterpret_cast<volatile unsigned*>(c.hsig.host);
	unsigned seq[2] = {0, 0};
	unsigned* announce = nullptr; unsigned announce_seq = 0;             // completion word the next launch raises
	auto gram_args = [&](int i) {
		tsqrmi::GramArgs ga{};
		ga.a = mt.a(i); ga.lda = lda; ga.m = m; ga.n = (int)n; ga.nchunks = nchunks; ga.cpw = g.cpw; ga.nwaves = g.nwaves;
		ga.part = part[i & 1];
		ga.announce = announce; ga.announce_seq = announce_seq; announce = nullptr;
		return ga;
	};
	auto chol_args = [&](int i) {
		tsqrmi::CholArgs ca{};
		ca.r = mt.r(i); ca.ldr = ldr; ca.z = c.wq + c.L.z;
		ca.status = c.status_dev(i & 1);
		ca.host_status = c.hsig.dev + 4 * (i & 1);
		ca.gsum = c.gsum();
		ca.rows = c.rows_global;
		ca.n = (int)n; ca.NT = 4; ca.level = 2; ca.scond_floor = g_set.bf16_scond_floor;
		return ca;
	};
	// one step = the launches of call i behind its Gram pass: chain(i) (with the Gram pass of call i + 1 when there is one), apply(i)
	auto step = [&](int i) -> int {
		if (i + 1 < count) {
			tsqrmi::ChainArgs ch{};
			ch.chol = chol_args(i);
			ch.part = part[i & 1]; ch.nparts = nparts; ch.ticket = ticket; ch.nred = nred;
			ProfScope ps(KC_GRAM, c.st);
			hipLaunchKernelGGL(tsqrmi::gram_blk_chain_kernel, dim3(nred + nparts), dim3(256), tsqrmi::GB_LDS_BYTES, c.st, gram_args(i + 1), ch);
		} else {
			// the last call: no Gram pass left to hide behind -- the chain as launches of its own (and the completion word of the call
			// before, which no Gram kernel is there to raise)
			if (announce) { hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, announce, announce_seq); announce = nullptr; }
			ProfScope ps(KC_CHOL, c.st);
			hipLaunchKernelGGL(tsqrmi::gram_reduce1_kernel, dim3(nred), dim3(256), 0, c.st, c.gsum(), part[i & 1], nparts, nelem, (double)m,
			                   nullptr, (size_t)0, nullptr, 0);
			hipLaunchKernelGGL(tsqrmi::chol16_kernel, dim3(1), dim3(1024), 0, c.st, chol_args(i));
		}
		HIPCHK(hipGetLastError());
		c.slot = i & 1;
		const int rc = apply_rinv(c, engine, mt.q(i), ldq, mt.a(i), lda, mt.r(i), ldr, m, n, /*z_ready=*/true, c.status_dev(i & 1));
		if (rc) return rc;
		unsigned sq = ++g_seq;
		if (sq == 0) sq = ++g_seq;
		seq[i & 1] = sq;
		words[4 * (i & 1) + 3] = 0;
		if (i + 1 < count) { announce = c.hsig.dev + 4 * (i & 1) + 3; announce_seq = sq; }   // raised by the first launch of step i + 1
		else hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, c.hsig.dev + 4 * (i & 1) + 3, sq);
		HIPCHK(hipGetLastError());
		return 0;
	};
	{
		ProfScope ps(KC_GRAM, c.st);
		hipLaunchKernelGGL(tsqrmi::gram_blk_kernel, dim3(nparts), dim3(256), tsqrmi::GB_LDS_BYTES, c.st, gram_args(0));
	}
	HIPCHK(hipGetLastError());
	int rc = step(0);
	if (rc) return rc;
	for (int i = 0; i < count; i++) {
		if (i + 1 < count) { rc = step(i + 1); if (rc) return rc; }
		rc = wait_word(words + 4 * (i & 1) + 3, seq[i & 1], c.st);
		if (rc) return rc;
		if (words[4 * (i & 1)] != 0) return rejected_tail(CallEnv{}, mt, count, cl, i, words, c.st, done);
		mt.state(i, TSQR_MI_SUCCESS);
		*done = i + 1;
	}
	t_last_engine = 3;
	prof_collect();
	return TSQR_MI_SUCCESS;
}

// The chained schedule for a stream of 128-column calls (one GPU; every 64-row block full: m % 64 == 0, the fast Gram form): the
// two-block factorisation of call i (45 us on one workgroup) rides in the Gram launch of call i + 1 (gram_wide_chain_kernel); the
// reduction of the partials stays a launch of its own in front of it:
//     gram(0) reduce(0) | [chol(0) + gram(1)]  reduce(1)  apply(0) | [chol(1) + gram(2)]  reduce(2)  apply(1) | ... | chol(last)  apply(last)
// Returns NOT_MINE when the stream is not one for it (nothing enqueued); a rejected verdict ends the schedule at that call (rejected_tail:
// its blocking call falls back to 64-column panels and overwrites A, as the blocking call does).
static int chained128(const Mats& mt, int count, const Call& cl, int* done) {
	*done = 0;
	const size_t m = cl.m, n = cl.n, lda = cl.lda, ldq = cl.ldq, ldr = cl.ldr;
	const int engine = engine_of(cl.mode);
	if (count < 3 || engine < 0 || cl.reorth || n != 2 * PW || m < n || m % 64 != 0 || m / 64 < 2 * (size_t)WIDE_MAX_WGS || lda > ((size_t)1 << 23) ||
	    lda < m || ldq < m || ldr < n || lda % 4 != 0)
		return NOT_MINE;
	for (int i = 0; i < (mt.same() ? 1 : count); i++)
		if ((reinterpret_cast<uintptr_t>(mt.a(i)) & 15) != 0) return NOT_MINE;
	if (!chain_order_safe(mt, count, cl) || !chain_fits_cache(mt, cl)) return NOT_MINE;
	Ctx c;
	init_ctx(c, cl.wq, cl.wr, m, n, cl.stream);
	c.rows_global = (double)m;
	resolve_host_sig(c, cl.h_wl, m);
	if (!(c.wide && c.policy == 0 && c.gram_level == 2 && c.hsig.dev && !g_set.debug)) return NOT_MINE;
	static DevOnce attr;
	if (attr.need(c.dev)) {
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_wide_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GW_LDS_BYTES));
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_wide_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GWC_LDS_BYTES));
		attr.done(c.dev);
	}
	float* w = c.wq + c.L.wide;
	double* gsum = reinterpret_cast<double*>(w);
	float* zw = w + WIDE_OFF_ZW;
	float* zf2 = w + WIDE_OFF_ZF2;
	const int nblk = (int)(m / 64), wgs = WIDE_MAX_WGS, nelem = 36 * 256;
	double* part = reinterpret_cast<double*>(c.wr);
	volatile unsigned* words = reinterpret_cast<volatile unsigned*>(c.hsig.host);
	unsigned seq[2] = {0, 0};
	unsigned* announce = nullptr; unsigned announce_seq = 0;
	auto gram_args = [&](int i) {
		tsqrmi::GramWideArgs ga{};
		ga.a = mt.a(i); ga.lda = lda; ga.m = m; ga.n = (int)n; ga.blk0 = 0; ga.nblk = nblk; ga.part = part;
		ga.announce = announce; ga.announce_seq = announce_seq; announce = nullptr;
		return ga;
	};
	auto chol_args = [&](int i) {
		tsqrmi::CholWideArgs wa{};
		wa.gsum = gsum; wa.r = mt.r(i); wa.ldr = ldr; wa.n = (int)n; wa.zf1 = c.wq + c.L.z; wa.zf2 = zf2; wa.zw = zw;
		wa.st1 = c.status_dev(2); wa.st2 = c.status_dev(3); wa.status = c.status_dev(i & 1);
		wa.host_status = c.hsig.dev + 4 * (i & 1);
		wa.rows = (double)m; wa.scond_floor = g_set.bf16_scond_floor;
		return wa;
	};
	auto reduce = [&]() {
		ProfScope ps(KC_CHOL, c.st);
		hipLaunchKernelGGL(tsqrmi::gram_reduce1_kernel, dim3(nelem / 16), dim3(256), 0, c.st, gsum, part, wgs, nelem, (double)m, nullptr, (size_t)0, nullptr, 0);
	};
	auto step = [&](int i) -> int {
		if (i + 1 < count) {
			{
				ProfScope ps(KC_GRAM, c.st);
				hipLaunchKernelGGL(tsqrmi::gram_wide_chain_kernel, dim3(1 + wgs), dim3(512), tsqrmi::GWC_LDS_BYTES, c.st, gram_args(i + 1), chol_args(i));
			}
			reduce();
		} else {
			if (announce) { hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, announce, announce_seq); announce = nullptr; }
			ProfScope ps(KC_CHOL, c.st);
			hipLaunchKernelGGL(tsqrmi::chol_wide_kernel, dim3(1), dim3(1024), 0, c.st, chol_args(i));
		}
		HIPCHK(hipGetLastError());
		tsqrmi::ApplyArgs aa{};
		aa.a = mt.a(i); aa.lda = lda; aa.q = mt.q(i); aa.ldq = ldq; aa.m = m; aa.n = (int)n; aa.z = zw; aa.skip_status = c.status_dev(i & 1);
		int rc;
		{
			ProfScope ps(KC_APPLY, c.st);
			rc = (engine == 0) ? launch_apply_wide<0>(c, aa) : (engine == 1 ? launch_apply_wide<1>(c, aa) : launch_apply_wide<2>(c, aa));
		}
		if (rc) return rc;
		HIPCHK(hipGetLastError());
		unsigned sq = ++g_seq;
		if (sq == 0) sq = ++g_seq;
		seq[i & 1] = sq;
		words[4 * (i & 1) + 3] = 0;
		if (i + 1 < count) { announce = c.hsig.dev + 4 * (i & 1) + 3; announce_seq = sq; }
		else hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, c.hsig.dev + 4 * (i & 1) + 3, sq);
		HIPCHK(hipGetLastError());
		return 0;
	};
	{
		ProfScope ps(KC_GRAM, c.st);
		hipLaunchKernelGGL((tsqrmi::gram_wide_kernel<true>), dim3(wgs), dim3(512), tsqrmi::GW_LDS_BYTES, c.st, gram_args(0));
	}
	reduce();
	HIPCHK(hipGetLastError());
	int rc = step(0);
	if (rc) return rc;
	for (int i = 0; i < count; i++) {
		if (i + 1 < count) { rc = step(i + 1); if (rc) return rc; }
		rc = wait_word(words + 4 * (i & 1) + 3, seq[i & 1], c.st);
		if (rc) return rc;
		if (words[4 * (i & 1)] != 0) return rejected_tail(CallEnv{}, mt, count, cl, i, words, c.st, done);
		mt.state(i, TSQR_MI_SUCCESS);
		*done = i + 1;
	}
	t_last_engine = 5;
	prof_collect();
	return TSQR_MI_SUCCESS;
}

// The chained schedule of a ROW-PARTITIONED stream (every rank: a full 64-column block of 128 k <= 2^20 rows).  The all-reduce sits inside
// the R-factor chain, so only the factorisation can ride in the next call's Gram launch (gram_blk_chain_kernel, `direct`):
//     gram(0) reduce(0) allreduce(0) | [chol(0) + gram(1)]  reduce(1) allreduce(1)  apply(0) | [chol(1) + gram(2)]  reduce(2) allreduce(2)  apply(1) | ...
//     ... | chol(last) (a launch of its own)  apply(last)
// -- the Cholesky launch (17.5 us + its ramp) leaves the critical path of every call but the last.  The collectives are enqueued in a
// different order than by the plain stream (allreduce(i + 1) before apply(i)), so ALL ranks must take this schedule or none.  The vote costs
// no collective of its own: the first all-reduce of the stream -- the Gram tiles of call 0, the same launch in either schedule -- carries one
// more double, 1.0 from every rank that is eligible (a rank that is not sends zeros instead of a Gram pass: the result is then thrown away);
// a one-thread kernel behind it puts the count into the pinned words and the host decides when it sees it, with Gram pass and all-reduce of
// call 0 already done.  Returns NOT_MINE when any rank is not eligible (the stream is then idle and holds nothing of this call).
// A rejected verdict -- the same on every rank -- ends the schedule at that call (rejected_tail).
static int chained_dist(const CallEnv& env, const Mats& mt, int count, const Call& cl, int* done) {
	*done = 0;
	const size_t m = cl.m, n = cl.n, lda = cl.lda, ldq = cl.ldq, ldr = cl.ldr;
	const int engine = engine_of(cl.mode);
	if (engine < 0 || m == 0 || n == 0 || n > PW || env.nranks < 1 || env.nranks > 255) return NOT_MINE;   // (conditions every rank shares; the plain path reports errors)
	Ctx c;
	env_ctx(c, env, cl.wq, cl.wr, m, n, nullptr, cl.stream, /*keep_in_flight=*/false);
	bool mine = c.hsig.dev && !cl.reorth && n == PW && m % 128 == 0 && m <= ((size_t)1 << 20) && lda % 4 == 0 && lda <= ((size_t)1 << 24) && lda >= m && ldq >= m &&
	            ldr >= n && c.policy == 0 && c.gram_level == 2 && !g_set.debug && chain_order_safe(mt, count, cl);
	for (int i = 0; mine && i < (mt.same() ? 1 : count); i++) mine = (reinterpret_cast<uintptr_t>(mt.a(i)) & 15) == 0;
	c.fold_cor = (engine == 1);
	static DevOnce attr;
	if (attr.need(c.dev)) {
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_blk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GB_LDS_BYTES));
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_blk_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GB_LDS_BYTES));
		attr.done(c.dev);
	}
	const GramPlan g = gram_plan(m, PW);
	const int nchunks = (int)(m / 128), nparts = std::min(nchunks, g.nblocks), nelem = 10 * 256;
	double* part = reinterpret_cast<double*>(c.wr);
	volatile unsigned* words = reinterpret_cast<volatile unsigned*>(c.hsig.host);
	unsigned seq[2] = {0, 0};
	unsigned* announce = nullptr; unsigned announce_seq = 0;
	auto gram_args = [&](int i) {
		tsqrmi::GramArgs ga{};
		ga.a = mt.a(i); ga.lda = lda; ga.m = m; ga.n = (int)n; ga.nchunks = nchunks; ga.cpw = g.cpw; ga.nwaves = g.nwaves; ga.part = part;
		ga.announce = announce; ga.announce_seq = announce_seq; announce = nullptr;
		return ga;
	};
	auto chol_args = [&](int i) {
		tsqrmi::CholArgs ca{};
		ca.r = mt.r(i); ca.ldr = ldr; ca.z = c.wq + c.L.z;
		ca.status = c.status_dev(i & 1);
		ca.host_status = c.hsig.dev + 4 * (i & 1);
		ca.gsum = c.gsum();
		ca.rows_dev = c.gsum() + nelem;                  // the all-reduced row count (as chol_from_g)
		ca.rows = c.rows_global;
		ca.n = (int)n; ca.NT = 4; ca.level = 2; ca.scond_floor = g_set.bf16_scond_floor;
		return ca;
	};
	auto reduce_allreduce = [&](int extra) -> int {      // partials of the Gram pass just enqueued -> summed tiles + row count (+ `extra` doubles), over all ranks
		{
			ProfScope ps(KC_CHOL, c.st);
			hipLaunchKernelGGL(tsqrmi::gram_reduce1_kernel, dim3(nelem / 16), dim3(256), 0, c.st, c.gsum(), part, nparts, nelem, (double)m,
			                   nullptr, (size_t)0, nullptr, 0);
		}
		HIPCHK(hipGetLastError());
		if (extra) hipLaunchKernelGGL(tsqrmi::set_f64_kernel, dim3(1), dim3(1), 0, c.st, c.gsum() + nelem + 1, 1.0);
		ProfScope ps(KC_MISC, c.st);
		if (c.comm.allreduce_f64(c.gsum(), (size_t)nelem + 1 + extra, c.st)) { t_last_error = "all-reduce of the Gram tiles failed"; return -1; }
		return 0;
	};
	// ---- call 0 up to its all-reduce, with the vote in the payload ----
	int rc = 0;
	if (mine) {
		{
			ProfScope ps(KC_GRAM, c.st);
			hipLaunchKernelGGL(tsqrmi::gram_blk_kernel, dim3(nparts), dim3(256), tsqrmi::GB_LDS_BYTES, c.st, gram_args(0));
		}
		HIPCHK(hipGetLastError());
		rc = reduce_allreduce(1);
	} else {
		HIPCHK(hipMemsetAsync(c.gsum(), 0, sizeof(double) * ((size_t)nelem + 2), c.st));
		if (c.comm.allreduce_f64(c.gsum(), (size_t)nelem + 2, c.st)) { t_last_error = "all-reduce of the schedule flags failed"; rc = -1; }
	}
	if (rc) return rc;
	if (c.hsig.dev) {
		unsigned vs = ++g_seq;
		if ((vs & 0xffffffu) == 0) vs = ++g_seq;
		words[8] = 0;
		hipLaunchKernelGGL(tsqrmi::vote_out_kernel, dim3(1), dim3(1), 0, c.st, c.gsum() + nelem + 1, c.hsig.dev + 8, vs);
		HIPCHK(hipGetLastError());
		unsigned w = 0;
		for (;;) {
			bool seen = false;
			for (int k = 0; k < 20000 && !seen; k++) {
				w = words[8];
				seen = (w >> 8) == (vs & 0xffffffu);
				if (!seen) __builtin_ia32_pause();
			}
			if (seen) break;
			const hipError_t e = hipStreamQuery(c.st);
			if (e == hipSuccess) { w = words[8]; break; }
			if (e != hipErrorNotReady) HIPCHK(e);
		}
		if ((int)(w & 0xffu) != env.nranks || (w >> 8) != (vs & 0xffffffu)) { prof_collect(); return NOT_MINE; }
	} else {                                             // (no pinned words on this rank: it voted "no"; it still has to leave with an idle stream)
		HIPCHK(hipStreamSynchronize(c.st));
		return NOT_MINE;
	}
	auto step = [&](int i) -> int {
		if (i + 1 < count) {
			tsqrmi::ChainArgs ch{};
			ch.chol = chol_args(i); ch.direct = 1;
			{
				ProfScope ps(KC_GRAM, c.st);
				hipLaunchKernelGGL(tsqrmi::gram_blk_chain_kernel, dim3(1 + nparts), dim3(256), tsqrmi::GB_LDS_BYTES, c.st, gram_args(i + 1), ch);
			}
			HIPCHK(hipGetLastError());
			const int rc2 = reduce_allreduce(0);
			if (rc2) return rc2;
		} else {
			if (announce) { hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, announce, announce_seq); announce = nullptr; }
			ProfScope ps(KC_CHOL, c.st);
			hipLaunchKernelGGL(tsqrmi::chol16_kernel, dim3(1), dim3(1024), 0, c.st, chol_args(i));
			HIPCHK(hipGetLastError());
		}
		c.slot = i & 1;
		const int rc2 = apply_rinv(c, engine, mt.q(i), ldq, mt.a(i), lda, mt.r(i), ldr, m, n, /*z_ready=*/true, c.status_dev(i & 1));
		if (rc2) return rc2;
		unsigned sq = ++g_seq;
		if (sq == 0) sq = ++g_seq;
		seq[i & 1] = sq;
		words[4 * (i & 1) + 3] = 0;
		if (i + 1 < count) { announce = c.hsig.dev + 4 * (i & 1) + 3; announce_seq = sq; }   // raised by the chained launch of step i + 1
		else hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, c.hsig.dev + 4 * (i & 1) + 3, sq);
		HIPCHK(hipGetLastError());
		return 0;
	};
	rc = step(0);
	if (rc) return rc;
	for (int i = 0; i < count; i++) {
		if (i + 1 < count) { rc = step(i + 1); if (rc) return rc; }
		rc = wait_word(words + 4 * (i & 1) + 3, seq[i & 1], c.st);
		if (rc) return rc;
		if (words[4 * (i & 1)] != 0) return rejected_tail(env, mt, count, cl, i, words, c.st, done);
		mt.state(i, TSQR_MI_SUCCESS);
		*done = i + 1;
	}
	t_last_engine = 3;
	prof_collect();
	return TSQR_MI_SUCCESS;
}

// Two calls in flight: call i + 1 is submitted before call i is finished; the completion word of call i is raised by the first kernel of
// call i + 1 (stream order: it starts when call i has finished), only the last call carries a completion kernel of its own.  Stream order
// is the blocking order, so any operand overlap the blocking calls allow is fine here.  Row-partitioned calls: every rank takes the same
// verdicts (they come from the all-reduced Gram matrix), hence the same path through this loop and the same order of collectives.
static int two_in_flight(const CallEnv& env, const Mats& mt, int count, const Call& cl) {
	auto submit = [&](int i, tsqr_mi_ticket* t, tsqr_mi_ticket* announce, bool own_flag) {
		return submit_impl(env, cl.mode, cl.reorth, mt.q(i), cl.ldq, mt.r(i), cl.ldr, mt.a(i), cl.lda, cl.m, cl.n, cl.wq, cl.wr, cl.h_wl, cl.stream, t, announce, own_flag);
	};
	tsqr_mi_ticket tk[2];
	int first = 0;
	int st = submit(0, &tk[0], nullptr, /*own_flag=*/count == 1);
	if (st) { for (int k = 0; k < count; k++) mt.state(k, st); return st; }           // (invalid size / mode: the same for every call of the stream)
	for (int i = 0; i < count; i++) {
		tsqr_mi_ticket* cur = &tk[i & 1];
		tsqr_mi_ticket* nxt = (i + 1 < count) ? &tk[(i + 1) & 1] : nullptr;
		if (nxt) {
			st = submit(i + 1, nxt, cur, /*own_flag=*/i + 2 == count);
			if (st) { (void)finish_impl(env, cur); return st; }
		}
		st = finish_impl(env, cur);
		mt.state(i, st);
		if (st && !first) first = st;
		if (st < 0 || (st && mt.same())) { if (nxt) (void)finish_impl(env, nxt); return st; }   // (a loop over one triple stops at its first non-zero state)
	}
	return first;
}

// `count` calls as a stream.  Depth 3: the chained schedules as far as they go -- a rejected call ends one, the stream goes on behind it
// with a fresh one (after a second rejection: two in flight for the rest, every attempt of a chained schedule would be thrown away).
static int stream_of_calls(const CallEnv& env, const Mats& mt, int count, const Call& cl) {
	const int depth = g_set.loop_depth.load();
	int first = 0;
	if (count < 2 || depth < 2) {
		for (int i = 0; i < count; i++) {
			const int st = blocking_one(env, mt, i, cl);
			mt.state(i, st);
			if (st && !first) first = st;
			if (st < 0 || (st && mt.same())) return st;
		}
		return first;
	}
	int pos = 0, rejections = 0;
	while (depth >= 3 && count - pos >= 3 && rejections < 2) {
		const Mats sub = mt.from(pos);
		int done = 0, st;
		if (env.dist) st = chained_dist(env, sub, count - pos, cl, &done);
		else {
			st = chained64(sub, count - pos, cl, &done);
			if (st == NOT_MINE) st = chained128(sub, count - pos, cl, &done);
		}
		if (st == NOT_MINE) break;
		if (st < 0) return st;
		if (st && !first) first = st;
		if (pos + done < count) rejections++;
		pos += done;
	}
	if (pos < count) {
		const int st = two_in_flight(env, mt.from(pos), count - pos, cl);
		if (st < 0) return st;
		if (st && !first) first = st;
	}
	return first;
}

void tsqr_mi_set_loop_depth(int depth) { g_set.loop_depth = depth < 2 ? 1 : (depth == 2 ? 2 : 3); }

// `count` calls with the same arguments: the reference's speed protocol (src/test.cu:299-309 is such a C++ loop around its call).
// Returns the first non-zero state.
int tsqr_mi_qr_f32_loop(int count, int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                        size_t m, size_t n, void* wq_v, void* wr_v, float* reorth_w, unsigned* d_wl, unsigned* h_wl, void* stream) {
	(void)reorth_w; (void)d_wl;
	Mats mt;
	mt.q0 = q; mt.r0 = r; mt.a0 = a;
	return stream_of_calls(CallEnv{}, mt, count, Call{mode, reorth, ldq, ldr, lda, m, n, wq_v, wr_v, h_wl, stream});
}

// `count` DIFFERENT matrices of one shape (include/tsqr_mi.h): the caller of mtk::qr::qr with many matrices -- the stream of calls above
// over one (q, r, a) triple per call.
int tsqr_mi_qr_f32_batch(int count, int mode, int reorth, float* const* q, size_t ldq, float* const* r, size_t ldr, float* const* a, size_t lda,
                         size_t m, size_t n, void* wq_v, void* wr_v, float* reorth_w, unsigned* d_wl, unsigned* h_wl, void* stream, int* states) {
	(void)reorth_w; (void)d_wl;
	if (count < 0 || (count > 0 && (!q || !r || !a))) return TSQR_MI_ERROR_INVALID_SIZE;
	Mats mt;
	mt.qs = q; mt.rs = r; mt.as = a; mt.states = states;
	return stream_of_calls(CallEnv{}, mt, count, Call{mode, reorth, ldq, ldr, lda, m, n, wq_v, wr_v, h_wl, stream});
}

// ---- fp16 I/O modes: reference mtk::qr::qr<fp16_notc | fp16_tc_nocor, Reorthogonalize> (src/blockqr.cu:437-449; io and working
// types half, src/tsqr.hpp:27-39).  The boundary converts, the factorisation is the fp32 pipeline: A is widened into the tail of
// wq (leading dimension = m rounded up to 128: full blocks for the fast kernels), Q and R are formed in fp32 next to it and
// rounded to fp16 on the way out.  fp16_notc runs the error-corrected bf16x3 apply engine (fp32-accurate products: the pipeline of
// fp32_tc_cor), fp16_tc_nocor the single-fp16-product engine (exact for fp16 data in A; inverse(R) rounded to fp16, no correction --
// the mode's meaning in the reference, src/tcqr32x16.cu:617-667).
// A is never modified.  Costs two conversion passes on top of the fp32 call -- unless the native path inside tsqr_mi_qr_f16 takes
// the call (one panel, one sweep, aligned columns, accepted by the bf16-split level's verdict). ----
namespace {
size_t f16_ld(size_t m) { return (m + 127) & ~(size_t)127; }
size_t f16_tail_offset(size_t m, size_t n) { return (tsqr_mi_working_q_size(m, n) + 63) & ~(size_t)63; }
int f16_engine_mode(int mode) {
	if (mode == TSQR_MI_FP16_NOTC) return TSQR_MI_FP32_TC_COR;
	if (mode == TSQR_MI_FP16_TC_NOCOR) return TSQR_MI_FP32_TC_NOCOR;
	return -1;
}
}  // namespace
size_t tsqr_mi_working_q_size_f16(size_t m, size_t n) {
	if (m == 0 || n == 0) return 0;
	return f16_tail_offset(m, n) + 2 * f16_ld(m) * n + ((n * n + 63) & ~(size_t)63);
}
size_t tsqr_mi_working_r_size_f16(size_t m, size_t n) { return tsqr_mi_working_r_size(m, n); }

int tsqr_mi_qr_f16(int mode, int reorth, void* q, size_t ldq, void* r, size_t ldr, const void* a, size_t lda,
                   size_t m, size_t n, void* wq_v, void* wr_v, void* reorth_w, unsigned* d_wl, unsigned* h_wl, void* stream) {
	(void)reorth_w;
	if (n > m || m == 0 || n == 0) return TSQR_MI_ERROR_INVALID_SIZE;
	const int emode = f16_engine_mode(mode);
	if (emode < 0) { t_last_error = "tsqr_mi_qr_f16 takes fp16_notc or fp16_tc_nocor"; return TSQR_MI_ERROR_UNSUPPORTED; }
	if (ldq < m || lda < m || ldr < n) return TSQR_MI_ERROR_INVALID_SIZE;
	hipStream_t st = reinterpret_cast<hipStream_t>(stream);
	const size_t ld32 = f16_ld(m);
	float* a32 = reinterpret_cast<float*>(wq_v) + f16_tail_offset(m, n);
	float* q32 = a32 + ld32 * n;
	float* r32 = q32 + ld32 * n;
	auto aligned16 = [](const void* p, size_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 8 == 0; };
	const unsigned grid = (unsigned)std::min<size_t>(4096, cdiv(cdiv(m, 8) * n, 256));
	// Native path (one panel, one sweep, auto policy, 16-byte aligned columns): the Gram pass takes the halves of A as MFMA operands
	// (gram_h_kernel: exact products, half the bytes), the apply pass reads halves and writes halves -- no conversion passes.  The
	// verdict is the bf16-split level's; a rejected matrix (A untouched) goes through the conversion path below and its whole ladder.
	if (n <= PW && !reorth && g_set.policy.load() == 0 && g_set.gram_level.load() == 2 && aligned16(a, lda) && aligned16(q, ldq)) {
		Ctx c;
		init_ctx(c, wq_v, wr_v, m, n, stream);
		c.rows_global = (double)m;
		resolve_host_sig(c, h_wl, m);
		c.slot = 0; c.prev_slot = -1;
		int rc = gram_g(c, reinterpret_cast<const float*>(a), lda, m, n, /*bf16=*/true, /*io_half=*/true);
		if (!rc) rc = chol_from_g(c, r32, n, n, 2);
		// (R to fp16: workgroup 0 of the apply pass does it -- scattered 2-byte stores inside the Cholesky kernel cost that 5 us, a
		// launch of its own 4.8 us)
		if (!rc) rc = apply_rinv(c, engine_of(f16_engine_mode(mode)), reinterpret_cast<float*>(q), ldq, reinterpret_cast<const float*>(a), lda,
		                         r32, n, m, n, /*z_ready=*/true, c.status_dev(0), /*io_half=*/true, r, ldr);
		if (rc) return rc;
		unsigned status = 1u;
		float scond = 0.0f;
		rc = read_status(c, 0, &status, &scond);
		if (rc) return rc;
		if (status == 0 && !(n <= 16 && scond > 32.0f)) {    // (n <= 16 with S > 32: the fp32 path adds a second sweep, see qr_core)
			t_last_engine = 3;
			return TSQR_MI_SUCCESS;
		}
	}
	hipLaunchKernelGGL(tsqrmi::widen_f16_kernel, dim3(grid), dim3(256), 0, st, a32, ld32, reinterpret_cast<const _Float16*>(a), lda, m, (int)n,
	                   aligned16(a, lda) ? 1 : 0);
	HIPCHK(hipGetLastError());
	const int rc = tsqr_mi_qr_f32(emode, reorth, q32, ld32, r32, n, a32, ld32, m, n, wq_v, wr_v, nullptr, d_wl, h_wl, stream);   // (blocking)
	if (rc) return rc;
	hipLaunchKernelGGL(tsqrmi::narrow_f16_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<_Float16*>(q), ldq, q32, ld32, m, (int)n,
	                   aligned16(q, ldq) ? 1 : 0);
	hipLaunchKernelGGL(tsqrmi::narrow_f16_kernel, dim3((unsigned)cdiv(cdiv(n, 8) * n, 256)), dim3(256), 0, st, reinterpret_cast<_Float16*>(r), ldr, r32, n,
	                   n, (int)n, aligned16(r, ldr) ? 1 : 0);
	HIPCHK(hipGetLastError());
	return tsqr_mi_stream_wait(stream);
}

// The fp16 calls as a stream, two in flight (the native path only: one panel, no reorth, aligned halves): Gram pass on the halves, reduction,
// Cholesky + verdict, apply pass (which also rounds R), the completion word of call i raised by the Gram kernel of call i + 1.  The
// verdict words alternate between the two halves of the pinned words.  Returns -2 when the call is not one for the native path
// (nothing enqueued); a rejected verdict drains the stream and finishes the count with blocking calls (conversion path, whole ladder).
// (mt: the half-typed operands of the calls, carried as float* -- one triple for a loop, one per call for a batch)
static int stream_of_calls_f16(const Mats& mt, int count, int mode, size_t ldq, size_t ldr, size_t lda,
                               size_t m, size_t n, void* wq_v, void* wr_v, unsigned* h_wl, void* stream) {
	auto aligned16 = [](const void* p, size_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 8 == 0; };
	const int emode = f16_engine_mode(mode);
	if (emode < 0 || n > m || m == 0 || n == 0 || n > PW || n <= 16 || ldq < m || lda < m || ldr < n) return NOT_MINE;
	for (int i = 0; i < (mt.same() ? 1 : count); i++)
		if (!aligned16(mt.a(i), lda) || !aligned16(mt.q(i), ldq)) return NOT_MINE;
	Ctx c;
	init_ctx(c, wq_v, wr_v, m, n, stream);
	c.rows_global = (double)m;
	resolve_host_sig(c, h_wl, m);
	if (!(c.policy == 0 && c.gram_level == 2 && c.hsig.dev && !t_prof.on && !g_set.debug)) return NOT_MINE;
	float* r32 = reinterpret_cast<float*>(wq_v) + f16_tail_offset(m, n) + 2 * f16_ld(m) * n;
	const int engine = engine_of(emode);
	volatile unsigned* words = reinterpret_cast<volatile unsigned*>(c.hsig.host);
	unsigned seq[2] = {0, 0};
	// n = 64, three calls or more: the chained schedule (tsqr_mi_qr_f32_loop's, with gram_h_chain_kernel) -- the R-factor chain of call i
	// inside the Gram launch of call i + 1, two sets of partials
	// (the chained launch order needs Q and R of call i clear of A of call i + 1, and two different matrices must share the Infinity Cache)
	const Call cl{mode, 0, ldq, ldr, lda, m, n, wq_v, wr_v, h_wl, stream};
	const bool chained = (n == PW && count >= 3 && g_set.loop_depth.load() >= 3 && chain_order_safe(mt, count, cl, sizeof(_Float16)) &&
	                      (mt.same() || (double)lda * (double)n * sizeof(_Float16) <= (double)g_set.chain_max_mib * 1048576.0));
	const GramPlan g = gram_plan(m, n);
	const int nelem = 10 * 256, nred = nelem / 16;
	double* part[2] = {reinterpret_cast<double*>(c.wr), reinterpret_cast<double*>(c.wr) + (size_t)g.nblocks * nelem};
	unsigned* ticket = c.status_dev(0) + 8;
	if (chained) HIPCHK(hipMemsetAsync(ticket, 0, sizeof(unsigned), c.st));
	auto gram_h_args = [&](int i) {
		tsqrmi::GramArgs ga{};
		ga.a = mt.a(i); ga.lda = lda; ga.m = m; ga.n = (int)n; ga.nchunks = g.nch; ga.cpw = g.cpw; ga.nwaves = g.nwaves;
		ga.part = part[i & 1];
		ga.announce = c.announce_word; ga.announce_seq = c.announce_seq; c.announce_word = nullptr;
		return ga;
	};
	auto chol_h_args = [&](int i) {
		tsqrmi::CholArgs ca{};
		ca.r = r32; ca.ldr = n; ca.z = c.wq + c.L.z;
		ca.status = c.status_dev(i & 1);
		ca.host_status = c.hsig.dev + 4 * (i & 1);
		ca.gsum = c.gsum();
		ca.rows = c.rows_global;
		ca.n = (int)n; ca.NT = 4; ca.level = 2; ca.scond_floor = g_set.bf16_scond_floor;
		return ca;
	};
	if (chained) {
		hipLaunchKernelGGL(tsqrmi::gram_h_kernel<4>, dim3(g.nblocks), dim3(256), 0, c.st, gram_h_args(0));
		HIPCHK(hipGetLastError());
	}
	auto step = [&](int i) -> int {                      // every launch of call i; c.announce_word (call i - 1's completion word) rides in its Gram kernel
		c.slot = i & 1; c.prev_slot = -1;
		int rc = 0;
		if (chained) {
			if (i + 1 < count) {
				tsqrmi::ChainArgs ch{};
				ch.chol = chol_h_args(i);
				ch.part = part[i & 1]; ch.nparts = g.nblocks; ch.ticket = ticket; ch.nred = nred;
				hipLaunchKernelGGL(tsqrmi::gram_h_chain_kernel, dim3(nred + g.nblocks), dim3(256), 0, c.st, gram_h_args(i + 1), ch);
			} else {
				if (c.announce_word) { hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, c.announce_word, c.announce_seq); c.announce_word = nullptr; }
				hipLaunchKernelGGL(tsqrmi::gram_reduce1_kernel, dim3(nred), dim3(256), 0, c.st, c.gsum(), part[i & 1], g.nblocks, nelem, (double)m,
				                   nullptr, (size_t)0, nullptr, 0);
				hipLaunchKernelGGL(tsqrmi::chol16_kernel, dim3(1), dim3(1024), 0, c.st, chol_h_args(i));
			}
			HIPCHK(hipGetLastError());
		} else {
			rc = gram_g(c, mt.a(i), lda, m, n, /*bf16=*/true, /*io_half=*/true);
			if (!rc) rc = chol_from_g(c, r32, n, n, 2);
		}
		if (!rc) rc = apply_rinv(c, engine, mt.q(i), ldq, mt.a(i), lda, r32, n, m, n, /*z_ready=*/true,
		                         c.status_dev(c.slot), /*io_half=*/true, mt.r(i), ldr);
		if (rc) return rc;
		unsigned sq = ++g_seq;
		if (sq == 0) sq = ++g_seq;
		seq[i & 1] = sq;
		words[4 * (i & 1) + 3] = 0;
		if (i + 1 < count) { c.announce_word = c.hsig.dev + 4 * (i & 1) + 3; c.announce_seq = sq; }
		else hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, c.st, c.hsig.dev + 4 * (i & 1) + 3, sq);
		HIPCHK(hipGetLastError());
		return 0;
	};
	int rc = step(0);
	if (rc) return rc;
	for (int i = 0; i < count; i++) {
		if (i + 1 < count) { rc = step(i + 1); if (rc) return rc; }
		for (bool done = false; !done;) {
			for (int k = 0; k < 20000 && !done; k++) {
				done = (words[4 * (i & 1) + 3] == seq[i & 1]);
				if (!done) __builtin_ia32_pause();
			}
			if (!done) {
				const hipError_t e = hipStreamQuery(c.st);
				if (e == hipSuccess) done = true;
				else if (e != hipErrorNotReady) HIPCHK(e);
			}
		}
		if (words[4 * (i & 1)] != 0) {
			// rejected (its apply pass skipped itself: A and Q untouched): drain; this call as a blocking call (conversion path, whole
			// ladder); the call behind it, enqueued in full, stands if it was accepted; the rest of the count as blocking calls
			HIPCHK(hipStreamSynchronize(c.st));
			const bool next_ok = i + 1 < count && words[4 * ((i + 1) & 1)] == 0;
			int first = 0;
			for (int k = i; k < count; k++) {
				int st = 0;
				if (!(k == i + 1 && next_ok))
					st = tsqr_mi_qr_f16(mode, 0, mt.q(k), ldq, mt.r(k), ldr, mt.a(k), lda, m, n, wq_v, wr_v, nullptr, nullptr, h_wl, stream);
				if (st < 0) return st;
				mt.state(k, st);
				if (st && !first) first = st;
				if (st && mt.same()) return st;
			}
			return first;
		}
		mt.state(i, TSQR_MI_SUCCESS);
	}
	t_last_engine = 3;
	return TSQR_MI_SUCCESS;
}

static int calls_f16(const Mats& mt, int count, int mode, int reorth, size_t ldq, size_t ldr, size_t lda, size_t m, size_t n,
                     void* wq_v, void* wr_v, void* reorth_w, unsigned* d_wl, unsigned* h_wl, void* stream) {
	if (count >= 2 && g_set.loop_depth.load() >= 2 && !reorth) {
		const int st = stream_of_calls_f16(mt, count, mode, ldq, ldr, lda, m, n, wq_v, wr_v, h_wl, stream);
		if (st != NOT_MINE) return st;
	}
	int first = 0;
	for (int i = 0; i < count; i++) {
		const int st = tsqr_mi_qr_f16(mode, reorth, mt.q(i), ldq, mt.r(i), ldr, mt.a(i), lda, m, n, wq_v, wr_v, reorth_w, d_wl, h_wl, stream);
		mt.state(i, st);
		if (st && !first) first = st;
		if (st < 0 || (st && mt.same())) return st;
	}
	return first;
}
int tsqr_mi_qr_f16_loop(int count, int mode, int reorth, void* q, size_t ldq, void* r, size_t ldr, const void* a, size_t lda,
                        size_t m, size_t n, void* wq_v, void* wr_v, void* reorth_w, unsigned* d_wl, unsigned* h_wl, void* stream) {
	Mats mt;
	mt.q0 = reinterpret_cast<float*>(q); mt.r0 = reinterpret_cast<float*>(r); mt.a0 = const_cast<float*>(reinterpret_cast<const float*>(a));
	return calls_f16(mt, count, mode, reorth, ldq, ldr, lda, m, n, wq_v, wr_v, reorth_w, d_wl, h_wl, stream);
}
// `count` DIFFERENT half-typed matrices of one shape (tsqr_mi_qr_f32_batch's counterpart for the fp16 I/O modes): calls the native path
// takes are issued as a stream, everything else as blocking calls; the same halves either way
int tsqr_mi_qr_f16_batch(int count, int mode, int reorth, void* const* q, size_t ldq, void* const* r, size_t ldr, const void* const* a, size_t lda,
                         size_t m, size_t n, void* wq_v, void* wr_v, void* reorth_w, unsigned* d_wl, unsigned* h_wl, void* stream, int* states) {
	if (count < 0 || (count > 0 && (!q || !r || !a))) return TSQR_MI_ERROR_INVALID_SIZE;
	Mats mt;                                             // (pointer arrays of the same layout: void* / float*, const or not)
	mt.qs = reinterpret_cast<float* const*>(q); mt.rs = reinterpret_cast<float* const*>(r); mt.as = reinterpret_cast<float* const*>(const_cast<void* const*>(a));
	mt.states = states;
	return calls_f16(mt, count, mode, reorth, ldq, ldr, lda, m, n, wq_v, wr_v, reorth_w, d_wl, h_wl, stream);
}

// ---- row-partitioned TSQR: one call per rank, the same ladder as tsqr_mi_qr_f32 with the exchange hooks switched on ----
int tsqr_mi_qr_f32_dist_fn(int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                           size_t m_local, size_t n, void* wq_v, void* wr_v, float* gather_buf,
                           void* nccl_comm, void* nccl_allreduce_fn, void* nccl_allgather_fn, int nranks, void* stream) {
	if (!nccl_comm || !nccl_allreduce_fn || !nccl_allgather_fn) {
		t_last_error = "row-partitioned call needs an ncclComm_t and the ncclAllReduce / ncclAllGather entry points of the library that created it";
		return TSQR_MI_ERROR_UNSUPPORTED;
	}
	Ctx c;
	c.comm.nccl = nccl_comm;
	c.comm.nccl_allreduce = reinterpret_cast<nccl_allreduce_t>(nccl_allreduce_fn);
	c.comm.nccl_allgather = reinterpret_cast<nccl_allgather_t>(nccl_allgather_fn);
	c.comm.gather_buf = gather_buf;
	return qr_dist_common(c, mode, reorth, q, ldq, r, ldr, a, lda, m_local, n, wq_v, wr_v, nranks, stream);
}
int tsqr_mi_qr_f32_dist(int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                        size_t m_local, size_t n, void* wq_v, void* wr_v, float* gather_buf,
                        void* nccl_comm, int nranks, void* stream) {
	void* ar = rccl_symbol("ncclAllReduce");
	void* ag = rccl_symbol("ncclAllGather");
	if (!ar || !ag) {
		t_last_error = "ncclAllReduce / ncclAllGather are not in the global symbol scope (the caller does not link RCCL): "
		               "pass the entry points of the library that created the communicator to tsqr_mi_qr_f32_dist_fn";
		return TSQR_MI_ERROR_UNSUPPORTED;
	}
	return tsqr_mi_qr_f32_dist_fn(mode, reorth, q, ldq, r, ldr, a, lda, m_local, n, wq_v, wr_v, gather_buf, nccl_comm, ar, ag, nranks, stream);
}
int tsqr_mi_qr_f32_dist_cb(int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                           size_t m_local, size_t n, void* wq_v, void* wr_v, float* gather_buf,
                           tsqr_mi_allreduce_f64_cb allreduce, tsqr_mi_allgather_f32_cb allgather, void* user, int nranks, void* stream) {
	if (!allreduce || !allgather) return TSQR_MI_ERROR_UNSUPPORTED;
	Ctx c;
	c.comm.cb_allreduce = allreduce; c.comm.cb_allgather = allgather; c.comm.cb_user = user;
	c.comm.gather_buf = gather_buf;
	return qr_dist_common(c, mode, reorth, q, ldq, r, ldr, a, lda, m_local, n, wq_v, wr_v, nranks, stream);
}
int tsqr_mi_qr_f32_dist_fn_loop(int count, int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                                size_t m_local, size_t n, void* wq_v, void* wr_v, float* gather_buf,
                                void* nccl_comm, void* nccl_allreduce_fn, void* nccl_allgather_fn, int nranks, void* stream) {
	if (!nccl_comm || !nccl_allreduce_fn || !nccl_allgather_fn) {
		t_last_error = "row-partitioned call needs an ncclComm_t and the ncclAllReduce / ncclAllGather entry points of the library that created it";
		return TSQR_MI_ERROR_UNSUPPORTED;
	}
	CallEnv env;
	env.dist = true; env.nranks = nranks;
	env.comm.nccl = nccl_comm;
	env.comm.nccl_allreduce = reinterpret_cast<nccl_allreduce_t>(nccl_allreduce_fn);
	env.comm.nccl_allgather = reinterpret_cast<nccl_allgather_t>(nccl_allgather_fn);
	env.comm.gather_buf = gather_buf;
	Mats mt;
	mt.q0 = q; mt.r0 = r; mt.a0 = a;
	return stream_of_calls(env, mt, count, Call{mode, reorth, ldq, ldr, lda, m_local, n, wq_v, wr_v, nullptr, stream});
}
int tsqr_mi_qr_f32_dist_cb_loop(int count, int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                                size_t m_local, size_t n, void* wq_v, void* wr_v, float* gather_buf,
                                tsqr_mi_allreduce_f64_cb allreduce, tsqr_mi_allgather_f32_cb allgather, void* user, int nranks, void* stream) {
	if (!allreduce || !allgather) return TSQR_MI_ERROR_UNSUPPORTED;
	CallEnv env;
	env.dist = true; env.nranks = nranks;
	env.comm.cb_allreduce = allreduce; env.comm.cb_allgather = allgather; env.comm.cb_user = user;
	env.comm.gather_buf = gather_buf;
	Mats mt;
	mt.q0 = q; mt.r0 = r; mt.a0 = a;
	return stream_of_calls(env, mt, count, Call{mode, reorth, ldq, ldr, lda, m_local, n, wq_v, wr_v, nullptr, stream});
}

// `count` DIFFERENT row-partitioned matrices (every rank: its row block of each; one block height for all): the stream of calls of the
// loop entries over one (q, r, a) triple per call.  Every rank passes the same count, the same loop depth and operands of the same
// eligibility (alignment, overlap) on the same calls' positions -- the verdicts, and so the path through the batch, are the same on
// all ranks by construction (they come from the all-reduced matrix).
int tsqr_mi_qr_f32_dist_fn_batch(int count, int mode, int reorth, float* const* q, size_t ldq, float* const* r, size_t ldr, float* const* a, size_t lda,
                                 size_t m_local, size_t n, void* wq_v, void* wr_v, float* gather_buf,
                                 void* nccl_comm, void* nccl_allreduce_fn, void* nccl_allgather_fn, int nranks, void* stream, int* states) {
	if (!nccl_comm || !nccl_allreduce_fn || !nccl_allgather_fn) {
		t_last_error = "row-partitioned call needs an ncclComm_t and the ncclAllReduce / ncclAllGather entry points of the library that created it";
		return TSQR_MI_ERROR_UNSUPPORTED;
	}
	if (count < 0 || (count > 0 && (!q || !r || !a))) return TSQR_MI_ERROR_INVALID_SIZE;
	CallEnv env;
	env.dist = true; env.nranks = nranks;
	env.comm.nccl = nccl_comm;
	env.comm.nccl_allreduce = reinterpret_cast<nccl_allreduce_t>(nccl_allreduce_fn);
	env.comm.nccl_allgather = reinterpret_cast<nccl_allgather_t>(nccl_allgather_fn);
	env.comm.gather_buf = gather_buf;
	Mats mt;
	mt.qs = q; mt.rs = r; mt.as = a; mt.states = states;
	return stream_of_calls(env, mt, count, Call{mode, reorth, ldq, ldr, lda, m_local, n, wq_v, wr_v, nullptr, stream});
}
int tsqr_mi_qr_f32_dist_cb_batch(int count, int mode, int reorth, float* const* q, size_t ldq, float* const* r, size_t ldr, float* const* a, size_t lda,
                                 size_t m_local, size_t n, void* wq_v, void* wr_v, float* gather_buf,
                                 tsqr_mi_allreduce_f64_cb allreduce, tsqr_mi_allgather_f32_cb allgather, void* user, int nranks, void* stream, int* states) {
	if (!allreduce || !allgather) return TSQR_MI_ERROR_UNSUPPORTED;
	if (count < 0 || (count > 0 && (!q || !r || !a))) return TSQR_MI_ERROR_INVALID_SIZE;
	CallEnv env;
	env.dist = true; env.nranks = nranks;
	env.comm.cb_allreduce = allreduce; env.comm.cb_allgather = allgather; env.comm.cb_user = user;
	env.comm.gather_buf = gather_buf;
	Mats mt;
	mt.qs = q; mt.rs = r; mt.as = a; mt.states = states;
	return stream_of_calls(env, mt, count, Call{mode, reorth, ldq, ldr, lda, m_local, n, wq_v, wr_v, nullptr, stream});
}

// ---- staged entry points (building blocks; every call builds its own context) ----
int tsqr_mi_local_r_f32(float* r, size_t ldr, const float* a, size_t lda, size_t m, size_t n,
                        void* wq, void* wr, void* stream) {
	if (m == 0 || n == 0 || n > PW) return TSQR_MI_ERROR_INVALID_SIZE;
	Ctx c;
	init_ctx(c, wq, wr, m, n, stream);
	return fold_r(c, r, ldr, a, lda, m, n, c.wr, c.wq);
}

int tsqr_mi_apply_rinv_f32(int mode, float* q, size_t ldq, const float* a, size_t lda, const float* r, size_t ldr,
                           size_t m, size_t n, void* wq, void* stream) {
	if (m == 0 || n == 0 || n > PW) return TSQR_MI_ERROR_INVALID_SIZE;
	const int engine = engine_of(mode);
	if (engine < 0) return TSQR_MI_ERROR_UNSUPPORTED;
	Ctx c;
	init_ctx(c, wq, nullptr, m, n, stream);
	return apply_rinv(c, engine, q, ldq, a, lda, r, ldr, m, n);
}

size_t tsqr_mi_gram_elems(size_t n) { const size_t NT = np_of(n) / 16; return NT * (NT + 1) / 2 * 256; }

int tsqr_mi_gram_f32(int level, double* gsum, const float* a, size_t lda, size_t m, size_t n, void* wq, void* wr, void* stream) {
	if (m == 0 || n == 0 || n > PW || (level != 1 && level != 2)) return TSQR_MI_ERROR_INVALID_SIZE;
	Ctx c;
	init_ctx(c, wq, wr, m, n, stream);
	const int rc = gram_g(c, a, lda, m, n, level == 2);
	if (rc) return rc;
	if (gsum && gsum != c.gsum())
		HIPCHK(hipMemcpyAsync(gsum, c.gsum(), sizeof(double) * tsqr_mi_gram_elems(n), hipMemcpyDeviceToDevice, c.st));
	return 0;
}

int tsqr_mi_chol_f32(int level, float* r, size_t ldr, const double* gsum, size_t m, size_t n, void* wq_v, unsigned* status_out, void* stream) {
	// level 3: shifted Cholesky of an fp64 (level-1) Gram matrix, G + s I with s from the row count m
	if (m == 0 || n == 0 || n > PW || (level != 1 && level != 2 && level != 3)) return TSQR_MI_ERROR_INVALID_SIZE;
	Ctx c;
	init_ctx(c, wq_v, nullptr, m, n, stream);
	c.rows_global = (double)m;
	if (gsum && gsum != c.gsum())
		HIPCHK(hipMemcpyAsync(c.gsum(), gsum, sizeof(double) * tsqr_mi_gram_elems(n), hipMemcpyDeviceToDevice, c.st));
	int rc = chol_from_g(c, r, ldr, n, level);
	if (rc) return rc;
	if (!status_out) return 0;                           // asynchronous: read the verdict later with tsqr_mi_chol_status
	unsigned status = 0;
	rc = read_status(c, 0, &status);                     // (no pinned words in this context: stream sync + copy)
	if (rc) return rc;
	*status_out = status;
	return 0;
}

// Blocks until everything enqueued on `stream` so far has completed: a one-thread kernel raises a word in the thread's pinned
// memory and the host spins on it (hipStreamSynchronize as the fallback).  For staged callers that end with an asynchronous call.
int tsqr_mi_stream_wait(void* stream) {
	Ctx c;
	c.st = reinterpret_cast<hipStream_t>(stream);
	if (t_own.get()) { c.hsig.host = t_own.host; c.hsig.dev = t_own.dev; }
	return wait_done(c);
}

int tsqr_mi_chol_status(const void* wq_v, size_t m, size_t n, unsigned* status_out, void* stream) {
	if (m == 0 || n == 0 || n > PW || !status_out) return TSQR_MI_ERROR_INVALID_SIZE;
	Ctx c;
	init_ctx(c, const_cast<void*>(wq_v), nullptr, m, n, stream);
	return read_status(c, 0, status_out);
}

int tsqr_mi_apply_z_f32(int mode, float* q, size_t ldq, const float* a, size_t lda, size_t m, size_t n, void* wq_v, void* stream) {
	if (m == 0 || n == 0 || n > PW) return TSQR_MI_ERROR_INVALID_SIZE;
	const int engine = engine_of(mode);
	if (engine < 0) return TSQR_MI_ERROR_UNSUPPORTED;
	Ctx c;
	init_ctx(c, wq_v, nullptr, m, n, stream);
	// under the auto policy a rejected Cholesky (status word != 0) turns a speculatively enqueued apply into a no-op
	const unsigned* skip = (c.policy == 0) ? c.status_dev(0) : nullptr;
	return apply_rinv(c, engine, q, ldq, a, lda, nullptr, 0, m, n, /*z_ready=*/true, skip);
}

// ---- harness support: the reference's accuracy metrics evaluated on the device in fp64 (src/validation.cu, src/test.cu:147-165) ----
// scratch: (n*n + 8) doubles of device memory; out_host[0..4] = ||Q^T Q - I||_F^2, its diagonal part, its off-diagonal part,
// ||Q R - A||_F^2, ||A||_F^2.  r / a may be null: then only the orthogonality sums are computed.  gram_out_host (n*n doubles) optional.
int tsqr_mi_validate_f32(const float* q, size_t ldq, const float* r, size_t ldr, const float* a, size_t lda,
                         size_t m, size_t n, double* scratch, double* out_host, double* gram_out_host, void* stream) {
	if (m == 0 || n == 0 || n > m) return TSQR_MI_ERROR_INVALID_SIZE;
	hipStream_t st = reinterpret_cast<hipStream_t>(stream);
	double* g = scratch;
	double* sums = scratch + n * n;
	HIPCHK(hipMemsetAsync(scratch, 0, sizeof(double) * (n * n + 8), st));
	const size_t rows_per_block = 32 * std::max<size_t>(1, cdiv(cdiv(m, 32), 2048));
	hipLaunchKernelGGL(tsqrmi::gramd_kernel, dim3((unsigned)cdiv(m, rows_per_block)), dim3(256), 0, st, g, q, ldq, m, (int)n, rows_per_block);
	hipLaunchKernelGGL(tsqrmi::orth_sums_kernel, dim3(1), dim3(256), 0, st, sums, g, (int)n);
	if (r && a) hipLaunchKernelGGL(tsqrmi::resid_kernel, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, st, sums, q, ldq, r, ldr, a, lda, m, (int)n);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(out_host, sums, sizeof(double) * 5, hipMemcpyDeviceToHost, st));
	if (gram_out_host) HIPCHK(hipMemcpyAsync(gram_out_host, g, sizeof(double) * n * n, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	return 0;
}

int tsqr_mi_rmul_f32(float* r, size_t ldr, const float* r2, size_t ldr2, size_t n, void* wq, void* stream) {
	if (n == 0) return TSQR_MI_ERROR_INVALID_SIZE;
	hipStream_t st = reinterpret_cast<hipStream_t>(stream);
	float* r1 = reinterpret_cast<float*>(wq);            // n*n floats at the start of wq
	const unsigned gb = (unsigned)std::min<size_t>(1024, cdiv(n * n, 256));
	hipLaunchKernelGGL(tsqrmi::copy2d_kernel, dim3(gb), dim3(256), 0, st, r1, n, r, ldr, (int)n, (int)n);
	launch_rmul(r, ldr, r2, ldr2, r1, n, n, st);
	HIPCHK(hipGetLastError());
	return 0;
}

}  // extern "C"
