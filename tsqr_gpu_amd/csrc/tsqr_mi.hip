// tsqr_mi.hip -- host orchestration and the extern "C" ABI of libtsqr_mi.so (declared in include/tsqr_mi.h).
//
// Host-side counterpart of the reference's block_qr_core / block_qr_reorthogonalization_core
// (reference src/blockqr.cu:45-178, 180-390) and tsqr16_geq32 (reference src/tsqr.cu:1064-1279),
// re-designed for MI355X:
//   * panel width 64 instead of 16: for n <= 64 there is no inter-panel coupling at all;
//   * R by a streaming Householder TSQR (fold_kernel) followed by a short fold tree over the per-wave
//     R factors -- the R-stack reduction of the reference, with fan-in 4 instead of 2;
//   * Q = A * inverse(R) on the MFMA units (apply_wg_kernel): "indirect TSQR".  Its loss of
//     orthogonality grows like cond(A)*eps, slower than the reference's 16-wide block Gram-Schmidt
//     without reorthogonalisation; Reorthogonalize=true runs a second sweep on Q (R <- R2*R), which
//     restores ||Q^T Q - I|| to O(eps) as the reference's BCGS2 does.
//   * no host synchronisation inside; one hipStreamSynchronize at the end (the reference call is blocking).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <string>

#include "../../include/tsqr_mi.h"
#include "tsqr_kernels.hip"
#include "validate.hip"

namespace {

thread_local std::string g_last_error;
int g_level0_waves = 2048;
int g_tree_cpw = 4;
int g_policy = 0;        // 0 auto (fp32_tc_cor: Gram engine with Householder fallback; fp32_notc: Householder), 1 Householder, 2 Gram
int g_last_engine = 0;   // 0 Householder TSQR, 1 fp64 Gram/Cholesky, 2 Gram broke down -> Householder fallback, 3 bf16-split Gram
int g_min_level = 2;     // lowest R-factor engine level the last call ended up using (2 bf16 Gram, 1 fp64 Gram, 0 Householder)
int g_gram_level = 2;    // first Gram level tried: 2 bf16-split (then fp64), 1 fp64 only
constexpr int GRAM_NSPLIT = 16;
int g_gram_waves = 2048;
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
int g_apply_rows = env_int("TSQR_MI_APPLY_ROWS", 128);   // rows per workgroup block of apply_wg_kernel (128 or 256)
int g_apply_wgs = env_int("TSQR_MI_APPLY_WGS", 0);     // 0: as many workgroups as are resident at once (256 CUs x LDS-limited 2 or 3)
float g_bf16_max_scond = (float)env_int("TSQR_MI_BF16_MAX_SCOND", 4);   // floor of the acceptance bound of the bf16 Gram level on S (chol_kernel)
// Acceptance bound on the scaled conditioning S for the bf16-split Gram level.  Measured (tools/policy_accuracy.py): the level's
// own contribution to ||Q^T Q - I||_F is about 8e-6 * S / sqrt(rows) (the fp32 roundings inside the per-K-step MFMA chains average
// out over the K-steps), so S <= 0.12 * sqrt(rows) keeps it near 1e-6; never below the floor (short matrices), never above 128.
float bf16_scond_limit(size_t rows) {
	return std::min(128.0f, std::max(g_bf16_max_scond, 0.12f * sqrtf((float)rows)));
}
int g_debug = env_int("TSQR_MI_DEBUG", 0);
unsigned g_seq = 0;                                   // sequence number of the completion flags
int g_host_status = env_int("TSQR_MI_HOST_STATUS", 1);   // Cholesky status words written straight into the pinned h_wl
int g_host_flag = env_int("TSQR_MI_HOST_FLAG", 1);       // end of call: spin on a pinned flag word instead of hipStreamSynchronize
// The Cholesky kernel also writes its status words (status, min pivot ratio, scaled cond) straight into the caller's pinned
// h_wl (mtk::qr::buffer::hl) so that the host needs no copy operation to read them after the stream sync.
// dev = device-visible alias of h_wl (null when it is not pinned host memory: then a 4-byte copy is enqueued as before).
struct HostSig { unsigned* host = nullptr; unsigned* dev = nullptr; };
HostSig g_hsig;
int g_shifted = env_int("TSQR_MI_SHIFTED", 1);         // shifted Cholesky QR (two-step) before the Householder fallback
bool g_used_shift = false, g_used_householder = false;
int g_fuse_gramq = env_int("TSQR_MI_FUSE_GRAMQ", 1);   // reorthogonalisation, n <= 64: the first sweep's apply kernel also accumulates Q^T Q
double* g_gramq_part = nullptr;                         // non-null: apply launches write per-workgroup Gram partials of their output there
int g_gramq_cap = 0, g_gramq_nparts = 0;               // capacity of that buffer (workgroups), workgroups of the last fused launch
bool g_gramq_ready = false;                             // the next bf16-level Gram request can skip its pass (partials are in place)
int g_spec_reorth = env_int("TSQR_MI_SPEC_REORTH", 1);   // reorthogonalisation, n <= 64: both sweeps enqueued speculatively (status slots, device-side skips)
int g_slot = 0, g_prev_slot = -1;                      // status slot of the sweep being enqueued / of the sweep it depends on (-1: none)
int g_reduce1 = env_int("TSQR_MI_REDUCE1", 1);         // partials -> G in one launch (gram_reduce1_kernel) instead of two

// ---- optional per-kernel-class timing with HIP events on the caller's stream (bench.py's roofline leg) ----
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
} g_prof;
struct ProfScope {                 // brackets one kernel launch (or a short launch group) with two events
	int idx = -1; hipStream_t st;
	ProfScope(int kc, hipStream_t s) : st(s) {
		if (g_prof.on && g_prof.n < Prof::MAXEV) {
			idx = g_prof.n++;
			g_prof.cls[idx] = kc;
			(void)hipEventRecord(g_prof.ev[2 * idx], st);
		}
	}
	~ProfScope() { if (idx >= 0) (void)hipEventRecord(g_prof.ev[2 * idx + 1], st); }
};
void prof_collect() {              // after the stream is idle
	for (int i = 0; i < g_prof.n; i++) {
		float t = 0.0f;
		if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) == hipSuccess) {
			g_prof.ms[g_prof.cls[i]] += t;
			g_prof.launches[g_prof.cls[i]] += 1;
		}
	}
	g_prof.n = 0;
}

constexpr size_t PW = 64;          // panel width

inline int fail(hipError_t e, const char* what) {
	g_last_error = std::string(what) + ": " + hipGetErrorString(e);
	return -(int)e;
}
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_, #expr); } while (0)

inline size_t cdiv(size_t a, size_t b) { return (a + b - 1) / b; }
inline size_t np_of(size_t n) { return 16 * cdiv(std::min(n, PW), 16); }

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
	size_t rows = m;
	int lv = 0;
	// a wave turns cpw*64 source rows into NP rows: cpw*64 >= 2*NP keeps every level shrinking
	const size_t cpw_min = std::max<size_t>(1, cdiv(2 * p.NP, 64));
	for (;;) {
		const size_t nch = cdiv(rows, 64);
		// tree levels over 64-row triangular blocks are binary: the first block is copied into R (FoldArgs::tri_init), one fold per level
		size_t cpw = (lv == 0) ? std::max(cpw_min, cdiv(nch, (size_t)g_level0_waves))
		                       : (p.NP == 64 ? (size_t)2 : std::max(cpw_min, (size_t)g_tree_cpw));
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

// Gram engine geometry: waves / workgroups of gram_kernel and the size of its per-workgroup partials (in floats)
struct GramPlan { int nch, cpw, nwaves, nblocks, ntri; size_t part_floats; };
GramPlan gram_plan(size_t m, size_t n) {
	GramPlan g{};
	const size_t NT = np_of(n) / 16;
	g.nch = (int)cdiv(m, 64);
	g.cpw = (int)std::max<size_t>(1, cdiv((size_t)g.nch, (size_t)g_gram_waves));
	g.nwaves = (int)cdiv((size_t)g.nch, (size_t)g.cpw);
	g.nblocks = (g.nwaves + 3) / 4;
	g.ntri = (int)(NT * (NT + 1) / 2);
	g.part_floats = (size_t)(g.nblocks + 1) * g.ntri * 256 * 2;   // (+1: the ragged last rows of the LDS-DMA Gram pass go through a one-workgroup launch)
	return g;
}

// layout of wq (floats): [stack_b][Z: 4096][S: 4096][part: NSLAB*4096][R1 copy: n*n][R2: n*n][gram sub-sums][status]
struct WqLayout { size_t z, s, part, r1, r2, r3, r4, gsub, status, total; };
WqLayout wq_layout(size_t m, size_t n) {
	const Plan p = make_plan(m, n);
	WqLayout L{};
	size_t o = p.stack_b;
	o = (o + 63) & ~(size_t)63;
	L.z = o; o += 4096;
	L.s = o; o += 4096;
	L.part = o;                                          // (unused since the MFMA coupling kernels)
	L.r1 = o; o += n * n;
	L.r2 = o; o += n * n;
	L.r3 = o; o += 4096;                                 // panel-local R1, R2 of the shifted-Cholesky two-step (<= 64 x 64 each)
	L.r4 = o; o += 4096;
	o = (o + 63) & ~(size_t)63;
	L.gsub = o; o += (size_t)(GRAM_NSPLIT + 1) * 16 * 256 * 2;    // sub-sums + summed tiles (Gram: 10 tiles, coupling: 16)
	L.status = o; o += 64;
	L.total = o;
	return L;
}

template <int NT> int launch_fold(const tsqrmi::FoldArgs& a, hipStream_t st) {
	const int blocks = (a.nwaves + 3) / 4;
	if constexpr (NT == 4) {
		if (a.tri_init) { hipLaunchKernelGGL((tsqrmi::fold_kernel<4, true>), dim3(blocks), dim3(256), 0, st, a); return 0; }
	}
	hipLaunchKernelGGL((tsqrmi::fold_kernel<NT, false>), dim3(blocks), dim3(256), 0, st, a);
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

// R (n x n, ldr; full block written, zeros below the diagonal) of src (m x n), n <= 64
int fold_r(float* r, size_t ldr, const float* src, size_t ld, size_t m, size_t n,
           float* wq, float* wr, hipStream_t st) {
	const Plan p = make_plan(m, n);
	const int NT = (int)(p.NP / 16);
	const float* cur = src; size_t cur_ld = ld;
	for (int lv = 0; lv < p.nlevels; lv++) {
		tsqrmi::FoldArgs a{};
		a.src = cur; a.ld = cur_ld; a.m = p.rows[lv];
		a.n = (int)n;                                    // stacks are NP wide, only the first n columns carry data
		a.nchunks = p.nch[lv]; a.cpw = p.cpw[lv]; a.nwaves = p.nw[lv];
		a.tri_init = (lv > 0 && p.NP == 64) ? 1 : 0;
		if (p.nw[lv] == 1) {
			a.dst = r; a.dst_ld = ldr; a.rows_store = (int)n; a.cols_store = (int)n;
		} else {
			float* stack = (lv % 2 == 0) ? wr : wq;
			a.dst = stack; a.dst_ld = (size_t)p.nw[lv] * p.NP; a.rows_store = (int)p.NP; a.cols_store = (int)p.NP;
			cur = stack; cur_ld = a.dst_ld;
		}
		{
			ProfScope ps(lv == 0 ? KC_FOLD0 : KC_TREE, st);
			dispatch_fold(NT, a, st);
		}
		HIPCHK(hipGetLastError());
	}
	return 0;
}

int g_gram_dma = env_int("TSQR_MI_GRAM_DMA", 0);       // bf16-level Gram pass: 1 per-wave LDS-DMA bounce (default), 2 workgroup ring, 0 register loads
inline bool lda_ok(const float* p, size_t ld) { return (reinterpret_cast<uintptr_t>(p) % 16 == 0) && (ld % 4 == 0) && ld < ((size_t)1 << 28); }
template <int NT> void launch_gram_dma(const tsqrmi::GramArgs& a, int grid, hipStream_t st) {
	constexpr int lds = tsqrmi::GramDmaCfg<NT>::LDS_BYTES;
	static bool attr_done = false;
	if (!attr_done) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_dma_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		attr_done = true;
	}
	hipLaunchKernelGGL(tsqrmi::gram_dma_kernel<NT>, dim3(grid), dim3(256), lds, st, a);
}
template <int NT> void launch_gram_bounce(const tsqrmi::GramArgs& a, int nblocks, hipStream_t st) {
	constexpr int NTRI = NT * (NT + 1) / 2;
	constexpr int lds = (4 * NT * 4096 > 2 * NTRI * 256 * 8) ? 4 * NT * 4096 : 2 * NTRI * 256 * 8;
	static bool attr_done = false;
	if (!attr_done) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_bounce_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
		attr_done = true;
	}
	hipLaunchKernelGGL(tsqrmi::gram_bounce_kernel<NT>, dim3(nblocks), dim3(256), lds, st, a);
}
template <int NT> int launch_gram(const tsqrmi::GramArgs& a, int nblocks, bool bf16, hipStream_t st) {
	if (bf16) hipLaunchKernelGGL(tsqrmi::gram_bf16_kernel<NT>, dim3(nblocks), dim3(256), 0, st, a);
	else hipLaunchKernelGGL(tsqrmi::gram_kernel<NT>, dim3(nblocks), dim3(256), 0, st, a);
	return nblocks;
}

// the last m % 64 rows of an LDS-DMA Gram pass: one workgroup of gram_bf16_kernel writing partial number `slot`
int launch_gram_tail(const tsqrmi::GramArgs& a, const float* src, size_t m, int NT, int slot, int ntri, hipStream_t st) {
	tsqrmi::GramArgs t = a;
	t.a = src + (m / 64) * 64; t.m = m % 64; t.nchunks = 1; t.cpw = 1; t.nwaves = 1;
	t.part = a.part + (size_t)slot * ntri * 256;
	switch (NT) {
		case 1: launch_gram<1>(t, 1, true, st); break;
		case 2: launch_gram<2>(t, 1, true, st); break;
		case 3: launch_gram<3>(t, 1, true, st); break;
		default: launch_gram<4>(t, 1, true, st); break;
	}
	return 1;
}

// Gram engine: R (n x n, ldr) and Z = inverse(R) (NP x NP in z_buf) of src (m x n); status -> wq[L.status]
// Gram matrix of src (m x n) in MFMA-accumulator order -> gsum (ntri*256 doubles).  bf16 = true: bf16x3-split MFMA
// (memory-bound, f32 C/D layout), false: fp64 MFMA (f64 C/D layout).
int gram_g(double* gsum, const float* src, size_t ld, size_t m, size_t n, float* wq, float* wr, const WqLayout& L, bool bf16, hipStream_t st) {
	const GramPlan g = gram_plan(m, n);
	const int NT = (int)(np_of(n) / 16);
	tsqrmi::GramArgs a{};
	a.a = src; a.lda = ld; a.m = m; a.n = (int)n; a.nchunks = g.nch; a.cpw = g.cpw; a.nwaves = g.nwaves;
	a.part = reinterpret_cast<double*>(wr);
	a.skip_status = g_prev_slot >= 0 ? reinterpret_cast<const unsigned*>(wq + L.status) + 16 * g_prev_slot : nullptr;
	int nparts = g.nblocks;                              // workgroups that wrote a partial
	const bool dma_ok = bf16 && g_gram_dma && n % 16 == 0 && m >= 64 && lda_ok(src, ld);
	if (bf16 && g_gramq_ready) {                         // the previous sweep's apply kernel accumulated this very Gram matrix
		g_gramq_ready = false;
		nparts = g_gramq_nparts;
	} else if (dma_ok && g_gram_dma == 1) {
		// per-wave LDS-DMA bounce (gram_dma.hip, gram_bounce_kernel): same geometry and partials as gram_bf16_kernel
		ProfScope ps(KC_GRAM, st);
		tsqrmi::GramArgs d = a;
		d.nchunks = (int)(m / 64);
		switch (NT) {
			case 1: launch_gram_bounce<1>(d, g.nblocks, st); break;
			case 2: launch_gram_bounce<2>(d, g.nblocks, st); break;
			case 3: launch_gram_bounce<3>(d, g.nblocks, st); break;
			default: launch_gram_bounce<4>(d, g.nblocks, st); break;
		}
		nparts = g.nblocks;
		if (m % 64) nparts += launch_gram_tail(a, src, m, NT, g.nblocks, g.ntri, st);
	} else if (dma_ok) {
		// workgroup-cooperative LDS-DMA pass over the full 64-row blocks (gram_dma.hip), the last m % 64 rows through the per-wave kernel
		ProfScope ps(KC_GRAM, st);
		const size_t nblk = m / 64;
		const int grid = (int)std::min<size_t>(nblk, (size_t)g.nblocks);
		tsqrmi::GramArgs d = a;
		d.nchunks = (int)nblk;
		switch (NT) {
			case 1: launch_gram_dma<1>(d, grid, st); break;
			case 2: launch_gram_dma<2>(d, grid, st); break;
			case 3: launch_gram_dma<3>(d, grid, st); break;
			default: launch_gram_dma<4>(d, grid, st); break;
		}
		nparts = grid;
		if (m % 64) nparts += launch_gram_tail(a, src, m, NT, grid, g.ntri, st);
	} else {
		ProfScope ps(KC_GRAM, st);
		switch (NT) {
			case 1: nparts = launch_gram<1>(a, g.nblocks, bf16, st); break;
			case 2: nparts = launch_gram<2>(a, g.nblocks, bf16, st); break;
			case 3: nparts = launch_gram<3>(a, g.nblocks, bf16, st); break;
			default: nparts = launch_gram<4>(a, g.nblocks, bf16, st); break;
		}
	}
	HIPCHK(hipGetLastError());
	const int nelem = g.ntri * 256;
	const int nsplit = std::min(GRAM_NSPLIT, nparts);
	double* sub = reinterpret_cast<double*>(wq + L.gsub);
	{
		ProfScope ps(KC_CHOL, st);
		if (g_reduce1) {
			hipLaunchKernelGGL(tsqrmi::gram_reduce1_kernel, dim3((nelem + 15) / 16), dim3(256), 0, st, gsum, a.part, nparts, nelem);
		} else {
			hipLaunchKernelGGL(tsqrmi::gram_reduce_kernel, dim3((nelem + 255) / 256, nsplit), dim3(256), 0, st,
			                   sub, a.part, nparts, nelem, nsplit);
			hipLaunchKernelGGL(tsqrmi::gram_reduce2_kernel, dim3((nelem + 255) / 256), dim3(256), 0, st, gsum, sub, nelem, nsplit);
		}
	}
	HIPCHK(hipGetLastError());
	return 0;
}

// R = chol(G) (n x n, ldr), Z = inverse(R) (NP x NP in z_buf), status word -> wq[L.status]
int chol_from_g(float* r, size_t ldr, float* z_buf, const double* gsum, size_t rows, size_t n, float* wq, const WqLayout& L, bool bf16,
                hipStream_t st, unsigned* host_status = nullptr, double shift_coef = 0.0) {
	const int NT = (int)(np_of(n) / 16);
	{
		ProfScope ps(KC_CHOL, st);
		unsigned* sdev = reinterpret_cast<unsigned*>(wq + L.status) + 16 * g_slot;
		const unsigned* sprev = g_prev_slot >= 0 ? reinterpret_cast<const unsigned*>(wq + L.status) + 16 * g_prev_slot : nullptr;
		hipLaunchKernelGGL(tsqrmi::chol_kernel, dim3(1), dim3(256), 0, st, r, ldr, z_buf,
		                   sdev, gsum, (int)n, NT, bf16 ? 1 : 0,
		                   shift_coef > 0.0 ? 0.0f : (bf16 ? 0.03125f : 9.094947017729282e-13f),
		                   bf16 ? bf16_scond_limit(rows) : INFINITY, host_status ? host_status + 4 * g_slot : nullptr, shift_coef, sprev,
		                   bf16 ? (double)rows * 0x1p-90 : 0.0);   // bf16 level: mean squared entry of every column >= 2^-90
	}
	HIPCHK(hipGetLastError());
	return 0;
}

// Gram engine: R (n x n, ldr) and Z = inverse(R) (NP x NP in z_buf) of src (m x n); status -> wq[L.status]
int gram_r(float* r, size_t ldr, float* z_buf, const float* src, size_t ld, size_t m, size_t n,
           float* wq, float* wr, const WqLayout& L, bool bf16, hipStream_t st) {
	double* gsum = reinterpret_cast<double*>(wq + L.gsub) + (size_t)GRAM_NSPLIT * 16 * 256;
	const int rc = gram_g(gsum, src, ld, m, n, wq, wr, L, bf16, st);
	if (rc) return rc;
	return chol_from_g(r, ldr, z_buf, gsum, m, n, wq, L, bf16, st, g_hsig.dev);
}

// apply_wg_kernel launcher: args.nchunks = row blocks of ROWS, args.nwaves = workgroups (persistent grid)
template <int E, int NT, bool UPD, int ROWS, bool GRAMQ> constexpr auto apply_wg_entry() {
	if constexpr (GRAMQ) return &tsqrmi::apply_wg_gramq_kernel<E, NT, UPD, ROWS>;
	else return &tsqrmi::apply_wg_kernel<E, NT, UPD, ROWS>;
}
template <int E, int NT, bool UPD, int ROWS, bool GRAMQ = false> int launch_apply_wg(tsqrmi::ApplyArgs a, hipStream_t st) {
	constexpr auto kernel = apply_wg_entry<E, NT, UPD, ROWS, GRAMQ>();
	constexpr int NP = 16 * NT, KT = (NP + 31) / 32;
	constexpr int NB = (!UPD && NT == 4) ? 6 : KT * NT;  // operand blocks of Z kept in LDS (apply_wg_kernel: COMPACT)
	size_t lds = sizeof(float) * NP * (ROWS + 4) +
	             (E == 0 ? sizeof(float) * NP * (NP + 16) : (size_t)(E == 2 ? 1 : 3) * NB * 512 * 2);
	if (GRAMQ) lds = std::max(lds, sizeof(double) * 2 * (NT * (NT + 1) / 2) * 256);   // the final workgroup reduction aliases the block
	static bool attr_done = false;
	if (!attr_done) {
		HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
		                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		attr_done = true;
	}
	const size_t nblk = cdiv(a.m, (size_t)ROWS);
	a.nchunks = (int)nblk;
	// persistent grid: as many workgroups as are resident on the 256 CUs at once (LDS / register bound: 2 or 3 per CU), unless overridden
	static int per_cu = 0;
	if (per_cu == 0) {
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kernel),
		                                                 256, lds) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 2; }
		per_cu = std::min(nb, E == 0 ? 2 : 3);           // measured: the fp32-MFMA engine is slower with three per CU (132 vs 112 us)
	}
	const size_t want = g_apply_wgs > 0 ? (size_t)g_apply_wgs : (size_t)256 * per_cu;
	a.nwaves = (int)std::min<size_t>(nblk, want);
	a.cpw = 0;
	if constexpr (GRAMQ) {
		a.nwaves = std::min(a.nwaves, g_gramq_cap);      // one partial per workgroup: never more than the buffer holds
		a.gpart = g_gramq_part;
		g_gramq_nparts = a.nwaves;
	}
	hipLaunchKernelGGL(kernel, dim3(a.nwaves), dim3(256), lds, st, a);
	return 0;
}
template <int E, int NT, bool UPD> int launch_apply_any(const tsqrmi::ApplyArgs& a, hipStream_t st) {
	if constexpr (!UPD && E != 0) {                      // (the fp32-MFMA engine's fused variant spills and loses: 0.29 vs 0.22 ms per apply)
		if (g_gramq_part && g_gramq_cap > 0) return launch_apply_wg<E, NT, UPD, 128, true>(a, st);
	}
	if (g_apply_rows == 256) return launch_apply_wg<E, NT, UPD, 256>(a, st);
	return launch_apply_wg<E, NT, UPD, 128>(a, st);
}
template <int E> int dispatch_apply_nt(int NT, const tsqrmi::ApplyArgs& a, hipStream_t st) {
	switch (NT) {
		case 1: return launch_apply_any<E, 1, false>(a, st);
		case 2: return launch_apply_any<E, 2, false>(a, st);
		case 3: return launch_apply_any<E, 3, false>(a, st);
		default: return launch_apply_any<E, 4, false>(a, st);
	}
}

// q = a * inverse(r); n <= 64; z_buf: 4096 floats of scratch
int apply_rinv(int engine, float* q, size_t ldq, const float* a, size_t lda, const float* r, size_t ldr,
               size_t m, size_t n, float* z_buf, hipStream_t st, bool z_ready = false, const unsigned* skip_status = nullptr) {
	const size_t NP = np_of(n);
	const int NT = (int)(NP / 16);
	if (!z_ready) {
		ProfScope ps(KC_TRINV, st);
		hipLaunchKernelGGL(tsqrmi::trinv_kernel, dim3(1), dim3(256), 0, st, z_buf, r, ldr, (int)n, (int)NP);
	}
	HIPCHK(hipGetLastError());
	tsqrmi::ApplyArgs aa{};
	aa.a = a; aa.lda = lda; aa.q = q; aa.ldq = ldq; aa.m = m; aa.n = (int)n; aa.z = z_buf; aa.skip_status = skip_status;
	int rc;
	{
		ProfScope ps(KC_APPLY, st);
		rc = (engine == 0) ? dispatch_apply_nt<0>(NT, aa, st) : (engine == 1 ? dispatch_apply_nt<1>(NT, aa, st) : dispatch_apply_nt<2>(NT, aa, st));
	}
	if (rc) return rc;
	HIPCHK(hipGetLastError());
	return 0;
}

// Is h_wl pinned host memory the device can write?  (mtk::qr::buffer allocates it with hipHostMalloc; anything else
// falls back to copy + stream sync.)  The answer is cached per pointer.
void resolve_host_sig(unsigned* h_wl) {
	g_hsig.host = h_wl; g_hsig.dev = nullptr;            // queried on every call (sub-microsecond): the caller may have re-allocated
	if (!h_wl || !g_host_status) return;
	hipPointerAttribute_t at{};
	if (hipPointerGetAttributes(&at, h_wl) != hipSuccess) { (void)hipGetLastError(); return; }
	if (at.type == hipMemoryTypeHost && at.devicePointer) g_hsig.dev = reinterpret_cast<unsigned*>(at.devicePointer);
}
// End of a call on the fast path: a one-thread kernel behind the last kernel raises h_wl[3]; the host spins on it (about
// 5 us cheaper than hipStreamSynchronize, tools/launch_cost.py) and polls the stream now and then so that a failed launch
// cannot hang the caller.  Returns 1 when the flag path is not available (caller then synchronises the stream).
int signal_and_wait(hipStream_t st) {
	if (!g_host_flag || !g_hsig.dev || g_prof.on) return 1;
	unsigned seq = ++g_seq;
	if (seq == 0) seq = ++g_seq;
	volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(g_hsig.host) + 3;
	*flag = 0;
	hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, st, g_hsig.dev + 3, seq);
	if (hipGetLastError() != hipSuccess) return 1;
	for (;;) {
		for (int i = 0; i < 20000; i++) {
			if (*flag == seq) return 0;
			__builtin_ia32_pause();
		}
		const hipError_t e = hipStreamQuery(st);
		if (e == hipSuccess) return 0;
		if (e != hipErrorNotReady) HIPCHK(e);
	}
}

int engine_of(int mode) {
	if (mode == TSQR_MI_FP32_NOTC) return 0;
	if (mode == TSQR_MI_FP32_TC_COR) return 1;
	if (mode == TSQR_MI_FP32_TC_NOCOR) return 2;         // R factor as fp32_tc_cor, Q = A * inverse(R) with fp16 operands and no correction
	return -1;
}

int signal_and_wait(hipStream_t st);
// read the Gram engine's status word (0 ok / 1 breakdown) after draining the stream
int read_status(const float* wq, const WqLayout& L, unsigned* h_pinned, hipStream_t st, unsigned* out) {
	if (g_debug) {
		unsigned w3[3];
		HIPCHK(hipStreamSynchronize(st));
		HIPCHK(hipMemcpy(w3, wq + L.status, sizeof(w3), hipMemcpyDeviceToHost));
		float ratio, scond;
		memcpy(&ratio, &w3[1], 4); memcpy(&scond, &w3[2], 4);
		fprintf(stderr, "[tsqr_mi] chol status %u  min pivot ratio %.4g  scaled cond S %.4g\n", w3[0], ratio, scond);
	}
	if (h_pinned && h_pinned == g_hsig.host && g_hsig.dev) {      // the Cholesky kernel wrote the words to h_wl itself
		const int w = signal_and_wait(st);                   // spin on the completion flag (no OS wake-up in the path), else a stream sync
		if (w < 0) return w;
		if (w == 1) HIPCHK(hipStreamSynchronize(st));
		*out = reinterpret_cast<volatile unsigned*>(h_pinned)[0];
		return 0;
	}
	if (!h_pinned) {                                     // staged API: library-owned pinned words (a pageable 4-byte copy costs ~15 us)
		static unsigned* own = nullptr;
		static unsigned* own_dev = nullptr;
		if (!own) {
			if (hipHostMalloc(reinterpret_cast<void**>(&own), 64, hipHostMallocDefault) != hipSuccess) { own = nullptr; (void)hipGetLastError(); }
			else if (hipHostGetDevicePointer(reinterpret_cast<void**>(&own_dev), own, 0) != hipSuccess) { own_dev = nullptr; (void)hipGetLastError(); }
		}
		if (own && own_dev && g_host_flag && !g_prof.on) {
			// a one-thread kernel copies the status words and raises a flag; the host spins on it (no copy engine, no OS wake-up:
			// hipStreamSynchronize showed sporadic multi-millisecond stalls on the box, which a rank of a multi-GPU run cannot afford)
			unsigned seq = ++g_seq;
			if (seq == 0) seq = ++g_seq;
			volatile unsigned* flag = own + 3;
			*flag = 0;
			hipLaunchKernelGGL(tsqrmi::host_status_flag_kernel, dim3(1), dim3(1), 0, st, own_dev,
			                   reinterpret_cast<const unsigned*>(wq + L.status), seq);
			if (hipGetLastError() == hipSuccess) {
				for (;;) {
					bool seen = false;
					for (int i = 0; i < 20000 && !seen; i++) { seen = (*flag == seq); if (!seen) __builtin_ia32_pause(); }
					if (seen) break;
					const hipError_t e = hipStreamQuery(st);
					if (e == hipSuccess) break;
					if (e != hipErrorNotReady) HIPCHK(e);
				}
				*out = reinterpret_cast<volatile unsigned*>(own)[0];
				return 0;
			}
		}
		h_pinned = own;
	}
	if (h_pinned) {
		HIPCHK(hipMemcpyAsync(h_pinned, wq + L.status, sizeof(unsigned), hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		*out = h_pinned[0];
	} else {
		HIPCHK(hipStreamSynchronize(st));
		HIPCHK(hipMemcpy(out, wq + L.status, sizeof(unsigned), hipMemcpyDeviceToHost));
	}
	return 0;
}

// r <- r2 * r1 (upper triangular n x n, fp64 accumulation; r may not alias r1 / r2)
void launch_rmul(float* r, size_t ldr, const float* r2, size_t ldr2, const float* r1, size_t ldr1, size_t n, hipStream_t st) {
	if (n > 128) {
		const unsigned t = (unsigned)cdiv(n, 32);
		hipLaunchKernelGGL(tsqrmi::rmul_tiled_kernel, dim3(t, t), dim3(256), 0, st, r, ldr, r2, ldr2, r1, ldr1, (int)n);
	} else {
		const unsigned gb = (unsigned)std::min<size_t>(1024, cdiv(n * n, 256));
		hipLaunchKernelGGL(tsqrmi::rmul_kernel, dim3(gb), dim3(256), 0, st, r, ldr, r2, ldr2, r1, ldr1, (int)n);
	}
}
constexpr int R_SHIFT_DIRECT = 9;
// R factor (and Q) of one <= 64-column panel.  use_gram: Gram/Cholesky engine, otherwise the Householder TSQR engine.
// check_now: verify the Gram engine's status immediately (one stream sync) and fall back to Householder on breakdown.
int panel_qr(int engine, int r_engine, bool check_now, float* qp, size_t ldq, float* rpp, size_t ldr, const float* ap, size_t lda,
             size_t m, size_t c, float* wq, float* wr, const WqLayout& L, unsigned* h_pinned, hipStream_t st) {
	int rc;
	// R_SHIFT_DIRECT: the caller has just seen the fp64 Gram level reject this very panel (speculative single-panel mode); its
	// Gram matrix is still in the work buffer, so go straight to the shifted-Cholesky step
	const bool direct_shift = (r_engine == R_SHIFT_DIRECT);
	if (direct_shift) r_engine = 0;
	for (int e = r_engine; e >= 1; e--) {                // 2: bf16-split Gram, 1: fp64 Gram; with check_now a rejected level escalates
		rc = gram_r(rpp, ldr, wq + L.z, ap, lda, m, c, wq, wr, L, e == 2, st);
		if (rc) return rc;
		bool ok = true;
		if (check_now) {
			unsigned status = 0;
			rc = read_status(wq, L, h_pinned, st, &status);
			if (rc) return rc;
			ok = (status == 0);
		}
		if (ok) {
			g_min_level = std::min(g_min_level, e);
			// speculative (unchecked) launch under the auto policy: the kernel itself skips the pass when the level was rejected
			const unsigned* skip = (!check_now && g_policy == 0) ? reinterpret_cast<const unsigned*>(wq + L.status) + 16 * g_slot : nullptr;
			return apply_rinv(engine, qp, ldq, ap, lda, rpp, ldr, m, c, wq + L.z, st, /*z_ready=*/true, skip);
		}
	}
	if ((direct_shift || (r_engine >= 1 && check_now)) && g_shifted && g_policy == 0) {
		// Both Gram levels rejected the panel (cond beyond ~1e6, or rank deficient).  Shifted Cholesky QR: the fp64 Gram matrix is
		// still in the work buffer; R1 = chol(G + s I) always exists, Q1 = A inverse(R1) has cond(Q1) <~ 1e5, and one unshifted fp64
		// sweep on Q1 in place finishes the panel: A = Q (R2 R1).  About 2x faster than the Householder fold below and, after that
		// second step, at least as orthogonal as its single indirect sweep.
		double* gsum = reinterpret_cast<double*>(wq + L.gsub) + (size_t)GRAM_NSPLIT * 16 * 256;
		float* r1 = wq + L.r3; float* r2 = wq + L.r4;
		const double coef = 11.0 * ((double)m * (double)c + (double)c * (double)(c + 1)) * 1.1102230246251565e-16;
		rc = chol_from_g(r1, c, wq + L.z, gsum, m, c, wq, L, /*bf16=*/false, st, g_hsig.dev, coef);
		if (rc) return rc;
		unsigned status = 0;
		rc = read_status(wq, L, h_pinned, st, &status);
		if (rc) return rc;
		if (status == 0) {
			rc = apply_rinv(engine, qp, ldq, ap, lda, r1, c, m, c, wq + L.z, st, /*z_ready=*/true);
			if (rc) return rc;
			rc = gram_r(r2, c, wq + L.z, qp, ldq, m, c, wq, wr, L, /*bf16=*/false, st);
			if (rc) return rc;
			rc = read_status(wq, L, h_pinned, st, &status);
			if (rc) return rc;
			if (status == 0) {
				rc = apply_rinv(engine, qp, ldq, qp, ldq, r2, c, m, c, wq + L.z, st, /*z_ready=*/true);
			} else {
				// Q1 is still numerically rank deficient: the input has an (almost) exactly dependent column whose rounding residue is
				// itself dependent (e.g. two constant columns).  No triangular solve can make an orthonormal column out of that; a
				// second SHIFTED step keeps everything bounded instead -- the other columns come out orthonormal, the residual stays
				// at rounding level, R shows the deficiency as a tiny diagonal entry and that one column of Q is left un-normalised.
				rc = chol_from_g(r2, c, wq + L.z, gsum, m, c, wq, L, /*bf16=*/false, st, g_hsig.dev, coef);
				if (rc) return rc;
				rc = read_status(wq, L, h_pinned, st, &status);
				if (rc) return rc;
				if (status == 0) {
					rc = apply_rinv(engine, qp, ldq, qp, ldq, r2, c, m, c, wq + L.z, st, /*z_ready=*/true);
				} else {                                     // non-finite data: last resort, the Householder engine on Q1
					g_used_householder = true;
					rc = fold_r(r2, c, qp, ldq, m, c, wq, wr, st);
					if (rc) return rc;
					rc = apply_rinv(engine, qp, ldq, qp, ldq, r2, c, m, c, wq + L.z, st);
				}
			}
			if (rc) return rc;
			const unsigned gbp = (unsigned)std::min<size_t>(1024, cdiv(c * c, 256));
			hipLaunchKernelGGL(tsqrmi::rmul_kernel, dim3(gbp), dim3(256), 0, st, rpp, ldr, r2, c, r1, c, (int)c);
			HIPCHK(hipGetLastError());
			g_min_level = 0;
			g_used_shift = true;
			return 0;
		}
	}
	g_min_level = 0;
	g_used_householder = true;
	rc = fold_r(rpp, ldr, ap, lda, m, c, wq, wr, st);
	if (rc) return rc;
	return apply_rinv(engine, qp, ldq, ap, lda, rpp, ldr, m, c, wq + L.z, st);
}

// one sweep of 64-wide-panel block QR:  (q, r) <- qr(a);  a is overwritten for n > 64; q may alias a.
int sweep(int engine, int r_engine, bool check_now, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda, size_t m, size_t n,
          float* wq, float* wr, const WqLayout& L, unsigned* h_pinned, hipStream_t st) {
	const size_t npanels = cdiv(n, PW);
	for (size_t pi = 0; pi < npanels; pi++) {
		const size_t P = pi * PW, c = std::min(PW, n - P);
		float* ap = a + P * lda;
		for (size_t bi = 0; bi < pi; bi++) {             // block modified Gram-Schmidt against finished panels
			const size_t B = bi * PW;
			ProfScope ps(KC_COUPLE, st);
			// S = Qb^T Ap  (exact fp32 MFMA), written into R(B:B+64, P:P+c); then Ap <- Ap - Qb * S on the mode's MFMA engine
			const GramPlan g = gram_plan(m, PW);
			tsqrmi::CrossArgs ca{};
			ca.x = q + B * ldq; ca.ldx = ldq; ca.y = ap; ca.ldy = lda; ca.m = m; ca.ny = (int)c;
			ca.nchunks = g.nch; ca.cpw = g.cpw; ca.nwaves = g.nwaves; ca.part = reinterpret_cast<double*>(wr);
			hipLaunchKernelGGL(tsqrmi::cross_kernel, dim3(g.nblocks), dim3(256), 0, st, ca);
			const int nelem = 16 * 256;
			const int nsplit = std::min(GRAM_NSPLIT, g.nblocks);
			double* sub = reinterpret_cast<double*>(wq + L.gsub);
			double* gsum = sub + (size_t)GRAM_NSPLIT * nelem;
			hipLaunchKernelGGL(tsqrmi::gram_reduce_kernel, dim3((nelem + 255) / 256, nsplit), dim3(256), 0, st,
			                   sub, ca.part, g.nblocks, nelem, nsplit);
			hipLaunchKernelGGL(tsqrmi::gram_reduce2_kernel, dim3((nelem + 255) / 256), dim3(256), 0, st, gsum, sub, nelem, nsplit);
			hipLaunchKernelGGL(tsqrmi::cross_finish_kernel, dim3(16), dim3(256), 0, st, r + P * ldr + B, ldr, wq + L.s, gsum, (int)c);
			HIPCHK(hipGetLastError());
			tsqrmi::ApplyArgs ua{};
			ua.a = q + B * ldq; ua.lda = ldq; ua.q = ap; ua.ldq = lda; ua.m = m; ua.n = (int)PW; ua.z = wq + L.s; ua.n_out = (int)c;
			const int rc2 = (engine == 0) ? launch_apply_any<0, 4, true>(ua, st)
			                              : (engine == 1 ? launch_apply_any<1, 4, true>(ua, st) : launch_apply_any<2, 4, true>(ua, st));
			if (rc2) return rc2;
			HIPCHK(hipGetLastError());
		}
		const int rc = panel_qr(engine, r_engine, check_now, q + P * ldq, ldq, r + P * ldr + P, ldr, ap, lda, m, c, wq, wr, L, h_pinned, st);
		if (rc) return rc;
	}
	return 0;
}

}  // namespace

extern "C" {

int tsqr_mi_version(void) { return 100; }
const char* tsqr_mi_last_error(void) { return g_last_error.c_str(); }

size_t tsqr_mi_batch_size_log2(size_t m) { return ref_bs_log2(m); }
size_t tsqr_mi_batch_size(size_t m) { return ref_bs(m); }

size_t tsqr_mi_working_q_size(size_t m, size_t n) {
	if (m == 0 || n == 0) return 0;
	return std::max(ref_wq(m, n), wq_layout(m, n).total);
}
size_t tsqr_mi_working_r_size(size_t m, size_t n) {
	if (m == 0 || n == 0) return 0;
	size_t need = 0;
	for (size_t P = 0; P < n; P += PW) {
		need = std::max(need, make_plan(m, std::min(PW, n - P)).stack_a);
		need = std::max(need, gram_plan(m, std::min(PW, n - P)).part_floats);
		if (n > PW) need = std::max(need, (size_t)gram_plan(m, PW).nblocks * 16 * 256 * 2);
	}
	// the stack of a dist/gathered fold is tiny; nothing extra needed
	return std::max(ref_wr(m, n), need);
}
size_t tsqr_mi_working_l_size(size_t m) { return m == 0 ? 0 : std::max<size_t>(ref_bs(m) + 1, 8); }   // >= 3 words: the Cholesky status words
size_t tsqr_mi_working_reorth_size(size_t m) { return 16 * 16 * 2 + m * 16; }

void tsqr_mi_profile_enable(int on) {
	if (on && !g_prof.created) {
		for (int i = 0; i < 2 * Prof::MAXEV; i++) (void)hipEventCreate(&g_prof.ev[i]);
		g_prof.created = true;
	}
	g_prof.on = on != 0;
	g_prof.n = 0;
	for (int k = 0; k < KC_COUNT; k++) { g_prof.ms[k] = 0; g_prof.launches[k] = 0; }
}
int tsqr_mi_profile_read(double* ms, long* launches, int max_classes) {
	prof_collect();
	const int k = std::min(max_classes, (int)KC_COUNT);
	for (int i = 0; i < k; i++) { ms[i] = g_prof.ms[i]; launches[i] = g_prof.launches[i]; }
	return k;
}

void tsqr_mi_set_policy(int policy) {
	switch (policy) {
		case 0: g_policy = 0; g_gram_level = 2; break;    // auto
		case 1: g_policy = 1; g_gram_level = 2; break;    // always Householder TSQR
		case 2: g_policy = 2; g_gram_level = 1; break;    // always fp64 Gram (no fallback)
		case 3: g_policy = 2; g_gram_level = 2; break;    // always bf16-split Gram (no check, no fallback)
		case 4: g_policy = 0; g_gram_level = 1; break;    // auto without the bf16-split level
		default: break;
	}
}
int tsqr_mi_last_engine(void) { return g_last_engine; }

void tsqr_mi_set_tuning(int level0_waves, int tree_chunks_per_wave) {
	if (level0_waves > 0) g_level0_waves = level0_waves;
	if (tree_chunks_per_wave > 1) g_tree_cpw = tree_chunks_per_wave;
}
void tsqr_mi_set_tuning2(int gram_waves, int apply_waves) {
	if (gram_waves > 0) g_gram_waves = gram_waves;
	if (apply_waves > 0) g_apply_wgs = std::max(1, apply_waves / 4);   // the apply kernel is launched as a persistent grid of workgroups (4 waves each)
}

int tsqr_mi_qr_f32(int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                   size_t m, size_t n, void* wq_v, void* wr_v, float* reorth_w, unsigned* d_wl, unsigned* h_wl,
                   void* stream) {
	(void)reorth_w; (void)d_wl;
	if (n > m || m == 0 || n == 0) return TSQR_MI_ERROR_INVALID_SIZE;     // reference src/blockqr.cu:409-411
	const int engine = engine_of(mode);
	if (engine < 0) { g_last_error = "compute_mode not implemented on gfx950"; return TSQR_MI_ERROR_UNSUPPORTED; }
	hipStream_t st = reinterpret_cast<hipStream_t>(stream);
	float* wq = reinterpret_cast<float*>(wq_v);
	float* wr = reinterpret_cast<float*>(wr_v);
	const WqLayout L = wq_layout(m, n);
	// auto policy: every mode starts at the bf16-split Gram level (exact products, fp64 accumulation across K-steps: more accurate
	// than any plain fp32 evaluation of A^T A, accepted only for well-conditioned panels), then the fp64 Gram level, the shifted
	// Cholesky QR step and the Householder fold; the mode selects the MFMA engine of the apply pass.  Policy 4 skips the bf16 level.
	const bool use_gram = (g_policy == 2) || (g_policy == 0);
	const bool may_fall_back = use_gram && g_policy == 0;
	// single panel, single sweep: run speculatively and look at the status at the final sync (A is untouched for n <= 64);
	// otherwise (several panels or a second sweep consuming Q) verify each panel right away.
	const bool deferred = may_fall_back && n <= PW && !reorth;
	const bool check_now = may_fall_back && !deferred;
	const unsigned gb = (unsigned)std::min<size_t>(1024, cdiv(n * n, 256));
	g_last_engine = 0;
	g_min_level = 2;
	g_used_shift = g_used_householder = false;
	resolve_host_sig(h_wl);

	// R-factor engine levels: 2 bf16-split Gram (memory-bound), 1 fp64 Gram, 0 Householder TSQR.  Deferred mode runs a level
	// speculatively and steps down when chol_kernel rejected it; with check_now panel_qr escalates per panel by itself.
	const int first_level = !use_gram ? 0 : g_gram_level;
	for (int level = first_level; level >= 0; level--) {
		int rc;
		if (!reorth) {
			// level 0 reached in the speculative (deferred) mode: both Gram levels were rejected and the fp64 Gram matrix of A is
			// still in the work buffer: panel_qr takes the shifted-Cholesky path on it before the Householder fold
			const bool retry_checked = (level == 0 && deferred && g_shifted && first_level >= 1);
			rc = sweep(engine, retry_checked ? R_SHIFT_DIRECT : level, check_now || retry_checked, q, ldq, r, ldr, a, lda, m, n, wq, wr, L, h_wl, st);
			if (rc) return rc;
			if (n > PW) hipLaunchKernelGGL(tsqrmi::zero_lower_kernel, dim3(gb), dim3(256), 0, st, r, ldr, (int)n);
		} else {
			// two sweeps: A = Q1 R1, then Q1 = Q R2 in place, R = R2 * R1 (the reference's BCGS2 plays this role).  R1 and R2 live in
			// the work buffer (packed, ld n); every sweep writes their upper triangles in full and rmul_kernel reads nothing else,
			// so neither needs zero-filling, and the product is written straight into the caller's r (zeros below the diagonal)
			float* r1 = wq + L.r1; float* r2 = wq + L.r2;
			// single panel: the first sweep's (last) apply launch accumulates Q^T Q while the block is in LDS, so that the second
			// sweep's bf16-level Gram pass over Q is not needed
			const bool fuse = g_fuse_gramq && n <= PW && g_policy == 0 && level == 2;
			if (g_spec_reorth && n <= PW && g_policy == 0 && level == first_level && !g_prof.on) {
				// Optimistic attempt: both sweeps, the R product and the completion flag are enqueued without looking at a verdict.
				// Device-side chain: apply 1 skips when Cholesky 1 rejected; Cholesky 2 then reports "rejected" at once; apply 2 (in
				// place) skips when Cholesky 2 rejected -- so A stays intact and Q holds Q1 or garbage, never a half-applied state.
				if (fuse) { g_gramq_part = reinterpret_cast<double*>(wr); g_gramq_cap = gram_plan(m, n).nblocks; g_gramq_nparts = 0; }
				g_slot = 0; g_prev_slot = -1;
				rc = sweep(engine, level, /*check_now=*/false, q, ldq, r1, n, a, lda, m, n, wq, wr, L, h_wl, st);
				g_gramq_part = nullptr; g_gramq_cap = 0;
				if (!rc) {
					g_gramq_ready = fuse && g_gramq_nparts > 0;
					g_slot = 1; g_prev_slot = 0;
					rc = sweep(engine, level, /*check_now=*/false, q, ldq, r2, n, q, ldq, m, n, wq, wr, L, h_wl, st);
					g_gramq_ready = false;
				}
				g_slot = 0; g_prev_slot = -1;
				if (rc) return rc;
				launch_rmul(r, ldr, r2, n, r1, n, n, st);
				HIPCHK(hipGetLastError());
				unsigned s01[2] = {1u, 1u};
				rc = signal_and_wait(st);
				if (rc < 0) return rc;
				if (rc == 0) {
					s01[0] = reinterpret_cast<volatile unsigned*>(g_hsig.host)[0];
					s01[1] = reinterpret_cast<volatile unsigned*>(g_hsig.host)[4];
				} else {
					HIPCHK(hipStreamSynchronize(st));
					unsigned w[17];
					HIPCHK(hipMemcpy(w, wq + L.status, sizeof(w), hipMemcpyDeviceToHost));
					s01[0] = w[0]; s01[1] = w[16];
				}
				if (s01[0] == 0 && s01[1] == 0) break;       // both sweeps accepted: done (g_min_level was set by panel_qr)
				g_min_level = 2;
				if (s01[0] == 0) {
					// the first sweep stands (Q holds Q1, r1 is valid); only the second one must be redone, now checked and below
					// the level that was just rejected
					rc = sweep(engine, level - 1, /*check_now=*/true, q, ldq, r2, n, q, ldq, m, n, wq, wr, L, h_wl, st);
					if (rc) return rc;
					g_min_level = std::min(g_min_level, level);  // (first sweep ran at `level`)
					launch_rmul(r, ldr, r2, n, r1, n, n, st);
					HIPCHK(hipGetLastError());
					rc = signal_and_wait(st);
					if (rc < 0) return rc;
					if (rc == 1) HIPCHK(hipStreamSynchronize(st));
					break;
				}
				level = std::max(level - 1, 0);              // first sweep rejected at this level: checked path from the next one
			}
			const bool fuse2 = g_fuse_gramq && n <= PW && g_policy == 0 && first_level == 2;
			if (fuse2) { g_gramq_part = reinterpret_cast<double*>(wr); g_gramq_cap = gram_plan(m, n).nblocks; g_gramq_nparts = 0; }
			rc = sweep(engine, level, check_now, q, ldq, r1, n, a, lda, m, n, wq, wr, L, h_wl, st);
			g_gramq_part = nullptr; g_gramq_cap = 0;
			if (rc) return rc;
			g_gramq_ready = fuse2 && g_gramq_nparts > 0;     // the second sweep always starts at the first level again (Q1 is well conditioned)
			rc = sweep(engine, first_level, check_now, q, ldq, r2, n, q, ldq, m, n, wq, wr, L, h_wl, st);
			g_gramq_ready = false;
			if (rc) return rc;
			launch_rmul(r, ldr, r2, n, r1, n, n, st);
		}
		HIPCHK(hipGetLastError());
		if (level > 0 && deferred) {
			unsigned status = 0;
			rc = signal_and_wait(st);
			if (rc < 0) return rc;
			if (rc == 0) status = reinterpret_cast<volatile unsigned*>(g_hsig.host)[0];   // written by the Cholesky kernel
			else {
				rc = read_status(wq, L, h_wl, st, &status);
				if (rc) return rc;
			}
			if (status != 0) { g_min_level = 2; continue; }       // rejected: step down and redo
		} else {
			rc = signal_and_wait(st);                        // completion flag in the pinned h_wl, or (rc == 1) a plain stream sync
			if (rc < 0) return rc;
			if (rc == 1) HIPCHK(hipStreamSynchronize(st));
		}
		break;
	}
	g_last_engine = !use_gram ? 0 : (g_used_householder ? 2 : (g_used_shift ? 4 : (g_min_level == 2 ? 3 : (g_min_level == 1 ? 1 : 2))));
	prof_collect();
	return TSQR_MI_SUCCESS;
}

int tsqr_mi_local_r_f32(float* r, size_t ldr, const float* a, size_t lda, size_t m, size_t n,
                        void* wq, void* wr, void* stream) {
	if (m == 0 || n == 0 || n > PW) return TSQR_MI_ERROR_INVALID_SIZE;
	return fold_r(r, ldr, a, lda, m, n, reinterpret_cast<float*>(wq), reinterpret_cast<float*>(wr),
	              reinterpret_cast<hipStream_t>(stream));
}

int tsqr_mi_apply_rinv_f32(int mode, float* q, size_t ldq, const float* a, size_t lda, const float* r, size_t ldr,
                           size_t m, size_t n, void* wq, void* stream) {
	if (m == 0 || n == 0 || n > PW) return TSQR_MI_ERROR_INVALID_SIZE;
	const int engine = engine_of(mode);
	if (engine < 0) return TSQR_MI_ERROR_UNSUPPORTED;
	const WqLayout L = wq_layout(m, n);
	return apply_rinv(engine, q, ldq, a, lda, r, ldr, m, n, reinterpret_cast<float*>(wq) + L.z,
	                  reinterpret_cast<hipStream_t>(stream));
}

// ---- staged Gram engine (row-partitioned multi-GPU path: local Gram -> all-reduce of G -> Cholesky -> apply) ----
int tsqr_mi_gram_f32(int level, double* gsum, const float* a, size_t lda, size_t m, size_t n, void* wq, void* wr, void* stream) {
	if (m == 0 || n == 0 || n > PW || (level != 1 && level != 2)) return TSQR_MI_ERROR_INVALID_SIZE;
	const WqLayout L = wq_layout(m, n);
	return gram_g(gsum, a, lda, m, n, reinterpret_cast<float*>(wq), reinterpret_cast<float*>(wr), L, level == 2,
	              reinterpret_cast<hipStream_t>(stream));
}

int tsqr_mi_chol_f32(int level, float* r, size_t ldr, const double* gsum, size_t m, size_t n, void* wq_v, unsigned* status_out, void* stream) {
	// level 3: shifted Cholesky of an fp64 (level-1) Gram matrix, G + s I with s from the row count m passed here (the caller
	// passes the GLOBAL row count for a row-partitioned matrix); the caller then runs one more plain sweep on the resulting Q
	if (m == 0 || n == 0 || n > PW || (level != 1 && level != 2 && level != 3)) return TSQR_MI_ERROR_INVALID_SIZE;
	hipStream_t st = reinterpret_cast<hipStream_t>(stream);
	float* wq = reinterpret_cast<float*>(wq_v);
	const WqLayout L = wq_layout(m, n);
	const double coef = (level == 3) ? 11.0 * ((double)m * (double)n + (double)n * (double)(n + 1)) * 1.1102230246251565e-16 : 0.0;
	int rc = chol_from_g(r, ldr, wq + L.z, gsum, m, n, wq, L, level == 2, st, nullptr, coef);   // m = rows (local is conservative for levels 1, 2)
	if (rc) return rc;
	if (!status_out) return 0;                           // asynchronous: read the verdict later with tsqr_mi_chol_status
	unsigned status = 0;
	rc = read_status(wq, L, nullptr, st, &status);       // blocking: the caller decides on the next level
	if (rc) return rc;
	*status_out = status;
	return 0;
}

// Blocks until everything enqueued on `stream` so far has completed: a one-thread kernel raises a word in library-owned pinned
// memory and the host spins on it (hipStreamSynchronize as the fallback).  For staged callers that end with an asynchronous call.
int tsqr_mi_stream_wait(void* stream) {
	hipStream_t st = reinterpret_cast<hipStream_t>(stream);
	static unsigned* own = nullptr;
	static unsigned* own_dev = nullptr;
	if (!own) {
		if (hipHostMalloc(reinterpret_cast<void**>(&own), 64, hipHostMallocDefault) != hipSuccess) { own = nullptr; (void)hipGetLastError(); }
		else if (hipHostGetDevicePointer(reinterpret_cast<void**>(&own_dev), own, 0) != hipSuccess) { own_dev = nullptr; (void)hipGetLastError(); }
	}
	if (own && own_dev && g_host_flag && !g_prof.on) {
		unsigned seq = ++g_seq;
		if (seq == 0) seq = ++g_seq;
		volatile unsigned* flag = own;
		*flag = 0;
		hipLaunchKernelGGL(tsqrmi::host_flag_kernel, dim3(1), dim3(1), 0, st, own_dev, seq);
		if (hipGetLastError() == hipSuccess) {
			for (;;) {
				bool seen = false;
				for (int i = 0; i < 20000 && !seen; i++) { seen = (*flag == seq); if (!seen) __builtin_ia32_pause(); }
				if (seen) return 0;
				const hipError_t e = hipStreamQuery(st);
				if (e == hipSuccess) return 0;
				if (e != hipErrorNotReady) HIPCHK(e);
			}
		}
	}
	HIPCHK(hipStreamSynchronize(st));
	return 0;
}

int tsqr_mi_chol_status(const void* wq_v, size_t m, size_t n, unsigned* status_out, void* stream) {
	if (m == 0 || n == 0 || n > PW || !status_out) return TSQR_MI_ERROR_INVALID_SIZE;
	const WqLayout L = wq_layout(m, n);
	return read_status(reinterpret_cast<const float*>(wq_v), L, nullptr, reinterpret_cast<hipStream_t>(stream), status_out);
}

int tsqr_mi_apply_z_f32(int mode, float* q, size_t ldq, const float* a, size_t lda, size_t m, size_t n, void* wq_v, void* stream) {
	if (m == 0 || n == 0 || n > PW) return TSQR_MI_ERROR_INVALID_SIZE;
	const int engine = engine_of(mode);
	if (engine < 0) return TSQR_MI_ERROR_UNSUPPORTED;
	float* wq = reinterpret_cast<float*>(wq_v);
	const WqLayout L = wq_layout(m, n);
	// under the auto policy a rejected Cholesky (status word != 0) turns a speculatively enqueued apply into a no-op
	const unsigned* skip = (g_policy == 0) ? reinterpret_cast<const unsigned*>(wq + L.status) : nullptr;
	return apply_rinv(engine, q, ldq, a, lda, nullptr, 0, m, n, wq + L.z, reinterpret_cast<hipStream_t>(stream), /*z_ready=*/true, skip);
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

size_t tsqr_mi_gram_elems(size_t n) { const size_t NT = np_of(n) / 16; return NT * (NT + 1) / 2 * 256; }

int tsqr_mi_rmul_f32(float* r, size_t ldr, const float* r2, size_t ldr2, size_t n, void* wq, void* stream) {
	if (n == 0) return TSQR_MI_ERROR_INVALID_SIZE;
	hipStream_t st = reinterpret_cast<hipStream_t>(stream);
	float* r1 = reinterpret_cast<float*>(wq);            // n*n floats at the start of wq
	const unsigned gb = (unsigned)std::min<size_t>(1024, cdiv(n * n, 256));
	hipLaunchKernelGGL(tsqrmi::copy2d_kernel, dim3(gb), dim3(256), 0, st, r1, n, r, ldr, (int)n, (int)n);
	hipLaunchKernelGGL(tsqrmi::rmul_kernel, dim3(gb), dim3(256), 0, st, r, ldr, r2, ldr2, r1, n, (int)n);
	HIPCHK(hipGetLastError());
	return 0;
}

// ---- RCCL path: the library is resolved lazily so that libtsqr_mi.so loads without librccl ----
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
static void* rccl_symbol(const char* name) {
	static void* h = nullptr;
	if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
	if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	return h ? dlsym(h, name) : nullptr;
}
static nccl_allgather_t resolve_allgather() {
	static nccl_allgather_t fn = nullptr;
	if (!fn) fn = reinterpret_cast<nccl_allgather_t>(rccl_symbol("ncclAllGather"));
	return fn;
}
static nccl_allreduce_t resolve_allreduce() {
	static nccl_allreduce_t fn = nullptr;
	if (!fn) fn = reinterpret_cast<nccl_allreduce_t>(rccl_symbol("ncclAllReduce"));
	return fn;
}

int tsqr_mi_qr_f32_dist(int mode, int reorth, float* q, size_t ldq, float* r, size_t ldr, float* a, size_t lda,
                        size_t m_local, size_t n, void* wq_v, void* wr_v, float* gather_buf,
                        void* nccl_comm, int nranks, void* stream) {
	if (m_local == 0 || n == 0 || n > PW || nranks < 1) return TSQR_MI_ERROR_INVALID_SIZE;
	const int engine = engine_of(mode);
	if (engine < 0) return TSQR_MI_ERROR_UNSUPPORTED;
	hipStream_t st = reinterpret_cast<hipStream_t>(stream);
	float* wq = reinterpret_cast<float*>(wq_v);
	float* wr = reinterpret_cast<float*>(wr_v);
	nccl_allgather_t allgather = resolve_allgather();
	if (!allgather) { g_last_error = "librccl.so / ncclAllGather not found"; return TSQR_MI_ERROR_UNSUPPORTED; }
	const WqLayout L = wq_layout(std::max(m_local, (size_t)nranks * n), n);
	float* rl = wq + L.r2;                               // local R, n x n packed (ld n)
	const float* src = a; size_t ld_src = lda;
	const bool use_gram = (g_policy == 2) || (g_policy == 0);
	const int first_level = g_gram_level;
	nccl_allreduce_t allreduce = use_gram ? resolve_allreduce() : nullptr;
	for (int it = 0; it < (reorth ? 2 : 1); it++) {
		int rc;
		if (use_gram && allreduce) {
			// Gram engine: local Gram tiles -> all-reduce (fp64 sum) -> Cholesky on every rank -> apply; rejected levels step down
			double* gsum = reinterpret_cast<double*>(wq + L.gsub) + (size_t)GRAM_NSPLIT * 16 * 256;
			const size_t gelems = (np_of(n) / 16) * (np_of(n) / 16 + 1) / 2 * 256;
			bool done = false;
			for (int level = first_level; level >= 1 && !done; level--) {
				rc = gram_g(gsum, src, ld_src, m_local, n, wq, wr, L, level == 2, st);
				if (rc) return rc;
				// ncclFloat64 == 8, ncclSum == 0 in nccl.h/rccl.h
				if (allreduce(gsum, gsum, gelems, 8, 0, nccl_comm, st) != 0) { g_last_error = "ncclAllReduce failed"; return -1; }
				float* rdst = (it == 0) ? r : rl;
				rc = chol_from_g(rdst, (it == 0) ? ldr : n, wq + L.z, gsum, m_local * (size_t)nranks, n, wq, L, level == 2, st);
				if (rc) return rc;
				// q does not alias the source: apply speculatively, then look at the status (one sync per sweep, no idle gap)
				const bool speculative = (q != src);
				if (speculative) {
					const unsigned* skip = (g_policy == 0) ? reinterpret_cast<const unsigned*>(wq + L.status) : nullptr;
					rc = apply_rinv(engine, q, ldq, src, ld_src, nullptr, 0, m_local, n, wq + L.z, st, /*z_ready=*/true, skip);
					if (rc) return rc;
				}
				unsigned status = 0;
				rc = read_status(wq, L, nullptr, st, &status);
				if (rc) return rc;
				if (status == 0 || g_policy == 2) {
					if (!speculative) {
						rc = apply_rinv(engine, q, ldq, src, ld_src, nullptr, 0, m_local, n, wq + L.z, st, /*z_ready=*/true);
						if (rc) return rc;
					}
					done = true;
				}
			}
			if (done) {
				if (it == 1) {
					float* r1 = wq + L.r1;
					hipLaunchKernelGGL(tsqrmi::copy2d_kernel, dim3(16), dim3(256), 0, st, r1, n, r, ldr, (int)n, (int)n);
					hipLaunchKernelGGL(tsqrmi::rmul_kernel, dim3(16), dim3(256), 0, st, r, ldr, rl, n, r1, n, (int)n);
				}
				src = q; ld_src = ldq;
				continue;
			}
		}
		rc = fold_r(rl, n, src, ld_src, m_local, n, wq, wr, st);
		if (rc) return rc;
		// ncclFloat32 == 7 in nccl.h/rccl.h
		if (allgather(rl, gather_buf, n * n, 7, nccl_comm, st) != 0) { g_last_error = "ncclAllGather failed"; return -1; }
		// gather_buf is [rank][col][row]; viewed column-major with ld n it is a (nranks*n) x n stack only per rank,
		// so fold the ranks' blocks as one tall matrix of n-row blocks: rows = nranks*n, ld = n is wrong for that ->
		// restack into wr as a proper column-major (nranks*n) x n matrix.
		float* stack = wr;
		for (int k = 0; k < nranks; k++)
			hipLaunchKernelGGL(tsqrmi::copy2d_kernel, dim3(16), dim3(256), 0, st,
			                   stack + (size_t)k * n, (size_t)nranks * n, gather_buf + (size_t)k * n * n, n, (int)n, (int)n);
		float* rdst = (it == 0) ? r : rl;
		const size_t ldd = (it == 0) ? ldr : n;
		rc = fold_r(rdst, ldd, stack, (size_t)nranks * n, (size_t)nranks * n, n, wq, wr + (size_t)nranks * n * n, st);
		if (rc) return rc;
		rc = apply_rinv(engine, q, ldq, src, ld_src, rdst, ldd, m_local, n, wq + L.z, st);
		if (rc) return rc;
		if (it == 1) {
			float* r1 = wq + L.r1;
			hipLaunchKernelGGL(tsqrmi::copy2d_kernel, dim3(16), dim3(256), 0, st, r1, n, r, ldr, (int)n, (int)n);
			hipLaunchKernelGGL(tsqrmi::rmul_kernel, dim3(16), dim3(256), 0, st, r, ldr, rl, n, r1, n, (int)n);
		}
		src = q; ld_src = ldq;
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(st));
	return TSQR_MI_SUCCESS;
}

}  // extern "C"
