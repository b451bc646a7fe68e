// tsqr_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the tall-skinny QR engine.
//
// Replaces, with a different (MI355X-first) algorithm, the device side of the reference hot path:
//   qr32x16_batched_kernel / qr32x16_core      reference src/tcqr32x16.cu:1373-1581   -> fold_kernel
//   tsqr_backward / tsqr_backward_layer0       reference src/tsqr.cu:143-204, 591-656  -> apply_wg_kernel
//   cuBLAS GEMMs between panels                reference src/blockqr.cu:92-116         -> proj_* / update_kernel
//
// Data layout used by every streaming kernel ("(c,q) layout"): a wave owns a chunk of 64 rows x NP
// columns (NP = 16*NT).  Lane l = 16*q + c holds, for every 16-column tile ct, column 16*ct+c and the
// sixteen rows {16*rt + 4*q + i : rt,i in 0..3} in registers p[ct][4*rt+i].  This is exactly the
// C/D layout of the 16x16 MFMAs (col = lane&15, row = 4*(lane>>4)+reg), loads/stores are 16 B per lane
// along the column-major leading dimension, a column reduction is 16 in-lane FMAs + one 4-lane (q) sum,
// and "broadcast the pivot column to every column of the tile" is one DPP row_newbcast.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace tsqrmi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte global access
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. makes every wave wait for its global
// loads AND stores in flight -- a prefetch issued before the barrier, or the streaming stores of the previous block, would then be
// waited for at every barrier.  Global data never travels between the waves of a workgroup in these kernels, so the LDS counter is
// all a barrier has to wait for.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		static_for<I + 1, N>(f);
	}
}

// value of lane (16*q + K) for every lane of DPP row q  (v_mov_b32_dpp row_newbcast:K)
template <int K>
__device__ __forceinline__ float bcast16(float x) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + K, 0xf, 0xf, true));
}

// sum over the four lanes {c, c+16, c+32, c+48}; result in all four  (v_permlane32_swap + v_permlane16_swap).
// Inline asm on purpose: __builtin_amdgcn_permlane{32,16}_swap returned the same register for both results
// under hipcc / ROCm 7.2 (sum degenerated to 2*x on MI355X; tests/test_gpu_primitives.py pins the asm form).
// The two v_nop are the wait states a VALU write needs before v_permlane*_swap reads it.
__device__ __forceinline__ float xq_sum(float x) {
	float a = x, b = x;
	asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
	const float y = a + b;
	float c = y, d = y;
	asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
	return c + d;
}

// ---------------------------------------------------------------------------------------------
// chunk load / store in the (c,q) layout
// ---------------------------------------------------------------------------------------------
// NTL: nontemporal loads -- for the stacks of R factors of the Householder engine, which are read once and must not displace the
// caller's A from the Infinity Cache (the apply pass reads A again); never for A itself.
template <int NT, bool NTL = false>
__device__ __forceinline__ void load_chunk(float (&p)[NT][16], const float* __restrict__ src, size_t ld,
                                           size_t row0, size_t m, int n, int c, int q) {
	const bool full = (row0 + 64 <= m);
#pragma unroll
	for (int ct = 0; ct < NT; ct++) {
		const int col = 16 * ct + c;
		const float* base = src + (size_t)col * ld + row0 + 4 * q;
		if (col < n) {
			if (full) {
#pragma unroll
				for (int rt = 0; rt < 4; rt++) {
					const f32x4u v = NTL ? __builtin_nontemporal_load(reinterpret_cast<const f32x4u*>(base + 16 * rt))
					                     : *reinterpret_cast<const f32x4u*>(base + 16 * rt);
					p[ct][4 * rt + 0] = v[0]; p[ct][4 * rt + 1] = v[1]; p[ct][4 * rt + 2] = v[2]; p[ct][4 * rt + 3] = v[3];
				}
			} else {
#pragma unroll
				for (int rt = 0; rt < 4; rt++)
#pragma unroll
					for (int i = 0; i < 4; i++) {
						const size_t row = row0 + 16 * rt + 4 * q + i;
						p[ct][4 * rt + i] = (row < m) ? base[16 * rt + i] : 0.0f;
					}
			}
		} else {
#pragma unroll
			for (int r = 0; r < 16; r++) p[ct][r] = 0.0f;
		}
	}
}

// ---- bf16 split helpers (shared by the fold, Gram and apply kernels) ----
__device__ __forceinline__ unsigned f2bf(float x) {    // round-to-nearest-even bf16 bits (finite inputs)
	const unsigned u = __builtin_bit_cast(unsigned, x);
	return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
	h = f2bf(x);
	const float r1 = x - __builtin_bit_cast(float, h << 16);
	m = f2bf(r1);
	const float r2 = r1 - __builtin_bit_cast(float, m << 16);
	l = f2bf(r2);
}

// the same 3-way split for a PAIR of values with v_cvt_pk_bf16_f32 (RNE); results are packed MFMA operand dwords
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
	const f32x2_t v = {a, b};
	return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
	// (the two subtractions of a stage are one packed fp32 operation, v_pk_add_f32)
	h = cvt_pk_bf16(a, b);
	const f32x2_t x = {a, b};
	const f32x2_t hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
	const f32x2_t r = x - hf;
	m = cvt_pk_bf16(r[0], r[1]);
	const f32x2_t mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
	const f32x2_t t = r - mf;
	l = cvt_pk_bf16(t[0], t[1]);
}

// 3-way bf16 split of NPAIR independent pairs, written stage by stage (scheduling barriers in between) so that the
// NPAIR dependency chains cvt -> unpack -> subtract -> cvt ... overlap instead of running back to back
template <int NPAIR>
__device__ __forceinline__ void split3_pairs(const float (&x)[2 * NPAIR], unsigned (&h)[NPAIR], unsigned (&m)[NPAIR], unsigned (&l)[NPAIR]) {
	float ra[NPAIR], rb[NPAIR];
#pragma unroll
	for (int i = 0; i < NPAIR; i++) h[i] = cvt_pk_bf16(x[2 * i], x[2 * i + 1]);
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int i = 0; i < NPAIR; i++) {
		ra[i] = x[2 * i] - __builtin_bit_cast(float, h[i] << 16);
		rb[i] = x[2 * i + 1] - __builtin_bit_cast(float, h[i] & 0xffff0000u);
	}
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int i = 0; i < NPAIR; i++) m[i] = cvt_pk_bf16(ra[i], rb[i]);
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int i = 0; i < NPAIR; i++) {
		ra[i] -= __builtin_bit_cast(float, m[i] << 16);
		rb[i] -= __builtin_bit_cast(float, m[i] & 0xffff0000u);
	}
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int i = 0; i < NPAIR; i++) l[i] = cvt_pk_bf16(ra[i], rb[i]);
	__builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------
// fold_kernel: streaming TPQRT.  Every wave folds `cpw` consecutive 64-row chunks of src into one
// upper-triangular R (NP x NP, kept packed in LDS):   R <- R-factor of [R ; chunk]
// with Householder reflectors whose top part is e_k (R stays triangular; only row k of R changes in
// step k).  Blocked: the 16 columns of the active tile are factored on the VALU (pivot broadcast fused
// into the FMAs by DPP row_newbcast, 4-lane sums by v_permlane swaps); the other tiles receive the
// block reflector I - V T^T V^T through v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains), with every
// operand taken from registers in the (c,q) layout -- the 16x16 transposes V needs are MFMAs against
// identity slices.  The register tiles rotate after each panel so the same 16-step code serves all of
// them (keeps the kernel inside the instruction cache).  The panel arithmetic is fp32 in both compute modes; the block reflector
// is applied with exact fp32 MFMA (fp32_notc, the role of reference src/tcqr32x16.cu:464-496) or with the bf16x3 error-corrected
// MFMA products (fp32_tc_cor, the role of src/tcqr32x16.cu:228-274 + 669-819: split operands, correction products first).
// ---------------------------------------------------------------------------------------------
struct FoldArgs {
	const float* src; size_t ld; size_t m; int n;     // source matrix (m x n, column-major)
	int nchunks; int cpw; int nwaves;                 // chunk = 64 rows; wave w folds chunks [w*cpw, (w+1)*cpw)
	float* dst; size_t dst_ld; int rows_store; int cols_store;   // wave w writes rows [w*rows_store, ...) of dst
	int tri_init;                                     // tree levels over 64-row upper-triangular blocks: the wave's first block IS its
	                                                  // initial R (copied, not folded), which makes a binary tree cost one fold per level
	int cor;                                          // fp32_tc_cor: block reflectors applied with the bf16x3 error-corrected MFMA products
};

// acc += (lane 16q+K of src) * other     -- one v_fmac_f32_dpp.  FIRST=true pads the two wait states a
// VALU-written DPP source needs (hipcc does not look inside asm).
template <int K, bool FIRST>
__device__ __forceinline__ void fmac_bcast(float& acc, float src, float other) {
	if constexpr (FIRST)
		asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
		             : "+v"(acc) : "v"(src), "v"(other), "n"(K));
	else
		asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
		             : "+v"(acc) : "v"(src), "v"(other), "n"(K));
}

__device__ __forceinline__ float fast_rcp(float a) {     // v_rcp_f32 + one Newton step (~0.5 ulp)
	const float r = __builtin_amdgcn_rcpf(a);
	return fmaf(r, fmaf(-a, r, 1.0f), r);
}

// One Householder step on the active tile.  KK: column inside the tile (compile time, DPP control).
//   p0   : the tile, 16 rows per lane (column c = lane&15)        Trow: row c of the panel's T (upper triangular)
//   sc   : scale of this lane's own column (v_c = p0 * sc once column c has been factored)
//   Rrow : packed row K of R in LDS, entry (K, first column of the tile + c) at Rrow[c - KK]
//   rkk, rkc : Rrow[0] and Rrow[c - KK], loaded one step ahead by the caller (hides the LDS latency)
// The reflector is kept orthogonal to rounding whatever the accuracy of v_sqrt/v_rcp: beta only fixes v = [1; x*inv],
// tau is then computed from that v (tau = 2 / (1 + inv^2 ||x||^2)) and the pivot entry is updated like any other
// column (r_kk - tau*w_k), so an inexact beta shows up as a tiny backward error, never as loss of orthogonality.
//   glim : last 4-register group (16-row tile) of the chunk that carries data in this panel (3 for a dense chunk; the panel
//          index when the chunk is an upper-triangular R block: rows below the panel's own row tile are zero)
// Cooperative fold (fold_coop_kernel): the WAVES waves of a workgroup hold one 64-row block each of a stack [R; B_1; ...; B_{WAVES-1}]
// and run ONE 64-step reflector chain for the whole stack.  What a reflector needs from all blocks is a sum over the waves: the
// column inner products x_k^T x_c of a panel step (sixteen floats per wave) and V^T B of a block reflector (one accumulator tile
// per wave and trailing tile).  Every wave adds the partial results of all waves in the same fixed order, so all of them hold
// bit-identical reflectors and update their own copy of R in LDS identically -- no shared state besides the two exchange arrays.
#ifndef TSQR_FOLD_COOP_WAVES
#define TSQR_FOLD_COOP_WAVES 8
#endif
constexpr int FOLD_COOP_WAVES = TSQR_FOLD_COOP_WAVES;    // (the macro: experiments with the fan-in, profiles/r04_experiment_log.md)
struct CoopCtx {
	float* xd;                           // [2][WAVES][64]: column inner products of a step, double buffered by the step's parity
	float* xw;                           // [3][WAVES][64][4]: V^T B accumulator tiles of a panel's trailing tiles
	int wave;
	int step;                            // running count of panel steps (parity selects the xd buffer)
};
template <bool COOP>
__device__ __forceinline__ float coop_sum(float d, CoopCtx* cc) {
	if constexpr (COOP) {
		const int lane = threadIdx.x & 63;
		float* buf = cc->xd + (cc->step & 1) * FOLD_COOP_WAVES * 64;
		cc->step++;
		buf[cc->wave * 64 + lane] = d;
		lds_barrier();                                   // a buffer is rewritten two steps later: every wave has read it by then
		float v[FOLD_COOP_WAVES];
#pragma unroll
		for (int w = 0; w < FOLD_COOP_WAVES; w++) v[w] = buf[w * 64 + lane];
		float t = v[0];
#pragma unroll
		for (int w = 1; w < FOLD_COOP_WAVES; w++) t += v[w];    // (fixed order: bit-identical on every wave)
		return t;
	} else {
		return d;
	}
}
template <bool COOP>
__device__ __forceinline__ f32x4 coop_sum4(f32x4 v, CoopCtx* cc, int slot) {
	if constexpr (COOP) {
		const int lane = threadIdx.x & 63;
		f32x4* buf = reinterpret_cast<f32x4*>(cc->xw) + slot * FOLD_COOP_WAVES * 64;
		buf[cc->wave * 64 + lane] = v;
		lds_barrier();                                   // (a slot is rewritten one panel = at least sixteen step barriers later)
		f32x4 u[FOLD_COOP_WAVES];
#pragma unroll
		for (int w = 0; w < FOLD_COOP_WAVES; w++) u[w] = buf[w * 64 + lane];
		f32x4 t = u[0];
#pragma unroll
		for (int w = 1; w < FOLD_COOP_WAVES; w++) t += u[w];
		return t;
	} else {
		return v;
	}
}

template <int KK, bool TRI, bool COOP = false>
__device__ __forceinline__ void panel_step(float (&p0)[16], float (&Trow)[16], float& sc, float* __restrict__ Rrow, int c,
                                           float rkk, float rkc, int glim, CoopCtx* cc = nullptr) {
	float acc = 0.0f;
	static_for<0, 4>([&](auto g) {
		constexpr int G = decltype(g)::value;
		if (!TRI || G <= glim)                           // wave-uniform; compiled out for dense chunks
			static_for<0, 4>([&](auto r) { fmac_bcast<KK, (TRI ? decltype(r)::value == 0 : 4 * G + decltype(r)::value == 0)>(acc, p0[4 * G + decltype(r)::value], p0[4 * G + decltype(r)::value]); });
	});
	const float d = coop_sum<COOP>(xq_sum(acc), cc);     // x_k^T x_c for every column c of the tile (COOP: over the blocks of all waves)
	const float ss = bcast16<KK>(d);                     // ||x_k||^2
	const float nrm = __builtin_amdgcn_sqrtf(fmaf(rkk, rkk, ss));
	const bool nz = nrm > 0.0f;
	const float beta = (rkk >= 0.0f) ? -nrm : nrm;       // -sign(rkk)*||.||, sign(0) = +1
	const float inv = nz ? fast_rcp(rkk - beta) : 0.0f;  // v_k = x_k * inv (top entry of v is 1)
	const float tau = nz ? 2.0f * fast_rcp(fmaf(inv * inv, ss, 1.0f)) : 0.0f;
	const float w = fmaf(inv, d, rkc);                   // w_c = r_kc + v_k^T b_c   (c == KK: r_kk + inv*||x||^2)
	const float tw = tau * w;
	const bool act = c > KK;
	const float g = act ? -tw * inv : 0.0f;              // b_c -= tau*w*v_k  ==  b_c += x_k * g
	if (c >= KK) Rrow[c - KK] = rkc - tw;                // predicated: lanes c < KK hold a stale prefetch of a finished entry
	static_for<0, 4>([&](auto gg) {
		constexpr int G = decltype(gg)::value;
		if (!TRI || G <= glim)
			static_for<0, 4>([&](auto r) { fmac_bcast<KK, (TRI && decltype(r)::value == 0)>(p0[4 * G + decltype(r)::value], p0[4 * G + decltype(r)::value], g); });
	});
	// T(0:KK, KK) = -tau * T(0:KK, 0:KK) * (V(:, 0:KK)^T v_k);   lane c owns row c of T
	if constexpr (KK > 0) {
		const float z = (c < KK) ? d * sc * inv : 0.0f;
		float t = 0.0f;
		static_for<0, KK>([&](auto l) { fmac_bcast<decltype(l)::value, decltype(l)::value == 0>(t, z, Trow[decltype(l)::value]); });
		Trow[KK] = (c < KK) ? -tau * t : ((c == KK) ? tau : 0.0f);
	} else {
		Trow[0] = (c == 0) ? tau : 0.0f;
	}
	sc = (c == KK) ? inv : sc;
}

// block reflector of the finished panel applied to one trailing tile (all operands in registers / LDS rows of R)
//   pj: trailing tile;  v: V of the panel in (c,q) layout (scaled);  vt[rt]: -V^T pieces (lane <-> row in tile rt)
//   ta[r] = T[4q+r][c];  Rp: packed R in LDS;  K0: first column of the panel;  colj: first column of the trailing tile
template <bool TRI, bool COOP = false>
__device__ __forceinline__ void trail_update(float (&pj)[16], const float (&v)[16], const f32x4 (&vt)[4], const float (&ta)[4],
                                             float* __restrict__ Rp, int K0, int colj, int NP, int c, int q, int glim,
                                             CoopCtx* cc = nullptr, int slot = 0) {
	// W0 = rows K0..K0+15 of R restricted to this tile, D layout (row 4q+r, column c)
	int idx[4];
	f32x4 w0;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const int k = K0 + 4 * q + r;
		idx[r] = k * NP - (k * (k - 1)) / 2 + (colj + c - k);
		w0[r] = Rp[idx[r]];
	}
	f32x4 w = w0;                                        // W = W0 + V^T B
	if constexpr (COOP) { if (cc->wave != 0) w = f32x4{0.f, 0.f, 0.f, 0.f}; }   // (the rows of R enter the sum once)
#pragma unroll
	for (int rt = 0; rt < 4; rt++)
		if (!TRI || rt <= glim) {                        // V is zero in the row tiles below the panel of a triangular block
#pragma unroll
			for (int r = 0; r < 4; r++) w = __builtin_amdgcn_mfma_f32_16x16x4f32(v[4 * rt + r], pj[4 * rt + r], w, 0, 0, 0);
		}
	w = coop_sum4<COOP>(w, cc, slot);
	f32x4 wp = {0.f, 0.f, 0.f, 0.f};                     // W' = T^T W   (k-slot (q, r) <-> panel column 4q+r)
#pragma unroll
	for (int r = 0; r < 4; r++) wp = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[r], w[r], wp, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 4; r++) Rp[idx[r]] = w0[r] - wp[r];
#pragma unroll
	for (int rt = 0; rt < 4; rt++) {                     // B -= V W'
		if (TRI && rt > glim) continue;
		f32x4 acc = {pj[4 * rt], pj[4 * rt + 1], pj[4 * rt + 2], pj[4 * rt + 3]};
#pragma unroll
		for (int r = 0; r < 4; r++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[rt][r], wp[r], acc, 0, 0, 0);
		pj[4 * rt] = acc[0]; pj[4 * rt + 1] = acc[1]; pj[4 * rt + 2] = acc[2]; pj[4 * rt + 3] = acc[3];
	}
}

// The same block reflector with the reference's error-corrected matrix-core arithmetic (fp32_tc_cor: reference
// src/tcqr32x16.cu:669-819 splits H and the updated tile into fp16 hi/lo and adds the correction products first).  Here both
// operands of the two large products are split three ways into bf16 (hi, mid, lo: 24 bits, fp32 exponent range) and multiplied on
// v_mfma_f32_16x16x32_bf16, six products smallest first:
//   W = W0 + V^T B : K = the 64 rows as two K-steps of 32 (registers 8kt .. 8kt+7 of both operands: the row order inside a K-step
//                    is the same permutation on both sides, so the contraction is unaffected);
//   B -= V W'      : K = the 16 panel columns, padded to 32: lane (row, q) contributes its own four columns 4q .. 4q+3 of -V^T
//                    (vt, from the exact fp32-MFMA transposition) in k-slots 0..3 and zeros in 4..7, W' supplies the same slots;
//   W' = T^T W     : 16 x 16 x 16, stays on the exact fp32 MFMA.
// vh/vm/vl[kt]: split of V per K-step; th/tm/tl[rt]: split of the padded -V^T slices (both prepared once per panel).
template <bool TRI, bool COOP = false>
__device__ __forceinline__ void trail_update_cor(float (&pj)[16], const bf16x8 (&vh)[2], const bf16x8 (&vm)[2], const bf16x8 (&vl)[2],
                                                 const bf16x8 (&th)[4], const bf16x8 (&tm)[4], const bf16x8 (&tl)[4], const float (&ta)[4],
                                                 float* __restrict__ Rp, int K0, int colj, int NP, int c, int q, int glim,
                                                 CoopCtx* cc = nullptr, int slot = 0) {
	int idx[4];
	f32x4 w0;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const int k = K0 + 4 * q + r;
		idx[r] = k * NP - (k * (k - 1)) / 2 + (colj + c - k);
		w0[r] = Rp[idx[r]];
	}
	f32x4 w = w0;                                        // W = W0 + V^T B
	if constexpr (COOP) { if (cc->wave != 0) w = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
	for (int kt = 0; kt < 2; kt++)
		if (!TRI || 2 * kt <= glim) {                    // rows 32 kt .. are zero in a triangular block beyond row tile glim
			u32x4 hh, mm, ll;
#pragma unroll
			for (int jp = 0; jp < 4; jp++) {
				unsigned h, m, lo;
				split3_pair(pj[8 * kt + 2 * jp], pj[8 * kt + 2 * jp + 1], h, m, lo);
				hh[jp] = h; mm[jp] = m; ll[jp] = lo;
			}
			const bf16x8 bh = __builtin_bit_cast(bf16x8, hh), bm = __builtin_bit_cast(bf16x8, mm), bl = __builtin_bit_cast(bf16x8, ll);
			w = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vm[kt], bm, w, 0, 0, 0);
			w = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[kt], bl, w, 0, 0, 0);
			w = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl[kt], bh, w, 0, 0, 0);
			w = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[kt], bm, w, 0, 0, 0);
			w = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vm[kt], bh, w, 0, 0, 0);
			w = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh[kt], bh, w, 0, 0, 0);
		}
	w = coop_sum4<COOP>(w, cc, slot);
	f32x4 wp = {0.f, 0.f, 0.f, 0.f};                     // W' = T^T W   (k-slot (q, r) <-> panel column 4q+r), exact fp32
#pragma unroll
	for (int r = 0; r < 4; r++) wp = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[r], w[r], wp, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 4; r++) Rp[idx[r]] = w0[r] - wp[r];
	// B operand of the second product: this lane's four rows of W' in k-slots 0..3, zeros in 4..7
	unsigned h0, m0, l0, h1, m1, l1;
	split3_pair(wp[0], wp[1], h0, m0, l0);
	split3_pair(wp[2], wp[3], h1, m1, l1);
	const bf16x8 wh = __builtin_bit_cast(bf16x8, u32x4{h0, h1, 0u, 0u}), wm = __builtin_bit_cast(bf16x8, u32x4{m0, m1, 0u, 0u}),
	             wl = __builtin_bit_cast(bf16x8, u32x4{l0, l1, 0u, 0u});
#pragma unroll
	for (int rt = 0; rt < 4; rt++) {                     // B -= V W'
		if (TRI && rt > glim) continue;
		f32x4 acc = {pj[4 * rt], pj[4 * rt + 1], pj[4 * rt + 2], pj[4 * rt + 3]};
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tm[rt], wm, acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th[rt], wl, acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tl[rt], wh, acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th[rt], wm, acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tm[rt], wh, acc, 0, 0, 0);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(th[rt], wh, acc, 0, 0, 0);
		pj[4 * rt] = acc[0]; pj[4 * rt + 1] = acc[1]; pj[4 * rt + 2] = acc[2]; pj[4 * rt + 3] = acc[3];
	}
}

// One TPQRT fold of the chunk in p (64 rows x NP columns, (c,q) layout) into the packed upper-triangular R of this wave (Rw, LDS):
// R <- R-factor of [R ; chunk].  TRI: the chunk is itself a 64-row upper-triangular block (panel S is zero below row tile S).
// COR: fp32_tc_cor -- the block reflector is applied with the bf16x3 error-corrected MFMA products (trail_update_cor), otherwise
// with exact fp32 MFMA (fp32_notc).  p is consumed (the register tiles rotate).
template <int NT, bool TRI, bool COR, bool COOP = false>
__device__ __forceinline__ void fold_one(float (&p)[NT][16], float* __restrict__ Rw, float* __restrict__ Tl, int n, int c, int q,
                                         CoopCtx* cc = nullptr) {
	constexpr int NP = 16 * NT;
	const int ntile = (n + 15) >> 4;                     // tiles that carry data
#pragma unroll 1
	for (int S = 0; S < ntile; S++) {
		const int K0 = 16 * S;
		const int glim = TRI ? S : 3;                    // triangular source block: panel S is zero below row tile S
		float Trow[16];
#pragma unroll
		for (int l = 0; l < 16; l++) Trow[l] = 0.0f;
		float sc = 1.0f;
		int offK = K0 * NP - (K0 * (K0 - 1)) / 2;        // packed offset of entry (K0, K0)
		float rkk = Rw[offK], rkc = Rw[offK + c];        // row K0, prefetched
		static_for<0, 16>([&](auto kk) {
			constexpr int KK = decltype(kk)::value;
			const int offN = offK + NP - (K0 + KK);      // row K+1
			float rkk_n = 0.0f, rkc_n = 0.0f;
			if (KK < 15) { rkk_n = Rw[offN]; rkc_n = Rw[offN + c - (KK + 1)]; }   // not touched by step KK
			if (K0 + KK < n) panel_step<KK, TRI, COOP>(p[0], Trow, sc, Rw + offK, c, rkk, rkc, glim, cc);
			offK = offN; rkk = rkk_n; rkc = rkc_n;
		});
		const int ntrail = ntile - 1 - S;
		if (ntrail > 0) {
#pragma unroll
			for (int r = 0; r < 16; r++) p[0][r] *= sc;                  // V of the panel
			if (q == 0) {
#pragma unroll
				for (int l = 0; l < 16; l += 4) {
					f32x4 tv = {Trow[l], Trow[l + 1], Trow[l + 2], Trow[l + 3]};
					*reinterpret_cast<f32x4*>(&Tl[c * 16 + l]) = tv;
				}
			}
			__builtin_amdgcn_wave_barrier();
			float ta[4];
#pragma unroll
			for (int r = 0; r < 4; r++) ta[r] = Tl[(4 * q + r) * 16 + c];
			f32x4 vt[4];                                      // -V^T per 16-row tile via MFMA against identity slices
#pragma unroll
			for (int rt = 0; rt < 4; rt++) {
				f32x4 t = {0.f, 0.f, 0.f, 0.f};
				if (!TRI || rt <= glim) {
#pragma unroll
					for (int r = 0; r < 4; r++)
						t = __builtin_amdgcn_mfma_f32_16x16x4f32(p[0][4 * rt + r], (c == 4 * q + r) ? -1.0f : 0.0f, t, 0, 0, 0);
				}
				vt[rt] = t;
			}
			if constexpr (COR) {
				bf16x8 vh[2], vm[2], vl[2], th[4], tm[4], tl[4];
#pragma unroll
				for (int kt = 0; kt < 2; kt++) {
					u32x4 hh, mm, ll;
#pragma unroll
					for (int jp = 0; jp < 4; jp++) {
						unsigned h, m, lo;
						split3_pair(p[0][8 * kt + 2 * jp], p[0][8 * kt + 2 * jp + 1], h, m, lo);
						hh[jp] = h; mm[jp] = m; ll[jp] = lo;
					}
					vh[kt] = __builtin_bit_cast(bf16x8, hh); vm[kt] = __builtin_bit_cast(bf16x8, mm); vl[kt] = __builtin_bit_cast(bf16x8, ll);
				}
#pragma unroll
				for (int rt = 0; rt < 4; rt++) {
					unsigned h0, m0, l0, h1, m1, l1;
					split3_pair(vt[rt][0], vt[rt][1], h0, m0, l0);
					split3_pair(vt[rt][2], vt[rt][3], h1, m1, l1);
					th[rt] = __builtin_bit_cast(bf16x8, u32x4{h0, h1, 0u, 0u});
					tm[rt] = __builtin_bit_cast(bf16x8, u32x4{m0, m1, 0u, 0u});
					tl[rt] = __builtin_bit_cast(bf16x8, u32x4{l0, l1, 0u, 0u});
				}
				static_for<1, NT>([&](auto jj) {
					constexpr int J = decltype(jj)::value;
					if (J <= ntrail) trail_update_cor<TRI, COOP>(p[J], vh, vm, vl, th, tm, tl, ta, Rw, K0, K0 + 16 * J, NP, c, q, glim, cc, J - 1);
				});
			} else {
				static_for<1, NT>([&](auto jj) {
					constexpr int J = decltype(jj)::value;
					if (J <= ntrail) trail_update<TRI, COOP>(p[J], p[0], vt, ta, Rw, K0, K0 + 16 * J, NP, c, q, glim, cc, J - 1);
				});
			}
			__builtin_amdgcn_wave_barrier();
		}
		// rotate the tiles: the next panel becomes p[0]
		static_for<0, NT - 1>([&](auto jj) {
			constexpr int J = decltype(jj)::value;
#pragma unroll
			for (int r = 0; r < 16; r++) p[J][r] = p[J + 1][r];
		});
	}
}

// write the packed R of a wave: lane <-> row, four columns per iteration so the LDS reads overlap
template <int NP>
__device__ __forceinline__ void store_packed_r(float* __restrict__ dst, size_t dst_ld, const float* __restrict__ Rw, int rows_store, int cols_store, int lane) {
	if (lane < rows_store) {
		const int off = lane * NP - (lane * (lane - 1)) / 2;
		for (int col0 = 0; col0 < cols_store; col0 += 4) {
			float v[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int col = col0 + u;
				v[u] = (lane <= col && lane < NP && col < NP) ? Rw[off + col - lane] : 0.0f;
			}
#pragma unroll
			for (int u = 0; u < 4; u++)
				if (col0 + u < cols_store) __builtin_nontemporal_store(v[u], &dst[(size_t)(col0 + u) * dst_ld + lane]);   // (keeps A in the Infinity Cache)
		}
	}
}

// TRI: the folded blocks are 64-row upper-triangular R factors (tree levels with NT == 4): the wave's first block is copied
// into R instead of folded and every panel skips the row tiles that are structurally zero.
template <int NT, bool TRI = false, bool COR = false>
__global__ __launch_bounds__(256, 2) void fold_kernel(const FoldArgs a) {
	constexpr int NP = 16 * NT;
	constexpr int RP = (NP * (NP + 1)) / 2 + 16;         // packed upper triangle (+ slack for masked reads)
	__shared__ float Rs[4][RP];
	__shared__ float Ts[4][256];
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int gw = blockIdx.x * 4 + wv;
	if (gw >= a.nwaves) return;                          // whole waves leave; no barrier in this kernel
	const int c = lane & 15, q = lane >> 4;
	float* Rw = Rs[wv];
	float* Tl = Ts[wv];
	for (int i = lane; i < RP; i += 64) Rw[i] = 0.0f;

	float p[NT][16];
	const int ch_end = min(a.nchunks, (gw + 1) * a.cpw);
	int ch_first = gw * a.cpw;
	if (TRI && ch_first < ch_end) {
		// the first block is the initial R: load it like any chunk (16 wide loads in flight) and scatter its upper triangle
		// into the packed rows -- register (ct, 4rt+i) of lane (c,q) is entry (row 16rt+4q+i, column 16ct+c)
		load_chunk<NT>(p, a.src, a.ld, (size_t)ch_first * 64, a.m, a.n, c, q);
#pragma unroll
		for (int ct = 0; ct < NT; ct++)
#pragma unroll
			for (int rho = 0; rho < 16; rho++) {
				const int row = 16 * (rho >> 2) + 4 * q + (rho & 3), col = 16 * ct + c;
				if (col >= row) Rw[row * NP - (row * (row - 1)) / 2 + col - row] = p[ct][rho];
			}
		ch_first++;
		__builtin_amdgcn_wave_barrier();
	}
	for (int ch = ch_first; ch < ch_end; ch++) {
		load_chunk<NT>(p, a.src, a.ld, (size_t)ch * 64, a.m, a.n, c, q);
		fold_one<NT, TRI, COR>(p, Rw, Tl, a.n, c, q);
	}
	store_packed_r<NP>(a.dst + (size_t)gw * a.rows_store, a.dst_ld, Rw, a.rows_store, a.cols_store, lane);
}

// ---------------------------------------------------------------------------------------------
// fold_coop_kernel (round 3): the binary R-stack reduction of the reference (src/tsqr.cu:1121-1172: one launch per level, 2 n x n
// tiles each) as COOPERATIVE folds.  A binary tree over 2048 level-0 factors is eleven dependent 64-step reflector chains of ~27 us
// each whoever launches them (round 2's fold_tree_kernel: 320-350 us); here a workgroup of eight waves folds EIGHT stacked
// triangular 64 x 64 blocks in ONE chain -- block 0 is the running R (every wave keeps an identical copy in LDS), block w is wave
// w's chunk, and what a reflector needs from all blocks travels through LDS once per step (CoopCtx) -- so 2048 factors need four
// chains (2048 -> 256 -> 32 -> 4 -> 1, ~50 us each: 195-205 us).  NT == 4 (49 <= n <= 64) only; narrower panels keep the
// per-level launches of fold_kernel.
// ---------------------------------------------------------------------------------------------
struct FoldTreeArgs {
	const float* src; size_t ld;        // stack of nblocks upper-triangular 64 x 64 blocks: block b = rows 64 b .. 64 b + 63, column-major, leading dimension ld
	int nblocks, n;
	float* dst; size_t dst_ld;          // workgroup g writes its R at rows g * rows_store of dst
	int rows_store, cols_store;
};
constexpr int FOLD_COOP_LDS_FLOATS = FOLD_COOP_WAVES * ((64 * 65) / 2 + 16) + FOLD_COOP_WAVES * 256 + 2 * FOLD_COOP_WAVES * 64 + 3 * FOLD_COOP_WAVES * 256;
template <bool COR>
__global__ __launch_bounds__(64 * FOLD_COOP_WAVES) void fold_coop_kernel(const FoldTreeArgs a) {
	constexpr int NT = 4, NP = 64, WAVES = FOLD_COOP_WAVES;
	constexpr int RP = (NP * (NP + 1)) / 2 + 16;
	extern __shared__ __attribute__((aligned(16))) char coop_smem[];     // Rs[WAVES][RP] | Ts[WAVES][256] | xd[2][WAVES][64] | xw[3][WAVES][256]
	float* Rs = reinterpret_cast<float*>(coop_smem);
	float* Ts = Rs + WAVES * RP;
	CoopCtx cc;
	cc.xd = Ts + WAVES * 256;
	cc.xw = cc.xd + 2 * WAVES * 64;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	cc.wave = wv; cc.step = 0;
	const int c = lane & 15, q = lane >> 4;
	float* Rw = Rs + wv * RP;
	float* Tl = Ts + wv * 256;
	for (int i = lane; i < RP; i += 64) Rw[i] = 0.0f;
	const int first = blockIdx.x * WAVES;                // this workgroup's blocks: first .. first + WAVES - 1 (those that exist)
	const size_t mrows = (size_t)a.nblocks * 64;
	float p[NT][16];
	// block `first` is the initial R of every wave: loaded like any chunk and scattered into the packed rows
	load_chunk<NT, true>(p, a.src, a.ld, (size_t)first * 64, mrows, a.n, c, q);
#pragma unroll
	for (int ct = 0; ct < NT; ct++)
#pragma unroll
		for (int rho = 0; rho < 16; rho++) {
			const int row = 16 * (rho >> 2) + 4 * q + (rho & 3), col = 16 * ct + c;
			if (col >= row) Rw[row * NP - (row * (row - 1)) / 2 + col - row] = p[ct][rho];
		}
	// wave w > 0 folds block first + w; wave 0 (and a wave whose block does not exist) contributes a zero chunk
	if (wv > 0 && first + wv < a.nblocks) {
		load_chunk<NT, true>(p, a.src, a.ld, (size_t)(first + wv) * 64, mrows, a.n, c, q);
	} else {
#pragma unroll
		for (int ct = 0; ct < NT; ct++)
#pragma unroll
			for (int rho = 0; rho < 16; rho++) p[ct][rho] = 0.0f;
	}
	__builtin_amdgcn_wave_barrier();
	fold_one<NT, true, COR, true>(p, Rw, Tl, a.n, c, q, &cc);
	if (wv == 0) store_packed_r<NP>(a.dst + (size_t)blockIdx.x * a.rows_store, a.dst_ld, Rw, a.rows_store, a.cols_store, lane);
}

// ---------------------------------------------------------------------------------------------
// Gram engine (R-factor engine of fp32_tc_cor):  G = A^T A on the fp64 matrix cores, R = chol(G) in fp64.
// Products of fp32 inputs are exact in fp64 and the accumulation is fp64, so R is as backward-accurate as an fp32
// Householder R while cond(A)^2 * 2^-53 * n << 1; chol16_kernel reports breakdown and the host falls back to fold_kernel.
//   gram_kernel        : every wave accumulates the upper-triangular 16x16 tiles of its strip's Gram matrix with
//                        v_mfma_f64_16x16x4_f64 (both operands straight from the (c,q) registers), the four waves of a
//                        workgroup are summed through LDS, one partial per workgroup goes to HBM.
//   gram_reduce_kernel : partials -> NSPLIT sub-sums per entry (fixed order: deterministic).
//   chol16_kernel      : sub-sums -> G (LDS, fp64) -> R (fp32, user layout), Z = inverse(R) (fp32, NP x NP), status.
// ---------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

// Per-workgroup partial sums are written once and read once a few microseconds later.  Written with plain stores they displace
// about as many bytes of A from the Infinity Cache, and the apply pass then runs 15 us slower at 2^20 x 64 (tools/mix_bench.py):
// nontemporal on both sides keeps them out.
__device__ __forceinline__ void part_store(double* p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ double part_load(const double* p) { return __builtin_nontemporal_load(p); }

struct GramArgs {
	const float* a; size_t lda; size_t m; int n;
	int nchunks; int cpw; int nwaves;
	double* part;                        // [gridDim.x][NTRI][256]  (tile, lane, reg) order of the MFMA accumulators
	const unsigned* skip_status;         // optional: return at once when *skip_status != 0 (an earlier, speculatively enqueued
	                                     // sweep this pass depends on was rejected; its successor Cholesky reports "rejected" too)
	unsigned* announce; unsigned announce_seq;   // optional: completion word (pinned host memory) of the CALL IN FRONT of this one in the
	                                     // stream and the value to raise it to
	int old_share;                       // 0 (= GB_OLD_SHARE); tools/gram_balance.py: share of a CU's blocks (in 32nds) for its first workgroup
};

// A stream of calls (tsqr_mi_qr_f32_loop): the first kernel of call i + 1 starts when the last kernel of call i has finished (stream
// order), so it can raise call i's completion word itself -- a one-thread kernel (4 us of launch ramp) less per call.
__device__ __forceinline__ void announce_previous_call(unsigned* word, unsigned seq) {
	if (word && blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<volatile unsigned*>(word) = seq;
}

template <int NT>
__global__ __launch_bounds__(256) void gram_kernel(const GramArgs a) {
	constexpr int NTRI = (NT * (NT + 1)) / 2;
	__shared__ double red[2][NTRI * 256];
	announce_previous_call(a.announce, a.announce_seq);
	if (a.skip_status && a.skip_status[0] != 0) return;
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int gw = blockIdx.x * 4 + wv;
	const int c = lane & 15, q = lane >> 4;
	f64x4 acc[NTRI];
#pragma unroll
	for (int t = 0; t < NTRI; t++) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
	if (gw < a.nwaves) {
		float p[NT][16];
		const int ch_end = a.nchunks, ch_step = a.nwaves, ch_begin = gw;      // interleaved: consecutive chunks go to consecutive waves
		for (int ch = ch_begin; ch < ch_end; ch += ch_step) {
			load_chunk<NT>(p, a.a, a.lda, (size_t)ch * 64, a.m, a.n, c, q);
#pragma unroll
			for (int rho = 0; rho < 16; rho++) {
				double pd[NT];
#pragma unroll
				for (int t = 0; t < NT; t++) pd[t] = (double)p[t][rho];
				int idx = 0;
#pragma unroll
				for (int ti = 0; ti < NT; ti++)
#pragma unroll
					for (int tj = ti; tj < NT; tj++) {
						acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(pd[ti], pd[tj], acc[idx], 0, 0, 0);
						idx++;
					}
			}
		}
	}
	// workgroup sum: waves 2,3 -> LDS, waves 0,1 add; wave 1 -> LDS, wave 0 adds and stores the partial
	if (wv >= 2) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[wv - 2][(t * 4 + r) * 64 + lane] = acc[t][r];
	}
	__syncthreads();
	if (wv < 2) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) acc[t][r] += red[wv][(t * 4 + r) * 64 + lane];
	}
	__syncthreads();
	if (wv == 1) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[0][(t * 4 + r) * 64 + lane] = acc[t][r];
	}
	__syncthreads();
	if (wv == 0) {
		double* out = a.part + (size_t)blockIdx.x * NTRI * 256;
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) part_store(&out[(t * 4 + r) * 64 + lane], acc[t][r] + red[0][(t * 4 + r) * 64 + lane]);
	}
}

// gram_bf16_kernel: the same Gram tiles on v_mfma_f32_16x16x32_bf16 with the 3-way bf16 split of both operands
// (six exact-product terms per tile).  Every K-step (32 rows) is one MFMA chain that starts from zero; its fp32 result is
// added to fp64 totals on the vector units, so the only fp32 roundings are those inside one chain and they average out over
// the K-steps (2^20 rows: U(0,1) input 5.5e-7 instead of 9.2e-7, U(0,1)+10 1.5e-5 instead of 2e-4 with fp32 totals).
// The host accepts the result only when the Cholesky pivots show nearly orthogonal columns (chol16_kernel thresholds).
// Partials use the f32 MFMA C/D layout (row = 4*(lane>>4) + reg).
// Measured alternatives (round 2, 2^20 x 64; tools/seq_bench.py, git history): a workgroup LDS-DMA ring (61 us: sharing a block
// between waves duplicates the split), a per-wave LDS-DMA bounce with full-line requests and a prefetched next chunk (54.6 us)
// against 53.6 us here: the pass is bound by vector + matrix issue at the clock the chip holds (~1.6 GHz), not by its requests.
// (Round 3, VERDICT r02 item 1e: one chain per 64-row chunk -- 12 products, half the fp64 flushes -- changes neither the accuracy
// (2^23 rows, same-sign inputs included: tools/gram_flush_accuracy.py in git history, profiles/r03_experiment_log.md) nor the call's
// time (0.1629 / 0.1629 vs 0.1629 / 0.1623 ms): left as it was.)
template <int NT>
__global__ __launch_bounds__(256) void gram_bf16_kernel(const GramArgs a) {
	constexpr int NTRI = (NT * (NT + 1)) / 2;
	__shared__ double red[2][NTRI * 256];
	announce_previous_call(a.announce, a.announce_seq);
	if (a.skip_status && a.skip_status[0] != 0) return;
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int gw = blockIdx.x * 4 + wv;
	const int c = lane & 15, q = lane >> 4;
	// the MFMA's own fp32 accumulation is biased (measured: ||Q^T Q - I|| grows with the accumulation length), so the
	// length of an MFMA accumulation chain must not depend on m: one K-step per chain, fp64 from there on
	f32x4 acc[NTRI];
	f64x4 tot[NTRI];
#pragma unroll
	for (int t = 0; t < NTRI; t++) {
		acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
		tot[t] = f64x4{0.0, 0.0, 0.0, 0.0};
	}
	if (gw < a.nwaves) {
		float p[NT][16];
		const int ch_end = a.nchunks, ch_step = a.nwaves, ch_begin = gw;
		for (int ch = ch_begin; ch < ch_end; ch += ch_step) {
			load_chunk<NT>(p, a.a, a.lda, (size_t)ch * 64, a.m, a.n, c, q);
#pragma unroll
			for (int kt = 0; kt < 2; kt++) {             // K-step of 32 rows: registers 8kt .. 8kt+7 of every lane
				bf16x8 oh[NT], om[NT], ol[NT];
#pragma unroll
				for (int t = 0; t < NT; t++) {
					u32x4 hh, mm, ll;
#pragma unroll
					for (int jp = 0; jp < 4; jp++) {
						unsigned h, m, lo;
						split3_pair(p[t][8 * kt + 2 * jp], p[t][8 * kt + 2 * jp + 1], h, m, lo);
						hh[jp] = h; mm[jp] = m; ll[jp] = lo;
					}
					oh[t] = __builtin_bit_cast(bf16x8, hh);
					om[t] = __builtin_bit_cast(bf16x8, mm);
					ol[t] = __builtin_bit_cast(bf16x8, ll);
				}
				// six of the nine partial products of (h+m+l)x(h+m+l), smallest first: mm hl lh hm mh hh; consecutive MFMAs hit
				// different accumulators
#pragma unroll
				for (int pass = 3; pass < 9; pass++) {
					int idx = 0;
#pragma unroll
					for (int ti = 0; ti < NT; ti++)
#pragma unroll
						for (int tj = ti; tj < NT; tj++) {
							const bf16x8 av = (pass == 4 || pass == 6 || pass == 8) ? oh[ti] : ((pass == 3 || pass == 7) ? om[ti] : ol[ti]);
							const bf16x8 bv = (pass == 5 || pass == 7 || pass == 8) ? oh[tj] : ((pass == 3 || pass == 6) ? om[tj] : ol[tj]);
							acc[idx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[idx], 0, 0, 0);
							idx++;
						}
				}
#pragma unroll
				for (int t = 0; t < NTRI; t++) {
#pragma unroll
					for (int r = 0; r < 4; r++) tot[t][r] += (double)acc[t][r];
					acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
				}
			}
		}
	}
	// workgroup sum in fp64: waves 2,3 -> LDS, waves 0,1 add; wave 1 -> LDS, wave 0 adds and stores the partial
	double dacc[NTRI][4];
#pragma unroll
	for (int t = 0; t < NTRI; t++)
#pragma unroll
		for (int r = 0; r < 4; r++) dacc[t][r] = (double)tot[t][r];
	if (wv >= 2) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[wv - 2][(t * 4 + r) * 64 + lane] = dacc[t][r];
	}
	__syncthreads();
	if (wv < 2) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) dacc[t][r] += red[wv][(t * 4 + r) * 64 + lane];
	}
	__syncthreads();
	if (wv == 1) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[0][(t * 4 + r) * 64 + lane] = dacc[t][r];
	}
	__syncthreads();
	if (wv == 0) {
		double* out = a.part + (size_t)blockIdx.x * NTRI * 256;
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) part_store(&out[(t * 4 + r) * 64 + lane], dacc[t][r] + red[0][(t * 4 + r) * 64 + lane]);
	}
}

// gram_h_kernel (round 3, fp16 I/O modes): the Gram tiles of an fp16 matrix straight from its halves -- lane (c,q) loads rows
// 8q .. 8q+7 of a 32-row K-step of column 16t+c as ONE 16-byte access, which IS the operand of v_mfma_f32_16x16x32_f16: no
// widening pass, no split (products of fp16 values are exact in the fp32 accumulator), ten MFMAs per K-step instead of sixty, half
// the bytes of the fp32 pass.  One MFMA chain per 64-row chunk (64 accumulations per entry: far inside the chain lengths validated
// for the bf16-split pass), fp64 totals; the next chunk's operands are requested before the current chunk's MFMAs.  Partials in the
// format of gram_bf16_kernel.  The host sends lda % 8 == 0 and a 16-byte aligned base only (tsqr_mi_qr_f16 otherwise converts).
// (the body takes its workgroup number as an argument: gram_h_chain_kernel runs it on a part of its grid; red: [2][NTRI * 256] doubles of LDS)
template <int NT>
__device__ __forceinline__ void gram_h_body(const GramArgs& a, double (*red)[((NT * (NT + 1)) / 2) * 256], const int wg) {
	constexpr int NTRI = (NT * (NT + 1)) / 2;
	if (a.skip_status && a.skip_status[0] != 0) return;
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int gw = wg * 4 + wv;
	const int c = lane & 15, q = lane >> 4;
	const _Float16* A = reinterpret_cast<const _Float16*>(a.a);
	f32x4 acc[NTRI];
	f64x4 tot[NTRI];
#pragma unroll
	for (int t = 0; t < NTRI; t++) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; tot[t] = f64x4{0.0, 0.0, 0.0, 0.0}; }
	auto load_chunk_h = [&](f16x8 (&op)[NT][2], int ch) {
#pragma unroll
		for (int t = 0; t < NT; t++)
#pragma unroll
			for (int kt = 0; kt < 2; kt++) {
				const size_t row = (size_t)ch * 64 + 32 * kt + 8 * q;
				const int col = 16 * t + c;
				f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
				if (col < a.n) {
					const _Float16* src = A + (size_t)col * a.lda + row;
					if (row + 7 < a.m) v = *reinterpret_cast<const f16x8*>(src);
					else {
#pragma unroll
						for (int i = 0; i < 8; i++)
							if (row + i < a.m) v[i] = src[i];
					}
				}
				op[t][kt] = v;
			}
	};
	if (gw < a.nwaves) {
		f16x8 cur[NT][2], nxt[NT][2];
		int ch = gw;
		if (ch < a.nchunks) load_chunk_h(cur, ch);
		for (; ch < a.nchunks; ch += a.nwaves) {
			const bool more = ch + a.nwaves < a.nchunks;     // wave-uniform
			if (more) load_chunk_h(nxt, ch + a.nwaves);
#pragma unroll
			for (int kt = 0; kt < 2; kt++) {
				int idx = 0;
#pragma unroll
				for (int ti = 0; ti < NT; ti++)
#pragma unroll
					for (int tj = ti; tj < NT; tj++) {
						acc[idx] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[ti][kt], cur[tj][kt], acc[idx], 0, 0, 0);
						idx++;
					}
			}
#pragma unroll
			for (int t = 0; t < NTRI; t++) {
#pragma unroll
				for (int r = 0; r < 4; r++) tot[t][r] += (double)acc[t][r];
				acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
			}
			if (more) {
#pragma unroll
				for (int t = 0; t < NT; t++) { cur[t][0] = nxt[t][0]; cur[t][1] = nxt[t][1]; }
			}
		}
	}
	// workgroup sum in fp64: waves 2,3 -> LDS, waves 0,1 add; wave 1 -> LDS, wave 0 adds and stores the partial
	if (wv >= 2) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[wv - 2][(t * 4 + r) * 64 + lane] = tot[t][r];
	}
	__syncthreads();
	if (wv < 2) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) tot[t][r] += red[wv][(t * 4 + r) * 64 + lane];
	}
	__syncthreads();
	if (wv == 1) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[0][(t * 4 + r) * 64 + lane] = tot[t][r];
	}
	__syncthreads();
	if (wv == 0) {
		double* out = a.part + (size_t)wg * NTRI * 256;
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) part_store(&out[(t * 4 + r) * 64 + lane], tot[t][r] + red[0][(t * 4 + r) * 64 + lane]);
	}
}
template <int NT>
__global__ __launch_bounds__(256) void gram_h_kernel(const GramArgs a) {
	constexpr int NTRI = (NT * (NT + 1)) / 2;
	__shared__ double red[2][NTRI * 256];
	announce_previous_call(a.announce, a.announce_seq);
	gram_h_body<NT>(a, red, blockIdx.x);
}

// gram_blk_kernel (round 3): the same Gram tiles for FULL 64-column matrices, with the block pattern of the apply pass on the load
// side.  gram_bf16_kernel's (c,q) register layout feeds the MFMAs without any exchange, but an instruction of it can only ask for
// 64 contiguous bytes per column (sixteen columns, four lanes each): its load-only skeleton takes 47.6 us at 2^20 x 64 where the
// block pattern (512 contiguous bytes per column and instruction) takes 40 (tools/seq_bench.py).  Here a workgroup loads 128-row x
// 64-column blocks in the block pattern, stages them as fp32 in LDS (double buffered: ONE LDS-only barrier per block), and wave w
// takes the K-step of rows 32 w .. 32 w + 31 for ALL ten tile pairs: two ds_read_b128 per column tile (column stride 136 floats:
// the eight lanes of a pass fall on different banks), the bf16 split of every element exactly once (by the wave that consumes it),
// 60 MFMAs.  Two blocks per workgroup in flight in registers, rotated by unrolling (see gram_wide_kernel).
// Totals as in gram_bf16_kernel: one MFMA chain per K-step, fp64 from there on; the ten accumulators are handled in two groups of
// five (the first group is flushed while the second group's MFMAs run: twenty accumulator registers instead of forty -- the kernel
// sits at 253 registers, two waves per SIMD).  (fp32 round-to-nearest wave totals, with fp64 only from the workgroup sum on, save
// 40 registers and a third of the vector instructions, do not change the call time, and cost accuracy on same-sign inputs as the
// rows per wave grow: U(0,1)+0.25 gives 9.5e-7 / 1.9e-6 / 2.4e-6 at 2^20 / 2^23 / 2^24 rows against 7.8e-7 / 9.5e-7 with fp64
// totals -- tools/gram_accuracy.py, profiles/r03_experiment_log.md.)
// Partials in the same format as gram_bf16_kernel.
constexpr int GB_ROWS = 128, GB_RS = 136;
constexpr int GB_OLD_SHARE = 20;                       // of 32: the share of a CU's blocks its first (older) workgroup takes (gram_blk_body)
constexpr int GB_LDS_BYTES = 2 * 64 * GB_RS * 4;
// (the body takes its workgroup number and the number of Gram workgroups as arguments: gram_blk_chain_kernel below runs it on a part of
// its grid)
__device__ __forceinline__ void gram_blk_body(const GramArgs& a, float* gb_as, const int wg, const int nwg) {
	constexpr int NTRI = 10;
	static_assert(GB_LDS_BYTES >= (int)sizeof(double) * 2 * NTRI * 256, "the final workgroup reduction aliases the block buffers");
	if (a.skip_status && a.skip_status[0] != 0) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int c = lane & 15, q = lane >> 4;
	const int lcol = lane >> 5, lrow = 4 * (lane & 31);
	const int nblk = a.nchunks;                          // 128-row blocks (the host sends only m % 128 == 0, n == 64, 16-byte aligned columns, lda < 2^23)
	// Which blocks are this workgroup's.  The grid is twice the CUs and the dispatcher fills every CU once, then again: workgroups w and
	// w + nwg / 2 share a CU, and the instruction arbiter prefers the older one -- with equal shares the first workgroup of a CU was
	// done at 35 us, the second at 44.5, its last 9 us alone on a CU that needs two workgroups to hide its latencies (stamps:
	// tools/gram_balance.py).  So the blocks of a pair p, p + half, p + 2 half, ... are split unevenly: the first GB_OLD_SHARE / 32 of
	// them to the older workgroup, the rest to the younger.  A fixed function of (w, nwg, blocks): the partials are the same
	// whichever launch runs this body.
	int first = wg, step = nwg, bend = nblk;
	if ((nwg & 1) == 0 && nblk >= 2 * nwg) {
		const int half = nwg >> 1, p = wg < half ? wg : wg - half;
		const int K = (nblk - p + half - 1) / half;
		const int H = (K * (a.old_share ? a.old_share : GB_OLD_SHARE) + 16) >> 5;
		step = half;
		first = wg < half ? p : p + half * H;
		bend = wg < half ? p + half * H : p + half * K;
	}
	// this workgroup's last block: where a look-ahead load past its end goes (a workgroup without blocks loads some valid block and uses nothing)
	const int blast = first < bend ? first + ((bend - 1 - first) / step) * step : min(first, nblk - 1);
	f64x4 tot[NTRI];
#pragma unroll
	for (int t = 0; t < NTRI; t++) tot[t] = f64x4{0.0, 0.0, 0.0, 0.0};
	// buffer loads: descriptor on the block (wave-uniform), ONE loop-invariant 32-bit per-thread offset, the column group in the
	// scalar offset -- no 64-bit address arithmetic in vector registers.  (With flat loads the allocator took the eight address pairs
	// of a set from registers of the set in flight and the waits it then has to insert drained the queue once per block.)
	const unsigned voff = (unsigned)(((size_t)lcol * a.lda + lrow) * sizeof(float));
	const unsigned soff0 = (unsigned)((size_t)(2 * wv) * a.lda * sizeof(float)), soffk = (unsigned)(8 * a.lda * sizeof(float));
	auto load_block = [&](f32x4 (&v)[8], int b) {
		// (no branch at all: past the end the last block is simply loaded again and never used)
		const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.a + (size_t)min(b, blast) * GB_ROWS), 0, -1, 0x00020000);
#pragma unroll
		for (int k = 0; k < 8; k++) v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff0 + k * soffk, 0));
	};
	auto stage = [&](const f32x4 (&v)[8], float* buf) {
#pragma unroll
		for (int k = 0; k < 8; k++) *reinterpret_cast<f32x4*>(&buf[(2 * wv + lcol + 8 * k) * GB_RS + lrow]) = v[k];
	};
	auto products = [&](const float* buf) {
		bf16x8 oh[4], om[4], ol[4];
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const float* src = &buf[(16 * t + c) * GB_RS + 32 * wv + 8 * q];
			const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
			u32x4 hh, mm, ll;
			unsigned h, m, lo;
			split3_pair(x0[0], x0[1], h, m, lo); hh[0] = h; mm[0] = m; ll[0] = lo;
			split3_pair(x0[2], x0[3], h, m, lo); hh[1] = h; mm[1] = m; ll[1] = lo;
			split3_pair(x1[0], x1[1], h, m, lo); hh[2] = h; mm[2] = m; ll[2] = lo;
			split3_pair(x1[2], x1[3], h, m, lo); hh[3] = h; mm[3] = m; ll[3] = lo;
			oh[t] = __builtin_bit_cast(bf16x8, hh);
			om[t] = __builtin_bit_cast(bf16x8, mm);
			ol[t] = __builtin_bit_cast(bf16x8, ll);
		}
		// tile pairs in the order of the partials, in two groups of five accumulators: (0,0) (0,1) (0,2) (0,3) (1,1) | (1,2) (1,3)
		// (2,2) (2,3) (3,3); per pair six of the nine partial products, smallest first: mm hl lh hm mh hh (as gram_bf16_kernel).
		// (A software-pipelined order -- split of tile t+1 interleaved with the MFMAs of the pairs that need only tiles <= t; vector
		// instructions do issue in the shadow of MFMAs, tools/issue_overlap.py -- measured the same call time: not kept.)
		static_for<0, 2>([&](auto gg) {
			constexpr int g = decltype(gg)::value;
			f32x4 acc[5];
#pragma unroll
			for (int i = 0; i < 5; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int pass = 3; pass < 9; pass++)
				static_for<0, 5>([&](auto ii) {
					constexpr int idx = 5 * g + decltype(ii)::value;
					constexpr int ti = idx < 4 ? 0 : (idx < 7 ? 1 : (idx < 9 ? 2 : 3));
					constexpr int tj = idx < 4 ? idx : (idx < 7 ? idx - 3 : (idx < 9 ? idx - 5 : 3));
					const bf16x8 av = (pass == 4 || pass == 6 || pass == 8) ? oh[ti] : ((pass == 3 || pass == 7) ? om[ti] : ol[ti]);
					const bf16x8 bv = (pass == 5 || pass == 7 || pass == 8) ? oh[tj] : ((pass == 3 || pass == 6) ? om[tj] : ol[tj]);
					acc[idx - 5 * g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[idx - 5 * g], 0, 0, 0);
				});
#pragma unroll
			for (int i = 0; i < 5; i++)
#pragma unroll
				for (int r = 0; r < 4; r++) tot[5 * g + i][r] += (double)acc[i][r];
		});
	};
	// register sets X, Y and two block buffers; the loop head sits between "stage the next block" and "barrier" (gram_wide_kernel)
	constexpr int BUF = 64 * GB_RS;
	f32x4 vx[8], vy[8];
	int bi = first, it = 0;
	load_block(vx, bi);
	asm volatile("" ::: "memory");
	load_block(vy, bi + step);
	asm volatile("" ::: "memory");
	if (bi < bend) stage(vx, gb_as);
	while (bi < bend) {
		lds_barrier();                                   // block `bi` is staged; every wave is done with the other buffer
		load_block(vx, bi + 2 * step);
		asm volatile("" ::: "memory");
		products(gb_as + (it & 1) * BUF);
		bi += step; it++;
		if (bi >= bend) break;
		stage(vy, gb_as + (it & 1) * BUF);
		lds_barrier();
		load_block(vy, bi + 2 * step);
		asm volatile("" ::: "memory");
		products(gb_as + (it & 1) * BUF);
		bi += step; it++;
		if (bi >= bend) break;
		stage(vx, gb_as + (it & 1) * BUF);
	}
	// workgroup sum in fp64: waves 2,3 -> LDS, waves 0,1 add; wave 1 -> LDS, wave 0 adds and stores the partial
	__syncthreads();
	double* red = reinterpret_cast<double*>(gb_as);      // [2][NTRI * 256]
	if (wv >= 2) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[(wv - 2) * NTRI * 256 + (t * 4 + r) * 64 + lane] = tot[t][r];
	}
	__syncthreads();
	if (wv < 2) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) tot[t][r] += red[wv * NTRI * 256 + (t * 4 + r) * 64 + lane];
	}
	__syncthreads();
	if (wv == 1) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[(t * 4 + r) * 64 + lane] = tot[t][r];
	}
	__syncthreads();
	if (wv == 0) {
		double* out = a.part + (size_t)wg * NTRI * 256;
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) part_store(&out[(t * 4 + r) * 64 + lane], tot[t][r] + red[(t * 4 + r) * 64 + lane]);
	}
}
__global__ __launch_bounds__(256, 2) void gram_blk_kernel(const GramArgs a) {
	extern __shared__ __attribute__((aligned(16))) float gb_as[];        // [buffer][column][GB_RS]
	announce_previous_call(a.announce, a.announce_seq);
	gram_blk_body(a, gb_as, blockIdx.x, gridDim.x);
}

// gram_reduce1_kernel: partials -> G in ONE launch (deterministic: fixed partition, fixed tree).  A workgroup owns 16
// consecutive entries; thread (e = tid & 15, s = tid >> 4) sums the partials b = s, s+16, s+32, ... of entry e with four
// independent accumulators, then the 16 s-sums of every entry are added through LDS in a fixed pairwise tree.
// gout[nelem] receives `rows` (the local row count as a double): a row-partitioned run all-reduces nelem + 1 doubles, and the
// Cholesky kernel then reads the GLOBAL row count for its thresholds from the same payload -- every rank takes the same verdict.
// Coupling tiles (S = Qb^T Ap of a panel pair, 16 tiles): with fin_zneg != nullptr the summed entry also leaves in its final forms --
// -S as the 64 x 64 operand of the update pass (zero beyond column fin_ny) and S as the block of R -- what cross_finish_kernel does
// in a launch of its own when an all-reduce sits between the sum and its use (row-partitioned runs).
template <bool WT = false>                               // WT: the sums leave as device-scope write-through stores (gram_blk_chain_kernel)
__device__ __forceinline__ void gram_reduce1_body(const int blk, double (*red)[17], double* __restrict__ gout, const double* __restrict__ part,
                                                  int nparts, int nelem, double rows, float* __restrict__ fin_r, size_t fin_ldr,
                                                  float* __restrict__ fin_zneg, int fin_ny) {
	if (blk == 0 && threadIdx.x == 0) gout[nelem] = rows;
	const int e = threadIdx.x & 15, s = threadIdx.x >> 4;
	const int el = blk * 16 + e;
	double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
	if (el < nelem) {
		// the first 512 partials of this thread's stride: U loads issued back to back (one memory round trip instead of eight),
		// then summed in the fixed order s_k += partial(s + 16 (4 i + k)), i ascending.  U = 32 covers 512 partials; launches with few
		// partials (small row counts: 16 partials at 4096 rows) take a copy with fewer loads (round 4: the reduction of the coupling
		// tiles of fifteen trailing panels issued 32 loads per thread where one was needed, 12.6 us per launch) -- the terms a shorter
		// copy leaves out are exact zeros in the longer one, so the sums are the same bits.
		auto head = [&](auto uu) {
			constexpr int U = decltype(uu)::value;
			double v[U];
#pragma unroll
			for (int u = 0; u < U; u++) {
				const int b = s + 16 * u;                    // (unconditional loads from a clamped index: no control flow between them)
				v[u] = part_load(&part[(size_t)min(b, nparts - 1) * nelem + el]);
			}
#pragma unroll
			for (int u = 0; u < U; u++)
				if (s + 16 * u >= nparts) v[u] = 0.0;
#pragma unroll
			for (int u = 0; u < U; u++) {
				if ((u & 3) == 0) s0 += v[u];
				else if ((u & 3) == 1) s1 += v[u];
				else if ((u & 3) == 2) s2 += v[u];
				else s3 += v[u];
			}
		};
		if (nparts <= 16) head(std::integral_constant<int, 1>{});
		else if (nparts <= 32) head(std::integral_constant<int, 2>{});
		else if (nparts <= 64) head(std::integral_constant<int, 4>{});
		else if (nparts <= 128) head(std::integral_constant<int, 8>{});
		else if (nparts <= 256) head(std::integral_constant<int, 16>{});
		else head(std::integral_constant<int, 32>{});
		for (int b = s + 512; b < nparts; b += 16) s0 += part_load(&part[(size_t)b * nelem + el]);
	}
	red[s][e] = (s0 + s1) + (s2 + s3);
	__syncthreads();
	if (s == 0 && el < nelem) {
		double v[16];
#pragma unroll
		for (int k = 0; k < 16; k++) v[k] = red[k][e];
		const double sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) +
		                   (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
		if (WT) __hip_atomic_store(&gout[el], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else gout[el] = sum;
		if (fin_zneg) {                                  // (entry el of the tile array = row i, column j of S: cross_finish_kernel's map)
			const int t = el >> 8, reg = (el >> 6) & 3, l = el & 63;
			const int i = 16 * (t >> 2) + 4 * (l >> 4) + reg, j = 16 * (t & 3) + (l & 15);
			const float f = (float)sum;
			fin_zneg[(size_t)j * 64 + i] = (j < fin_ny) ? -f : 0.0f;
			if (j < fin_ny) fin_r[(size_t)j * fin_ldr + i] = f;
		}
	}
}
__global__ __launch_bounds__(256) void gram_reduce1_kernel(double* __restrict__ gout, const double* __restrict__ part, int nparts, int nelem,
                                                           double rows, float* __restrict__ fin_r = nullptr, size_t fin_ldr = 0,
                                                           float* __restrict__ fin_zneg = nullptr, int fin_ny = 0) {
	__shared__ double red[16][17];
	gram_reduce1_body<false>(blockIdx.x, red, gout, part, nparts, nelem, rows, fin_r, fin_ldr, fin_zneg, fin_ny);
}

// The Cholesky step: sub-sums -> G (fp64) -> R = chol(G) and M = R^-T by the same row operations (forward elimination of
// [R^T | I]), all in fp64; Z = inverse(R) = M^T.  chol_body16 below; chol16_kernel is its launch for the 64-column path,
// chol_wide_kernel (tsqr_wide.hip) runs it twice inside one launch.
__device__ __forceinline__ double bcast_lane_f64(double x, int lane_const) {
	const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
	const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, lane_const);
	const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane_const);
	return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// In-kernel time stamps of the Cholesky step (diagnostic builds only: the selftest library is compiled with -DTSQR_CHOL_STAMPS, the
// product library never is).  Lane 0 of every wave notes the shader clock (s_memtime) in LDS; the kernel dumps the table at its end.
#ifdef TSQR_CHOL_STAMPS
__device__ unsigned long long* g_chol_stamp_out = nullptr;           // [4 waves][CHOL_NSTAMP] + [4] s_memrealtime pairs
constexpr int CHOL_NSTAMP = 160;
__shared__ unsigned long long chol_stamp_lds[4 * CHOL_NSTAMP];
#define CHOL_STAMP16(slot) do { if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 4) chol_stamp_lds[(threadIdx.x >> 6) * CHOL_NSTAMP + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CHOL_STAMP16(slot) do {} while (0)
#endif
typedef double f64x2c __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// chol_section4: the owner's section of a group of four pivots K0 .. K0+3 -- the critical path of the whole step.  The wave holds, per
// lane j, column j of the group's four rows of G (g[0..3]) and of M = R^-T (mm[0..3]).  Rounds 1-3 walked the four pivots one after
// the other, each waiting for lane broadcasts of values the pivot before it had just produced (pivot -> rsq -> scale -> broadcast ->
// fma -> next pivot: ~310 cycles a pivot, in-kernel stamps).  Round 4: the 4 x 4 diagonal block of the group is broadcast ONCE, up front
// (ten lane reads, all independent), and every lane factors it privately -- a chain of four rsq + Newton steps with no cross-lane
// traffic at all -- while the scalings and eliminations of its own column follow in the chain's shadow.  Same operations on the same
// operands as before (the trailing matrix is exactly symmetric, so the private copy l[v][u] IS what lane K0+v computes for R[K0+u][K0+v]):
// R, Z and the verdict are bit for bit those of rounds 1-3.
//   out: rk[u] = R[K0+u][j], mk[u] = M[K0+u][j] (zero for rows >= n; also written to rr / mr, the group's published rows in LDS),
//        piv0[u] = the pivot before clamping (verdict: pv[])
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void chol_section4(double (&g)[4], double (&mm)[4], const int K0, const int j, const int n,
                                              double (&rk)[4], double (&mk)[4], double (&piv0)[4], double* __restrict__ rr, double* __restrict__ mr) {
	// a[u][v] = G'[K0+u][K0+v], v <= u: register u of lane K0+v
	double a00 = bcast_lane_f64(g[0], K0);
	double a10 = bcast_lane_f64(g[1], K0), a11 = bcast_lane_f64(g[1], K0 + 1);
	double a20 = bcast_lane_f64(g[2], K0), a21 = bcast_lane_f64(g[2], K0 + 1), a22 = bcast_lane_f64(g[2], K0 + 2);
	double a30 = bcast_lane_f64(g[3], K0), a31 = bcast_lane_f64(g[3], K0 + 1), a32 = bcast_lane_f64(g[3], K0 + 2), a33 = bcast_lane_f64(g[3], K0 + 3);
	auto inv_sqrt = [](double piv) {
		double y = __builtin_amdgcn_rsq(piv);
		return fma(0.5 * y, fma(-piv * y, y, 1.0), y);       // one Newton step (v_rsq_f64 is good to ~2^-23: 2^-45 after it)
	};
	const bool live0 = K0 < n, live1 = K0 + 1 < n, live2 = K0 + 2 < n, live3 = K0 + 3 < n;
	// Row K of R is g * y right of the diagonal, piv * y on it, zero left of it; row K of M is mm * y.  Every finished row is published
	// (rr / mr: the group's rows in LDS) as soon as it exists: in-kernel stamps showed the LDS writes of a group taking ~500 cycles to be
	// acknowledged behind the other waves' reads when all eight were issued at the end of the section.
	// pivot 0
	piv0[0] = a00;
	const double p0 = (a00 > 0.0) ? a00 : 1.0;               // keeps the arithmetic finite; breakdown is flagged from pv[] afterwards
	const double y0 = inv_sqrt(p0);
	const double l10 = live0 ? a10 * y0 : 0.0, l20 = live0 ? a20 * y0 : 0.0, l30 = live0 ? a30 * y0 : 0.0;
	rk[0] = !live0 ? 0.0 : ((j > K0) ? g[0] * y0 : ((j == K0) ? p0 * y0 : 0.0));
	mk[0] = live0 ? mm[0] * y0 : 0.0;
	rr[0 * 64 + j] = rk[0]; mr[0 * 64 + j] = mk[0];
	a11 = fma(-l10, l10, a11); a21 = fma(-l20, l10, a21); a31 = fma(-l30, l10, a31);
	a22 = fma(-l20, l20, a22); a32 = fma(-l30, l20, a32); a33 = fma(-l30, l30, a33);
	g[1] = fma(-l10, rk[0], g[1]); mm[1] = fma(-l10, mk[0], mm[1]);
	g[2] = fma(-l20, rk[0], g[2]); mm[2] = fma(-l20, mk[0], mm[2]);
	g[3] = fma(-l30, rk[0], g[3]); mm[3] = fma(-l30, mk[0], mm[3]);
	// pivot 1
	piv0[1] = a11;
	const double p1 = (a11 > 0.0) ? a11 : 1.0;
	const double y1 = inv_sqrt(p1);
	const double l21 = live1 ? a21 * y1 : 0.0, l31 = live1 ? a31 * y1 : 0.0;
	rk[1] = !live1 ? 0.0 : ((j > K0 + 1) ? g[1] * y1 : ((j == K0 + 1) ? p1 * y1 : 0.0));
	mk[1] = live1 ? mm[1] * y1 : 0.0;
	rr[1 * 64 + j] = rk[1]; mr[1 * 64 + j] = mk[1];
	a22 = fma(-l21, l21, a22); a32 = fma(-l31, l21, a32); a33 = fma(-l31, l31, a33);
	g[2] = fma(-l21, rk[1], g[2]); mm[2] = fma(-l21, mk[1], mm[2]);
	g[3] = fma(-l31, rk[1], g[3]); mm[3] = fma(-l31, mk[1], mm[3]);
	// pivot 2
	piv0[2] = a22;
	const double p2 = (a22 > 0.0) ? a22 : 1.0;
	const double y2 = inv_sqrt(p2);
	const double l32 = live2 ? a32 * y2 : 0.0;
	rk[2] = !live2 ? 0.0 : ((j > K0 + 2) ? g[2] * y2 : ((j == K0 + 2) ? p2 * y2 : 0.0));
	mk[2] = live2 ? mm[2] * y2 : 0.0;
	rr[2 * 64 + j] = rk[2]; mr[2 * 64 + j] = mk[2];
	a33 = fma(-l32, l32, a33);
	g[3] = fma(-l32, rk[2], g[3]); mm[3] = fma(-l32, mk[2], mm[3]);
	// pivot 3
	piv0[3] = a33;
	const double p3 = (a33 > 0.0) ? a33 : 1.0;
	const double y3 = inv_sqrt(p3);
	rk[3] = !live3 ? 0.0 : ((j > K0 + 3) ? g[3] * y3 : ((j == K0 + 3) ? p3 * y3 : 0.0));
	mk[3] = live3 ? mm[3] * y3 : 0.0;
	rr[3 * 64 + j] = rk[3]; mr[3 * 64 + j] = mk[3];
}

// ---------------------------------------------------------------------------------------------
// chol_body16: the step on SIXTEEN waves -- thread (w = wave, j = lane) holds column j of the four rows 4w .. 4w+3 of G and of
// M = R^-T, i.e. wave w owns exactly one "group" of four pivots.  The owner factors its four rows against each other in registers
// (lane broadcasts, no LDS: v_rsq_f64 + one Newton step per pivot), publishes the four finished rows in LDS (double buffered), and
// after ONE LDS-only barrier the waves behind it apply the rank-4 update to their rows -- the next owner as one straight-line block
// with its own section, so that its update of M fills the latency gaps of the pivot chain.  A wave that is off the path streams the
// finished rows of R and Z to global memory and grows the verdict sum.  Why sixteen waves: in-kernel stamps of the four-wave form of
// rounds 1-2 (sixteen rows per wave; tools/chol_stamps.py, profiles/r03_experiment_log.md) showed a group costing the owner's
// section (~1.25 K cycles) PLUS every wave's rank-4 update of 16 rows (128 fp64 FMAs = ~1.4 K cycles: a lone wave issues one
// v_fma_f64 per ~10 cycles) before the next owner could start; with one group per wave that update is four rows, no rows rotate,
// and the waves share the fp64 pipes four per SIMD: 19.2 -> 17.4 us in the call.
//   LOADG      : functor e -> G tile entry e (accumulator order of the Gram pass: f32_layout 1 = bf16 MFMA, 0 = fp64 MFMA)
//   shift_coef : > 0 = shifted Cholesky (Fukaya et al., "Shifted Cholesky QR for computing the QR factorization of ill-conditioned
//                matrices", SIAM J. Sci. Comput. 2020): G + s I with s = shift_coef * trace(G) >= 11 (mn + n(n+1)) u ||A||_2^2 is
//                safely positive definite for any fp32 input; the caller runs a second (unshifted) sweep on the resulting Q
//   min_diag   : (bf16-split level) a column whose squared norm is so small that its fp32 products live near the denormal range was
//                not accumulated accurately (measured: entries ~1e-22 gave ||Q^T Q - I|| = 1e-2) -> rejected, the fp64 level is exact
//   status[0]  : 0 ok, 1 rejected: a pivot fell below min_ratio of its diagonal entry (2^-40 for the fp64 Gram matrix: cond(A)^2 is
//                beyond fp64 Cholesky; 2^-5 for the bf16-split Gram matrix) or S exceeds max_scond (bf16-split Gram matrix only: its
//                fp32 accumulation is good enough for nearly orthogonal columns only)
//   status[1]  : bit pattern of the smallest pivot ratio (float), status[2]: of S (float), the scaled conditioning
//                S = || D inverse(R) ||_F^2 / n with D = diag(sqrt(g_jj)): 1 for orthogonal columns of any scaling, ~cond^2 of the
//                column-scaled matrix otherwise (an entry-wise error eps sqrt(g_ii g_jj) of G perturbs Q^T Q by <= eps n S)
//   host_status: optional device-visible alias of pinned host memory that receives the three status words as well.  No fence: the
//                host reads them only after the completion word of a LATER kernel on the stream, or after a stream synchronisation
//                (a system-scope release here wrote back the whole L2 on the critical path of every call)
//   gs_out     : receives the address of the fp64 LDS image of Z (chol_wide_kernel goes on with it)
// ---------------------------------------------------------------------------------------------
template <class LOADG>
__device__ __forceinline__ void chol_body16(float* __restrict__ r, size_t ldr, float* __restrict__ z, unsigned* __restrict__ status,
                                            unsigned* __restrict__ host_status, LOADG loadg, int n, int NT, int f32_layout, float min_ratio,
                                            float max_scond, double shift_coef = 0.0, double min_diag = 0.0, double** gs_out = nullptr) {
	__shared__ double Gs[64 * 65];               // symmetric G (assembly); afterwards the fp64 image of Z: Gs[K * 65 + j] = Z[j][K]
	if (gs_out) *gs_out = Gs;                    // (a caller in the same kernel may go on with that image: chol_wide_kernel)
	__shared__ double Rrow[2 * 256], Mrow[2 * 256], dg[64], pv[64], sred[16];
	const int t = threadIdx.x;
	const int j = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
	const int NP = 16 * NT;
	CHOL_STAMP16(0);
#ifdef TSQR_CHOL_STAMPS
	if ((threadIdx.x & 63) == 0 && w < 4) chol_stamp_lds[w * CHOL_NSTAMP + 5] = __builtin_amdgcn_s_memrealtime();
#endif
	// the first four waves assemble G (one value per thread and tile, loads first); the others wait at the barriers
	double gv[10];
	if (t < 256) {
		int idx = 0;
		for (int ti = 0; ti < 4; ti++)
			for (int tj = ti; tj < 4; tj++) {
				if (ti < NT && tj < NT) { gv[ti * 4 + tj - (ti * (ti + 1)) / 2] = loadg(idx * 256 + t); idx++; }
				else gv[ti * 4 + tj - (ti * (ti + 1)) / 2] = 0.0;
			}
	}
	if (NT < 4)
		for (int i = t; i < 64 * 65; i += 1024) Gs[i] = 0.0;
	for (int e = n * NP + t; e < NP * NP; e += 1024) z[e] = 0.0f;         // padding rows of Z
	__syncthreads();
	if (t < 256) {
		const int reg = t >> 6, l = t & 63;
#pragma unroll
		for (int ti = 0; ti < 4; ti++)
#pragma unroll
			for (int tj = ti; tj < 4; tj++) {
				if (ti < NT && tj < NT) {
					const int row = 16 * ti + (f32_layout ? 4 * (l >> 4) + reg : (l >> 4) + 4 * reg);
					const int col = 16 * tj + (l & 15);
					const double v = gv[ti * 4 + tj - (ti * (ti + 1)) / 2];
					// C/D layouts: f64 MFMA row = (lane>>4) + 4*reg, f32/bf16 MFMA row = 4*(lane>>4) + reg; col = lane&15.  A diagonal
					// tile holds (i,j) and (j,i); in the bf16-split Gram matrix they can differ by an ulp (cross terms are added in
					// opposite order), so only the upper-triangle owner writes both mirror positions
					if (row <= col) {
						Gs[row * 65 + col] = v;
						Gs[col * 65 + row] = v;
					}
				}
			}
	}
	__syncthreads();
	if (shift_coef > 0.0) {                              // shifted Cholesky: G + s I, s = shift_coef * trace(G)
		if (w == 0) {
			double tr = (j < n) ? Gs[j * 65 + j] : 0.0;
			for (int o = 32; o > 0; o >>= 1) tr += __shfl_xor(tr, o);
			if (j < n) Gs[j * 65 + j] += shift_coef * tr;
		}
		__syncthreads();
	}
	// Round 4: the rows are PACKED.  Of row i = 4w + u the elimination only ever needs G[i][j] for j >= 4w (upper triangle + the wave's
	// own 4 x 4 diagonal block) and M[i][j] for j < 4w (M = R^-T is lower triangular, and its own diagonal block is touched by nothing
	// but the owner's section): lane j of x[u] holds G[i][j] for j >= 4w and M[i][j] for j < 4w, md[u] the diagonal block of M (identity
	// until the section).  A rank-4 update is then 16 fp64 FMAs per wave instead of 32 -- the step is bound by the fp64 issue of the four
	// SIMDs of its one CU (stamps: a group cost ~2000 cycles = four waves x 32 FMAs x ~10 cycles + the owner's section per SIMD).  Every
	// element sees the same operations in the same order as before: R, Z and the verdict are unchanged bit for bit.
	const bool gpart = j >= 4 * w;
	double x[4], md[4];
#pragma unroll
	for (int u = 0; u < 4; u++) {
		const int i = 4 * w + u;
		x[u] = gpart ? Gs[i * 65 + j] : 0.0;
		md[u] = (i == j) ? 1.0 : 0.0;
	}
	if (t < 64) { dg[t] = Gs[t * 65 + t]; pv[t] = 1.0; }
	const double dgj = Gs[j * 65 + j];
	double s_acc = 0.0;
	__syncthreads();
	CHOL_STAMP16(1);
	const int ngroups = (n + 3) >> 2;
	// owner section of group gi: the four pivots, rows factored against each other in registers (lane broadcasts), published
	auto section = [&](int gi) __attribute__((always_inline)) {      // (called from two places: left out of line, x / md would live in scratch memory)
		const int K0 = 4 * gi;
		double* rr = Rrow + (gi & 1) * 256;              // [4][64]
		double* mr = Mrow + (gi & 1) * 256;
		double g[4], mm[4], rk[4], mk[4], piv0[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {                    // (K0 = 4w: the G lanes are j >= K0, where the section reads them)
			const double xv = x[u], mv = md[u];              // (values first, then the select: a select between the two ADDRESSES puts the arrays in scratch memory)
			g[u] = xv;
			mm[u] = gpart ? mv : xv;
		}
		chol_section4(g, mm, K0, j, n, rk, mk, piv0, rr, mr);
		if (gi > 0) CHOL_STAMP16(8 + 8 * (gi - 1) + 5);
#pragma unroll
		for (int u = 0; u < 4; u++)
			if (j == 0 && K0 + u < n) pv[K0 + u] = piv0[u];
	};
	// rank-4 update of this wave's four rows with the published rows of group gi: R part first (an owner's pivots wait for it), then M
	auto update = [&](int gi) __attribute__((always_inline)) {
		const double* rr = Rrow + (gi & 1) * 256;
		const double* mr = Mrow + (gi & 1) * 256;
		// (stamps, round 4: a group's period is the LDS pipe -- every wave behind the owner re-reads the published rows.  A lane needs
		// row K of R where it holds G and row K of M where it holds M: ONE read per row from the array its side of the packing selects)
		const double* yr = gpart ? rr : mr;
		double rki[4][4], yv[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const f64x2c a01 = *reinterpret_cast<const f64x2c*>(&rr[u * 64 + 4 * w]);
			const f64x2c a23 = *reinterpret_cast<const f64x2c*>(&rr[u * 64 + 4 * w + 2]);
			rki[u][0] = a01[0]; rki[u][1] = a01[1]; rki[u][2] = a23[0]; rki[u][3] = a23[1];      // R[K0+u][4w .. 4w+3]
			yv[u] = yr[u * 64 + j];
		}
#pragma unroll
		for (int sl = 0; sl < 4; sl++)
#pragma unroll
			for (int u = 0; u < 4; u++) x[sl] = fma(-rki[u][sl], yv[u], x[sl]);
	};
	if (w == 0) section(0);
#pragma unroll 1
	for (int gi = 0; gi < ngroups; gi++) {
		CHOL_STAMP16(8 + 8 * gi + 1);
		lds_barrier();                                   // group gi's rows are published (LDS only: result stores stay in flight)
		CHOL_STAMP16(8 + 8 * gi + 2);
		if (w == gi + 1 && gi + 1 < ngroups) {
			// the next owner: update and owner section as ONE block of straight-line code -- the update of its rows of M has no
			// part in the pivot chain and fills the chain's latency gaps instead of standing in front of it.  (Round 4: raising this
			// wave's issue priority with s_setprio for the length of the block changed nothing: 15.9 vs 15.7 us.)
			update(gi);
			CHOL_STAMP16(8 + 8 * gi + 4);
			section(gi + 1);
			CHOL_STAMP16(8 + 8 * gi + 7);
		} else if (w > gi) {
			update(gi);
		}
		if (w == ((gi + 15) & 15)) {
			// a wave that is off the path (the previous owner: its rows are finished; for the first group the last wave, which is
			// fifteen groups away from owning): the finished rows leave for global memory, the verdict sum grows
			const int K0 = 4 * gi;
			const double* rr = Rrow + (gi & 1) * 256;
			const double* mr = Mrow + (gi & 1) * 256;
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int K = K0 + u;
				const double rkj = rr[u * 64 + j], mkc = mr[u * 64 + j];
				if (K < n) {
					if (j < NP) z[(size_t)K * NP + j] = (j <= K) ? (float)mkc : 0.0f;     // Z[j][K] = M[K][j]
					if (j < n) r[(size_t)j * ldr + K] = (j >= K) ? (float)rkj : 0.0f;
					Gs[K * 65 + j] = (j <= K) ? mkc : 0.0;                                 // (fp64 image, read on by chol_wide_kernel)
					if (j <= K) s_acc = fma(dgj * mkc, mkc, s_acc);                        // sum of g_jj * Z[j][K]^2
				}
			}
		}
		CHOL_STAMP16(8 + 8 * gi + 3);
	}
	CHOL_STAMP16(2);
	// scaled conditioning S and the status words
	for (int o = 32; o > 0; o >>= 1) s_acc += __shfl_xor(s_acc, o);
	if (j == 0) sred[w] = s_acc;
	lds_barrier();                                       // (LDS only: the verdict arithmetic runs while the last rows' stores are acknowledged)
	if (w == 0) {
		const double d0 = dg[j], p0 = pv[j];
		float ratio = (j < n) ? ((d0 > 0.0) ? (float)(p0 / d0) : 0.0f) : 1.0f;
		if (j < n && !(d0 >= min_diag)) ratio = 0.0f;
		for (int o = 32; o > 0; o >>= 1) ratio = fminf(ratio, __shfl_xor(ratio, o));
		if (j == 0) {
			double ssum = 0.0;
#pragma unroll
			for (int k = 0; k < 16; k++) ssum += sred[k];
			const float scond = (float)(ssum / (double)n);
			const unsigned s0 = (ratio > min_ratio && scond <= max_scond) ? 0u : 1u;     // NaN compares false -> rejected
			status[0] = s0;
			status[1] = __builtin_bit_cast(unsigned, ratio);
			status[2] = __builtin_bit_cast(unsigned, scond);
			if (host_status) {
				volatile unsigned* hs = host_status;
				hs[1] = __builtin_bit_cast(unsigned, ratio);
				hs[2] = __builtin_bit_cast(unsigned, scond);
				hs[0] = s0;
			}
		}
	}
	CHOL_STAMP16(3);
#ifdef TSQR_CHOL_STAMPS
	if ((threadIdx.x & 63) == 0 && w < 4) chol_stamp_lds[w * CHOL_NSTAMP + 4] = __builtin_amdgcn_s_memrealtime();
	__syncthreads();
	if (g_chol_stamp_out)
		for (int i = threadIdx.x; i < 4 * CHOL_NSTAMP; i += 1024) g_chol_stamp_out[i] = chol_stamp_lds[i];
#endif
}

struct CholArgs {
	float* r; size_t ldr;                // R out: n x n, full block written (zeros below the diagonal)
	float* z;                            // Z = inverse(R) out: NP x NP column-major (ld NP), zero padded
	unsigned* status;                    // [0] 0 accepted / 1 rejected, [1] min pivot ratio (float bits), [2] S (float bits)
	unsigned* host_status;               // optional device-visible alias of pinned host words receiving the same three values
	const double* gsum;                  // summed Gram tiles, (tile, reg, lane) accumulator order
	const unsigned* prev_status;         // optional: status word of the sweep this one depends on (rejected -> report rejected at once)
	const double* rows_dev;              // optional: the row count (summed over the ranks of a row-partitioned run) as a double in
	                                     // device memory -- overrides `rows`, so that every rank applies identical thresholds
	double rows;                         // rows of the factored matrix: sets the bf16-level acceptance bound and the shift
	double shift_coef;                   // > 0: shifted Cholesky, s = shift_coef * (rows * n + n (n + 1)) * trace(G)
	int n, NT;
	int level;                           // 2 bf16-split Gram matrix (f32 accumulator layout; pivot ratio > 2^-5, S bound, column norms >= 2^-90 rows),
	                                     // 1 fp64 Gram matrix (f64 accumulator layout; ratio > 2^-40), 3 shifted fp64 (ratio > 0: rejects only non-finite input)
	float scond_floor;                   // bf16 level: S <= min(128, max(scond_floor, 0.12 sqrt(rows)))
	int relax;                           // level 2 only: 1 = ANOTHER SWEEP FOLLOWS on the Q this factor produces (reorthogonalised calls), so Q need
	                                     // only come out well conditioned, not orthonormal: pivot ratio > 2^-20, S <= min(1.6e7, 8000 sqrt(rows))
	                                     // (loss of orthogonality of this sweep ~8e-6 S / sqrt(rows) <= 0.06: the next sweep sees cond(Q) ~ 1)
	int retry_shift;                     // level 2 only, 1: a matrix the (relaxed) rule rejects is factored again at once, in the same launch, as
	                                     // G + s I, s = c trace(G), c = 8 * 2^-23 / sqrt(rows) -- shifted Cholesky QR
	                                     // on the bf16-split Gram matrix (qr_core); `rows` is the all-reduced count of a row-partitioned call, so every
	                                     // rank applies the same shift.  status[3] = 1 and host word 0 = 2 tell "shifted": two more sweeps must follow.
	                                     // The host words are written once, at the end of the launch.
};

// the 64-column path's launch of the step.  prev_status: status word of an earlier factorisation this one depends on (speculatively
// enqueued second sweep): when that one was rejected this one reports "rejected" at once, so that everything enqueued behind it skips as well
__global__ __launch_bounds__(1024) void chol16_kernel(const CholArgs a) {
	if (a.prev_status && a.prev_status[0] != 0) {
		if (threadIdx.x == 0) {
			a.status[0] = 1u; a.status[1] = 0u; a.status[2] = 0u;
			if (a.host_status) { volatile unsigned* hs = a.host_status; hs[1] = 0u; hs[2] = 0u; hs[0] = 1u; }
		}
		return;
	}
	const double rows = a.rows_dev ? a.rows_dev[0] : a.rows;
	float min_ratio = 0.0f, max_scond = INFINITY;
	double min_diag = 0.0, shift = 0.0;
	if (a.level == 2) {
		min_ratio = 0.03125f;
		max_scond = fminf(128.0f, fmaxf(a.scond_floor, 0.12f * sqrtf((float)rows)));
		min_diag = rows * 0x1p-90;
		if (a.relax) { min_ratio = 0x1p-20f; max_scond = fminf(1.6e7f, 8000.0f * sqrtf((float)rows)); }
	} else if (a.level == 1) {
		min_ratio = 9.094947017729282e-13f;              // 2^-40
	} else {
		shift = a.shift_coef * (rows * (double)a.n + (double)a.n * (double)(a.n + 1));
	}
	const bool retry = a.level == 2 && a.retry_shift != 0;
	auto loadg = [&](int e) { return a.gsum[e]; };
	chol_body16(a.r, a.ldr, a.z, a.status, retry ? nullptr : a.host_status, loadg, a.n, a.NT, a.level == 2 ? 1 : 0, min_ratio, max_scond, shift, min_diag);
	if (!retry) return;
	// (thread 0 wrote the verdict itself: program order)
	__shared__ unsigned again;
	__syncthreads();
	if (threadIdx.x == 0) { again = a.status[0]; a.status[3] = 0u; }
	__syncthreads();
	if (again) {
		// rejected: the same Gram matrix, shifted.  Accepted whenever every pivot is positive (non-finite input stays rejected); the
		// column-norm floor of the bf16-split level still holds (products near the denormal range were not accumulated accurately)
		const double coef = 8.0 * 0x1p-23 / sqrt(fmax(rows, 1.0));
		chol_body16(a.r, a.ldr, a.z, a.status, nullptr, loadg, a.n, a.NT, 1, 0.0f, INFINITY, coef, min_diag);
		__syncthreads();
		if (threadIdx.x == 0 && a.status[0] == 0u) a.status[3] = 1u;
	}
	if (threadIdx.x == 0 && a.host_status) {
		volatile unsigned* hs = a.host_status;
		hs[1] = a.status[1];
		hs[2] = a.status[2];
		hs[0] = (a.status[0] == 0u && a.status[3] != 0u) ? 2u : a.status[0];
	}
}

// ---------------------------------------------------------------------------------------------
// The same step on FOUR waves (round 2's form of it: wave w owns four consecutive rows of each 16-row block), for the one place where
// the Cholesky step has to live in a 256-thread workgroup: gram_blk_chain_kernel.  Its LDS comes from the caller (42.5 KB).
// ---------------------------------------------------------------------------------------------
// Row ownership of the elimination kernels: thread (w, j) holds column j of the rows  row(w, s) = 16*(s>>2) + 4*w + (s&3),
// i.e. every wave owns FOUR consecutive rows of each 16-row block.  A "group" = those four rows: its owner factors them
// against each other in registers (lane broadcasts, no LDS), publishes the four finished rows, and after ONE barrier all
// waves apply the four rank-1 updates to their remaining rows.  16 barriers for 64 rows; after the four groups of a block
// the register rows rotate by four so the active block is always slots 0..3.
template <int U>
__device__ __forceinline__ void chol_group4(double (&g)[16], double (&mm)[16], double* Rrow, double* Mrow, float* __restrict__ r, size_t ldr,
                                           float* __restrict__ z, int NP, double* Zd,
                                           double* pv, int w, int j, int n, int kk, double dgj, double& s_acc) {
	const int K0 = 16 * kk + 4 * U;
	if (K0 >= n) return;                                 // uniform over the workgroup (the barrier below included)
	double* rr = Rrow + (U & 1) * 256;                   // [4][64]
	double* mr = Mrow + (U & 1) * 256;
	if (w == U) {
		// the owner's section is the critical path (three waves wait at the barrier): nothing but the pivots, the two row
		// scalings, the in-group eliminations and the publication of the rows (chol_section4); fp32 copies, Z and the verdict sums
		// are taken from the published rows by a wave that is off the path (below)
		double g4[4] = {g[0], g[1], g[2], g[3]}, m4[4] = {mm[0], mm[1], mm[2], mm[3]};
		double rk[4], mk[4], piv0[4];
		chol_section4(g4, m4, K0, j, n, rk, mk, piv0, rr, mr);
#pragma unroll
		for (int u = 0; u < 4; u++) {
			g[u] = g4[u]; mm[u] = m4[u];
			if (j == 0 && K0 + u < n) pv[K0 + u] = piv0[u];
		}
	}
	lds_barrier();                                       // (LDS only: the result stores below stay in flight across the groups)
	double rkj[4], mkc[4];
#pragma unroll
	for (int u = 0; u < 4; u++) { rkj[u] = rr[u * 64 + j]; mkc[u] = mr[u * 64 + j]; }
	if (w == ((U + 3) & 3)) {                            // the previous owner: not the next one, which is on the critical path
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int K = K0 + u;
			if (K < n) {
				// the finished rows leave for global memory at once (round 3: the stores overlap the rest of the elimination instead
				// of forming a 5 K-cycle tail): column K of Z = row K of M, contiguous; row K of R, strided by ldr, exact zeros
				// below the diagonal
				if (j < NP) z[(size_t)K * NP + j] = (j <= K) ? (float)mkc[u] : 0.0f;     // Z[j][K] = M[K][j]
				if (j < n) r[(size_t)j * ldr + K] = (j >= K) ? (float)rkj[u] : 0.0f;
				Zd[K * 65 + j] = (j <= K) ? mkc[u] : 0.0;                            // (fp64 image, read on by chol_wide_kernel)
				if (j <= K) s_acc = fma(dgj * mkc[u], mkc[u], s_acc);                // sum of g_jj * Z[j][K]^2
			}
		}
	}
	const int nlive = 16 - 4 * kk;                       // register rows that still exist
#pragma unroll
	for (int s = 0; s < 16; s++) {
		if (s < nlive && !(s < 4 && w <= U)) {            // wave-uniform; slots 0..3 of waves <= U are finished rows
			const int i = 16 * (kk + (s >> 2)) + 4 * w + (s & 3);
			double acc_g = g[s], acc_m = mm[s];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const double rki = rr[u * 64 + i];           // R[K0+u][i]; rows i > K0+3 here
				acc_g = fma(-rki, rkj[u], acc_g);
				acc_m = fma(-rki, mkc[u], acc_m);
			}
			g[s] = acc_g; mm[s] = acc_m;
		}
	}
}

// LOADG: functor e -> G tile entry e (accumulator order); host_status: optional device-visible alias of pinned host memory
// that receives the three status words as well (the host then needs no copy operation to read them).
template <class LOADG>
__device__ __forceinline__ void chol_body4(float* __restrict__ r, size_t ldr, float* __restrict__ z, unsigned* __restrict__ status,
                                          unsigned* __restrict__ host_status, LOADG loadg, int n, int NT, int f32_layout, float min_ratio,
                                          float max_scond, double shift_coef, double min_diag, double* lds) {
	double* Gs = lds;                            // [64 * 65] symmetric G (assembly); afterwards the fp64 image of Z: Gs[K * 65 + j] = Z[j][K]
	double* Rrow = Gs + 64 * 65;                 // [2 * 256]
	double* Mrow = Rrow + 2 * 256;               // [2 * 256]
	double* dg = Mrow + 2 * 256;                 // [64]
	double* pv = dg + 64;                        // [64]
	const int t = threadIdx.x;
	const int j = t & 63, w = t >> 6;
	const int NP = 16 * NT;
	// issue the loads of G first (one value per thread and tile), then initialise LDS while they are in flight
	double gv[10];
	{
		int idx = 0;
		for (int ti = 0; ti < 4; ti++)
			for (int tj = ti; tj < 4; tj++) {
				if (ti < NT && tj < NT) { gv[ti * 4 + tj - (ti * (ti + 1)) / 2] = loadg(idx * 256 + t); idx++; }
				else gv[ti * 4 + tj - (ti * (ti + 1)) / 2] = 0.0;
			}
	}
	if (NT < 4)                                          // with all ten tiles present every entry of Gs is written below
		for (int i = t; i < 64 * 65; i += 256) Gs[i] = 0.0;
	for (int e = n * NP + t; e < NP * NP; e += 256) z[e] = 0.0f;          // padding rows of Z
	__syncthreads();
	{
		const int reg = t >> 6, l = t & 63;
#pragma unroll
		for (int ti = 0; ti < 4; ti++)
#pragma unroll
			for (int tj = ti; tj < 4; tj++) {
				if (ti < NT && tj < NT) {
					// C/D layouts: f64 MFMA row = (lane>>4) + 4*reg, f32/bf16 MFMA row = 4*(lane>>4) + reg; col = lane&15
					const int row = 16 * ti + (f32_layout ? 4 * (l >> 4) + reg : (l >> 4) + 4 * reg);
					const int col = 16 * tj + (l & 15);
					const double v = gv[ti * 4 + tj - (ti * (ti + 1)) / 2];
					// a diagonal tile holds (i,j) and (j,i); in the bf16-split Gram matrix they can differ by an ulp (cross terms
					// are added in opposite order), so only the upper-triangle owner writes both mirror positions
					if (row <= col) {
						Gs[row * 65 + col] = v;
						Gs[col * 65 + row] = v;
					}
				}
			}
	}
	__syncthreads();
	if (shift_coef > 0.0) {
		// shifted Cholesky (Fukaya et al., "Shifted Cholesky QR for computing the QR factorization of ill-conditioned matrices",
		// SIAM J. Sci. Comput. 2020): G + s I with s = shift_coef * trace(G) >= 11 (mn + n(n+1)) u ||A||_2^2 is safely positive
		// definite for any fp32 input; the caller runs a second (unshifted) sweep on the resulting Q
		if (w == 0) {
			double tr = (j < n) ? Gs[j * 65 + j] : 0.0;
			for (int o = 32; o > 0; o >>= 1) tr += __shfl_xor(tr, o);
			if (j < n) Gs[j * 65 + j] += shift_coef * tr;
		}
		__syncthreads();
	}
	double g[16], mm[16];
#pragma unroll
	for (int s = 0; s < 16; s++) {
		const int i = 16 * (s >> 2) + 4 * w + (s & 3);
		g[s] = Gs[i * 65 + j];
		mm[s] = (i == j) ? 1.0 : 0.0;
	}
	if (t < 64) { dg[t] = Gs[t * 65 + t]; pv[t] = 1.0; }
	const double dgj = Gs[j * 65 + j];
	double s_acc = 0.0;
	__syncthreads();
#pragma unroll 1
	for (int kk = 0; kk < 4; kk++) {
		static_for<0, 4>([&](auto u) { chol_group4<decltype(u)::value>(g, mm, Rrow, Mrow, r, ldr, z, NP, Gs, pv, w, j, n, kk, dgj, s_acc); });
#pragma unroll
		for (int s = 0; s < 12; s++) { g[s] = g[s + 4]; mm[s] = mm[s + 4]; }   // the next 16-row block moves to slots 0..3
	}
	// scaled conditioning S = || D * inverse(R) ||_F^2 / n with D = diag(sqrt(g_jj)): 1 for orthogonal columns of any scaling,
	// ~cond^2 of the column-scaled matrix otherwise.  An entry-wise error eps*sqrt(g_ii g_jj) of G perturbs Q^T Q by <= eps*n*S.
	for (int o = 32; o > 0; o >>= 1) s_acc += __shfl_xor(s_acc, o);
	__syncthreads();
	if (j == 0) Rrow[w] = s_acc;
	__syncthreads();
	// status[0]: 0 ok, 1 rejected: a pivot fell below min_ratio of its diagonal entry (2^-40 for the fp64 Gram matrix:
	//            cond(A)^2 beyond fp64 Cholesky; 2^-5 for the bf16-split Gram matrix) or S exceeds max_scond (bf16-split Gram
	//            matrix only: its fp32 accumulation is good enough for nearly orthogonal columns only)
	// status[1]: bit pattern of the smallest pivot ratio (float), status[2]: of S (float) -- diagnostics
	if (w == 0) {
		const double d0 = dg[j], p0 = pv[j];
		float ratio = (j < n) ? ((d0 > 0.0) ? (float)(p0 / d0) : 0.0f) : 1.0f;
		// min_diag (bf16-split level): a column whose squared norm is so small that its fp32 products live near the denormal range
		// was not accumulated accurately (measured: entries ~1e-22 gave ||Q^T Q - I|| = 1e-2) -> reject, the fp64 level is exact
		if (j < n && !(d0 >= min_diag)) ratio = 0.0f;
		for (int o = 32; o > 0; o >>= 1) ratio = fminf(ratio, __shfl_xor(ratio, o));
		if (j == 0) {
			const float scond = (float)(((Rrow[0] + Rrow[1]) + (Rrow[2] + Rrow[3])) / (double)n);
			const unsigned s0 = (ratio > min_ratio && scond <= max_scond) ? 0u : 1u;     // NaN compares false -> rejected
			status[0] = s0;
			status[1] = __builtin_bit_cast(unsigned, ratio);
			status[2] = __builtin_bit_cast(unsigned, scond);
			if (host_status) {
				// (no fence: the host reads these words only after the completion word of a LATER kernel on the stream, or after a
				// stream synchronisation -- a system-scope release here wrote back the whole L2 on the critical path of every call)
				volatile unsigned* hs = host_status;
				hs[1] = __builtin_bit_cast(unsigned, ratio);
				hs[2] = __builtin_bit_cast(unsigned, scond);
				hs[0] = s0;
			}
		}
	}
	// (R and Z have left for global memory row by row during the elimination: chol_group)
}


// ---------------------------------------------------------------------------------------------
// A STREAM of 2^k x 64 calls (tsqr_mi_qr_f32_loop): the R-factor chain of call i -- reduction of its Gram partials, Cholesky, verdict;
// 22 us during which one workgroup works and the rest of the chip idles -- runs in the shadow of the Gram pass of call i + 1, in ONE
// launch: the first `nred` workgroups of the grid reduce 16 entries each (gram_reduce1_body: same partition, same order of additions
// as the launch of its own) and take a ticket; the one that draws the last ticket factors (last-adder pattern: no workgroup ever
// waits for another).  All other workgroups are the Gram pass of the next call, numbered from 0 as in gram_blk_kernel.
// ---------------------------------------------------------------------------------------------
// The hand-over of the reduced sums inside these launches (write-through stores, `s_waitcnt vmcnt(0)`, a relaxed ticket, device-scope loads)
// is NOT a release / acquire pair of the HIP memory model: it relies on gfx9's store accounting -- `vmcnt` counts stores, and a store with
// `sc1` is acknowledged only once it is visible device-wide -- which gfx10+ (a separate store counter) does not give.  This library is
// built for gfx950 only; any other device target must not compile this file silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "tsqr_kernels.hip: the chained launches rely on gfx942 / gfx950 memory-counter semantics (vmcnt covers stores); port the hand-over before building for another target"
#endif
struct ChainArgs {
	CholArgs chol;                       // (level 2; chol.gsum receives the reduced matrix)
	const double* part; int nparts;      // Gram partials of the call being factored
	unsigned* ticket;                    // arrival counter, zero at launch; the last adder re-arms it
	int nred;                            // reduction workgroups = entries / 16
	int direct;                          // 1: chol.gsum holds the reduced (row-partitioned: all-reduced) matrix already -- ONE chain workgroup,
	                                     // which factors at once (nred, part, ticket unused)
};
__global__ __launch_bounds__(256, 2) void gram_blk_chain_kernel(const GramArgs a, const ChainArgs ch) {
	extern __shared__ __attribute__((aligned(16))) float gb_as[];
	static_assert(GB_LDS_BYTES >= (int)sizeof(double) * (64 * 65 + 4 * 256 + 128), "chol_body4's arrays alias the block buffers");
	announce_previous_call(a.announce, a.announce_seq);
	const int nchain = ch.direct ? 1 : ch.nred;
	if ((int)blockIdx.x >= nchain) {
		gram_blk_body(a, gb_as, (int)blockIdx.x - nchain, (int)gridDim.x - nchain);
		return;
	}
	const CholArgs& c = ch.chol;
	if (ch.direct) {
		// row-partitioned stream: the all-reduce sits between the reduction and this point, so only the factorisation rides along
		const double rows = c.rows_dev ? c.rows_dev[0] : c.rows;
		const float max_scond = fminf(128.0f, fmaxf(c.scond_floor, 0.12f * sqrtf((float)rows)));
		const double* g = c.gsum;
		chol_body4(c.r, c.ldr, c.z, c.status, c.host_status, [&](int e) { return g[e]; }, c.n, c.NT, /*f32_layout=*/1, 0.03125f, max_scond, 0.0,
		           rows * 0x1p-90, reinterpret_cast<double*>(gb_as));
		return;
	}
	gram_reduce1_body<true>(blockIdx.x, reinterpret_cast<double (*)[17]>(gb_as), const_cast<double*>(c.gsum), ch.part, ch.nparts, 10 * 256, c.rows,
	                        nullptr, 0, nullptr, 0);
	// The sums left as device-scope write-through stores; once they are acknowledged (vmcnt) the workgroup takes its ticket.  No
	// fence anywhere: a release / acquire fence here writes back / invalidates a whole L2 per wave, 640 times per launch, under the
	// Gram pass that runs beside it (measured: the launch took 85 us instead of 49).
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	__shared__ unsigned last;
	if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(ch.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ch.nred - 1)) ? 1u : 0u;
	__syncthreads();
	if (!last) return;
	if (threadIdx.x == 0) __hip_atomic_store(ch.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
	const double rows = c.rows;
	const float max_scond = fminf(128.0f, fmaxf(c.scond_floor, 0.12f * sqrtf((float)rows)));
	const double* g = c.gsum;                            // (written by other workgroups of this launch: device-scope loads)
	chol_body4(c.r, c.ldr, c.z, c.status, c.host_status, [&](int e) { return __hip_atomic_load(&g[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, c.n, c.NT, /*f32_layout=*/1, 0.03125f, max_scond, 0.0,
	           rows * 0x1p-90, reinterpret_cast<double*>(gb_as));
}

// The same launch for a stream of fp16 calls (tsqr_mi_qr_f16_loop, n = 64): the R-factor chain of call i beside gram_h_kernel's body for
// call i + 1.  One LDS array serves the three roles (reduction scratch, the Gram role's workgroup sum, chol_body4).
__global__ __launch_bounds__(256) void gram_h_chain_kernel(const GramArgs a, const ChainArgs ch) {
	__shared__ double hc_lds[64 * 65 + 4 * 256 + 128];
	static_assert(sizeof(hc_lds) >= sizeof(double) * 2 * 10 * 256, "the Gram role's workgroup sum fits");
	announce_previous_call(a.announce, a.announce_seq);
	if ((int)blockIdx.x >= ch.nred) {
		gram_h_body<4>(a, reinterpret_cast<double (*)[10 * 256]>(hc_lds), (int)blockIdx.x - ch.nred);
		return;
	}
	const CholArgs& c = ch.chol;
	gram_reduce1_body<true>(blockIdx.x, reinterpret_cast<double (*)[17]>(hc_lds), const_cast<double*>(c.gsum), ch.part, ch.nparts, 10 * 256, c.rows,
	                        nullptr, 0, nullptr, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (write-through sums acknowledged, then the ticket: gram_blk_chain_kernel)
	__syncthreads();
	__shared__ unsigned last;
	if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(ch.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ch.nred - 1)) ? 1u : 0u;
	__syncthreads();
	if (!last) return;
	if (threadIdx.x == 0) __hip_atomic_store(ch.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const double rows = c.rows;
	const float max_scond = fminf(128.0f, fmaxf(c.scond_floor, 0.12f * sqrtf((float)rows)));
	const double* g = c.gsum;
	chol_body4(c.r, c.ldr, c.z, c.status, c.host_status, [&](int e) { return __hip_atomic_load(&g[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, c.n, c.NT,
	           /*f32_layout=*/1, 0.03125f, max_scond, 0.0, rows * 0x1p-90, hc_lds);
}

// ---------------------------------------------------------------------------------------------
// Panel coupling for n > 64 (block modified Gram-Schmidt between 64-column panels; replaces the two cuBLAS GEMMs of
// reference src/blockqr.cu:92-116):
//   cross_kernel        : S = Qb^T Ap (64 x c) on v_mfma_f32_16x16x32_bf16 with the 3-way split of both operands (six products per
//                         tile, as the Gram engine: 24-bit products at 1/8 of the exact-fp32 MFMA's cycles -- 154 -> ~90 us at
//                         2^20 x 128), both operands straight from the (c,q) registers; per-workgroup partials like the Gram engine
//   cross_finish_kernel : summed tiles -> S into R (ldr) and -S as a 64 x 64 column-major matrix for the update
//   apply_wg_kernel<E,4,true,ROWS> : Ap <- Ap - Qb * S   (the apply kernel with a C input and a full, non-triangular Z)
// ---------------------------------------------------------------------------------------------
struct CrossArgs {
	const float* x; size_t ldx; const float* y; size_t ldy; size_t m; int ny;     // X: m x 64, Y: m x ny
	int nchunks; int cpw; int nwaves;
	double* part;                        // [gridDim.y][gridDim.x][16][256]
	int ny_total;                        // > 0 (round 4, right-looking coupling): Y is the WHOLE trailing matrix, m x ny_total; workgroups (., j) take its
	                                     // 64-column panel j (the last one may be narrower) -- one launch per finished panel instead of one per panel pair
};

__global__ __launch_bounds__(256, 2) void cross_kernel(const CrossArgs a) {
	__shared__ float red[2][16 * 256];
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int gw = blockIdx.x * 4 + wv;
	const int c = lane & 15, q = lane >> 4;
	const float* const ypan = a.y + (size_t)blockIdx.y * 64 * a.ldy;                     // (blockIdx.y == 0 unless ny_total is set)
	const int ny = a.ny_total > 0 ? min(64, a.ny_total - 64 * (int)blockIdx.y) : a.ny;
	f32x4 acc[16];
#pragma unroll
	for (int t = 0; t < 16; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
	if (gw < a.nwaves) {
		const int ch_end = min(a.nchunks, (gw + 1) * a.cpw);
		for (int ch = gw * a.cpw; ch < ch_end; ch++) {
			float px[4][16], py[4][16];
			load_chunk<4>(px, a.x, a.ldx, (size_t)ch * 64, a.m, 64, c, q);
			load_chunk<4>(py, ypan, a.ldy, (size_t)ch * 64, a.m, ny, c, q);
#pragma unroll
			for (int kt = 0; kt < 2; kt++) {                 // K-step of 32 rows: registers 8kt .. 8kt+7 of every lane (as gram_bf16_kernel)
				bf16x8 xh[4], xm[4], xl[4], yh[4], ym[4], yl[4];
#pragma unroll
				for (int t = 0; t < 4; t++) {
					u32x4 hh, mm, ll;
#pragma unroll
					for (int jp = 0; jp < 4; jp++) {
						unsigned h, m, lo;
						split3_pair(px[t][8 * kt + 2 * jp], px[t][8 * kt + 2 * jp + 1], h, m, lo);
						hh[jp] = h; mm[jp] = m; ll[jp] = lo;
					}
					xh[t] = __builtin_bit_cast(bf16x8, hh); xm[t] = __builtin_bit_cast(bf16x8, mm); xl[t] = __builtin_bit_cast(bf16x8, ll);
#pragma unroll
					for (int jp = 0; jp < 4; jp++) {
						unsigned h, m, lo;
						split3_pair(py[t][8 * kt + 2 * jp], py[t][8 * kt + 2 * jp + 1], h, m, lo);
						hh[jp] = h; mm[jp] = m; ll[jp] = lo;
					}
					yh[t] = __builtin_bit_cast(bf16x8, hh); ym[t] = __builtin_bit_cast(bf16x8, mm); yl[t] = __builtin_bit_cast(bf16x8, ll);
				}
				// six split products per tile, smallest first (mm hl lh hm mh hh); fp32 accumulation over this wave's rows as before
#pragma unroll
				for (int pass = 3; pass < 9; pass++)
#pragma unroll
					for (int ti = 0; ti < 4; ti++)
#pragma unroll
						for (int tj = 0; tj < 4; tj++) {
							const bf16x8 av = (pass == 4 || pass == 6 || pass == 8) ? xh[ti] : ((pass == 3 || pass == 7) ? xm[ti] : xl[ti]);
							const bf16x8 bv = (pass == 5 || pass == 7 || pass == 8) ? yh[tj] : ((pass == 3 || pass == 6) ? ym[tj] : yl[tj]);
							acc[4 * ti + tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[4 * ti + tj], 0, 0, 0);
						}
			}
		}
	}
	if (wv >= 2) {
#pragma unroll
		for (int t = 0; t < 16; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[wv - 2][(t * 4 + r) * 64 + lane] = acc[t][r];
	}
	__syncthreads();
	if (wv < 2) {
#pragma unroll
		for (int t = 0; t < 16; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) acc[t][r] += red[wv][(t * 4 + r) * 64 + lane];
	}
	__syncthreads();
	if (wv == 1) {
#pragma unroll
		for (int t = 0; t < 16; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[0][(t * 4 + r) * 64 + lane] = acc[t][r];
	}
	__syncthreads();
	if (wv == 0) {
		double* out = a.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 * 256;
#pragma unroll
		for (int t = 0; t < 16; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) part_store(&out[(t * 4 + r) * 64 + lane], (double)(acc[t][r] + red[0][(t * 4 + r) * 64 + lane]));
	}
}

// the reduction of cross_kernel's partials for ALL trailing panels of a finished panel in one launch (grid (256, trailing panels)):
// slice j sums its nparts partials exactly as gram_reduce1_kernel does (same body) into gout + j * CROSS_GSTRIDE and -- one GPU -- writes
// the block of R (rows of the finished panel, columns of trailing panel j) and the operand -S of the update (fin_zneg + j * 4096)
constexpr int CROSS_GSTRIDE = 16 * 256 + 8;            // (the body puts the row count behind the 4096 sums of a slice)
__global__ __launch_bounds__(256) void cross_reduce_multi_kernel(double* __restrict__ gout, const double* __restrict__ part, int nparts, double rows,
                                                                 float* __restrict__ fin_r, size_t fin_ldr, float* __restrict__ fin_zneg, int ny_total) {
	__shared__ double red[16][17];
	const int j = blockIdx.y;
	gram_reduce1_body<false>(blockIdx.x, red, gout + (size_t)j * CROSS_GSTRIDE, part + (size_t)j * nparts * (16 * 256), nparts, 16 * 256, rows,
	                         fin_zneg ? fin_r + (size_t)j * 64 * fin_ldr : nullptr, fin_ldr, fin_zneg ? fin_zneg + (size_t)j * 4096 : nullptr,
	                         min(64, ny_total - 64 * j));
}

// gsum: 16 tiles x 256 doubles in f32-MFMA accumulator order (tile 4*ti+tj, row = 4*(lane>>4)+reg, col = lane&15)
// (ny_total > 0: grid (16, trailing panels) -- slice j reads gsum + j * CROSS_GSTRIDE and writes R's block j and zneg + j * 4096)
__global__ __launch_bounds__(256) void cross_finish_kernel(float* __restrict__ r, size_t ldr, float* __restrict__ zneg,
                                                           const double* __restrict__ gsum, int ny, int ny_total = 0) {
	if (ny_total > 0) {
		const int j = blockIdx.y;
		r += (size_t)j * 64 * ldr; zneg += (size_t)j * 4096; gsum += (size_t)j * CROSS_GSTRIDE; ny = min(64, ny_total - 64 * j);
	}
	for (int e = blockIdx.x * 256 + threadIdx.x; e < 16 * 256; e += 256 * gridDim.x) {
		const int t = e >> 8, reg = (e >> 6) & 3, l = e & 63;
		const int i = 16 * (t >> 2) + 4 * (l >> 4) + reg;
		const int j = 16 * (t & 3) + (l & 15);
		const float v = (float)gsum[e];
		zneg[(size_t)j * 64 + i] = (j < ny) ? -v : 0.0f;
		if (j < ny) r[(size_t)j * ldr + i] = v;
	}
}

// ---------------------------------------------------------------------------------------------
// trinv_kernel: Z = inverse of the n x n upper-triangular R (fp64 arithmetic, fp32 in/out), written zero-padded to
// NP x NP column-major (ld NP).  Forward elimination of [R^T | I] with the rows interleaved over the four waves in
// registers (same scheme as chol_body16: M = R^-T, Z = M^T), one barrier per step.
// ---------------------------------------------------------------------------------------------
template <int U>
__device__ __forceinline__ void trinv_group(double (&mm)[16], double* Mrow, const double* Rs, const double* rdiag,
                                            float* __restrict__ z, int w, int j, int n, int NP, int kk) {
	const int K0 = 16 * kk + 4 * U;
	if (K0 >= n) return;                                 // uniform over the workgroup
	double* mr = Mrow + (U & 1) * 256;                   // [4][64]
	if (w == U) {
		static_for<0, 4>([&](auto uu) {
			constexpr int u = decltype(uu)::value;
			const int K = K0 + u;
			const bool live = K < n;
			const double mk = live ? mm[u] * rdiag[K] : 0.0;
			static_for<u + 1, 4>([&](auto vv) {
				constexpr int v = decltype(vv)::value;
				mm[v] = fma(-Rs[K * 65 + K0 + v], mk, mm[v]);
			});
			mr[u * 64 + j] = mk;
			if (live && j < NP) z[(size_t)K * NP + j] = (j <= K) ? (float)mk : 0.0f;   // Z[j][K] = M[K][j]
		});
	}
	__syncthreads();
	double mkc[4];
#pragma unroll
	for (int u = 0; u < 4; u++) mkc[u] = mr[u * 64 + j];
	const int nlive = 16 - 4 * kk;
#pragma unroll
	for (int s = 0; s < 16; s++) {
		if (s < nlive && !(s < 4 && w <= U)) {
			const int i = 16 * (kk + (s >> 2)) + 4 * w + (s & 3);
			double acc = mm[s];
#pragma unroll
			for (int u = 0; u < 4; u++) acc = fma(-Rs[(K0 + u) * 65 + i], mkc[u], acc);
			mm[s] = acc;
		}
	}
}

__global__ __launch_bounds__(256) void trinv_kernel(float* __restrict__ z, const float* __restrict__ r, size_t ldr,
                                                    int n, int NP) {
	__shared__ double Rs[64 * 65];               // Rs[row * 65 + col]
	__shared__ double Mrow[2 * 256], rdiag[64];
	const int t = threadIdx.x;
	const int j = t & 63, w = t >> 6;
	for (int e = t; e < 64 * 64; e += 256) {
		const int row = e & 63, col = e >> 6;
		Rs[row * 65 + col] = (row <= col && col < n) ? (double)r[(size_t)col * ldr + row] : 0.0;
	}
	for (int e = n * NP + t; e < NP * NP; e += 256) z[e] = 0.0f;          // padding rows of Z
	__syncthreads();
	if (t < 64) rdiag[t] = (t < n) ? 1.0 / Rs[t * 65 + t] : 0.0;
	double mm[16];
#pragma unroll
	for (int s = 0; s < 16; s++) mm[s] = (16 * (s >> 2) + 4 * w + (s & 3) == j) ? 1.0 : 0.0;
	__syncthreads();
#pragma unroll 1
	for (int kk = 0; kk < 4; kk++) {
		static_for<0, 4>([&](auto u) { trinv_group<decltype(u)::value>(mm, Mrow, Rs, rdiag, z, w, j, n, NP, kk); });
#pragma unroll
		for (int s = 0; s < 12; s++) mm[s] = mm[s + 4];
	}
}

// ---------------------------------------------------------------------------------------------
// Q[rows, 0:n] = A[rows, 0:n] * Z   (Z = NP x NP upper triangular, zero padded).
// ENGINE 0 (fp32_notc):   v_mfma_f32_16x16x4_f32, exact fp32 FMA chains.
// ENGINE 1 (fp32_tc_cor): v_mfma_f32_16x16x32_bf16 on a 3-way bf16 split (hi, mid, lo) of both operands;
//   six products per tile, accumulated smallest terms first:  (mid*mid + hi*lo + lo*hi) + (hi*mid + mid*hi) + hi*hi
//   -- the error-correction idea of the reference's fp32_tc_cor (src/tcqr32x16.cu:669-819) carried to ~24 bits.
// ---------------------------------------------------------------------------------------------
struct ApplyArgs {
	const float* a; size_t lda; float* q; size_t ldq; size_t m; int n;
	const float* z;                     // NP x NP, ld NP
	int nchunks; int cpw; int nwaves;
	int n_out;                          // UPD only: columns of the output / C input (n is then the contraction length, 64)
	int multi_cols;                     // UPD only, > 0 (round 4, right-looking coupling): q is the WHOLE trailing matrix, m x multi_cols, z an array of
	                                    // 64 x 64 operands (-S_j, 4096 floats each); workgroups (., j) update its 64-column panel j
	double* gpart;                      // GRAMQ only: per-workgroup partial Gram tiles of the OUTPUT block rows (format of gram_bf16_kernel)
	const unsigned* skip_status;        // optional: the kernel returns at once when *skip_status != 0 (the Cholesky kernel
	                                    // rejected its Gram matrix: a speculatively enqueued apply then costs a launch, not a pass)
	const float* r32; void* r16; size_t ldr16;   // fp16 I/O only, optional: workgroup 0 also rounds the n x n factor r32 (ld n) to the caller's half-typed R
	int share[4];                       // all zero: blocks interleaved over the grid.  Otherwise (grid = four workgroups per CU, dispatched in four
	                                    // rounds): 64ths of a CU's blocks for its first .. fourth workgroup (apply_wg_body) and
	int even_share;                     // 128ths of the blocks of two neighbouring CUs (even XCD, odd XCD) for the one on the even XCD
	int plain_q;                        // 1: Q leaves with plain (cache-allocating) stores and
	int forward;                        // 1: the blocks are taken in ascending order (default: descending, nontemporal stores) -- the first sweep of a
	                                    // reorthogonalised call: the sweep behind it walks Q in DESCENDING order and so starts with the blocks written
	                                    // last, the half of Q the Infinity Cache still holds (round 4: C5 0.513 -> 0.498 ms, well-conditioned reorth
	                                    // 0.364 -> 0.357; plain stores alone, same order in both sweeps, had gained nothing: round 3's and this round's
	                                    // A/B -- the second sweep then starts with the blocks the cache has already dropped)
};

// ---------------------------------------------------------------------------------------------
// apply_wg_kernel: the product Q = A * Z organised per WORKGROUP for the DRAM access pattern.
// Measured (round 1, tools/pattern_bench*.py in git history, 2^20 x 64, lda = 2^20): a wave that touches 64 columns x 256 B per chunk copies at
// 4.4 TB/s, a workgroup that moves ROWS*4 contiguous bytes of ONE column per instruction (loads and stores) at 5.0-5.2 TB/s,
// independent of the power-of-two column stride.  So:
//   * a workgroup owns ROWS x NP blocks (interleaved over the grid); wave w loads columns {(w+4k)*CPI + ...}, each load
//     instruction = ROWS*4 contiguous bytes per column, written to LDS as As[col][row] (row index XOR-swizzled by bit 3
//     of the column so that the operand reads below are bank-conflict free);
//   * the next block is prefetched into registers before the products of the current one start;
//   * wave w multiplies rows [w*ROWS/4, (w+1)*ROWS/4): A operand = 4-byte LDS reads along k, six-product bf16 split (or exact fp32 / single fp16 product),
//     the result tile overwrites the wave's own rows of As in place;
//   * after a barrier the block leaves through the same linear mapping (UPD: + the C input, loaded linearly as well).
// ENGINE 2 (fp32_tc_nocor, reference src/tcqr32x16.cu:499-560's mode): v_mfma_f32_16x16x32_f16 on fp16-rounded operands, no
// correction terms (range and precision of fp16, like the reference's mode).
// ApplyArgs: nchunks = number of row blocks, nwaves = number of workgroups, cpw unused.
// ---------------------------------------------------------------------------------------------
// GRAMQ: additionally accumulate the Gram matrix Q^T Q of the rows this workgroup produces (bf16-split level, fp64 totals, same
// partial format as gram_bf16_kernel) -- the second sweep of a reorthogonalisation then needs no Gram pass of its own.
// IO = _Float16 (fp16 I/O modes): a.a and a.q hold halves (8-byte aligned rows of four: the host checks lda, ldq % 4 == 0 and the
// bases); a block is widened on its way into As and the result narrowed on its way out -- the products run on the same engines.
template <int ENGINE, int NT, bool UPD, int ROWS, bool GRAMQ, int NW = 4, class IO = float>
__device__ __forceinline__ void apply_wg_body(const ApplyArgs& a) {
	static_assert(!GRAMQ || NW == 4, "the fused Gram accumulation is written for four waves");
	static_assert(sizeof(IO) == 4 || (!UPD && !GRAMQ), "fp16 I/O: the plain product only");
	constexpr int NP = 16 * NT;
	constexpr int RS = ROWS + 4;                         // column stride of As (floats)
	constexpr int LPC = ROWS / 4, CPI = 64 / LPC;        // lanes per column, columns per load instruction
	constexpr int NI = NP / (NW * CPI);                   // load instructions per wave and block
	constexpr int SL = ROWS / (16 * NW);                       // 16-row slabs per wave
	constexpr int ZS = NP + 16;
	// fp32-MFMA engine, triangular 128 x 128 Z: only the part on and right of the diagonal tiles is kept (row k holds the columns
	// j >= 16 (k / 16)), 40 KiB instead of 72 -- two workgroups then share a CU.  The row stride of a 16-row group is its length padded
	// to 16 or 48 mod 64 floats, so that the four rows a B operand touches (k = 4t + q) fall on different banks.
	constexpr bool ZTRI = (ENGINE == 0 && NT == 8 && !UPD);
	auto ztri_stride = [](int g) { return g == 0 ? 144 : (g <= 2 ? 112 : (g <= 4 ? 80 : (g <= 6 ? 48 : 16))); };
	auto ztri_base = [&](int g) { int o = 0; for (int i = 0; i < g; i++) o += 16 * ztri_stride(i); return o; };
	constexpr int KT = (NP + 31) / 32;
	// blocks (kt, ct) of the MFMA-operand image of Z: for a triangular 64 x 64 Z the two blocks (1,0), (1,1) are zero and
	// are not stored -- 18.4 KB instead of 24.6 KB, which lets three workgroups share a CU's LDS
	constexpr bool COMPACT = (!UPD && (NT == 4 || NT == 8));
	constexpr int NB = COMPACT ? KT * NT - KT * (KT - 1) : KT * NT;      // triangular: row kt keeps the blocks ct >= 2 kt
	auto zblk = [](int kt, int ct) { return COMPACT ? kt * NT - kt * (kt - 1) + ct - 2 * kt : kt * NT + ct; };
	auto zblk_kt = [](int b) { int kt = 0; while (COMPACT && kt + 1 < KT && b >= (kt + 1) * NT - (kt + 1) * kt) kt++; return COMPACT ? kt : b / NT; };
	extern __shared__ __attribute__((aligned(16))) char smem[];
	float* As = reinterpret_cast<float*>(smem);
	char* zbase = smem + sizeof(float) * NP * RS;
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int c = lane & 15, q = lane >> 4;
	const int lcol = lane / LPC, lrow = 4 * (lane % LPC);

	const int nblk = a.nchunks, nwg = a.nwaves;
	// (block order does not matter to the Infinity Cache: tools/seq_bench.py; blk() below)
	auto swz = [](int col) { return ((col >> 3) & 1) << 4; };
	auto load_block = [&](f32x4 (&v)[NI], const auto* base, size_t ld, int ncols, int b) {
		using T = std::remove_cv_t<std::remove_pointer_t<decltype(base)>>;
		const size_t row = (size_t)b * ROWS + lrow;
#pragma unroll
		for (int k = 0; k < NI; k++) {
			const int col = (wv + NW * k) * CPI + lcol;
			v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
			if (col < ncols) {
				const T* src = base + (size_t)col * ld + row;
				if (row + 3 < a.m) {
					if constexpr (sizeof(T) == 2) {
						const f16x4 h = *reinterpret_cast<const f16x4*>(src);
						v[k] = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
					} else {
						v[k] = *reinterpret_cast<const f32x4u*>(src);
					}
				} else {
#pragma unroll
					for (int i = 0; i < 4; i++)
						if (row + i < a.m) v[k][i] = (float)src[i];
				}
			}
		}
	};
	const IO* a_in = reinterpret_cast<const IO*>(a.a);

	constexpr int NTRI = (NT * (NT + 1)) / 2;
	f64x4 gtot[GRAMQ ? NTRI : 1];
	if constexpr (GRAMQ) {
#pragma unroll
		for (int t = 0; t < NTRI; t++) gtot[t] = f64x4{0.0, 0.0, 0.0, 0.0};
	}
	// the first block's loads are issued before Z is staged: their HBM latency overlaps the staging work
	// prefetch: the next block in registers (v); 64-row blocks are small enough to keep the block after it in flight as well (v2)
	constexpr bool DEEP = (ROWS == 64 && !UPD) || NW == 8 || sizeof(IO) == 2;      // (half I/O: a block is half the bytes -- keep two in flight)
	f32x4 v[NI], v2[DEEP ? NI : 1];
	// Which blocks are this workgroup's: ordinals w, w + nwg, ... by default.  With a.share set (grid = four workgroups per CU,
	// blocks a multiple of the CUs) the shares are uneven, because the workgroups are (stamps of every workgroup, tools/gram_balance.py
	// apply; 2^20 x 64, equal shares: the pass ends at 84 us, the mean workgroup at 66):
	//   * not equally fast by XCD: workgroup w runs on XCD w mod 8, and the odd XCDs move their blocks 15 % slower than the even ones
	//     (mean end 72 against 62 us on every box looked at) -- the even XCDs then idle for the last 10 us of the pass;
	//   * not equals on a CU: the dispatcher places w, w + nwg/4, w + nwg/2, w + 3 nwg/4 on one CU in four rounds and the CU prefers
	//     the older ones (57 / 63 / 70 / 75 us).
	// Two neighbouring CU groups p = 2 pp (even XCD) and p + 1 (odd XCD) pool their blocks -- ordinals 2 pp + (j & 1) + (nwg/4)(j >> 1),
	// j = 0 .. 2K-1 -- the even group takes the first a.even_share / 128 of that list, and inside a group the four workgroups take
	// contiguous runs of a.share[rank] / 64.  (Every block is computed on its own: Q does not depend on who computes it.)
	const int quarter = nwg >> 2;
	// (compiled into the plain 64-row fp32 kernels only: elsewhere the extra index arithmetic cost the fp16-product kernels a workgroup per CU)
	constexpr bool CAN_SHARE = !UPD && !GRAMQ && ROWS == 64 && NW == 4 && sizeof(IO) == 4;
	const bool uneven = CAN_SHARE && a.share[0] != 0 && (nwg & 7) == 0 && nblk % quarter == 0 && nblk >= 16 * quarter;
	int bi = blockIdx.x, bstep = nwg, bend = nblk, pair_base = 0;
	if (uneven) {
		const int p = (int)blockIdx.x % quarter, rank = (int)blockIdx.x / quarter;
		const int K2 = 2 * (nblk / quarter);
		const int E = (K2 * a.even_share + 64) >> 7;
		const int g0 = (p & 1) ? E : 0, len = (p & 1) ? K2 - E : E;
		int lo = 0, hi = 0, acc = 0;
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int nxt = acc + a.share[r];
			if (r == rank) { lo = (len * acc + 32) >> 6; hi = (r == 3) ? len : (len * nxt + 32) >> 6; }
			acc = nxt;
		}
		pair_base = p & ~1;
		bstep = 1;
		bi = g0 + lo;
		bend = g0 + hi;
	}
	auto blk = [&](int i) {                              // i: position in this workgroup's progression -> block (reverse order: as good as any)
		const int ordinal = uneven ? pair_base + (i & 1) + quarter * (i >> 1) : i;
		return a.forward ? ordinal : nblk - 1 - ordinal;
	};
	if (bi < bend) load_block(v, a_in, a.lda, a.n, blk(bi));
	if constexpr (DEEP) { if (bi + bstep < bend) load_block(v2, a_in, a.lda, a.n, blk(bi + bstep)); }
	// (the verdict word is looked at only now: its round trip runs under the loads just issued; a skipped launch has merely
	// requested a block or two of an input that is valid either way)
	if (a.skip_status && a.skip_status[0] != 0) return;  // uniform over the grid
	if constexpr (sizeof(IO) == 2) {
		// (the half-typed R of the fp16 I/O modes: 4096 values -- a launch of its own costs 4.8 us, here it rides along)
		if (blockIdx.x == 0 && a.r16) {
			_Float16* r16 = reinterpret_cast<_Float16*>(a.r16);
			for (int e = threadIdx.x; e < a.n * a.n; e += 64 * NW) r16[(size_t)(e / a.n) * a.ldr16 + (e % a.n)] = (_Float16)a.r32[e];
		}
	}

	if constexpr (ENGINE == 0) {
		float* Zs = reinterpret_cast<float*>(zbase);
		// (loads first, all in flight, then the LDS writes: see the bf16 engine below)
		constexpr int ZE0 = (NP * NP) / (64 * NW);
		static_assert((NP * NP) % (64 * NW) == 0, "Z elements divide evenly over the workgroup");
		float zv0[ZE0];
#pragma unroll
		for (int u = 0; u < ZE0; u++) zv0[u] = a.z[threadIdx.x + u * 64 * NW];          // idx = j * NP + k: straight copy order
#pragma unroll
		for (int u = 0; u < ZE0; u++) {
			const int idx = threadIdx.x + u * 64 * NW;
			const int k = idx % NP, j = idx / NP;
			if constexpr (ZTRI) {
				const int g = k >> 4;
				if (j >= 16 * g) Zs[ztri_base(g) + (k & 15) * ztri_stride(g) + (j - 16 * g)] = zv0[u];
			} else {
				Zs[k * ZS + j] = zv0[u];
			}
		}
	} else if constexpr (ENGINE == 2) {
		// Zh[kt][ct][lane][8] : B operand of v_mfma_f32_16x16x32_f16, one fp16 image (no correction terms)
		_Float16* Zh = reinterpret_cast<_Float16*>(zbase);
		constexpr int ZE2 = (NB * 64 * 8) / (64 * NW);
		static_assert((NB * 64 * 8) % (64 * NW) == 0, "Z image elements divide evenly over the workgroup");
		float zv2[ZE2];
#pragma unroll
		for (int u = 0; u < ZE2; u++) {                      // (loads first, all in flight: see the bf16 engine below)
			const int idx = threadIdx.x + u * 64 * NW;
			const int jj = idx & 7, l = (idx >> 3) & 63, b = idx >> 9;
			const int kt = zblk_kt(b), ct = COMPACT ? b - (kt * NT - kt * (kt - 1)) + 2 * kt : b % NT;
			const int k = 32 * kt + 8 * (l >> 4) + jj, j = 16 * ct + (l & 15);
			zv2[u] = a.z[(size_t)j * NP + min(k, NP - 1)];
			if (k >= NP) zv2[u] = 0.0f;
		}
#pragma unroll
		for (int u = 0; u < ZE2; u++) Zh[threadIdx.x + u * 64 * NW] = (_Float16)zv2[u];
	} else {
		unsigned short* Zb = reinterpret_cast<unsigned short*>(zbase);
		// every load of a thread first (unconditional, from a clamped index: one L2 round trip for the whole staging instead of one
		// per element -- all workgroups stage Z at the same moment, on the critical path of the pass), then the splits
		constexpr int ZE = (NB * 64 * 8) / (64 * NW);
		static_assert((NB * 64 * 8) % (64 * NW) == 0, "Z image elements divide evenly over the workgroup");
		float zv[ZE];
#pragma unroll
		for (int u = 0; u < ZE; u++) {
			const int idx = threadIdx.x + u * 64 * NW;
			const int jj = idx & 7, l = (idx >> 3) & 63, b = idx >> 9;
			const int kt = zblk_kt(b), ct = COMPACT ? b - (kt * NT - kt * (kt - 1)) + 2 * kt : b % NT;
			const int k = 32 * kt + 8 * (l >> 4) + jj, j = 16 * ct + (l & 15);
			zv[u] = a.z[(size_t)j * NP + min(k, NP - 1)];
			if (k >= NP) zv[u] = 0.0f;
		}
#pragma unroll
		for (int u = 0; u < ZE; u++) {
			const int idx = threadIdx.x + u * 64 * NW;
			unsigned h, m, lo;
			split3(zv[u], h, m, lo);
			Zb[0 * NB * 512 + idx] = (unsigned short)h;      // o = (b * 64 + l) * 8 + jj = idx
			Zb[1 * NB * 512 + idx] = (unsigned short)m;
			Zb[2 * NB * 512 + idx] = (unsigned short)lo;
		}
	}

	for (; bi < bend; bi += bstep) {
		const int b = blk(bi);
#pragma unroll
		for (int k = 0; k < NI; k++) {
			const int col = (wv + NW * k) * CPI + lcol;
			*reinterpret_cast<f32x4*>(&As[col * RS + (lrow ^ swz(col))]) = v[k];
		}
		__syncthreads();                                 // (also orders the Z image on the first pass)
		if constexpr (DEEP) {
#pragma unroll
			for (int k = 0; k < NI; k++) v[k] = v2[k];
			if (bi + 2 * bstep < bend) load_block(v2, a_in, a.lda, a.n, blk(bi + 2 * bstep));
		} else {
			if (bi + bstep < bend) load_block(v, a_in, a.lda, a.n, blk(bi + bstep));
		}
		f32x4 cin[UPD ? NI : 1];
		if constexpr (UPD) load_block(cin, a.q, a.ldq, a.n_out, b);

#pragma unroll
		for (int s = 0; s < SL; s++) {
			const int rb = wv * (ROWS / NW) + 16 * s;
			f32x4 acc[NT];
#pragma unroll
			for (int ct = 0; ct < NT; ct++) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
			if constexpr (ENGINE == 0) {
				const float* Zs = reinterpret_cast<const float*>(zbase);
				// operands of k-step t+1 are requested before the products of k-step t are issued: an MFMA then never waits for its own
				// LDS read (one read per product otherwise: 96 instead of 32 cycles per product)
				auto fetch = [&](int t, float& av, float (&bv)[NT]) {
					const int k = 4 * t + q;
					av = As[k * RS + ((rb + c) ^ swz(k))];
#pragma unroll
					for (int ct = 0; ct < NT; ct++) {
						bv[ct] = 0.0f;
						if (UPD || 4 * t <= 16 * ct + 15) {
							if constexpr (ZTRI) bv[ct] = Zs[ztri_base(t / 4) + (4 * (t % 4) + q) * ztri_stride(t / 4) + 16 * (ct - t / 4) + c];
							else bv[ct] = Zs[k * ZS + 16 * ct + c];
						}
					}
				};
				float a0, a1, b0[NT], b1[NT];
				fetch(0, a0, b0);
				__builtin_amdgcn_sched_barrier(0);       // (keeps the scheduler from sinking the reads back next to their products)
#pragma unroll
				for (int t = 0; t < NP / 4; t += 2) {
					if (t + 1 < NP / 4) fetch(t + 1, a1, b1);
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int ct = 0; ct < NT; ct++)
						if (UPD || 4 * t <= 16 * ct + 15) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[ct], acc[ct], 0, 0, 0);
					__builtin_amdgcn_sched_barrier(0);
					if (t + 2 < NP / 4) fetch(t + 2, a0, b0);
					__builtin_amdgcn_sched_barrier(0);
					if (t + 1 < NP / 4) {
#pragma unroll
						for (int ct = 0; ct < NT; ct++)
							if (UPD || 4 * (t + 1) <= 16 * ct + 15) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[ct], acc[ct], 0, 0, 0);
					}
				}
			} else if constexpr (ENGINE == 2) {
				// fp32_tc_nocor: both operands rounded to fp16 (round to nearest even), ONE product per block, fp32 accumulation
				const _Float16* Zh = reinterpret_cast<const _Float16*>(zbase);
				f16x8 ah[KT];
#pragma unroll
				for (int kt = 0; kt < KT; kt++)
#pragma unroll
					for (int e = 0; e < 8; e++) {
						const int k0 = 32 * kt + 8 * q + e;
						ah[kt][e] = (_Float16)((k0 < NP) ? As[k0 * RS + ((rb + c) ^ swz(k0))] : 0.0f);
					}
#pragma unroll
				for (int kt = 0; kt < KT; kt++)
#pragma unroll
					for (int ct = 0; ct < NT; ct++) {
						if (UPD || 32 * kt <= 16 * ct + 15) {         // triangular Z: block (kt, ct) is zero when all its k > all its j
							const f16x8 bh = *reinterpret_cast<const f16x8*>(&Zh[(zblk(kt, ct) * 64 + lane) * 8]);
							acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kt], bh, acc[ct], 0, 0, 0);
						}
					}
			} else {
				const unsigned short* Zb = reinterpret_cast<const unsigned short*>(zbase);
				constexpr int PS = NB * 512;
				bf16x8 ah[KT], am[KT], al[KT];
				{
					float x[8 * KT];
					unsigned hh[4 * KT], mm[4 * KT], ll[4 * KT];
#pragma unroll
					for (int kt = 0; kt < KT; kt++)
#pragma unroll
						for (int e = 0; e < 8; e++) {
							const int k0 = 32 * kt + 8 * q + e;
							x[8 * kt + e] = (k0 < NP) ? As[k0 * RS + ((rb + c) ^ swz(k0))] : 0.0f;
						}
					split3_pairs<4 * KT>(x, hh, mm, ll);
#pragma unroll
					for (int kt = 0; kt < KT; kt++) {
						ah[kt] = __builtin_bit_cast(bf16x8, u32x4{hh[4 * kt], hh[4 * kt + 1], hh[4 * kt + 2], hh[4 * kt + 3]});
						am[kt] = __builtin_bit_cast(bf16x8, u32x4{mm[4 * kt], mm[4 * kt + 1], mm[4 * kt + 2], mm[4 * kt + 3]});
						al[kt] = __builtin_bit_cast(bf16x8, u32x4{ll[4 * kt], ll[4 * kt + 1], ll[4 * kt + 2], ll[4 * kt + 3]});
					}
				}
				auto pair = [&](auto KTc, auto CTc, auto KTd, auto CTd) {
					constexpr int k0 = decltype(KTc)::value, c0 = decltype(CTc)::value;
					constexpr int k1 = decltype(KTd)::value, c1 = decltype(CTd)::value;
					const int o0 = (zblk(k0, c0) * 64 + lane) * 8, o1 = (zblk(k1, c1) * 64 + lane) * 8;
					const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(&Zb[0 * PS + o0]);
					const bf16x8 bm0 = *reinterpret_cast<const bf16x8*>(&Zb[1 * PS + o0]);
					const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(&Zb[2 * PS + o0]);
					const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(&Zb[0 * PS + o1]);
					const bf16x8 bm1 = *reinterpret_cast<const bf16x8*>(&Zb[1 * PS + o1]);
					const bf16x8 bl1 = *reinterpret_cast<const bf16x8*>(&Zb[2 * PS + o1]);
					acc[c0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[k0], bm0, acc[c0], 0, 0, 0);
					acc[c1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[k1], bm1, acc[c1], 0, 0, 0);
					acc[c0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k0], bl0, acc[c0], 0, 0, 0);
					acc[c1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k1], bl1, acc[c1], 0, 0, 0);
					acc[c0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[k0], bh0, acc[c0], 0, 0, 0);
					acc[c1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[k1], bh1, acc[c1], 0, 0, 0);
					acc[c0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k0], bm0, acc[c0], 0, 0, 0);
					acc[c1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k1], bm1, acc[c1], 0, 0, 0);
					acc[c0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[k0], bh0, acc[c0], 0, 0, 0);
					acc[c1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[k1], bh1, acc[c1], 0, 0, 0);
					acc[c0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k0], bh0, acc[c0], 0, 0, 0);
					acc[c1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[k1], bh1, acc[c1], 0, 0, 0);
				};
				using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
				using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
				if constexpr (NT == 1) {
					pair(I0{}, I0{}, I0{}, I0{});
					acc[0] = acc[0] * 0.5f;
				} else if constexpr (NT == 2) {
					pair(I0{}, I0{}, I0{}, I1{});
				} else if constexpr (NT == 3) {
					pair(I0{}, I0{}, I0{}, I1{});
					pair(I0{}, I2{}, I1{}, I2{});
				} else if constexpr (NT == 4) {
					pair(I0{}, I0{}, I0{}, I1{});
					pair(I0{}, I2{}, I0{}, I3{});
					pair(I1{}, I2{}, I1{}, I3{});
					if constexpr (UPD) pair(I1{}, I0{}, I1{}, I1{});
				} else {
					// triangular 128 x 128 Z: k-tile kt meets the column tiles ct >= 2 kt, taken two at a time
					static_assert(NT == 8 && !UPD, "wide apply: 128 columns, triangular Z");
					static_for<0, KT>([&](auto kk) {
						constexpr int kt = decltype(kk)::value;
						static_for<kt, NT / 2>([&](auto cc) {
							constexpr int ct = 2 * decltype(cc)::value;
							pair(std::integral_constant<int, kt>{}, std::integral_constant<int, ct>{},
							     std::integral_constant<int, kt>{}, std::integral_constant<int, ct + 1>{});
						});
					});
				}
			}
			// the result tile replaces this wave's rows of As (same swizzle): D layout col = 16ct + c, rows rb + 4q + i
#pragma unroll
			for (int ct = 0; ct < NT; ct++) {
				const int col = 16 * ct + c;
				*reinterpret_cast<f32x4*>(&As[col * RS + ((rb + 4 * q) ^ swz(col))]) = acc[ct];
			}
		}
		if constexpr (GRAMQ) {
			// Gram tiles of this wave's freshly written rows (ROWS/4 rows = ROWS/128 K-steps of 32): lane (c,q) reads logical rows
			// 8q..8q+7 of column 16t+c (two 16-byte reads; the XOR swizzle keeps 8-row groups contiguous), one MFMA chain from zero per
			// K-step, added to fp64 totals -- exactly gram_bf16_kernel's arithmetic on the values that are about to be stored
#pragma unroll
			for (int ks = 0; ks < ROWS / 128; ks++) {
				const int rbk = wv * (ROWS / 4) + 32 * ks + 8 * q;
				bf16x8 oh[NT], om[NT], ol[NT];
#pragma unroll
				for (int t = 0; t < NT; t++) {
					const int col = 16 * t + c;
					const f32x4 x0 = *reinterpret_cast<const f32x4*>(&As[col * RS + (rbk ^ swz(col))]);
					const f32x4 x1 = *reinterpret_cast<const f32x4*>(&As[col * RS + ((rbk + 4) ^ swz(col))]);
					u32x4 hh, mm, ll;
					unsigned h, m, lo;
					split3_pair(x0[0], x0[1], h, m, lo); hh[0] = h; mm[0] = m; ll[0] = lo;
					split3_pair(x0[2], x0[3], h, m, lo); hh[1] = h; mm[1] = m; ll[1] = lo;
					split3_pair(x1[0], x1[1], h, m, lo); hh[2] = h; mm[2] = m; ll[2] = lo;
					split3_pair(x1[2], x1[3], h, m, lo); hh[3] = h; mm[3] = m; ll[3] = lo;
					oh[t] = __builtin_bit_cast(bf16x8, hh);
					om[t] = __builtin_bit_cast(bf16x8, mm);
					ol[t] = __builtin_bit_cast(bf16x8, ll);
				}
				f32x4 gacc[NTRI];
#pragma unroll
				for (int t = 0; t < NTRI; t++) gacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int pass = 3; pass < 9; pass++) {       // mm hl lh hm mh hh (smallest first), as in gram_bf16_kernel
					int idx = 0;
#pragma unroll
					for (int ti = 0; ti < NT; ti++)
#pragma unroll
						for (int tj = ti; tj < NT; tj++) {
							const bf16x8 av = (pass == 4 || pass == 6 || pass == 8) ? oh[ti] : ((pass == 3 || pass == 7) ? om[ti] : ol[ti]);
							const bf16x8 bv = (pass == 5 || pass == 7 || pass == 8) ? oh[tj] : ((pass == 3 || pass == 6) ? om[tj] : ol[tj]);
							gacc[idx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, gacc[idx], 0, 0, 0);
							idx++;
						}
				}
#pragma unroll
				for (int t = 0; t < NTRI; t++)
#pragma unroll
					for (int r = 0; r < 4; r++) gtot[t][r] += (double)gacc[t][r];
			}
		}
		__syncthreads();
		{
			const size_t row = (size_t)b * ROWS + lrow;
			const int nout = UPD ? a.n_out : a.n;
#pragma unroll
			for (int k = 0; k < NI; k++) {
				const int col = (wv + NW * k) * CPI + lcol;
				if (col < nout) {
					f32x4 x = *reinterpret_cast<const f32x4*>(&As[col * RS + (lrow ^ swz(col))]);
					if constexpr (UPD) x += cin[k];
					if constexpr (sizeof(IO) == 2) {
						_Float16* dh = reinterpret_cast<_Float16*>(a.q) + (size_t)col * a.ldq + row;
						if (row + 3 < a.m) {
							const f16x4 h = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
							__builtin_nontemporal_store(h, reinterpret_cast<f16x4*>(dh));
						} else {
#pragma unroll
							for (int i = 0; i < 4; i++)
								if (row + i < a.m) dh[i] = (_Float16)x[i];
						}
						continue;
					}
					float* dst = a.q + (size_t)col * a.ldq + row;
					if (row + 3 < a.m) {
						// Q must not displace A from the Infinity Cache: nontemporal.  The updated panel of a coupling step (UPD) is read
						// again at once by its own Gram and apply passes: plain stores keep it there.
						if (UPD || a.plain_q) *reinterpret_cast<f32x4u*>(dst) = x;
						else __builtin_nontemporal_store(x, reinterpret_cast<f32x4u*>(dst));
					} else {
#pragma unroll
						for (int i = 0; i < 4; i++)
							if (row + i < a.m) dst[i] = x[i];
					}
				}
			}
		}
		__syncthreads();
	}
	if constexpr (GRAMQ) {
		// workgroup sum in fp64 (LDS aliases As / the Z image: every wave has passed the loop's last barrier), as in gram_bf16_kernel
		double* red = reinterpret_cast<double*>(smem);   // [2][NTRI*256]
		if (wv >= 2) {
#pragma unroll
			for (int t = 0; t < NTRI; t++)
#pragma unroll
				for (int r = 0; r < 4; r++) red[(size_t)(wv - 2) * NTRI * 256 + (t * 4 + r) * 64 + lane] = gtot[t][r];
		}
		__syncthreads();
		if (wv < 2) {
#pragma unroll
			for (int t = 0; t < NTRI; t++)
#pragma unroll
				for (int r = 0; r < 4; r++) gtot[t][r] += red[(size_t)wv * NTRI * 256 + (t * 4 + r) * 64 + lane];
		}
		__syncthreads();
		if (wv == 1) {
#pragma unroll
			for (int t = 0; t < NTRI; t++)
#pragma unroll
				for (int r = 0; r < 4; r++) red[(t * 4 + r) * 64 + lane] = gtot[t][r];
		}
		__syncthreads();
		if (wv == 0) {
			double* out = a.gpart + (size_t)blockIdx.x * NTRI * 256;
#pragma unroll
			for (int t = 0; t < NTRI; t++)
#pragma unroll
				for (int r = 0; r < 4; r++) part_store(&out[(t * 4 + r) * 64 + lane], gtot[t][r] + red[(t * 4 + r) * 64 + lane]);
		}
	}
}

template <int ENGINE, int NT, bool UPD, int ROWS, bool GRAMQ = false>
__global__ __launch_bounds__(256) void apply_wg_kernel(const ApplyArgs a) {
	if constexpr (UPD) {
		ApplyArgs b = a;                                 // (uniform: the slice's own arguments, in scalar registers)
		if (a.multi_cols > 0) {
			const int j = blockIdx.y;
			b.q = a.q + (size_t)j * 64 * a.ldq; b.z = a.z + (size_t)j * 4096; b.n_out = min(64, a.multi_cols - 64 * j);
		}
		apply_wg_body<ENGINE, NT, UPD, ROWS, false>(b);
	} else {
		apply_wg_body<ENGINE, NT, UPD, ROWS, false>(a);
	}
}
// the variant that also accumulates Q^T Q: two waves per SIMD (the register allocator is told to stay within 256 registers)
// fp16 I/O modes: the plain product with halves at both ends (tsqr_mi_qr_f16's native path)
template <int ENGINE, int NT, int ROWS>
__global__ __launch_bounds__(256) void apply_wg_h_kernel(const ApplyArgs a) {
	apply_wg_body<ENGINE, NT, false, ROWS, false, 4, _Float16>(a);
}
template <int ENGINE, int NT, bool UPD, int ROWS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void apply_wg_gramq_kernel(const ApplyArgs a) {
	apply_wg_body<ENGINE, NT, UPD, ROWS, true>(a);
}

// 64 < n <= 128 in one pass: Q (m x n) = A (m x n) * Z with a triangular 128 x 128 Z.  Eight waves own 128-row blocks (one workgroup
// per CU: the block and the operand image of Z fill most of its LDS); wave w multiplies rows 16 w .. 16 w + 15.
template <int ENGINE>
__global__ __launch_bounds__(512) void apply_wide_kernel(const ApplyArgs a) {
	apply_wg_body<ENGINE, 8, false, 128, false, 8>(a);
}
// fp32-MFMA engine (fp32_notc): its product phase is four times longer per block, and with a single resident workgroup it does not
// overlap the memory phases (320 us).  Four waves on 64-row blocks with the compact triangular Z: two workgroups per CU.
__global__ __launch_bounds__(256, 2) void apply_wide_f32_kernel(const ApplyArgs a) {
	apply_wg_body<0, 8, false, 64, false, 4>(a);
}

// R <- R2 * R1 (n x n upper triangular, fp64 accumulation).  r1 is a packed copy (ld n) of the old R.
__global__ __launch_bounds__(256) void rmul_kernel(float* __restrict__ r, size_t ldr, const float* __restrict__ r2, size_t ldr2,
                                                   const float* __restrict__ r1, size_t ldr1, int n) {
	const size_t total = (size_t)n * n;
	for (size_t idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)256 * gridDim.x) {
		const int i = (int)(idx % n), j = (int)(idx / n);
		double acc = 0.0;
		if (i <= j)
			for (int k = i; k <= j; k++) acc += (double)r2[(size_t)k * ldr2 + i] * (double)r1[(size_t)j * ldr1 + k];
		r[(size_t)j * ldr + i] = (float)acc;
	}
}

// The same product for n <= 64 in one workgroup of sixteen waves: both factors staged in LDS as fp64 (one global round trip instead
// of one per term), wave w forms the 16 x 16 tile (w >> 2, w & 3) of R2 R1 with v_mfma_f64_16x16x4_f64 over the k range in which
// both triangular factors are non-zero.  12-14 us -> ~5 us in a reorthogonalised call.
__global__ __launch_bounds__(1024) void rmul64_kernel(float* __restrict__ r, size_t ldr, const float* __restrict__ r2, size_t ldr2,
                                                      const float* __restrict__ r1, size_t ldr1, int n,
                                                      const float* __restrict__ r3 = nullptr, size_t ldr3 = 0) {
	// r3 != nullptr (round 4, shifted CholeskyQR3: R = R3 R2 R1): both products in this launch, the inner one handed on through LDS in
	// fp64 (as two launches the chain cost 2 x 6.1 us at the end of the call, behind the last apply pass)
	__shared__ double A2[64 * 65], A1[64 * 65];          // A2[i * 65 + k] = R2[i][k], A1[k * 65 + j] = R1[k][j]; zero outside the upper triangles
	const int t = threadIdx.x;
	float v2[4], v1[4], v3[4];
#pragma unroll
	for (int u = 0; u < 4; u++) {                        // (loads first, all in flight)
		const int e = t + 1024 * u, i = e & 63, j = e >> 6;      // entry (i, j) of either factor, column-major
		const bool in = i <= j && j < n;
		v2[u] = in ? r2[(size_t)j * ldr2 + i] : 0.0f;
		v1[u] = in ? r1[(size_t)j * ldr1 + i] : 0.0f;
		v3[u] = (in && r3) ? r3[(size_t)j * ldr3 + i] : 0.0f;
	}
#pragma unroll
	for (int u = 0; u < 4; u++) {
		const int e = t + 1024 * u, i = e & 63, j = e >> 6;
		A2[i * 65 + j] = (double)v2[u];
		A1[i * 65 + j] = (double)v1[u];
	}
	__syncthreads();
	const int w = t >> 6, l = t & 63, ti = w >> 2, tj = w & 3, li = l & 15, lq = l >> 4;
	f64x4 c = f64x4{0.0, 0.0, 0.0, 0.0};                // c[reg] = (R2 R1)[16 ti + lq + 4 reg][16 tj + li]
	if (ti <= tj) {
		for (int ks = 4 * ti; ks < 4 * (tj + 1); ks++)   // R2[i][k] = 0 for k < i, R1[k][j] = 0 for k > j
			c = __builtin_amdgcn_mfma_f64_16x16x4f64(A2[(16 * ti + li) * 65 + 4 * ks + lq], A1[(4 * ks + lq) * 65 + 16 * tj + li], c, 0, 0, 0);
	}
	if (r3) {                                            // (uniform)
		__syncthreads();                                 // every wave is done with R2 and R1
#pragma unroll
		for (int reg = 0; reg < 4; reg++) {
			const int i = 16 * ti + lq + 4 * reg, j = 16 * tj + li;
			A1[i * 65 + j] = (i <= j) ? c[reg] : 0.0;    // P = R2 R1 (upper triangular), fp64
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const int e = t + 1024 * u, i = e & 63, j = e >> 6;
			A2[i * 65 + j] = (double)v3[u];
		}
		__syncthreads();
		c = f64x4{0.0, 0.0, 0.0, 0.0};
		if (ti <= tj) {
			for (int ks = 4 * ti; ks < 4 * (tj + 1); ks++)
				c = __builtin_amdgcn_mfma_f64_16x16x4f64(A2[(16 * ti + li) * 65 + 4 * ks + lq], A1[(4 * ks + lq) * 65 + 16 * tj + li], c, 0, 0, 0);
		}
	}
#pragma unroll
	for (int reg = 0; reg < 4; reg++) {
		const int i = 16 * ti + lq + 4 * reg, j = 16 * tj + li;
		if (i < n && j < n) r[(size_t)j * ldr + i] = (i <= j) ? (float)c[reg] : 0.0f;
	}
}

// completion signal: one thread stores seq to device-visible pinned host memory.  Enqueued behind the last kernel of a call
// so that the host can spin on the word instead of paying a stream synchronisation (stream order makes it a full barrier).
__global__ void host_flag_kernel(unsigned* __restrict__ host_flag, unsigned seq) {
	*reinterpret_cast<volatile unsigned*>(host_flag) = seq;
}
// row-partitioned loop entries: the eligibility vote of a chained stream travels with the first Gram all-reduce of the stream (one more
// double behind the row count); set_f64_kernel writes this rank's 1.0, vote_out_kernel puts the all-reduced count (<= 255 ranks) and a
// 24-bit sequence number into ONE pinned host word
__global__ void set_f64_kernel(double* __restrict__ p, double v) { *p = v; }
__global__ void vote_out_kernel(const double* __restrict__ count, unsigned* __restrict__ host_word, unsigned seq) {
	const double v = *count;
	const unsigned n = (v >= 0.0 && v < 255.5) ? (unsigned)(v + 0.5) : 255u;
	*reinterpret_cast<volatile unsigned*>(host_word) = ((seq & 0xffffffu) << 8) | n;
}

// fp16 I/O modes (reference mtk::qr::qr<fp16_notc | fp16_tc_nocor>: io type half, src/tsqr.hpp:38-39): the boundary converts, the
// factorisation runs on the fp32 pipeline.  One thread moves eight consecutive rows of one column: a 16-byte fp16 access when the
// fp16 side is 16-byte aligned (vec = 1: base pointer and leading dimension), element by element otherwise; ragged tails by element.
__global__ __launch_bounds__(256) void widen_f16_kernel(float* __restrict__ dst, size_t ldd, const _Float16* __restrict__ src, size_t lds,
                                                        size_t rows, int cols, int vec) {
	const size_t rb = (rows + 7) / 8, total = rb * (size_t)cols;
	for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)256 * gridDim.x) {
		const size_t col = idx / rb, row = (idx % rb) * 8;
		const _Float16* s = src + col * lds + row;
		float* d = dst + col * ldd + row;
		if (vec && row + 8 <= rows) {
			const f16x8 h = __builtin_nontemporal_load(reinterpret_cast<const f16x8*>(s));
			*reinterpret_cast<f32x4*>(d) = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
			*reinterpret_cast<f32x4*>(d + 4) = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
		} else {
			for (int i = 0; i < 8 && row + i < rows; i++) d[i] = (float)s[i];
		}
	}
}
// (round to nearest even; values beyond the fp16 range become infinities, as a half-typed R does in the reference)
__global__ __launch_bounds__(256) void narrow_f16_kernel(_Float16* __restrict__ dst, size_t ldd, const float* __restrict__ src, size_t lds,
                                                         size_t rows, int cols, int vec) {
	const size_t rb = (rows + 7) / 8, total = rb * (size_t)cols;
	for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)256 * gridDim.x) {
		const size_t col = idx / rb, row = (idx % rb) * 8;
		const float* s = src + col * lds + row;
		_Float16* d = dst + col * ldd + row;
		if (vec && row + 8 <= rows) {
			const f32x4 a = *reinterpret_cast<const f32x4u*>(s), b = *reinterpret_cast<const f32x4u*>(s + 4);
			const f16x8 h = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
			__builtin_nontemporal_store(h, reinterpret_cast<f16x8*>(d));
		} else {
			for (int i = 0; i < 8 && row + i < rows; i++) d[i] = (_Float16)s[i];
		}
	}
}

__global__ __launch_bounds__(256) void copy2d_kernel(float* __restrict__ dst, size_t ldd, const float* __restrict__ src, size_t lds,
                                                     int rows, int cols) {
	const size_t total = (size_t)rows * cols;
	for (size_t idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)256 * gridDim.x) {
		const int i = (int)(idx % rows), j = (int)(idx / rows);
		dst[(size_t)j * ldd + i] = src[(size_t)j * lds + i];
	}
}

__global__ __launch_bounds__(256) void zero_lower_kernel(float* __restrict__ r, size_t ldr, int n) {
	const size_t total = (size_t)n * n;
	for (size_t idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)256 * gridDim.x) {
		const int i = (int)(idx % n), j = (int)(idx / n);
		if (i > j) r[(size_t)j * ldr + i] = 0.0f;
	}
}

}  // namespace tsqrmi
