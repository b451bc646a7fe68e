// tsqr_wide.hip -- 64 < n <= 128 columns factored as ONE Cholesky-QR panel (included by tsqr_mi.hip behind tsqr_kernels.hip).
//
// The panel path couples two 64-column panels by block Gram-Schmidt (cross_kernel + update + second Gram pass: seven passes over
// panel-sized data, 2.95 GB at 2^20 x 128).  For a well-conditioned matrix the same factorisation is
//     G = A^T A (128 x 128)  ->  R = chol(G) in two 64 x 64 blocks  ->  Q = A * inverse(R)
// with ONE read of A for G and one read + one write for Q (1.6 GB):
//   gram_wide_kernel   : all 36 Gram tiles of a 64-row x 128-column block; the block is split ONCE into its three bf16 images in LDS
//                        and every wave takes nine tile pairs (tile rows w and 7-w) with both MFMA operands read from LDS
//   chol_wide_kernel   : one workgroup of sixteen waves, one launch: chol(G11) -> R11, Z11 [chol_body16]; R12 = Z11^T G12, G22' = G22 - R12^T R12;
//                        chol(G22') -> R22, Z22 [chol_body16]; Z12 = -Z11 R12 Z22, the 128 x 128 Z for the apply pass, the verdict
//                        over BOTH blocks (scaled conditioning S with the ORIGINAL diagonal of G, pivot ratios)
//   apply_wide_kernel  : Q = A * Z  (tsqr_kernels.hip)
// The acceptance rule is the bf16-split Gram level's; a rejected factorisation leaves A untouched and the
// caller falls back to the panel path.  Plays the role of reference src/blockqr.cu:45-178 for two panels at once.
namespace tsqrmi {

struct GramWideArgs {
	const float* a; size_t lda; size_t m; int n;
	int blk0, nblk;                      // this launch covers the 64-row blocks blk0 .. blk0 + nblk - 1
	double* part;                        // [gridDim.x][36][256] doubles, tile order of wide_tile() below
	const unsigned* skip_status;
	unsigned* announce; unsigned announce_seq;   // as in GramArgs: completion word of the call in front of this one
};

// destination tile of the pair (ti, tj), ti <= tj, in the summed array: [G11: 10 tiles, NT = 4 order][G22: 10 tiles][G12: 16 tiles row-major]
// -- the first diagonal block is then directly a chol_body16 input
__host__ __device__ constexpr int tri4(int ti, int tj) { return ti * 4 - (ti * (ti - 1)) / 2 + (tj - ti); }
__host__ __device__ constexpr int wide_tile(int ti, int tj) {
	return (tj < 4) ? tri4(ti, tj) : ((ti >= 4) ? 10 + tri4(ti - 4, tj - 4) : 20 + ti * 4 + (tj - 4));
}
constexpr int WIDE_TILES = 36;
constexpr int WIDE_G22 = 10 * 256, WIDE_G12 = 20 * 256;   // offsets (doubles) into the summed array
// Column stride of a bf16 image in dwords: 64 rows (32 dwords) + 8 of padding.  Measured with tools/lds_pattern.py (lane (c, q)
// reads 16 bytes at dword A c + 4 q): ds_read_b128 is free of bank conflicts for A = 40 but not for A = 36, 44, 52, 68 ... -- the
// sixteen lanes of a pass must agree in address bit 4 (or be pairwise contiguous); A = 36 gave SQ_LDS_BANK_CONFLICT = 8.3 M per launch.
constexpr int GW_CS = 40;

// Work split of the 36 tile pairs over four wave roles, nine pairs each, chosen so that a role touches few distinct column tiles
// (every tile costs three 16-byte LDS reads per lane and K-step: 63 tile reads for all four roles instead of 102 with a row-wise
// split -- the operand reads from LDS, not the MFMAs, bound this kernel):
//   role 0: rows {0,1,2} x columns {5,6,7}      role 1: rows {0,1,2} x columns {2,3,4}
//   role 2: rows {3,4} x columns {3..7}          role 3: the triangles of {5,6,7} and of {0,1}
// A role keeps up to three "row" tiles resident in registers (three bf16 images each) and streams its column tiles one at a time;
// a diagonal pair takes both operands from the streamed tile.
struct WideRole {
	int nres; int res[3];                // resident row tiles
	int ncol; int col[5];                // streamed column tiles
	int pa[9]; int pb[9];                // pair p = (row tile pa[p], column tile col[pb[p]]), pa[p] = index into res, or -1: the column tile itself
};
__host__ __device__ constexpr WideRole wide_role(int w) {
	return w == 0 ? WideRole{3, {0, 1, 2}, 3, {5, 6, 7, 0, 0}, {0, 1, 2, 0, 1, 2, 0, 1, 2}, {0, 0, 0, 1, 1, 1, 2, 2, 2}}
	     : w == 1 ? WideRole{3, {0, 1, 2}, 3, {2, 3, 4, 0, 0}, {0, 1, -1, 0, 1, 2, 0, 1, 2}, {0, 0, 0, 1, 1, 1, 2, 2, 2}}
	     : w == 2 ? WideRole{2, {3, 4, 0}, 5, {3, 4, 5, 6, 7}, {-1, 0, -1, 0, 1, 0, 1, 0, 1}, {0, 1, 1, 2, 2, 3, 3, 4, 4}}
	              : WideRole{3, {5, 6, 0}, 5, {5, 6, 7, 0, 1}, {-1, 0, -1, 0, 1, -1, -1, 2, -1}, {0, 1, 1, 2, 2, 2, 3, 4, 4}};
}
__host__ __device__ constexpr int wide_role_ti(int w, int p) { const WideRole R = wide_role(w); return R.pa[p] < 0 ? R.col[R.pb[p]] : R.res[R.pa[p]]; }
__host__ __device__ constexpr int wide_role_tj(int w, int p) { const WideRole R = wide_role(w); return R.col[R.pb[p]]; }

// the MFMA section of role W for one K-step (32 rows) of the block whose images start at `img`
template <int W>
__device__ __forceinline__ void gram_wide_step(f32x4 (&acc)[9], const unsigned* __restrict__ img, int ks, int lane) {
	constexpr WideRole R = wide_role(W);
	constexpr int IMG = 128 * GW_CS;                     // dwords per image
	const int c = lane & 15, q = lane >> 4;
	const unsigned* base = &img[c * GW_CS + 16 * ks + 4 * q];
	bf16x8 rh[R.nres], rm[R.nres], rl[R.nres];
	static_for<0, R.nres>([&](auto tt) {
		constexpr int t = decltype(tt)::value;
		const unsigned* src = base + 16 * R.res[t] * GW_CS;
		rh[t] = *reinterpret_cast<const bf16x8*>(src);
		rm[t] = *reinterpret_cast<const bf16x8*>(src + IMG);
		rl[t] = *reinterpret_cast<const bf16x8*>(src + 2 * IMG);
	});
	// the images of column tile b+1 are requested before the products of column tile b are issued (scheduling barriers keep the
	// compiler from sinking the reads back next to their first use): a product never waits for its own LDS read
	bf16x8 bimg[2][3];
	auto fetch_col = [&](int slot, int tile) {
		const unsigned* src = base + 16 * tile * GW_CS;
		bimg[slot][0] = *reinterpret_cast<const bf16x8*>(src);
		bimg[slot][1] = *reinterpret_cast<const bf16x8*>(src + IMG);
		bimg[slot][2] = *reinterpret_cast<const bf16x8*>(src + 2 * IMG);
	};
	fetch_col(0, R.col[0]);
	__builtin_amdgcn_sched_barrier(0);
	static_for<0, R.ncol>([&](auto bb) {
		constexpr int b = decltype(bb)::value;
		if constexpr (b + 1 < R.ncol) fetch_col((b + 1) & 1, R.col[b + 1]);
		__builtin_amdgcn_sched_barrier(0);
		const bf16x8 bh = bimg[b & 1][0], bm = bimg[b & 1][1], bl = bimg[b & 1][2];
		// six of the nine partial products, smallest first: mm hl lh hm mh hh (as gram_bf16_kernel)
		static_for<3, 9>([&](auto pp) {
			constexpr int pass = decltype(pp)::value;
			static_for<0, 9>([&](auto ii) {
				constexpr int p = decltype(ii)::value;
				if constexpr (R.pb[p] == b) {
					constexpr int ta = R.pa[p];
					bf16x8 ah_, am_, al_;
					if constexpr (ta < 0) { ah_ = bh; am_ = bm; al_ = bl; }
					else { ah_ = rh[ta]; am_ = rm[ta]; al_ = rl[ta]; }
					const bf16x8 av = (pass == 4 || pass == 6 || pass == 8) ? ah_ : ((pass == 3 || pass == 7) ? am_ : al_);
					const bf16x8 bv = (pass == 5 || pass == 7 || pass == 8) ? bh : ((pass == 3 || pass == 6) ? bm : bl);
					acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[p], 0, 0, 0);
				}
			});
		});
		__builtin_amdgcn_sched_barrier(0);
	});
}

// Eight waves, one workgroup per CU.  A 64-row x 128-column block is split ONCE into its three bf16 images in LDS (double buffered:
// one barrier per block, and that barrier waits for LDS only); waves 0-3 take the first 32 rows (K-step), waves 4-7 the second, wave
// role = wave & 3.  Two blocks per workgroup are kept in flight in registers; the register sets rotate by unrolling, never by moves
// (a move waits for the loads still in flight), and the loads of a full block are unconditional and back to back (a join between
// them makes the compiler wait for each load before it issues the next).  Measured at 2^20 x 128 (537 MB, beyond the Infinity
// Cache): 137 us = 3.9 TB/s; the loads alone take 130 us in this geometry.
// FAST: every block is full (64 rows inside the matrix, n == 128, 128 lda floats < 4 GiB).  !FAST: the general form (ragged rows, n < 128), one block in
// flight -- the host sends only what FAST cannot take there.
constexpr int GW_LDS_BYTES = 2 * 3 * 128 * GW_CS * 4;
// (the body takes its workgroup number and the number of Gram workgroups as arguments: gram_wide_chain_kernel runs it on a part of its grid)
template <bool FAST>
__device__ __forceinline__ void gram_wide_body(const GramWideArgs& a, unsigned* gw_img, const int wg, const int nwg) {
	if (a.skip_status && a.skip_status[0] != 0) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: the per-role MFMA sections are scalar branches
	const int role = wv & 3, ks = wv >> 2;
	const int lcol = lane >> 4, lrow = 4 * (lane & 15);
	constexpr int IMG = 128 * GW_CS;
	f32x4 acc[9];
	f64x4 tot[9];
#pragma unroll
	for (int t = 0; t < 9; t++) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; tot[t] = f64x4{0.0, 0.0, 0.0, 0.0}; }
	// FAST: buffer loads -- descriptor on the block (wave-uniform), one loop-invariant 32-bit per-thread offset, the column group in the
	// scalar offset.  (With flat loads the allocator recycled registers of the set in flight for the 64-bit addresses and the wait it
	// then needs drained the queue at every second block: s_waitcnt vmcnt(0) at the loop head; see gram_blk_kernel.)
	const unsigned voff = (unsigned)(((size_t)lcol * a.lda + lrow) * sizeof(float));
	const unsigned soff0 = (unsigned)((size_t)(4 * wv) * a.lda * sizeof(float)), soffk = (unsigned)(32 * a.lda * sizeof(float));
	auto load_block = [&](f32x4 (&v)[4], int b) {
		if constexpr (FAST) {
			// (no branch at all: past the end the last block is simply loaded again and never used)
			const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.a + (size_t)(a.blk0 + min(b, a.nblk - 1)) * 64), 0, -1, 0x00020000);
#pragma unroll
			for (int k = 0; k < 4; k++) v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff0 + k * soffk, 0));
		} else {
			const size_t row = (size_t)(a.blk0 + b) * 64 + lrow;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int col = (wv + 8 * k) * 4 + lcol;     // one instruction of a wave: 4 columns x 256 contiguous bytes
				v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
				if (b < a.nblk && col < a.n) {
					const float* src = a.a + (size_t)col * a.lda + row;
					if (row + 3 < a.m) v[k] = *reinterpret_cast<const f32x4u*>(src);
					else {
#pragma unroll
						for (int i = 0; i < 4; i++)
							if (row + i < a.m) v[k][i] = src[i];
					}
				}
			}
		}
	};
	const int step = nwg;
	int bi = wg, it = 0;
	auto split_block = [&](const f32x4 (&v)[4], unsigned* buf) {
		// split the block once: thread (column, four rows) -> two dwords per image
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int col = (wv + 8 * k) * 4 + lcol;
			unsigned h0, m0, l0, h1, m1, l1;
			split3_pair(v[k][0], v[k][1], h0, m0, l0);
			split3_pair(v[k][2], v[k][3], h1, m1, l1);
			unsigned* dst = &buf[col * GW_CS + (lrow >> 1)];
			*reinterpret_cast<uint2*>(dst) = uint2{h0, h1};
			*reinterpret_cast<uint2*>(dst + IMG) = uint2{m0, m1};
			*reinterpret_cast<uint2*>(dst + 2 * IMG) = uint2{l0, l1};
		}
	};
	auto products = [&](const unsigned* buf) {
		// the MFMA's own fp32 accumulation is biased over long chains: one chain per K-step, fp64 totals (as gram_bf16_kernel)
		switch (role) {
			case 0: gram_wide_step<0>(acc, buf, ks, lane); break;
			case 1: gram_wide_step<1>(acc, buf, ks, lane); break;
			case 2: gram_wide_step<2>(acc, buf, ks, lane); break;
			default: gram_wide_step<3>(acc, buf, ks, lane); break;
		}
#pragma unroll
		for (int t = 0; t < 9; t++) {
#pragma unroll
			for (int r = 0; r < 4; r++) tot[t][r] += (double)acc[t][r];
			acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
		}
	};
	// Two register sets X, Y and two image buffers.  The loop is rotated so that its head sits between "split of the next block"
	// and "barrier": there the only loads in flight are those of one set on every path into the head, and each wait inside the body
	// follows the issue of the other set in straight-line code -- the compiler's (positional) wait counters then let exactly the
	// younger set stay in flight instead of draining everything.  (The empty asm statements keep the sets from being interleaved.)
	f32x4 vx[4], vy[4];
	load_block(vx, bi);
	asm volatile("" ::: "memory");
	load_block(vy, bi + step);
	asm volatile("" ::: "memory");
	if (bi < a.nblk) split_block(vx, gw_img);
	while (bi < a.nblk) {
		lds_barrier();                                   // images of block `bi` complete; every wave is done with the other buffer
		load_block(vx, bi + 2 * step);
		asm volatile("" ::: "memory");
		products(gw_img + (it & 1) * 3 * IMG);
		bi += step; it++;
		if (bi >= a.nblk) break;
		split_block(vy, gw_img + (it & 1) * 3 * IMG);
		lds_barrier();
		load_block(vy, bi + 2 * step);
		asm volatile("" ::: "memory");
		products(gw_img + (it & 1) * 3 * IMG);
		bi += step; it++;
		if (bi >= a.nblk) break;
		split_block(vx, gw_img + (it & 1) * 3 * IMG);
	}
	// the two K-halves of every tile meet through LDS (fp64), then the role's nine tiles go to the workgroup's partial
	__syncthreads();
	double* red = reinterpret_cast<double*>(gw_img);     // [role][9][4][64]
	if (ks == 1) {
#pragma unroll
		for (int t = 0; t < 9; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[((role * 9 + t) * 4 + r) * 64 + lane] = tot[t][r];
	}
	__syncthreads();
	if (ks == 0) {
		double* out = a.part + (size_t)wg * WIDE_TILES * 256;
#pragma unroll
		for (int t = 0; t < 9; t++) {
			const int ti = role == 0 ? wide_role_ti(0, t) : (role == 1 ? wide_role_ti(1, t) : (role == 2 ? wide_role_ti(2, t) : wide_role_ti(3, t)));
			const int tj = role == 0 ? wide_role_tj(0, t) : (role == 1 ? wide_role_tj(1, t) : (role == 2 ? wide_role_tj(2, t) : wide_role_tj(3, t)));
			const int dst = wide_tile(ti, tj);
#pragma unroll
			for (int r = 0; r < 4; r++) part_store(&out[(dst * 4 + r) * 64 + lane], tot[t][r] + red[((role * 9 + t) * 4 + r) * 64 + lane]);
		}
	}
}
template <bool FAST>
__global__ __launch_bounds__(512) void gram_wide_kernel(const GramWideArgs a) {
	extern __shared__ __attribute__((aligned(16))) unsigned gw_img[];    // [buffer][image hi / mid / lo][column][row pair]
	announce_previous_call(a.announce, a.announce_seq);
	gram_wide_body<FAST>(a, gw_img, blockIdx.x, gridDim.x);
}

// chol_wide_kernel: the whole two-block factorisation in ONE workgroup and one launch -- chol(G11), the Schur complement
// R12 = Z11^T G12, G22' = G22 - R12^T R12, chol(G22'), Z12 = -Z11 R12 Z22, the 128 x 128 fp32 Z (ld 128) for apply_wide_kernel, the
// verdict over BOTH blocks and the zeros below the diagonal blocks of R -- with the intermediate results handed on through LDS.
// (As four launches -- chol, Schur, chol, Z12 -- the chain cost 24 + 16 + 24 + 21 us: a one-workgroup kernel pays a launch and at
// least one dependent memory round trip each time; the 64^3 fp64 products themselves are ~4 us each.)
//   gsum layout: [G11: 10 tiles][G22: 10 tiles][G12: 16 tiles], f32 accumulator layout (row = 4 (lane >> 4) + reg, col = lane & 15)
//   status out : [0] 0 accepted / 1 rejected, [1] smallest pivot ratio r_jj^2 / g_jj over all n columns (ORIGINAL diagonal of G),
//                [2] S = ||D inverse(R)||_F^2 / n
struct PtrLoad { const double* p; __device__ double operator()(int e) const { return p[e]; } };
struct CholWideArgs {
	const double* gsum;                  // [G11 | G22 | G12] summed tiles (+ the row count behind them)
	float* r; size_t ldr; int n;         // R out (n x n)
	float* zf1; float* zf2;              // fp32 Z11 / Z22 scratch (4096 floats each; chol_body16 writes them)
	float* zw;                           // 128 x 128 Z out
	unsigned* st1; unsigned* st2;        // verdict words of the two blocks (diagnostics)
	unsigned* status; unsigned* host_status;
	const unsigned* prev_status;
	const double* rows_dev; double rows; float scond_floor;
};
// Round 3: sixteen waves.  The two 64 x 64 factorisations are chol_body16 (one group of four pivots per wave); the four 64^3 fp64
// products run on the fp64 matrix cores: wave w owns the 16 x 16 output tile (w >> 2, w & 3) and chains v_mfma_f64_16x16x4_f64 over
// the inner index, both operands one ds_read_b64 per lane straight from the LDS images (a 2 x 2 register-block VALU form of the
// same products is LDS-bandwidth-bound: 4 reads per 4 FMAs, ~6.8 us per full product on sixteen waves).  G22' goes to the second
// factorisation through LDS in the fp64 accumulator layout the matrix cores leave it in.  67.5 -> 56 (VALU products) -> us.
template <class FA, class FB>
__device__ __forceinline__ f64x4 tile_product_f64(FA fa, FB fb, int ks0, int ks1, int lq) {
	f64x4 c = f64x4{0.0, 0.0, 0.0, 0.0};                // c[reg] = C[lq + 4 reg][lane & 15]
#pragma unroll 4
	for (int ks = ks0; ks < ks1; ks++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(fa(4 * ks + lq), fb(4 * ks + lq), c, 0, 0, 0);
	return c;
}

__global__ __launch_bounds__(1024) void chol_wide_kernel(const CholWideArgs a) {
	__shared__ double Gd[64 * 68];                       // G12: Gd[k * 68 + j]; later Z11 by columns: Cs[k * 68 + i] = Z11[i][k]
	__shared__ double Rs[64 * 68];                       // R12: Rs[i * 68 + j]; later T = R12 Z22
	__shared__ double G2s[10 * 256];                     // G22' in chol_body16's tile order, fp64 accumulator layout
	__shared__ double dg[128], red[32];
	__shared__ unsigned verdict[2];
	const int t = threadIdx.x;
	const int n2 = a.n - 64, NT2 = (n2 + 15) / 16;
	auto reject = [&]() {
		if (t == 0) {
			a.status[0] = 1u; a.status[1] = 0u; a.status[2] = 0u;
			if (a.host_status) { volatile unsigned* hs = a.host_status; hs[1] = 0u; hs[2] = 0u; hs[0] = 1u; }
		}
	};
	if (a.prev_status && a.prev_status[0] != 0) { reject(); return; }
	const double rows = a.rows_dev ? a.rows_dev[0] : a.rows;
	const double min_diag = rows * 0x1p-90;
	const int w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;
	const int ti = w >> 2, tj = w & 3, li = l & 15, lq = l >> 4;      // this wave's output tile; lane -> column li, rows lq + 4 reg
	// G12, G22 and the diagonal are requested now: their latency hides behind the first factorisation
	double gin[4], g2v[4] = {0.0, 0.0, 0.0, 0.0}, dgin = 0.0;
#pragma unroll
	for (int u = 0; u < 4; u++) gin[u] = a.gsum[WIDE_G12 + t + 1024 * u];
	if (ti <= tj) {                                      // tile (ti, tj) of G22 in the Gram pass's fp32 accumulator order: row lq + 4 reg -> (reg' = lq, lane' = 16 reg + li)
#pragma unroll
		for (int reg = 0; reg < 4; reg++) g2v[reg] = a.gsum[WIDE_G22 + tri4(ti, tj) * 256 + lq * 64 + 16 * reg + li];
	}
	if (t < 128) {                                       // original diagonal: tile (d, d), row = col = 16 d + c -> reg = c & 3, lane = 16 (c >> 2) + c
		const int blk = t >> 6, j = t & 63, d = j >> 4, cc = j & 15;
		dgin = a.gsum[(blk ? WIDE_G22 : 0) + tri4(d, d) * 256 + (cc & 3) * 64 + 16 * (cc >> 2) + cc];
	}
	// ---- block 1: R11, Z11 (fp64 image stays in LDS) ----
	double* Zi = nullptr;                                // chol_body16's LDS image of inverse(R): Zi[K * 65 + j] = Z[j][K], zero for j > K
	chol_body16(a.r, a.ldr, a.zf1, a.st1, nullptr, PtrLoad{a.gsum}, 64, 4, 1, 0.03125f, INFINITY, 0.0, min_diag, &Zi);
	if (t == 0) verdict[0] = a.st1[0];                   // (written by this very thread inside chol_body16)
	if (t < 128) dg[t] = dgin;
#pragma unroll
	for (int u = 0; u < 4; u++) {
		const int e = t + 1024 * u;
		const int tile = e >> 8, reg = (e >> 6) & 3, ll = e & 63;
		Gd[(16 * (tile >> 2) + 4 * (ll >> 4) + reg) * 68 + 16 * (tile & 3) + (ll & 15)] = gin[u];
	}
	__syncthreads();
	if (verdict[0] != 0) { reject(); return; }
	// ---- Schur complement: R12 = Z11^T G12, G22' = G22 - R12^T R12 ----
	{
		// R12[i][j] = sum_{k <= i} Z11[k][i] G12[k][j];  Z11[k][i] = Zi[i * 65 + k]
		const f64x4 c = tile_product_f64([&](int k) { return Zi[(16 * ti + li) * 65 + k]; }, [&](int k) { return Gd[k * 68 + 16 * tj + li]; }, 0, 4 * (ti + 1), lq);
#pragma unroll
		for (int reg = 0; reg < 4; reg++) {
			const int i = 16 * ti + lq + 4 * reg, j = 16 * tj + li;
			const double v = (j < n2) ? c[reg] : 0.0;
			Rs[i * 68 + j] = v;
			if (j < n2) a.r[(size_t)(64 + j) * a.ldr + i] = (float)v;
		}
	}
	lds_barrier();                                       // R12 complete; every wave is done with G12
	double* Cs = Gd;
	for (int e = t; e < 64 * 64; e += 1024) Cs[(e >> 6) * 68 + (e & 63)] = Zi[(e >> 6) * 65 + (e & 63)];   // Z11 survives the second factorisation here
	if (ti <= tj && tj < NT2) {
		const f64x4 c = tile_product_f64([&](int i) { return Rs[i * 68 + 16 * ti + li]; }, [&](int i) { return Rs[i * 68 + 16 * tj + li]; }, 0, 16, lq);
#pragma unroll
		for (int reg = 0; reg < 4; reg++) G2s[(ti * NT2 - (ti * (ti - 1)) / 2 + (tj - ti)) * 256 + reg * 64 + l] = g2v[reg] - c[reg];
	}
	lds_barrier();                                       // G22' complete; every wave is done with the image of Z11
	// ---- block 2: R22, Z22 (the same instantiation of chol_body16 as block 1, i.e. the same LDS arrays) ----
	chol_body16(a.r + 64 * a.ldr + 64, a.ldr, a.zf2, a.st2, nullptr, PtrLoad{G2s}, n2, NT2, 0, 0.03125f, INFINITY, 0.0, min_diag, &Zi);
	if (t == 0) verdict[1] = a.st2[0];
	__syncthreads();
	if (verdict[1] != 0) { reject(); return; }
	// ---- Z12 = -Z11 (R12 Z22), the 128 x 128 Z, the verdict over both blocks ----
	// T[i][y] = sum_{x <= y} R12[i][x] Z22[x][y];  Z22[x][y] = Zi[y * 65 + x] (rows y >= n2 of the image are stale: they only reach
	// columns y >= n2 of T and of Z12, which are not stored)
	const f64x4 tt = tile_product_f64([&](int x) { return Rs[(16 * ti + li) * 68 + x]; }, [&](int x) { return Zi[(16 * tj + li) * 65 + x]; }, 0, 4 * (tj + 1), lq);
	double s_acc = 0.0;
	float ratio = 1.0f;
	for (int e = t; e < 64 * 64; e += 1024) {
		const int j = e & 63, K = e >> 6;                // Z[j][K] of either block
		const double z1 = Cs[K * 68 + j];
		const double z2 = (K < n2) ? Zi[K * 65 + j] : 0.0;
		s_acc = fma(dg[j] * z1, z1, s_acc);
		s_acc = fma(dg[64 + j] * z2, z2, s_acc);
		if (j == K) {
			ratio = fminf(ratio, (float)(1.0 / (dg[j] * z1 * z1)));
			if (K < n2) ratio = fminf(ratio, (float)(1.0 / (dg[64 + j] * z2 * z2)));
		}
		a.zw[(size_t)K * 128 + j] = (float)z1;                       // Z11[j][K] (zero below the diagonal already)
		a.zw[(size_t)K * 128 + 64 + j] = 0.0f;
		a.zw[(size_t)(64 + K) * 128 + 64 + j] = (float)z2;           // Z22[j][K]
	}
	lds_barrier();
#pragma unroll
	for (int reg = 0; reg < 4; reg++) Rs[(16 * ti + lq + 4 * reg) * 68 + 16 * tj + li] = tt[reg];     // T[k][y]
	lds_barrier();
	{
		// Z12[i][y] = - sum_{k >= i} Z11[i][k] T[k][y];  Z11[i][k] = Cs[k * 68 + i]
		const f64x4 c = tile_product_f64([&](int k) { return Cs[k * 68 + 16 * ti + li]; }, [&](int k) { return Rs[k * 68 + 16 * tj + li]; }, 4 * ti, 16, lq);
#pragma unroll
		for (int reg = 0; reg < 4; reg++) {
			const int i = 16 * ti + lq + 4 * reg, y = 16 * tj + li;
			const double zz = (y < n2) ? -c[reg] : 0.0;
			a.zw[(size_t)(64 + y) * 128 + i] = (float)zz;
			s_acc = fma(dg[i] * zz, zz, s_acc);
		}
	}
	for (int e = t; e < 64 * n2; e += 1024) {            // R below the diagonal blocks: rows 64.., columns < 64
		const int i = e % n2, j = e / n2;
		a.r[(size_t)j * a.ldr + 64 + i] = 0.0f;
	}
	for (int o = 32; o > 0; o >>= 1) { s_acc += __shfl_xor(s_acc, o); ratio = fminf(ratio, __shfl_xor(ratio, o)); }
	if ((t & 63) == 0) { red[t >> 6] = s_acc; red[16 + (t >> 6)] = (double)ratio; }
	lds_barrier();
	if (t == 0) {
		double ssum = 0.0;
		float rmin = 1.0f;
#pragma unroll
		for (int k = 0; k < 16; k++) { ssum += red[k]; rmin = fminf(rmin, (float)red[16 + k]); }
		const float scond = (float)(ssum / (double)a.n);
		const float max_scond = fminf(128.0f, fmaxf(a.scond_floor, 0.12f * sqrtf((float)rows)));
		const unsigned s0 = (rmin > 0.03125f && scond <= max_scond) ? 0u : 1u;       // NaN compares false -> rejected
		a.status[0] = s0;
		a.status[1] = __builtin_bit_cast(unsigned, rmin);
		a.status[2] = __builtin_bit_cast(unsigned, scond);
		if (a.host_status) {
			volatile unsigned* hs = a.host_status;
			hs[1] = __builtin_bit_cast(unsigned, rmin);
			hs[2] = __builtin_bit_cast(unsigned, scond);
			hs[0] = s0;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// The two-block factorisation on FOUR waves, for the one place where it lives in a workgroup that is not its own launch: the chain role
// of gram_wide_chain_kernel (a stream of 128-column calls, tsqr_mi_qr_f32_loop).  chol_wide_kernel's steps in its order, with its
// arithmetic: chol_body4 for the two 64 x 64 factorisations (bitwise equal to chol_body16) and the same fp64-MFMA tile products,
// wave w taking the tiles (0..3, w) one after the other instead of one tile per wave -- R and Z come out bit for bit as from
// chol_wide_kernel (the sum S of the verdict is added up in another order).  LDS from the caller: WIDE4_LDS_DOUBLES doubles.
// ---------------------------------------------------------------------------------------------
constexpr int WIDE4_LDS_DOUBLES = (64 * 65 + 4 * 256 + 128) + 2 * 64 * 68 + 10 * 256 + 128 + 8 + 2;
__device__ __forceinline__ void chol_wide4_body(const CholWideArgs& a, double* lds) {
	double* Lc = lds;                                    // chol_body4's arrays; its first 64 * 65 doubles end up as the fp64 image of inverse(R)
	double* Gd = Lc + (64 * 65 + 4 * 256 + 128);         // G12: Gd[k * 68 + j]; later Z11 by columns
	double* Rs = Gd + 64 * 68;                           // R12; later T = R12 Z22
	double* G2s = Rs + 64 * 68;                          // G22' in chol_body4's tile order, fp64 accumulator layout
	double* dg = G2s + 10 * 256;                         // [128]
	double* red = dg + 128;                              // [8]
	unsigned* verdict = reinterpret_cast<unsigned*>(red + 8);
	const int t = threadIdx.x;                           // 0 .. 255
	const int n2 = a.n - 64, NT2 = (n2 + 15) / 16;
	auto reject = [&]() {
		if (t == 0) {
			a.status[0] = 1u; a.status[1] = 0u; a.status[2] = 0u;
			if (a.host_status) { volatile unsigned* hs = a.host_status; hs[1] = 0u; hs[2] = 0u; hs[0] = 1u; }
		}
	};
	if (a.prev_status && a.prev_status[0] != 0) { reject(); return; }
	const double rows = a.rows_dev ? a.rows_dev[0] : a.rows;
	const double min_diag = rows * 0x1p-90;
	const int w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;
	const int tj = w, li = l & 15, lq = l >> 4;          // this wave's output tiles: (ti, w), ti = 0 .. 3
	double gin[16], g2v[4][4], dgin = 0.0;
#pragma unroll
	for (int u = 0; u < 16; u++) gin[u] = a.gsum[WIDE_G12 + t + 256 * u];
#pragma unroll
	for (int ti = 0; ti < 4; ti++)
#pragma unroll
		for (int reg = 0; reg < 4; reg++) g2v[ti][reg] = (ti <= tj) ? a.gsum[WIDE_G22 + tri4(ti, tj) * 256 + lq * 64 + 16 * reg + li] : 0.0;
	if (t < 128) {
		const int blk = t >> 6, j = t & 63, d = j >> 4, cc = j & 15;
		dgin = a.gsum[(blk ? WIDE_G22 : 0) + tri4(d, d) * 256 + (cc & 3) * 64 + 16 * (cc >> 2) + cc];
	}
	// ---- block 1 ----
	double* Zi = Lc;
	chol_body4(a.r, a.ldr, a.zf1, a.st1, nullptr, PtrLoad{a.gsum}, 64, 4, 1, 0.03125f, INFINITY, 0.0, min_diag, Lc);
	if (t == 0) verdict[0] = a.st1[0];
	if (t < 128) dg[t] = dgin;
#pragma unroll
	for (int u = 0; u < 16; u++) {
		const int e = t + 256 * u;
		const int tile = e >> 8, reg = (e >> 6) & 3, ll = e & 63;
		Gd[(16 * (tile >> 2) + 4 * (ll >> 4) + reg) * 68 + 16 * (tile & 3) + (ll & 15)] = gin[u];
	}
	__syncthreads();
	if (verdict[0] != 0) { reject(); return; }
	// ---- Schur complement ----
#pragma unroll
	for (int ti = 0; ti < 4; ti++) {
		const f64x4 c = tile_product_f64([&](int k) { return Zi[(16 * ti + li) * 65 + k]; }, [&](int k) { return Gd[k * 68 + 16 * tj + li]; }, 0, 4 * (ti + 1), lq);
#pragma unroll
		for (int reg = 0; reg < 4; reg++) {
			const int i = 16 * ti + lq + 4 * reg, j = 16 * tj + li;
			const double v = (j < n2) ? c[reg] : 0.0;
			Rs[i * 68 + j] = v;
			if (j < n2) a.r[(size_t)(64 + j) * a.ldr + i] = (float)v;
		}
	}
	lds_barrier();
	double* Cs = Gd;
	for (int e = t; e < 64 * 64; e += 256) Cs[(e >> 6) * 68 + (e & 63)] = Zi[(e >> 6) * 65 + (e & 63)];
#pragma unroll
	for (int ti = 0; ti < 4; ti++) {
		if (ti <= tj && tj < NT2) {
			const f64x4 c = tile_product_f64([&](int i) { return Rs[i * 68 + 16 * ti + li]; }, [&](int i) { return Rs[i * 68 + 16 * tj + li]; }, 0, 16, lq);
#pragma unroll
			for (int reg = 0; reg < 4; reg++) G2s[(ti * NT2 - (ti * (ti - 1)) / 2 + (tj - ti)) * 256 + reg * 64 + l] = g2v[ti][reg] - c[reg];
		}
	}
	lds_barrier();
	// ---- block 2 ----
	chol_body4(a.r + 64 * a.ldr + 64, a.ldr, a.zf2, a.st2, nullptr, PtrLoad{G2s}, n2, NT2, 0, 0.03125f, INFINITY, 0.0, min_diag, Lc);
	if (t == 0) verdict[1] = a.st2[0];
	__syncthreads();
	if (verdict[1] != 0) { reject(); return; }
	// ---- Z12, the 128 x 128 Z, the verdict ----
	f64x4 tt[4];
#pragma unroll
	for (int ti = 0; ti < 4; ti++)
		tt[ti] = tile_product_f64([&](int x) { return Rs[(16 * ti + li) * 68 + x]; }, [&](int x) { return Zi[(16 * tj + li) * 65 + x]; }, 0, 4 * (tj + 1), lq);
	double s_acc = 0.0;
	float ratio = 1.0f;
	for (int e = t; e < 64 * 64; e += 256) {
		const int j = e & 63, K = e >> 6;
		const double z1 = Cs[K * 68 + j];
		const double z2 = (K < n2) ? Zi[K * 65 + j] : 0.0;
		s_acc = fma(dg[j] * z1, z1, s_acc);
		s_acc = fma(dg[64 + j] * z2, z2, s_acc);
		if (j == K) {
			ratio = fminf(ratio, (float)(1.0 / (dg[j] * z1 * z1)));
			if (K < n2) ratio = fminf(ratio, (float)(1.0 / (dg[64 + j] * z2 * z2)));
		}
		a.zw[(size_t)K * 128 + j] = (float)z1;
		a.zw[(size_t)K * 128 + 64 + j] = 0.0f;
		a.zw[(size_t)(64 + K) * 128 + 64 + j] = (float)z2;
	}
	lds_barrier();
#pragma unroll
	for (int ti = 0; ti < 4; ti++)
#pragma unroll
		for (int reg = 0; reg < 4; reg++) Rs[(16 * ti + lq + 4 * reg) * 68 + 16 * tj + li] = tt[ti][reg];
	lds_barrier();
#pragma unroll
	for (int ti = 0; ti < 4; ti++) {
		const f64x4 c = tile_product_f64([&](int k) { return Cs[k * 68 + 16 * ti + li]; }, [&](int k) { return Rs[k * 68 + 16 * tj + li]; }, 4 * ti, 16, lq);
#pragma unroll
		for (int reg = 0; reg < 4; reg++) {
			const int i = 16 * ti + lq + 4 * reg, y = 16 * tj + li;
			const double zz = (y < n2) ? -c[reg] : 0.0;
			a.zw[(size_t)(64 + y) * 128 + i] = (float)zz;
			s_acc = fma(dg[i] * zz, zz, s_acc);
		}
	}
	for (int e = t; e < 64 * n2; e += 256) {
		const int i = e % n2, j = e / n2;
		a.r[(size_t)j * a.ldr + 64 + i] = 0.0f;
	}
	for (int o = 32; o > 0; o >>= 1) { s_acc += __shfl_xor(s_acc, o); ratio = fminf(ratio, __shfl_xor(ratio, o)); }
	if ((t & 63) == 0) { red[t >> 6] = s_acc; red[4 + (t >> 6)] = (double)ratio; }
	lds_barrier();
	if (t == 0) {
		double ssum = 0.0;
		float rmin = 1.0f;
#pragma unroll
		for (int k = 0; k < 4; k++) { ssum += red[k]; rmin = fminf(rmin, (float)red[4 + k]); }
		const float scond = (float)(ssum / (double)a.n);
		const float max_scond = fminf(128.0f, fmaxf(a.scond_floor, 0.12f * sqrtf((float)rows)));
		const unsigned s0 = (rmin > 0.03125f && scond <= max_scond) ? 0u : 1u;
		a.status[0] = s0;
		a.status[1] = __builtin_bit_cast(unsigned, rmin);
		a.status[2] = __builtin_bit_cast(unsigned, scond);
		if (a.host_status) {
			volatile unsigned* hs = a.host_status;
			hs[1] = __builtin_bit_cast(unsigned, rmin);
			hs[2] = __builtin_bit_cast(unsigned, scond);
			hs[0] = s0;
		}
	}
}

// A stream of 128-column calls: workgroup 0 of the launch is the two-block factorisation of call i (its waves 4 .. 7 leave at once -- a
// barrier counts the waves that are still there --, waves 0 .. 3 run chol_wide4_body in the LDS the Gram role stages its images
// in), every other workgroup is the Gram pass of call i + 1, numbered from 0 as in gram_wide_kernel.  The reduction of a call's
// partials stays a launch of its own in front (tsqr_mi.hip: stream_of_calls_wide_chained).
constexpr int GWC_LDS_BYTES = (GW_LDS_BYTES > WIDE4_LDS_DOUBLES * 8) ? GW_LDS_BYTES : WIDE4_LDS_DOUBLES * 8;
__global__ __launch_bounds__(512) void gram_wide_chain_kernel(const GramWideArgs a, const CholWideArgs cw) {
	extern __shared__ __attribute__((aligned(16))) unsigned gwc_lds[];
	announce_previous_call(a.announce, a.announce_seq);
	if (blockIdx.x != 0) {
		gram_wide_body<true>(a, gwc_lds, (int)blockIdx.x - 1, (int)gridDim.x - 1);
		return;
	}
	if (threadIdx.x >= 256) return;
	chol_wide4_body(cw, reinterpret_cast<double*>(gwc_lds));
}

}  // namespace tsqrmi
