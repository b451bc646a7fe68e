// chol_mfma.hip -- the n x n (n <= 64) step between the two streaming passes:  G = A^T A  ->  R = chol(G), Z = inverse(R).
//
// Plays the role of the reference's root of the R tree + first backward level (reference src/tsqr.cu:1164-1230: one tile QR that
// turns the reduced stack into R, and the products that carry it back towards Q), re-designed for one CDNA4 workgroup.
// Everything is fp64 on v_mfma_f64_16x16x4_f64, ONE wave, no barrier, no LDS traffic inside the factorisation:
//
//   * G, R, M = R^-T and N = Z^T live as 16 x 16 tiles in the f64 MFMA C/D layout (lane l = 16q + c holds column c and the rows
//     q + 4*reg, reg = 0..3).  In that layout register `reg` of a tile IS the K-slice `reg` of both MFMA operands, so every product
//     of the form X^T Y costs four MFMAs with no data movement at all -- and every product of the blocked algorithm is written in
//     that form (R_kj = T_k^T G_kj, G_ij -= R_ki^T R_kj, N_ji = -T_j^T sum_k R_kj^T N_ki with N_kk = M_k, T_k = M_k^T).
//   * the 16 pivots of a diagonal tile are a chain of rank-1 MFMA updates: the scaled pivot row stays in the lanes / register where
//     the C/D layout keeps it (K-slice k&3) and is used as A and B operand directly; the matrix unit does the outer product, so the
//     chain per pivot is  readlane -> rsq + one Newton step -> two multiplies -> MFMA.  The same row operations applied to an
//     identity tile give M_k = R_kk^-T.
//   * accept / reject verdict as before: smallest pivot ratio p_j / g_jj and scaled conditioning S = ||D Z||_F^2 / n.
//
// Measured on MI355X: see DESIGN.md section 4 (the four-wave elimination kernel of round 1 needed 21-26 us, most of it waiting
// on its 16 workgroup barriers and on the owner wave's serial pivot groups).
#pragma once
#include <hip/hip_runtime.h>

namespace tsqrmi {

typedef double cf64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double rl64(double x, int lane) {
	const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
	const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, lane);
	const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane);
	return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// C + X^T Y  for 16 x 16 tiles in the f64 C/D layout
__device__ __forceinline__ cf64x4 xty(const cf64x4& X, const cf64x4& Y, cf64x4 C) {
#pragma unroll
	for (int r = 0; r < 4; r++) C = __builtin_amdgcn_mfma_f64_16x16x4f64(X[r], Y[r], C, 0, 0, 0);
	return C;
}
__device__ __forceinline__ cf64x4 xty_neg(const cf64x4& X, const cf64x4& Y, cf64x4 C) {
#pragma unroll
	for (int r = 0; r < 4; r++) C = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[r], Y[r], C, 0, 0, 0);
	return C;
}

// Right-looking Cholesky of one symmetric 16 x 16 tile D (both triangles valid) as 16 rank-1 MFMA updates; the same row
// operations on an identity tile give M = R^-T.  Rk, Mk: R (upper; garbage below the diagonal) and M (lower) in the C/D layout.
// Only the A operand is masked (rows i > k, K-slice k&3): the other K-slices of B then do not matter, so B is the plain scaled
// register.  Row k of D and of the identity part is never touched after step k, so R = D * y_row and M = Mt * y_row at the end, and
// the diagonal of D then holds the pivots (for the verdict).  Pivots <= 0 (or NaN) are replaced by 1: the arithmetic stays finite.
__device__ __forceinline__ void chol16_chain(cf64x4& D, cf64x4& Rk, cf64x4& Mk, int lane) {
	const int c = lane & 15, q = lane >> 4;
	int code[4], qv = q;                                 // code[rg] = q where the lane's element (row q + 4 rg, column c) is strictly upper, else -1
#pragma unroll
	for (int r = 0; r < 4; r++) code[r] = (c > q + 4 * r) ? q : -1;
	// the 32 lane masks of a chain are each used once: keep the compiler from hoisting them out of the four chains into SGPRs (it
	// spilled 260 SGPRs through v_writelane/v_readlane when it did)
	asm volatile("" : "+v"(code[0]), "+v"(code[1]), "+v"(code[2]), "+v"(code[3]), "+v"(qv));
	cf64x4 Mt, ys;
#pragma unroll
	for (int r = 0; r < 4; r++) { Mt[r] = (q + 4 * r == c) ? 1.0 : 0.0; ys[r] = 1.0; }
	static_for<0, 16>([&](auto kk) {
		constexpr int k = decltype(kk)::value;
		constexpr int rg = k >> 2, rq = k & 3;           // row k of the tile: register rg of the lanes with q == rq
		const double p = rl64(D[rg], 16 * rq + k);       // element (k, k)
		const double pp = (p > 0.0) ? p : 1.0;
		double y = __builtin_amdgcn_rsq(pp);
		y = fma(0.5 * y, fma(-pp * y, y, 1.0), y);       // one Newton step: v_rsq_f64 is good to ~2^-26
		const double ny = -y;
		const double ua = (code[rg] == rq) ? D[rg] * y : 0.0;               // R[k][i] for i > k in K-slice rq, zero elsewhere
		D = __builtin_amdgcn_mfma_f64_16x16x4f64(ua, D[rg] * ny, D, 0, 0, 0);     // D[i][j] -= R[k][i] R[k][j]
		Mt = __builtin_amdgcn_mfma_f64_16x16x4f64(ua, Mt[rg] * ny, Mt, 0, 0, 0);  // M[i][:] -= R[k][i] M[k][:]
		ys[rg] = (qv == rq) ? y : ys[rg];
	});
#pragma unroll
	for (int r = 0; r < 4; r++) { Rk[r] = D[r] * ys[r]; Mk[r] = Mt[r] * ys[r]; }
}

struct CholArgs {
	float* r; size_t ldr;                // R out: n x n, full block written (zeros below the diagonal)
	float* z;                            // Z = inverse(R) out: NP x NP column-major (ld NP), zero padded
	unsigned* status;                    // [0] 0 accepted / 1 rejected, [1] min pivot ratio (float bits), [2] S (float bits)
	unsigned* host_status;               // optional device-visible alias of pinned host words receiving the same three values
	const double* gsum;                  // summed Gram tiles, (tile, reg, lane) accumulator order
	const unsigned* prev_status;         // optional: status word of the sweep this one depends on (rejected -> report rejected at once)
	const double* rows_dev;              // optional: the row count (sum over ranks) as a double in device memory; overrides `rows`
	double rows;                         // rows of the factored matrix: sets the bf16-level acceptance bound and the shift
	double shift_coef_per_row;           // > 0: shifted Cholesky, s = shift_coef_per_row * (rows * n + n (n + 1)) * trace(G)
	int n, NT;
	int f32_layout;                      // 1: tiles in the f32/bf16 MFMA C/D order (row = 4q + reg), 0: f64 MFMA order (row = q + 4 reg)
	int level;                           // 2 bf16-split Gram matrix (pivot ratio > 2^-5, S bound, column norms >= 2^-90), 1 fp64 (ratio > 2^-40),
	                                     // 3 shifted (ratio > 0: rejects only non-finite input)
	float scond_floor;                   // bf16 level: S <= min(128, max(scond_floor, 0.12 sqrt(rows)))
};

template <int NTC>
__device__ __forceinline__ void chol_mfma_body(const CholArgs& a, double* dgs, double* pvs) {
	const int lane = threadIdx.x & 63;
	const int c = lane & 15, q = lane >> 4;
	const int n = a.n;
	constexpr int NP = 16 * NTC;
	const double rows = a.rows_dev ? a.rows_dev[0] : a.rows;

	// ---- load the Gram tiles: G[bi][bj] (bi <= bj), diagonal tiles mirrored from their upper triangle, identity padding beyond n
	cf64x4 G[NTC][NTC];
	static_for<0, NTC>([&](auto bi_) {
		constexpr int bi = decltype(bi_)::value;
		static_for<bi, NTC>([&](auto bj_) {
			constexpr int bj = decltype(bj_)::value;
			constexpr int t = bi * NTC - (bi * (bi - 1)) / 2 + (bj - bi);
			const double* src = a.gsum + (size_t)t * 256;
#pragma unroll
			for (int r = 0; r < 4; r++) {
				int row = q + 4 * r, col = c;
				if (bi == bj && row > col) { const int x = row; row = col; col = x; }     // mirror: take (col, row) of the upper triangle
				const int idx = a.f32_layout ? ((row & 3) * 64 + 16 * (row >> 2) + col) : ((row >> 2) * 64 + 16 * (row & 3) + col);
				double v = src[idx];
				const int gr = 16 * bi + q + 4 * r, gc = 16 * bj + c;
				if (gr >= n || gc >= n) v = (gr == gc) ? 1.0 : 0.0;
				G[bi][bj][r] = v;
			}
		});
	});
	// diagonal of G -> dgs[0..63] (LDS), then one value per lane
	static_for<0, NTC>([&](auto b_) {
		constexpr int b = decltype(b_)::value;
#pragma unroll
		for (int r = 0; r < 4; r++)
			if (q + 4 * r == c) dgs[16 * b + c] = G[b][b][r];
	});
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	double dgv = (lane < NP) ? dgs[lane] : 1.0;
	if (a.shift_coef_per_row > 0.0) {
		// shifted Cholesky (Fukaya et al., SIAM J. Sci. Comput. 2020): G + s I with s = 11 (m n + n (n+1)) u ||A||_2^2-scale, here via trace(G)
		double tr = (lane < n) ? dgv : 0.0;
		for (int o = 32; o > 0; o >>= 1) tr += __shfl_xor(tr, o);
		const double s = a.shift_coef_per_row * (rows * (double)n + (double)n * (double)(n + 1)) * tr;
		if (lane < n) dgv += s;
		static_for<0, NTC>([&](auto b_) {
			constexpr int b = decltype(b_)::value;
#pragma unroll
			for (int r = 0; r < 4; r++)
				if (q + 4 * r == c && 16 * b + c < n) G[b][b][r] += s;
		});
	}
	double dgc[NTC];                                     // g_jj for j = 16 b + c (this lane's column in block column b)
	if (a.shift_coef_per_row > 0.0) {
		if (lane < NP) dgs[lane] = dgv;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}
#pragma unroll
	for (int b = 0; b < NTC; b++) dgc[b] = dgs[16 * b + c];

	// ---- blocked right-looking Cholesky; M_k = R_kk^-T and T_k = M_k^T = inverse(R_kk) per diagonal tile
	cf64x4 M[NTC], T[NTC];
	cf64x4 ident;
#pragma unroll
	for (int r = 0; r < 4; r++) ident[r] = (q + 4 * r == c) ? 1.0 : 0.0;
	const cf64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
	static_for<0, NTC>([&](auto kb_) {
		constexpr int kb = decltype(kb_)::value;
		cf64x4 Rk;
		chol16_chain(G[kb][kb], Rk, M[kb], lane);
#pragma unroll
		for (int r = 0; r < 4; r++)
			if (q + 4 * r == c) pvs[16 * kb + c] = G[kb][kb][r];          // the pivots are the diagonal the chain leaves behind
		G[kb][kb] = Rk;
		T[kb] = xty(M[kb], ident, zero4);                                // transpose
		static_for<kb + 1, NTC>([&](auto bj_) {                          // R_kj = T_k^T G_kj
			constexpr int bj = decltype(bj_)::value;
			G[kb][bj] = xty(T[kb], G[kb][bj], zero4);
		});
		static_for<kb + 1, NTC>([&](auto bi_) {                          // G_ij -= R_ki^T R_kj  (next diagonal tile first)
			constexpr int bi = decltype(bi_)::value;
			static_for<bi, NTC>([&](auto bj_) {
				constexpr int bj = decltype(bj_)::value;
				G[bi][bj] = xty_neg(G[kb][bi], G[kb][bj], G[bi][bj]);
			});
		});
	});

	// ---- N = Z^T:  N_jj = M_j,  N_ji = -T_j^T * sum_{k = i .. j-1} R_kj^T N_ki   (i < j)
	cf64x4 N[NTC][NTC];
	static_for<0, NTC>([&](auto bj_) {
		constexpr int bj = decltype(bj_)::value;
		N[bj][bj] = M[bj];
		static_for<0, bj>([&](auto bi_) {
			constexpr int bi = decltype(bi_)::value;
			cf64x4 S = zero4;
			static_for<bi, bj>([&](auto k_) {
				constexpr int k = decltype(k_)::value;
				S = xty(G[k][bj], N[k][bi], S);
			});
			N[bj][bi] = xty_neg(T[bj], S, zero4);
		});
	});

	// ---- verdict
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	const double pvv = (lane < NP) ? pvs[lane] : 1.0;
	float ratio = 1.0f;
	if (lane < n) {
		const double rr = pvv * __builtin_amdgcn_rcp(dgv);
		ratio = (dgv > 0.0 && pvv > 0.0) ? (float)rr : 0.0f;          // NaN pivots / diagonals compare false -> 0
		if (a.level == 2 && !(dgv >= rows * 0x1p-90)) ratio = 0.0f;   // bf16 level: products near the fp32 denormal range are not exact
	}
	for (int o = 32; o > 0; o >>= 1) ratio = fminf(ratio, __shfl_xor(ratio, o));
	double s_acc = 0.0;
	static_for<0, NTC>([&](auto bj_) {
		constexpr int bj = decltype(bj_)::value;
		static_for<0, bj + 1>([&](auto bi_) {
			constexpr int bi = decltype(bi_)::value;
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const int K = 16 * bj + q + 4 * r, j = 16 * bi + c;       // N_ji(row, col) = Z(j, K)
				const double zz = N[bj][bi][r];
				if (K < n && j < n) s_acc = fma(dgc[bi] * zz, zz, s_acc);
			}
		});
	});
	for (int o = 32; o > 0; o >>= 1) s_acc += __shfl_xor(s_acc, o);
	if (lane == 0) {
		const float scond = (float)(s_acc / (double)n);
		float min_ratio = 0.0f, max_scond = INFINITY;
		if (a.level == 2) { min_ratio = 0.03125f; max_scond = fminf(128.0f, fmaxf(a.scond_floor, 0.12f * sqrtf((float)rows))); }
		else if (a.level == 1) min_ratio = 9.094947017729282e-13f;       // 2^-40
		const unsigned s0 = (ratio > min_ratio && scond <= max_scond) ? 0u : 1u;   // NaN compares false -> rejected
		a.status[0] = s0;
		a.status[1] = __builtin_bit_cast(unsigned, ratio);
		a.status[2] = __builtin_bit_cast(unsigned, scond);
		if (a.host_status) {
			volatile unsigned* hs = a.host_status;
			hs[1] = __builtin_bit_cast(unsigned, ratio);
			hs[2] = __builtin_bit_cast(unsigned, scond);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // system scope: words 1, 2 are visible before the verdict word
			hs[0] = s0;
		}
	}

	// ---- Z (fp32, NP x NP column-major, zero padded): every position is written, no predicate needed
	static_for<0, NTC>([&](auto bi_) {
		constexpr int bi = decltype(bi_)::value;
		static_for<0, NTC>([&](auto bj_) {
			constexpr int bj = decltype(bj_)::value;
#pragma unroll
			for (int r = 0; r < 4; r++) {
				// this lane holds N[bi][bj](row, col) = Z(j, K) with K = 16 bi + row, j = 16 bj + col
				const int K = 16 * bi + q + 4 * r, j = 16 * bj + c;
				float zv = 0.0f;
				if constexpr (bj <= bi) zv = (j <= K && K < n) ? (float)N[bi][bj][r] : 0.0f;
				a.z[(size_t)K * NP + j] = zv;
			}
		});
	});
	// ---- R (fp32, n x n, exact zeros below the diagonal).  n == NP (the common case): plain stores; otherwise predicated ones.
	if (n == NP) {
		static_for<0, NTC>([&](auto bi_) {
			constexpr int bi = decltype(bi_)::value;
			static_for<0, NTC>([&](auto bj_) {
				constexpr int bj = decltype(bj_)::value;
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const int row = 16 * bi + q + 4 * r, col = 16 * bj + c;
					float v = 0.0f;
					if constexpr (bi <= bj) v = (row <= col) ? (float)G[bi][bj][r] : 0.0f;
					a.r[(size_t)col * a.ldr + row] = v;
				}
			});
		});
	} else {
		static_for<0, NTC>([&](auto bi_) {
			constexpr int bi = decltype(bi_)::value;
			static_for<0, NTC>([&](auto bj_) {
				constexpr int bj = decltype(bj_)::value;
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const int row = 16 * bi + q + 4 * r, col = 16 * bj + c;
					float v = 0.0f;
					if constexpr (bi <= bj) v = (row <= col) ? (float)G[bi][bj][r] : 0.0f;
					if (row < n && col < n) a.r[(size_t)col * a.ldr + row] = v;
				}
			});
		});
	}
}

// one wave; NTC = number of 16-column tiles (NP = 16 NTC).  LDS: 128 doubles.
template <int NTC>
__global__ __launch_bounds__(64) void chol_mfma_kernel(const CholArgs a) {
	__shared__ double dgs[64], pvs[64];
	if (a.prev_status && a.prev_status[0] != 0) {
		if (threadIdx.x == 0) {
			a.status[0] = 1u; a.status[1] = 0u; a.status[2] = 0u;
			if (a.host_status) { volatile unsigned* hs = a.host_status; hs[1] = 0u; hs[2] = 0u; hs[0] = 1u; }
		}
		return;
	}
	chol_mfma_body<NTC>(a, dgs, pvs);
}

inline void launch_chol_mfma(const CholArgs& a, hipStream_t st) {
	switch (a.NT) {
		case 1: hipLaunchKernelGGL(chol_mfma_kernel<1>, dim3(1), dim3(64), 0, st, a); break;
		case 2: hipLaunchKernelGGL(chol_mfma_kernel<2>, dim3(1), dim3(64), 0, st, a); break;
		case 3: hipLaunchKernelGGL(chol_mfma_kernel<3>, dim3(1), dim3(64), 0, st, a); break;
		default: hipLaunchKernelGGL(chol_mfma_kernel<4>, dim3(1), dim3(64), 0, st, a); break;
	}
}

}  // namespace tsqrmi
