// chol_wg.hip -- the n x n (n <= 64) step between the two streaming passes:  G = A^T A  ->  R = chol(G), Z = inverse(R), verdict.
//
// Plays the role of the reference's root of the R tree + first backward level (reference src/tsqr.cu:1164-1230: the tile QR that
// turns the reduced stack into R and the product that starts carrying it back towards Q).  One workgroup of four waves, fp64 on
// the vector units (v_fma_f64 outruns v_mfma_f64_16x16x4_f64 on this part: ~134 cycles per MFMA, measured with a complete
// MFMA-based version of this step -- 20.5 us), organised as a two-stage software pipeline instead of round 1's lock-step
// elimination (21-26 us: four pivots of ~600 cycles each per barrier on one wave while three waves waited):
//
//   * rows are kept column-per-lane (lane j = column j) in groups of four consecutive rows;
//   * wave 0 ("pivot wave") does nothing but the critical chain: take the next group's four rows, apply the previous group's
//     rank-4 update to them, factor them against each other in registers (readlane broadcasts, v_rsq_f64 + one Newton step),
//     publish the four rows of R (and a transposed copy for broadcast reads) in LDS -- one workgroup barrier per group;
//   * waves 1..3 ("update waves") own the trailing rows of G and all rows of M = R^-T (the same row operations applied to I):
//     after barrier g they apply group g to the rows they own (the group that becomes pivot next-but-one first, handing it to
//     the pivot wave through LDS), finish the four rows of M that belong to group g, and apply group g-1's finished M rows --
//     the inverse trails the factorisation by one group and never delays it;
//   * everything the pivot chain does not need (fp32 copies of R for the coalesced store, Z stores, the verdict sums) happens in
//     the shadow of the other stage.
// LDS: 38.5 KiB through a caller-provided pointer, so that the apply kernel's first workgroup can run this body inside the
// apply launch (tsqr_kernels.hip, apply_wg_kernel FUSED) while the other workgroups prefetch their first blocks of A.
#pragma once
#include <hip/hip_runtime.h>

namespace tsqrmi {

struct CholArgs {
	float* r; size_t ldr;                // R out: n x n, full block written (zeros below the diagonal)
	float* z;                            // Z = inverse(R) out: NP x NP column-major (ld NP), zero padded
	unsigned* status;                    // [0] 0 accepted / 1 rejected, [1] min pivot ratio (float bits), [2] S (float bits)
	unsigned* host_status;               // optional device-visible alias of pinned host words receiving the same three values
	const double* gsum;                  // summed Gram tiles, (tile, reg, lane) accumulator order
	const unsigned* prev_status;         // optional: status word of the sweep this one depends on (rejected -> report rejected at once)
	const double* rows_dev;              // optional: the row count (summed over ranks) as a double in device memory; overrides `rows`
	double rows;                         // rows of the factored matrix: sets the bf16-level acceptance bound and the shift
	double shift_coef;                   // > 0: shifted Cholesky, s = shift_coef * (rows * n + n (n + 1)) * trace(G)
	int n, NT;
	int f32_layout;                      // 1: tiles in the f32/bf16 MFMA C/D order (row = 4q + reg), 0: f64 MFMA order (row = q + 4 reg)
	int level;                           // 2 bf16-split Gram matrix (pivot ratio > 2^-5, S bound, column norms >= 2^-90), 1 fp64 (ratio > 2^-40),
	                                     // 3 shifted (ratio > 0: rejects only non-finite input)
	float scond_floor;                   // bf16 level: S <= min(128, max(scond_floor, 0.12 sqrt(rows)))
};

struct CholLds {
	double RR[3][4][64];                 // [g % 3][u][j]   = R[4g + u][j]
	double RT[3][64][4];                 // [g % 3][i][u]   = R[4g + u][i]   (multipliers of row i: two 16-B broadcast reads)
	double MM[2][4][64];                 // [g & 1][u][j]   = M[4g + u][j]   (finished rows of M = R^-T)
	double NX[2][4][64];                 // [g & 1][u][j]   = G rows of group g, updated through group g - 2, on their way to the pivot wave
	double YY[3][4];                     // [g % 3][u]      = 1 / R[4g+u][4g+u]
	double pv[64], dg[64];               // pivots, original diagonal
	double sred[4];
	float Rf[64][65];                    // fp32 R for the coalesced store
};
constexpr int CHOL_LDS_BYTES = (int)sizeof(CholLds);

__device__ __forceinline__ double rl64(double x, int lane) {
	const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
	const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, lane);
	const unsigned hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane);
	return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// entry (i, j) of the symmetric Gram matrix from the summed accumulator tiles; only upper-triangle positions are read (the
// bf16-split tiles may differ by an ulp between (i,j) and (j,i)); identity beyond n
__device__ __forceinline__ double chol_gram_entry(const CholArgs& a, int i, int j) {
	if (i >= a.n || j >= a.n) return (i == j) ? 1.0 : 0.0;
	const int lo = min(i, j), hi = max(i, j);
	const int ti = lo >> 4, tj = hi >> 4, row = lo & 15, col = hi & 15;
	const int t = ti * a.NT - (ti * (ti - 1)) / 2 + (tj - ti);
	const int idx = a.f32_layout ? ((row & 3) * 64 + 16 * (row >> 2) + col) : ((row >> 2) * 64 + 16 * (row & 3) + col);
	return a.gsum[(size_t)t * 256 + idx];
}

#ifdef TSQR_CHOL_DBG
__device__ long long g_chol_stamps[4][16][8];
#define CHOL_STAMP(g, k) do { if (j == 0) g_chol_stamps[w][g][k] = __builtin_readcyclecounter(); } while (0)
#else
#define CHOL_STAMP(g, k) do { } while (0)
#endif
// 256 threads.  All four waves must call it (it contains workgroup barriers).
__device__ __forceinline__ void chol_wg(const CholArgs& a, CholLds& L) {
	const int t = threadIdx.x;
	const int j = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);     // wave index as a scalar: the role branches are uniform
	const int n = a.n;
	const int NP = 16 * a.NT;
	const int NG = 4 * a.NT;                              // groups of four rows (identity rows beyond n)
	const double rows = a.rows_dev ? a.rows_dev[0] : a.rows;

	// ---- prologue: rows of G into registers (update waves: up to six groups each; pivot wave: group 0), M = I
	// update wave w (1..3) owns the groups  w - 1 + 3 s,  s = 0..5
	double Gr[6][4], Mr[6][4];                            // update waves
	double Pg[4], rk[4];                                  // pivot wave: the active group / its finished rows
	double dgj = chol_gram_entry(a, j, j);
	if (a.shift_coef > 0.0) {
		// shifted Cholesky (Fukaya et al., SIAM J. Sci. Comput. 2020): G + s I,  s = 11 (m n + n (n + 1)) u trace(G) is safely positive definite
		double tr = (j < n) ? dgj : 0.0;
		for (int o = 32; o > 0; o >>= 1) tr += __shfl_xor(tr, o);
		if (j < n) dgj += a.shift_coef * (rows * (double)n + (double)n * (double)(n + 1)) * tr;
	}
	if (w == 0) {
		L.dg[j] = dgj;
#pragma unroll
		for (int u = 0; u < 4; u++) { Pg[u] = (u == j) ? dgj : chol_gram_entry(a, u, j); rk[u] = 0.0; }
	} else {
#pragma unroll
		for (int s = 0; s < 6; s++) {
			const int g = w - 1 + 3 * s;
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int i = 4 * g + u;
				Gr[s][u] = (g < NG) ? ((i == j) ? dgj : chol_gram_entry(a, i, j)) : 0.0;
				Mr[s][u] = (i == j) ? 1.0 : 0.0;
			}
		}
		if (w == 2 && NG > 1) {                           // group 1 goes to the pivot wave as it is
#pragma unroll
			for (int u = 0; u < 4; u++) L.NX[1][u][j] = Gr[0][u];
		}
	}
	double s_acc = 0.0;

	for (int g = 0; g < NG; g++) {
		const int K0 = 4 * g, b3 = g % 3, p3 = (g + 2) % 3;      // p3: buffer of group g - 1
		CHOL_STAMP(g, 0);
		if (w == 0) {
			// ---- pivot wave: factor the four rows of group g against each other, publish
			double y[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int K = K0 + u;
				const double p = rl64(Pg[u], K);
				const double pp = (p > 0.0) ? p : 1.0;        // keeps the arithmetic finite; the verdict sees the real pivot
				double yy = __builtin_amdgcn_rsq(pp);
				yy = fma(0.5 * yy, fma(-pp * yy, yy, 1.0), yy);       // one Newton step: v_rsq_f64 is good to ~2^-26
				y[u] = yy;
				rk[u] = (j > K) ? Pg[u] * yy : ((j == K) ? pp * yy : 0.0);
				if (j == 0) L.pv[K] = p;
#pragma unroll
				for (int v = u + 1; v < 4; v++) Pg[v] = fma(-rl64(rk[u], K0 + v), rk[u], Pg[v]);
			}
#pragma unroll
			for (int u = 0; u < 4; u++) L.RR[b3][u][j] = rk[u];
			*reinterpret_cast<double2*>(&L.RT[b3][j][0]) = double2{rk[0], rk[1]};
			*reinterpret_cast<double2*>(&L.RT[b3][j][2]) = double2{rk[2], rk[3]};
			if (j < 4) L.YY[b3][j] = (j == 0) ? y[0] : ((j == 1) ? y[1] : ((j == 2) ? y[2] : y[3]));
		}
		CHOL_STAMP(g, 1);
		__syncthreads();                                  // B(g): group g is published; NX holds group g + 1 (updated through g - 1)
		CHOL_STAMP(g, 2);
		if (w == 0) {
			// fp32 copy for the store at the end (off the chain), then fetch group g + 1 and apply group g to it
#pragma unroll
			for (int u = 0; u < 4; u++) L.Rf[K0 + u][j] = (float)rk[u];
			if (g + 1 < NG) {
#pragma unroll
				for (int v = 0; v < 4; v++) {
					const double2 m01 = *reinterpret_cast<const double2*>(&L.RT[b3][K0 + 4 + v][0]);
					const double2 m23 = *reinterpret_cast<const double2*>(&L.RT[b3][K0 + 4 + v][2]);
					double x = L.NX[(g + 1) & 1][v][j];
					x = fma(-m01.x, rk[0], x); x = fma(-m01.y, rk[1], x);
					x = fma(-m23.x, rk[2], x); x = fma(-m23.y, rk[3], x);
					Pg[v] = x;
				}
			}
		} else {
			// ---- update waves
			double rj[4];
#pragma unroll
			for (int u = 0; u < 4; u++) rj[u] = L.RR[b3][u][j];
			// (1) rows of G: the group that becomes pivot next-but-one first (hand-off), then the later ones
#pragma unroll
			for (int s = 0; s < 6; s++) {
				const int go = w - 1 + 3 * s;                 // group held in slot s
				if (go >= g + 2 && go < NG) {                 // wave-uniform
#pragma unroll
					for (int v = 0; v < 4; v++) {
						const int i = 4 * go + v;
						const double2 m01 = *reinterpret_cast<const double2*>(&L.RT[b3][i][0]);
						const double2 m23 = *reinterpret_cast<const double2*>(&L.RT[b3][i][2]);
						double x = Gr[s][v];
						x = fma(-m01.x, rj[0], x); x = fma(-m01.y, rj[1], x);
						x = fma(-m23.x, rj[2], x); x = fma(-m23.y, rj[3], x);
						Gr[s][v] = x;
					}
					if (go == g + 2) {
#pragma unroll
						for (int v = 0; v < 4; v++) L.NX[go & 1][v][j] = Gr[s][v];
					}
				}
			}
			CHOL_STAMP(g, 3);
			// (2) rows of M: apply the finished rows of group g - 1 to every later row this wave owns ...
			if (g >= 1) {
				double mj[4];
#pragma unroll
				for (int u = 0; u < 4; u++) mj[u] = L.MM[(g - 1) & 1][u][j];
#pragma unroll
				for (int s = 0; s < 6; s++) {
					const int go = w - 1 + 3 * s;
					if (go >= g && go < NG) {
#pragma unroll
						for (int v = 0; v < 4; v++) {
							const int i = 4 * go + v;
							const double2 m01 = *reinterpret_cast<const double2*>(&L.RT[p3][i][0]);
							const double2 m23 = *reinterpret_cast<const double2*>(&L.RT[p3][i][2]);
							double x = Mr[s][v];
							x = fma(-m01.x, mj[0], x); x = fma(-m01.y, mj[1], x);
							x = fma(-m23.x, mj[2], x); x = fma(-m23.y, mj[3], x);
							Mr[s][v] = x;
						}
					}
				}
			}
			CHOL_STAMP(g, 4);
			// ... then finish the four rows of M that belong to group g (their owner only) and publish them
#pragma unroll
			for (int s = 0; s < 6; s++) {
				const int go = w - 1 + 3 * s;
				if (go == g) {
#pragma unroll
					for (int u = 0; u < 4; u++) {
						const int K = K0 + u;
						const double mk = Mr[s][u] * L.YY[b3][u];
#pragma unroll
						for (int v = u + 1; v < 4; v++) Mr[s][v] = fma(-L.RR[b3][u][K0 + v], mk, Mr[s][v]);
						L.MM[g & 1][u][j] = mk;
						Mr[s][u] = mk;                                // kept for the Z store after the loop (no global store in here:
						                                              // a workgroup barrier drains vmcnt, i.e. waits for its round trip)
						if (K < n && j <= K) s_acc = fma(L.dg[j] * mk, mk, s_acc);               // sum of g_jj * Z[j][K]^2
					}
				}
			}
		}
	}
	// ---- Z out (fp32, NP x NP column-major, zero padded): every update wave stores the rows of M it finished, Z[j][K] = M[K][j]
	if (w != 0) {
#pragma unroll
		for (int s = 0; s < 6; s++) {
			const int go = w - 1 + 3 * s;
			if (go < NG) {
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const int K = 4 * go + u;
					if (j < NP) a.z[(size_t)K * NP + j] = (K < n && j <= K) ? (float)Mr[s][u] : 0.0f;
				}
			}
		}
	}
	// ---- verdict: smallest pivot ratio and scaled conditioning S = || D inverse(R) ||_F^2 / n, D = diag(sqrt(g_jj))
	for (int o = 32; o > 0; o >>= 1) s_acc += __shfl_xor(s_acc, o);
	if (j == 0) L.sred[w] = s_acc;
	__syncthreads();
	if (w == 0) {
		const double d0 = L.dg[j], p0 = L.pv[j];
		float ratio = 1.0f;
		if (j < n) {
			ratio = (d0 > 0.0 && p0 > 0.0) ? (float)(p0 / d0) : 0.0f;       // NaN pivots / diagonals compare false -> 0
			if (a.level == 2 && !(d0 >= rows * 0x1p-90)) ratio = 0.0f;     // bf16 level: products near the fp32 denormal range are not exact
		}
		for (int o = 32; o > 0; o >>= 1) ratio = fminf(ratio, __shfl_xor(ratio, o));
		if (j == 0) {
			const float scond = (float)((L.sred[1] + L.sred[2] + L.sred[3]) / (double)n);
			float min_ratio = 0.0f, max_scond = INFINITY;
			if (a.level == 2) { min_ratio = 0.03125f; max_scond = fminf(128.0f, fmaxf(a.scond_floor, 0.12f * sqrtf((float)rows))); }
			else if (a.level == 1) min_ratio = 9.094947017729282e-13f;       // 2^-40
			const unsigned s0 = (ratio > min_ratio && scond <= max_scond) ? 0u : 1u;      // NaN compares false -> rejected
			a.status[0] = s0;
			a.status[1] = __builtin_bit_cast(unsigned, ratio);
			a.status[2] = __builtin_bit_cast(unsigned, scond);
			if (a.host_status) {
				volatile unsigned* hs = a.host_status;
				hs[1] = __builtin_bit_cast(unsigned, ratio);
				hs[2] = __builtin_bit_cast(unsigned, scond);
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");        // system scope: words 1, 2 are visible before the verdict word
				hs[0] = s0;
			}
		}
	}
	// ---- R out (fp32, exact zeros below the diagonal), coalesced along the rows
	{
		const int i = t & 63;
		if (i < n)
			for (int jj = t >> 6; jj < n; jj += 4) a.r[(size_t)jj * a.ldr + i] = (i <= jj) ? L.Rf[i][jj] : 0.0f;
	}
}

// the same step as a launch of its own (staged API, checked paths): one workgroup of 256 threads
__global__ __launch_bounds__(256) void chol_wg_kernel(const CholArgs a) {
	__shared__ CholLds lds;
	if (a.prev_status && a.prev_status[0] != 0) {
		if (threadIdx.x == 0) {
			a.status[0] = 1u; a.status[1] = 0u; a.status[2] = 0u;
			if (a.host_status) { volatile unsigned* hs = a.host_status; hs[1] = 0u; hs[2] = 0u; hs[0] = 1u; }
		}
		return;
	}
	chol_wg(a, lds);
}

}  // namespace tsqrmi
