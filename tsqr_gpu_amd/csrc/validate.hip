// validate.hip -- on-device evaluation of the reference's accuracy metrics (harness support, not on the hot path).
//   orthogonality: G = Q^T Q in fp64 (the reference converts Q to double and calls cublasDgemm: src/validation.cu:43-80),
//                  then ||G - I||_F^2 split into diagonal / off-diagonal parts (src/validation.cu:86-127)
//   residual     : ||Q R - A||_F^2 and ||A||_F^2 (src/test.cu:147-165), accumulated in fp64 here
// Generic kernels (any n, any leading dimension), fp64 FMA on the vector units; sums are combined with fp64 atomics.
#include <hip/hip_runtime.h>

namespace tsqrmi {

// G (n x n, column-major, zero-initialised by the caller) += Q(rows of this block)^T Q(rows of this block)
__global__ __launch_bounds__(256) void gramd_kernel(double* __restrict__ g, const float* __restrict__ q, size_t ldq, size_t m, int n,
                                                    size_t rows_per_block) {
	__shared__ float tile[32][129];                      // 32 rows x up to 128 columns per pass over a column panel
	const size_t r_begin = (size_t)blockIdx.x * rows_per_block;
	const size_t r_end = r_begin + rows_per_block < m ? r_begin + rows_per_block : m;
	// column panels of 128: (pi, pj) pairs with pi <= pj; each thread owns an 8 x 8 block of the 128 x 128 result
	const int ti = threadIdx.x & 15, tj = threadIdx.x >> 4;
	for (int pi = 0; pi < n; pi += 128)
		for (int pj = pi; pj < n; pj += 128) {
			double acc[8][8];
#pragma unroll
			for (int a = 0; a < 8; a++)
#pragma unroll
				for (int b = 0; b < 8; b++) acc[a][b] = 0.0;
			__shared__ float tile2[32][129];
			for (size_t r0 = r_begin; r0 < r_end; r0 += 32) {
				__syncthreads();
				for (int idx = threadIdx.x; idx < 32 * 128; idx += 256) {
					const int rr = idx & 31, cc = idx >> 5;
					const size_t row = r0 + rr;
					tile[rr][cc] = (row < r_end && pi + cc < n) ? q[(size_t)(pi + cc) * ldq + row] : 0.0f;
					tile2[rr][cc] = (row < r_end && pj + cc < n) ? q[(size_t)(pj + cc) * ldq + row] : 0.0f;
				}
				__syncthreads();
				for (int rr = 0; rr < 32; rr++) {
					double x[8], y[8];
#pragma unroll
					for (int a = 0; a < 8; a++) { x[a] = (double)tile[rr][ti + 16 * a]; y[a] = (double)tile2[rr][tj + 16 * a]; }
#pragma unroll
					for (int a = 0; a < 8; a++)
#pragma unroll
						for (int b = 0; b < 8; b++) acc[a][b] = fma(x[a], y[b], acc[a][b]);
				}
			}
#pragma unroll
			for (int a = 0; a < 8; a++)
#pragma unroll
				for (int b = 0; b < 8; b++) {
					const int i = pi + ti + 16 * a, j = pj + tj + 16 * b;
					if (i < n && j < n && acc[a][b] != 0.0) {
						atomicAdd(&g[(size_t)j * n + i], acc[a][b]);
						if (pi != pj) atomicAdd(&g[(size_t)i * n + j], acc[a][b]);
					}
				}
		}
}

// out[0] = ||G - I||_F^2, out[1] = diagonal part, out[2] = off-diagonal part
__global__ __launch_bounds__(256) void orth_sums_kernel(double* __restrict__ out, const double* __restrict__ g, int n) {
	__shared__ double sd[256], so[256];
	double d = 0.0, o = 0.0;
	for (size_t e = threadIdx.x; e < (size_t)n * n; e += 256) {
		const int i = (int)(e % n), j = (int)(e / n);
		const double v = g[e] - (i == j ? 1.0 : 0.0);
		if (i == j) d += v * v; else o += v * v;
	}
	sd[threadIdx.x] = d; so[threadIdx.x] = o;
	__syncthreads();
	for (int s = 128; s > 0; s >>= 1) {
		if (threadIdx.x < s) { sd[threadIdx.x] += sd[threadIdx.x + s]; so[threadIdx.x] += so[threadIdx.x + s]; }
		__syncthreads();
	}
	if (threadIdx.x == 0) { out[0] = sd[0] + so[0]; out[1] = sd[0]; out[2] = so[0]; }
}

// out[3] += ||Q R - A||_F^2 over this block's rows, out[4] += ||A||_F^2   (R upper triangular)
__global__ __launch_bounds__(256) void resid_kernel(double* __restrict__ out, const float* __restrict__ q, size_t ldq,
                                                    const float* __restrict__ r, size_t ldr, const float* __restrict__ a, size_t lda,
                                                    size_t m, int n) {
	__shared__ double s1[256], s2[256];
	const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
	double e2 = 0.0, a2 = 0.0;
	if (row < m) {
		for (int j = 0; j < n; j++) {
			double acc = 0.0;
			for (int k = 0; k <= j; k++) acc = fma((double)q[(size_t)k * ldq + row], (double)r[(size_t)j * ldr + k], acc);
			const double av = (double)a[(size_t)j * lda + row];
			const double dv = acc - av;
			e2 = fma(dv, dv, e2);
			a2 = fma(av, av, a2);
		}
	}
	s1[threadIdx.x] = e2; s2[threadIdx.x] = a2;
	__syncthreads();
	for (int s = 128; s > 0; s >>= 1) {
		if (threadIdx.x < s) { s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; }
		__syncthreads();
	}
	if (threadIdx.x == 0) { atomicAdd(&out[3], s1[0]); atomicAdd(&out[4], s2[0]); }
}

}  // namespace tsqrmi
