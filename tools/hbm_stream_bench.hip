// What a read-A / write-Q stream reaches beyond the Infinity Cache, by access pattern (round 4).
// Skeleton passes, no arithmetic: copy an m x 64 fp32 column-major matrix (leading dimension ld) into another one.
//   lin      : the matrix as one linear array, float4 per lane, grid-stride (the guide's "float4 copy": 6.29 TB/s)
//   col<R,V> : the apply pass's pattern -- a workgroup (four waves) moves R-row x 64-column blocks, interleaved over a persistent
//              grid; V floats per lane (V = 1: one column segment of 64 rows = 256 B per wave instruction, what apply_wg_kernel<.,.,.,64>
//              issues; V = 4: 16 B per lane, 64/(R/4) columns per instruction)
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_stream_bench tools/hbm_stream_bench.hip ; run: ./hbm_stream_bench [log2 m]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4u __attribute__((aligned(4)));

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void lin_kernel(float* __restrict__ q, const float* __restrict__ a, size_t n4) {
	const size_t stride = (size_t)gridDim.x * 256;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
		const f32x4 v = NTL ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a) + i) : reinterpret_cast<const f32x4*>(a)[i];
		if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(q) + i); else reinterpret_cast<f32x4*>(q)[i] = v;
	}
}
// linear, but every workgroup owns a contiguous slab (chunked, not interleaved)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void lin_slab_kernel(float* __restrict__ q, const float* __restrict__ a, size_t n4) {
	const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
	const size_t b = (size_t)blockIdx.x * per, e = b + per < n4 ? b + per : n4;
	for (size_t i = b + threadIdx.x; i < e; i += 256) {
		const f32x4 v = NTL ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a) + i) : reinterpret_cast<const f32x4*>(a)[i];
		if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(q) + i); else reinterpret_cast<f32x4*>(q)[i] = v;
	}
}

// MODE 0 copy, 1 load only, 2 store only
template <int ROWS, int V, int MODE, bool NTL, bool NTS, int DEPTH>
__global__ __launch_bounds__(256) void col_kernel(float* __restrict__ q, const float* __restrict__ a, size_t ld, size_t ldq, int nblocks) {
	constexpr int NP = 64;
	constexpr int LPC = ROWS / V;                 // lanes per column segment
	static_assert(LPC <= 64 && 64 % LPC == 0, "");
	constexpr int CPI = 64 / LPC;                 // columns per wave instruction
	constexpr int NI = NP / (4 * CPI);            // instructions per wave and block
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int lcol = lane / LPC, lrow = V * (lane % LPC);
	float acc = 0.f;
	typedef float vec __attribute__((ext_vector_type(V == 1 ? 2 : V)));   // (V == 1 handled with scalars below)
	for (int b0 = blockIdx.x; b0 < nblocks; b0 += gridDim.x * DEPTH) {
		float v[DEPTH][NI][V];
#pragma unroll
		for (int d = 0; d < DEPTH; d++) {
			const int b = b0 + d * gridDim.x;
			if (b < nblocks) {
#pragma unroll
				for (int k = 0; k < NI; k++) {
					const size_t off = (size_t)((wv + 4 * k) * CPI + lcol) * ld + (size_t)b * ROWS + lrow;
					if (MODE != 2) {
						if constexpr (V == 4) {
							const f32x4 t = NTL ? __builtin_nontemporal_load(reinterpret_cast<const f32x4u*>(a + off)) : *reinterpret_cast<const f32x4u*>(a + off);
							v[d][k][0] = t[0]; v[d][k][1] = t[1]; v[d][k][2] = t[2]; v[d][k][3] = t[3];
						} else {
							v[d][k][0] = NTL ? __builtin_nontemporal_load(a + off) : a[off];
						}
					} else {
#pragma unroll
						for (int u = 0; u < V; u++) v[d][k][u] = (float)(b + k + u);
					}
				}
			}
		}
#pragma unroll
		for (int d = 0; d < DEPTH; d++) {
			const int b = b0 + d * gridDim.x;
			if (b < nblocks) {
#pragma unroll
				for (int k = 0; k < NI; k++) {
					const size_t off = (size_t)((wv + 4 * k) * CPI + lcol) * ldq + (size_t)b * ROWS + lrow;
					if (MODE == 1) {
#pragma unroll
						for (int u = 0; u < V; u++) acc += v[d][k][u];
					} else if constexpr (V == 4) {
						const f32x4 t = {v[d][k][0], v[d][k][1], v[d][k][2], v[d][k][3]};
						if (NTS) __builtin_nontemporal_store(t, reinterpret_cast<f32x4u*>(q + off)); else *reinterpret_cast<f32x4u*>(q + off) = t;
					} else {
						if (NTS) __builtin_nontemporal_store(v[d][k][0], q + off); else q[off] = v[d][k][0];
					}
				}
			}
		}
	}
	if (MODE == 1 && acc == 123.456f) q[0] = acc;
}

struct Timer {
	hipEvent_t e0, e1;
	Timer() { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); }
	template <class F> double us(F&& f, int reps = 6) {
		f(); f();
		(void)hipDeviceSynchronize();
		double best = 1e30, sum = 0;
		for (int i = 0; i < reps; i++) {
			(void)hipEventRecord(e0, 0); f(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
			float ms; (void)hipEventElapsedTime(&ms, e0, e1);
			sum += ms * 1e3; if (ms * 1e3 < best) best = ms * 1e3;
		}
		return sum / reps;
	}
};

int main(int argc, char** argv) {
	const int lm = argc > 1 ? atoi(argv[1]) : 23;
	const size_t m = (size_t)1 << lm, n = 64;
	const size_t pad_max = 4096;
	float *a, *q;
	const size_t elems = (m + pad_max) * n;
	if (hipMalloc(&a, elems * 4) != hipSuccess || hipMalloc(&q, elems * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void)hipMemset(a, 0x3c, elems * 4); (void)hipMemset(q, 0, elems * 4);
	Timer T;
	const double bytes_copy = 2.0 * m * n * 4, bytes_one = 1.0 * m * n * 4;
	auto rep = [&](const char* name, double us, double bytes) { printf("%-64s %9.1f us  %6.2f TB/s\n", name, us, bytes / us * 1e-6); fflush(stdout); };
	printf("m = 2^%d, n = 64, %.0f MB per matrix\n", lm, bytes_one * 1e-6);
	const size_t n4 = m * n / 4;
	for (int g : {1024, 2048, 4096, 8192, 65536}) {
		char nm[128];
		snprintf(nm, sizeof nm, "lin  copy plain ld / nt st, grid %d", g);   rep(nm, T.us([&] { hipLaunchKernelGGL((lin_kernel<false, true>), dim3(g), dim3(256), 0, 0, q, a, n4); }), bytes_copy);
		snprintf(nm, sizeof nm, "lin  copy nt ld / nt st,    grid %d", g);   rep(nm, T.us([&] { hipLaunchKernelGGL((lin_kernel<true, true>), dim3(g), dim3(256), 0, 0, q, a, n4); }), bytes_copy);
		snprintf(nm, sizeof nm, "lin  copy plain ld / plain st, grid %d", g); rep(nm, T.us([&] { hipLaunchKernelGGL((lin_kernel<false, false>), dim3(g), dim3(256), 0, 0, q, a, n4); }), bytes_copy);
	}
	for (int g : {1024, 4096}) {
		char nm[128];
		snprintf(nm, sizeof nm, "slab copy plain ld / nt st, grid %d", g);   rep(nm, T.us([&] { hipLaunchKernelGGL((lin_slab_kernel<false, true>), dim3(g), dim3(256), 0, 0, q, a, n4); }), bytes_copy);
	}
	{
		const double us = T.us([&] { (void)hipMemcpyAsync(q, a, m * n * 4, hipMemcpyDeviceToDevice, 0); });
		rep("hipMemcpyAsync D2D", us, bytes_copy);
	}
#define COL(ROWS, V, MODE, NTL, NTS, DEPTH, G, LD, LDQ, label)                                                                                  \
	{                                                                                                                                       \
		char nm[160];                                                                                                                       \
		snprintf(nm, sizeof nm, "col  %s rows %d, %d B/lane, %s ld / %s st, depth %d, grid %d, ld m+%zu ldq m+%zu", label, ROWS, 4 * V,    \
		         NTL ? "nt" : "plain", NTS ? "nt" : "plain", DEPTH, G, (size_t)(LD)-m, (size_t)(LDQ)-m);                                   \
		const int nb = (int)(m / ROWS);                                                                                                     \
		rep(nm, T.us([&] { hipLaunchKernelGGL((col_kernel<ROWS, V, MODE, NTL, NTS, DEPTH>), dim3(G), dim3(256), 0, 0, q, a, (size_t)(LD), (size_t)(LDQ), nb); }), \
		    MODE == 0 ? bytes_copy : bytes_one);                                                                                            \
	}
	// the apply pass's own pattern and its neighbours
	COL(64, 1, 0, false, true, 2, 1024, m, m, "copy")
	COL(64, 1, 0, false, true, 1, 1024, m, m, "copy")
	COL(64, 1, 0, false, true, 2, 2048, m, m, "copy")
	COL(64, 4, 0, false, true, 2, 1024, m, m, "copy")
	COL(64, 4, 0, false, true, 1, 2048, m, m, "copy")
	COL(128, 4, 0, false, true, 1, 1024, m, m, "copy")
	COL(128, 4, 0, false, true, 2, 1024, m, m, "copy")
	COL(128, 4, 0, false, true, 1, 2048, m, m, "copy")
	COL(256, 4, 0, false, true, 1, 1024, m, m, "copy")
	COL(256, 4, 0, false, true, 1, 2048, m, m, "copy")
	COL(256, 4, 0, false, true, 2, 1024, m, m, "copy")
	COL(128, 4, 0, true, true, 1, 1024, m, m, "copy")
	COL(128, 4, 0, true, true, 2, 1024, m, m, "copy")
	COL(256, 4, 0, true, true, 2, 1024, m, m, "copy")
	COL(128, 4, 0, false, false, 1, 1024, m, m, "copy")
	// leading dimensions off the power of two
	COL(128, 4, 0, false, true, 1, 1024, m + 64, m + 64, "copy")
	COL(128, 4, 0, false, true, 1, 1024, m + 1024, m + 1024, "copy")
	COL(128, 4, 0, false, true, 1, 1024, m + 4096, m + 4096, "copy")
	COL(128, 4, 0, false, true, 1, 1024, m + 2048 + 64, m + 2048 + 64, "copy")
	COL(128, 4, 0, false, true, 1, 1024, m, m + 1024, "copy")
	COL(64, 1, 0, false, true, 2, 1024, m + 1024, m + 1024, "copy")
	COL(256, 4, 0, false, true, 2, 1024, m + 1024, m + 1024, "copy")
	// one direction only
	COL(128, 4, 1, false, true, 1, 1024, m, m, "load")
	COL(128, 4, 1, true, true, 1, 1024, m, m, "load")
	COL(128, 4, 1, false, true, 2, 1024, m, m, "load")
	COL(128, 4, 1, false, true, 1, 1024, m + 1024, m + 1024, "load")
	COL(128, 4, 2, false, true, 1, 1024, m, m, "store")
	COL(128, 4, 2, false, false, 1, 1024, m, m, "store")
	COL(128, 4, 2, false, true, 1, 1024, m + 1024, m + 1024, "store")
	{
		char nm[128];
		for (int g : {2048, 8192}) {
			snprintf(nm, sizeof nm, "lin  load only (copy kernel reading both), grid %d", g);
			(void)nm;
		}
	}
	(void)hipFree(a); (void)hipFree(q);
	return 0;
}
