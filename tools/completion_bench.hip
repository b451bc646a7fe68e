// completion_bench.hip -- how should a blocking call learn that its last kernel has finished?
// Measures, behind an ~80 us streaming copy (the shape of the apply pass: 268 MB read + 268 MB nontemporal write), the time
// per iteration of [launch copy; completion mechanism] for:
//   A  one-thread kernel that stores a sequence number to pinned host memory, host spins on the word   (what the engine did in r02)
//   B  hipStreamWriteValue32 on the pinned word (no kernel), host spins
//   C  hipEventRecord + spin on hipEventQuery
//   D  hipStreamSynchronize
//   E  spin on hipStreamQuery
//   F  hipEventRecord + hipEventSynchronize
// Build: hipcc --offload-arch=gfx950 -O3 -o completion_bench completion_bench.hip ; run: ./completion_bench [iters]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n4) {
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
		const f32x4 v = src[i];
		__builtin_nontemporal_store(v, &dst[i]);
	}
}
__global__ void flag_kernel(unsigned* flag, unsigned seq) { *reinterpret_cast<volatile unsigned*>(flag) = seq; }

int main(int argc, char** argv) {
	const int iters = argc > 1 ? atoi(argv[1]) : 300;
	const size_t bytes = (size_t)1 << 28;
	f32x4 *src, *dst;
	CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
	CK(hipMemset(src, 1, bytes));
	unsigned* hflag; unsigned* dflag;
	CK(hipHostMalloc(reinterpret_cast<void**>(&hflag), 64, hipHostMallocDefault));
	CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&dflag), hflag, 0));
	hipStream_t st; CK(hipStreamCreate(&st));
	hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	hipEvent_t evt; CK(hipEventCreate(&evt));
	const size_t n4 = bytes / 16;
	auto launch = [&]() { hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, st, src, dst, n4); };
	volatile unsigned* flag = hflag;
	unsigned seq = 0;
	const char* names[] = {"A flag kernel + spin", "B hipStreamWriteValue32 + spin", "C event record + spin on hipEventQuery",
	                       "D hipStreamSynchronize", "E spin on hipStreamQuery", "F event record + hipEventSynchronize",
	                       "G timing event record + spin on hipEventQuery"};
	for (int rep = 0; rep < 2; rep++)
		for (int v = 0; v < 7; v++) {
			// warm
			for (int i = 0; i < 50; i++) launch();
			CK(hipStreamSynchronize(st));
			bool ok = true;
			auto t0 = std::chrono::steady_clock::now();
			for (int i = 0; i < iters && ok; i++) {
				launch();
				switch (v) {
					case 0:
						seq++; *flag = 0;
						hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, st, dflag, seq);
						while (*flag != seq) _mm_pause();
						break;
					case 1: {
						seq++; *flag = 0;
						hipError_t e = hipStreamWriteValue32(st, dflag, seq, 0);
						if (e != hipSuccess) { printf("%s: %s\n", names[v], hipGetErrorString(e)); ok = false; (void)hipGetLastError(); break; }
						while (*flag != seq) _mm_pause();
						break;
					}
					case 2:
						CK(hipEventRecord(ev, st));
						while (hipEventQuery(ev) == hipErrorNotReady) _mm_pause();
						break;
					case 3: CK(hipStreamSynchronize(st)); break;
					case 4: while (hipStreamQuery(st) == hipErrorNotReady) _mm_pause(); break;
					case 5: CK(hipEventRecord(ev, st)); CK(hipEventSynchronize(ev)); break;
					case 6:
						CK(hipEventRecord(evt, st));
						while (hipEventQuery(evt) == hipErrorNotReady) _mm_pause();
						break;
				}
			}
			CK(hipStreamSynchronize(st));
			const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
			if (ok) printf("rep %d  %-48s %8.2f us per [copy + completion]\n", rep, names[v], us);
		}
	// the copy alone, back to back, for reference
	for (int i = 0; i < 50; i++) launch();
	CK(hipStreamSynchronize(st));
	auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < iters; i++) launch();
	CK(hipStreamSynchronize(st));
	printf("copy kernels back to back: %8.2f us each\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters);
	return 0;
}
