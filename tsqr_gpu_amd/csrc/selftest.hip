// selftest.hip -- GPU unit tests of the wave-level primitives and MFMA operand layouts used by tsqr_kernels.hip.
// Built into libtsqr_selftest.so; driven by tests/test_gpu_primitives.py.
#include <hip/hip_runtime.h>
#include "tsqr_kernels.hip"

namespace {
using namespace tsqrmi;

// out[0*64+l] = bcast16<5>(l), out[1*64+l] = bcast16<0>, out[2*64+l] = bcast16<15>, out[3*64+l] = xq_sum(l), out[4*64+l] = xq_sum(1<<q)
__global__ void prim_kernel(float* out) {
	const int l = threadIdx.x;
	const float x = (float)l;
	out[0 * 64 + l] = bcast16<5>(x);
	out[1 * 64 + l] = bcast16<0>(x);
	out[2 * 64 + l] = bcast16<15>(x);
	out[3 * 64 + l] = xq_sum(x);
	out[4 * 64 + l] = xq_sum((float)(1 << (4 * (l >> 4))) * (float)(1 + (l & 15)));
}

// D = A(16x4) * B(4x16) with v_mfma_f32_16x16x4_f32; a[i*4+k], b[k*16+j] row-major inputs; d[i*16+j]
__global__ void mfma_f32_kernel(float* d, const float* a, const float* b) {
	const int l = threadIdx.x;
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
	acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(l & 15) * 4 + (l >> 4)], b[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
	for (int i = 0; i < 4; i++) d[(4 * (l >> 4) + i) * 16 + (l & 15)] = acc[i];
}

// D = A(16x32) * B(32x16) with v_mfma_f32_16x16x32_bf16 on exactly representable inputs
__global__ void mfma_bf16_kernel(float* d, const float* a, const float* b) {
	const int l = threadIdx.x;
	bf16x8 av, bv;
	for (int j = 0; j < 8; j++) {
		av[j] = (short)f2bf(a[(l & 15) * 32 + 8 * (l >> 4) + j]);
		bv[j] = (short)f2bf(b[(8 * (l >> 4) + j) * 16 + (l & 15)]);
	}
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
	acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
	for (int i = 0; i < 4; i++) d[(4 * (l >> 4) + i) * 16 + (l & 15)] = acc[i];
}

// split3 round trip: out[3*i..] = hi, mid, lo as floats
__global__ void split_kernel(float* out, const float* in, int n) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	unsigned h, m, lo;
	split3(in[i], h, m, lo);
	out[3 * i] = __builtin_bit_cast(float, h << 16);
	out[3 * i + 1] = __builtin_bit_cast(float, m << 16);
	out[3 * i + 2] = __builtin_bit_cast(float, lo << 16);
}
}  // namespace

extern "C" {
int tsqr_selftest_prims(float* out) { hipLaunchKernelGGL(prim_kernel, dim3(1), dim3(64), 0, 0, out); return (int)hipDeviceSynchronize(); }
int tsqr_selftest_mfma_f32(float* d, const float* a, const float* b) { hipLaunchKernelGGL(mfma_f32_kernel, dim3(1), dim3(64), 0, 0, d, a, b); return (int)hipDeviceSynchronize(); }
int tsqr_selftest_mfma_bf16(float* d, const float* a, const float* b) { hipLaunchKernelGGL(mfma_bf16_kernel, dim3(1), dim3(64), 0, 0, d, a, b); return (int)hipDeviceSynchronize(); }
int tsqr_selftest_split(float* out, const float* in, int n) { hipLaunchKernelGGL(split_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, out, in, n); return (int)hipDeviceSynchronize(); }
}
