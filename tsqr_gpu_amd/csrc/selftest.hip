// selftest.hip -- GPU unit tests of the wave-level primitives and MFMA operand layouts used by tsqr_kernels.hip.
// Built into libtsqr_selftest.so; driven by tests/test_gpu_primitives.py.
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "tsqr_kernels.hip"
#include "tsqr_wide.hip"

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

// ---- chol16_kernel on summed Gram tiles (the step between the two streaming passes): one launch (results) + `reps` timed launches.
// level 2: tiles in the f32 accumulator layout (bf16-split Gram pass), 1 / 3: f64 accumulator layout (fp64 Gram pass, 3 = shifted)
extern "C" float tsqr_selftest_chol(float* r, size_t ldr, float* z, unsigned* status, const double* gsum, int n, int NT,
                                    int level, double rows, int reps) {
	tsqrmi::CholArgs a{};
	a.r = r; a.ldr = ldr; a.z = z; a.status = status; a.host_status = nullptr; a.gsum = gsum; a.prev_status = nullptr; a.rows_dev = nullptr;
	a.rows = rows; a.shift_coef = (level == 3) ? 11.0 * 1.1102230246251565e-16 : 0.0; a.n = n; a.NT = NT; a.level = level; a.scond_floor = 4.0f;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	auto launch = [&]() { hipLaunchKernelGGL(tsqrmi::chol16_kernel, dim3(1), dim3(1024), 0, 0, a); };
	launch();
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++) launch();
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0.f;
	hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1);
	if (hipGetLastError() != hipSuccess) return -1.0f;
	return reps > 0 ? ms / reps : 0.0f;
}


// ---- gram_blk_chain_kernel against the launches it merges (tsqr_mi_qr_f32_loop's chained schedule): m x 64 matrix a (m % 128 == 0).
// r / z / status [0]: gram_blk_kernel -> gram_reduce1_kernel -> chol16_kernel.  [1]: chain role of a fused launch on the same partials,
// whose Gram role writes a second set of partials; [2]: chain role of a SECOND fused launch on that second set (ticket re-armed by the
// first).  scratch: 2 * nparts * 2560 + 2 * 2568 doubles + 2 words; part_equal <- 1 when both sets of partials are bitwise equal.
extern "C" int tsqr_selftest_chain(const float* a, size_t lda, size_t m, int nparts, float* r3, float* z3, unsigned* status3, double* scratch, int* part_equal_host) {
	const int nelem = 2560, nred = nelem / 16;
	double* partA = scratch; double* partB = partA + (size_t)nparts * nelem;
	double* gsum1 = partB + (size_t)nparts * nelem; double* gsum2 = gsum1 + 2568;
	unsigned* ticket = reinterpret_cast<unsigned*>(gsum2 + 2568);
	(void)hipMemset(ticket, 0, 8);
	(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_blk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GB_LDS_BYTES);
	(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_blk_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GB_LDS_BYTES);
	tsqrmi::GramArgs ga{};
	ga.a = a; ga.lda = lda; ga.m = m; ga.n = 64; ga.nchunks = (int)(m / 128); ga.part = partA;
	auto chol = [&](int k, double* gs) {
		tsqrmi::CholArgs c{};
		c.r = r3 + 4096 * k; c.ldr = 64; c.z = z3 + 4096 * k; c.status = status3 + 16 * k; c.gsum = gs; c.rows = (double)m; c.n = 64; c.NT = 4; c.level = 2;
		c.scond_floor = 4.0f;
		return c;
	};
	hipLaunchKernelGGL(tsqrmi::gram_blk_kernel, dim3(nparts), dim3(256), tsqrmi::GB_LDS_BYTES, 0, ga);
	hipLaunchKernelGGL(tsqrmi::gram_reduce1_kernel, dim3(nred), dim3(256), 0, 0, gsum1, partA, nparts, nelem, (double)m, nullptr, (size_t)0, nullptr, 0);
	hipLaunchKernelGGL(tsqrmi::chol16_kernel, dim3(1), dim3(1024), 0, 0, chol(0, gsum1));
	tsqrmi::ChainArgs ch{};
	ch.chol = chol(1, gsum2); ch.part = partA; ch.nparts = nparts; ch.ticket = ticket; ch.nred = nred;
	ga.part = partB;
	hipLaunchKernelGGL(tsqrmi::gram_blk_chain_kernel, dim3(nred + nparts), dim3(256), tsqrmi::GB_LDS_BYTES, 0, ga, ch);
	ch.chol = chol(2, gsum2); ch.part = partB;
	ga.part = partA;                                     // (rewrites the first set with the same values)
	hipLaunchKernelGGL(tsqrmi::gram_blk_chain_kernel, dim3(nred + nparts), dim3(256), tsqrmi::GB_LDS_BYTES, 0, ga, ch);
	if (hipDeviceSynchronize() != hipSuccess) return -1;
	std::vector<double> ha((size_t)nparts * nelem), hb((size_t)nparts * nelem);
	(void)hipMemcpy(ha.data(), partA, ha.size() * 8, hipMemcpyDeviceToHost);
	(void)hipMemcpy(hb.data(), partB, hb.size() * 8, hipMemcpyDeviceToHost);
	*part_equal_host = memcmp(ha.data(), hb.data(), ha.size() * 8) == 0 ? 1 : 0;
	unsigned t = 1;
	(void)hipMemcpy(&t, ticket, 4, hipMemcpyDeviceToHost);
	if (t != 0) return -2;                               // the last adder did not re-arm the ticket
	return (int)hipGetLastError();
}

// ---- gram_blk_kernel's body with a start and an end stamp per workgroup (s_memrealtime, 100 MHz) and where it ran (XCC_ID, HW_ID):
// how evenly the static partition of the Gram pass finishes (tools/gram_balance.py) ----
__global__ __launch_bounds__(256, 2) void gram_blk_stamp_kernel(const tsqrmi::GramArgs a, unsigned long long* stamps) {
	extern __shared__ __attribute__((aligned(16))) float gb_as_st[];
	unsigned long long t0 = 0;
	if (threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
	tsqrmi::gram_blk_body(a, gb_as_st, blockIdx.x, gridDim.x);
	if (threadIdx.x == 0) {
		unsigned xcc, hw;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		stamps[4 * blockIdx.x + 0] = t0;
		stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
		stamps[4 * blockIdx.x + 2] = xcc;
		stamps[4 * blockIdx.x + 3] = hw;
	}
}
extern "C" int tsqr_selftest_gram_balance(const float* a, size_t lda, size_t m, int nparts, double* part, unsigned long long* stamps, int warm, int old_share) {
	(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_blk_stamp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GB_LDS_BYTES);
	tsqrmi::GramArgs ga{};
	ga.a = a; ga.lda = lda; ga.m = m; ga.n = 64; ga.nchunks = (int)(m / 128); ga.part = part; ga.old_share = old_share;
	for (int i = 0; i <= warm; i++) hipLaunchKernelGGL(gram_blk_stamp_kernel, dim3(nparts), dim3(256), tsqrmi::GB_LDS_BYTES, 0, ga, stamps);
	return (int)hipDeviceSynchronize();
}

// ---- the apply pass (apply_wg_kernel<1, 4, false, 64>: Q = A * Z, bf16x3) with a start and an end stamp per workgroup: how evenly the
// four workgroups of a CU finish (tools/gram_balance.py apply) ----
__global__ __launch_bounds__(256) void apply_stamp_kernel(const tsqrmi::ApplyArgs a, unsigned long long* stamps) {
	unsigned long long t0 = 0;
	if (threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
	tsqrmi::apply_wg_body<1, 4, false, 64, false>(a);
	if (threadIdx.x == 0) {
		unsigned xcc, hw;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		stamps[4 * blockIdx.x + 0] = t0;
		stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
		stamps[4 * blockIdx.x + 2] = xcc;
		stamps[4 * blockIdx.x + 3] = hw;
	}
}
extern "C" int tsqr_selftest_apply_balance(float* q, const float* a, size_t ld, size_t m, const float* z, int nwg, unsigned long long* stamps, int warm,
                                           int s0, int s1, int s2, int s3, int even_share) {
	constexpr size_t lds = sizeof(float) * 64 * (64 + 4) + (size_t)3 * 6 * 512 * 2;
	(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&apply_stamp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	tsqrmi::ApplyArgs aa{};
	aa.a = a; aa.lda = ld; aa.q = q; aa.ldq = ld; aa.m = m; aa.n = 64; aa.z = z;
	aa.nchunks = (int)((m + 63) / 64); aa.nwaves = nwg; aa.cpw = 0;
	aa.share[0] = s0; aa.share[1] = s1; aa.share[2] = s2; aa.share[3] = s3; aa.even_share = even_share;
	for (int i = 0; i <= warm; i++) hipLaunchKernelGGL(apply_stamp_kernel, dim3(nwg), dim3(256), lds, 0, aa, stamps);
	return (int)hipDeviceSynchronize();
}

// ---- gram_wide_chain_kernel against the launches it merges (a stream of 128-column calls): m x 128 matrix a (m % 64 == 0).
// [0]: gram_wide_kernel<true> -> gram_reduce1_kernel -> chol_wide_kernel.  [1]: the chain role of a fused launch on the same summed
// tiles (its Gram role writes a second set of partials, compared bitwise with the first).  r3: 3 x 128 x 128, zw3: 3 x 128 x 128,
// status3: 3 x 16 words; scratch: 2 * nwg * 36 * 256 + 36 * 256 + 16 doubles + 3 * 2 * 4096 floats + 64 words.
extern "C" int tsqr_selftest_wide_chain(const float* a, size_t lda, size_t m, int nwg, float* r3, float* zw3, unsigned* status3, double* scratch, int* part_equal_host) {
	const int nelem = 36 * 256;
	double* partA = scratch; double* partB = partA + (size_t)nwg * nelem;
	double* gsum = partB + (size_t)nwg * nelem;
	float* zf = reinterpret_cast<float*>(gsum + nelem + 16);
	unsigned* st12 = reinterpret_cast<unsigned*>(zf + 3 * 2 * 4096);
	(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_wide_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GW_LDS_BYTES);
	(void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tsqrmi::gram_wide_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, tsqrmi::GWC_LDS_BYTES);
	tsqrmi::GramWideArgs ga{};
	ga.a = a; ga.lda = lda; ga.m = m; ga.n = 128; ga.blk0 = 0; ga.nblk = (int)(m / 64); ga.part = partA;
	auto cw = [&](int k) {
		tsqrmi::CholWideArgs c{};
		c.gsum = gsum; c.r = r3 + 16384 * k; c.ldr = 128; c.n = 128; c.zf1 = zf + 8192 * k; c.zf2 = zf + 8192 * k + 4096; c.zw = zw3 + 16384 * k;
		c.st1 = st12 + 32 * k; c.st2 = st12 + 32 * k + 16; c.status = status3 + 16 * k; c.rows = (double)m; c.scond_floor = 4.0f;
		return c;
	};
	hipLaunchKernelGGL(tsqrmi::gram_wide_kernel<true>, dim3(nwg), dim3(512), tsqrmi::GW_LDS_BYTES, 0, ga);
	hipLaunchKernelGGL(tsqrmi::gram_reduce1_kernel, dim3(nelem / 16), dim3(256), 0, 0, gsum, partA, nwg, nelem, (double)m, nullptr, (size_t)0, nullptr, 0);
	hipLaunchKernelGGL(tsqrmi::chol_wide_kernel, dim3(1), dim3(1024), 0, 0, cw(0));
	ga.part = partB;
	hipLaunchKernelGGL(tsqrmi::gram_wide_chain_kernel, dim3(1 + nwg), dim3(512), tsqrmi::GWC_LDS_BYTES, 0, ga, cw(1));
	if (hipDeviceSynchronize() != hipSuccess) return -1;
	std::vector<double> ha((size_t)nwg * nelem), hb((size_t)nwg * nelem);
	(void)hipMemcpy(ha.data(), partA, ha.size() * 8, hipMemcpyDeviceToHost);
	(void)hipMemcpy(hb.data(), partB, hb.size() * 8, hipMemcpyDeviceToHost);
	*part_equal_host = memcmp(ha.data(), hb.data(), ha.size() * 8) == 0 ? 1 : 0;
	return (int)hipGetLastError();
}

// ---- in-kernel time stamps of chol16_kernel (this library is built with -DTSQR_CHOL_STAMPS): out[4][160] shader-clock values ----
extern "C" int tsqr_selftest_chol_stamps(unsigned long long* out_dev, float* r, size_t ldr, float* z, unsigned* status, const double* gsum, int n, int NT,
                                         int level, double rows) {
#ifdef TSQR_CHOL_STAMPS
	tsqrmi::CholArgs a{};
	a.r = r; a.ldr = ldr; a.z = z; a.status = status; a.host_status = nullptr; a.gsum = gsum; a.prev_status = nullptr; a.rows_dev = nullptr;
	a.rows = rows; a.shift_coef = 0.0; a.n = n; a.NT = NT; a.level = level; a.scond_floor = 4.0f;
	unsigned long long* null_out = nullptr;
	hipMemcpyToSymbol(HIP_SYMBOL(tsqrmi::g_chol_stamp_out), &null_out, sizeof(null_out));
	auto launch = [&]() { hipLaunchKernelGGL(tsqrmi::chol16_kernel, dim3(1), dim3(1024), 0, 0, a); };
	for (int i = 0; i < 20; i++) launch();               // warm: clocks, instruction cache
	hipMemcpyToSymbol(HIP_SYMBOL(tsqrmi::g_chol_stamp_out), &out_dev, sizeof(out_dev));
	launch();
	hipMemcpyToSymbol(HIP_SYMBOL(tsqrmi::g_chol_stamp_out), &null_out, sizeof(null_out));
	return (int)hipDeviceSynchronize();
#else
	return -1;
#endif
}

// ---- launch-path costs: wall time per iteration of {k dependent tiny kernels [+ 4-byte D2H copy] + stream sync} ----
__global__ void tiny_kernel(unsigned* p, unsigned* hostflag) {
	if (threadIdx.x == 0) { p[0] += 1; if (hostflag) { __threadfence_system(); *reinterpret_cast<volatile unsigned*>(hostflag) = p[0]; } }
}
#include <chrono>
extern "C" double tsqr_selftest_launch_cost(unsigned* dev_word, unsigned* pinned, int nkernels, int with_copy, int host_flag_spin, int iters) {
	hipStream_t st = 0;
	(void)hipMemset(dev_word, 0, 4);
	(void)hipDeviceSynchronize();
	unsigned expect = 0;
	const auto t0 = std::chrono::steady_clock::now();
	for (int it = 0; it < iters; it++) {
		for (int k = 0; k < nkernels; k++) {
			const bool last = (k == nkernels - 1);
			hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st, dev_word, (host_flag_spin == 1 && last) ? pinned : nullptr);
		}
		expect += (unsigned)nkernels;
		if (host_flag_spin == 1) {
			while (*reinterpret_cast<volatile unsigned*>(pinned) != expect) { }
		} else if (host_flag_spin == 2) {
			if (with_copy) (void)hipMemcpyAsync(pinned, dev_word, 4, hipMemcpyDeviceToHost, st);
			while (hipStreamQuery(st) == hipErrorNotReady) { }
		} else {
			if (with_copy) (void)hipMemcpyAsync(pinned, dev_word, 4, hipMemcpyDeviceToHost, st);
			(void)hipStreamSynchronize(st);
		}
	}
	const auto t1 = std::chrono::steady_clock::now();
	(void)hipDeviceSynchronize();
	return std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
}


// ---- a sequence of skeleton passes over the same A / Q, each timed separately (what the two streaming passes of a call can reach
// when they alternate): pass p = {mode: 0 copy (nt stores) / 1 load only / 2 store only / 3 load only in the per-wave (c,q) chunk pattern,
// dir: 0 forward / 1 backward block order, nt_load: 1 = nontemporal loads}
template <int MODE, bool NTL>
__global__ __launch_bounds__(256) void seq_wg_kernel(float* q, const float* a, size_t ld, size_t m, int nblocks, int nwg, int backward) {
	constexpr int ROWS = 128, NP = 64;
	constexpr int LPC = ROWS / 4, CPI = 64 / LPC, NI = NP / (4 * CPI);
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int lcol = lane / LPC, lrow = 4 * (lane % LPC);
	tsqrmi::f32x4 acc = {0.f, 0.f, 0.f, 0.f};
	if (MODE == 3) {
		// per-wave chunks of 64 rows x 64 columns: lane (c,q) loads 16 B of column 16 ct + c at rows 16 rt + 4 q (64-B runs per column and instruction)
		const int c = lane & 15, qq = lane >> 4;
		const int nch = (int)(m / 64), gw = blockIdx.x * 4 + wv, nw = nwg * 4;
		for (int ch0 = gw; ch0 < nch; ch0 += nw) {
			const int ch = backward ? nch - 1 - ch0 : ch0;
#pragma unroll
			for (int ct = 0; ct < 4; ct++)
#pragma unroll
				for (int rt = 0; rt < 4; rt++) {
					const float* src = a + (size_t)(16 * ct + c) * ld + (size_t)ch * 64 + 16 * rt + 4 * qq;
					acc += NTL ? __builtin_nontemporal_load(reinterpret_cast<const tsqrmi::f32x4u*>(src)) : *reinterpret_cast<const tsqrmi::f32x4u*>(src);
				}
		}
	} else {
		for (int b0 = blockIdx.x; b0 < nblocks; b0 += nwg) {
			const int b = backward ? nblocks - 1 - b0 : b0;
			tsqrmi::f32x4 v[NI];
#pragma unroll
			for (int k = 0; k < NI; k++) {
				const size_t off = (size_t)((wv + 4 * k) * CPI + lcol) * ld + (size_t)b * ROWS + lrow;
				if (MODE != 2) v[k] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const tsqrmi::f32x4u*>(a + off)) : *reinterpret_cast<const tsqrmi::f32x4u*>(a + off);
				else v[k] = tsqrmi::f32x4{(float)b, (float)k, 1.f, 2.f};
			}
#pragma unroll
			for (int k = 0; k < NI; k++) {
				const size_t off = (size_t)((wv + 4 * k) * CPI + lcol) * ld + (size_t)b * ROWS + lrow;
				if (MODE == 0 || MODE == 2) __builtin_nontemporal_store(v[k], reinterpret_cast<tsqrmi::f32x4u*>(q + off));
				else acc += v[k];
			}
		}
	}
	if ((MODE == 1 || MODE == 3) && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) q[0] = acc[0];
}
__global__ void idle_spin_kernel(float* q, int ticks) {       // one wave spinning for ticks x 10 ns (s_memrealtime runs at 100 MHz)
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
	if (ticks < 0) q[0] = 1.f;
}
template <int MODE> static void launch_seq(float* q, const float* a, size_t ld, size_t m, int nwg, int backward, int ntl) {
	const int nblocks = (int)(m / 128);
	if (ntl) hipLaunchKernelGGL((seq_wg_kernel<MODE, true>), dim3(nwg), dim3(256), 0, 0, q, a, ld, m, nblocks, nwg, backward);
	else hipLaunchKernelGGL((seq_wg_kernel<MODE, false>), dim3(nwg), dim3(256), 0, 0, q, a, ld, m, nblocks, nwg, backward);
}
extern "C" int tsqr_selftest_seq(float* q, const float* a, size_t ld, size_t m, int nwg, int npass, const int* modes, const int* dirs, const int* ntls,
                                 int reps, float* out_us) {
	hipEvent_t ev[2 * 8];
	if (npass > 8) return -1;
	for (int i = 0; i < 2 * npass; i++) (void)hipEventCreate(&ev[i]);
	for (int p = 0; p < npass; p++) out_us[p] = 0.f;
	for (int it = 0; it < reps + 2; it++) {
		for (int p = 0; p < npass; p++) {
			(void)hipEventRecord(ev[2 * p], 0);
			switch (modes[p]) {
				case 0: launch_seq<0>(q, a, ld, m, nwg, dirs[p], ntls[p]); break;
				case 1: launch_seq<1>(q, a, ld, m, nwg, dirs[p], ntls[p]); break;
				case 2: launch_seq<2>(q, a, ld, m, nwg, dirs[p], ntls[p]); break;
				case 4: hipLaunchKernelGGL(idle_spin_kernel, dim3(1), dim3(64), 0, 0, q, 100 * dirs[p]); break;   // dirs = microseconds
				default: launch_seq<3>(q, a, ld, m, nwg, dirs[p], ntls[p]); break;
			}
			(void)hipEventRecord(ev[2 * p + 1], 0);
		}
		(void)hipDeviceSynchronize();
		if (it >= 2)
			for (int p = 0; p < npass; p++) { float ms = 0.f; (void)hipEventElapsedTime(&ms, ev[2 * p], ev[2 * p + 1]); out_us[p] += ms * 1e3f / reps; }
	}
	for (int i = 0; i < 2 * npass; i++) (void)hipEventDestroy(ev[i]);
	return (int)hipGetLastError();
}

// ---- achievable v_mfma_f64_16x16x4_f64 rate: 10 independent accumulators per wave, no memory traffic ----
__global__ __launch_bounds__(256) void mfma_f64_rate_kernel(double* out, int iters) {
	tsqrmi::f64x4 acc[10];
#pragma unroll
	for (int t = 0; t < 10; t++) acc[t] = tsqrmi::f64x4{0.0, 0.0, 0.0, 0.0};
	double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int t = 0; t < 10; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
	}
	double s = 0.0;
#pragma unroll
	for (int t = 0; t < 10; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
	if (s == 123.456) out[0] = s;
}
extern "C" float tsqr_selftest_mfma_f64_rate(double* out, int wgs, int iters) {
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(wgs), dim3(256), 0, 0, out, iters);
	(void)hipEventRecord(e0, 0);
	hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(wgs), dim3(256), 0, 0, out, iters);
	(void)hipEventRecord(e1, 0);
	(void)hipEventSynchronize(e1);
	float ms = 0.f;
	(void)hipEventElapsedTime(&ms, e0, e1);
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	return ms;
}

// ---- do vector instructions issue in the shadow of MFMAs?  Per iteration ten independent v_mfma_f32_16x16x32_bf16 and / or thirty
// independent vector instructions (v_pk_add_f32, v_cvt_pk_bf16_f32 + v_and), interleaved 1 : 3 or in two blocks; shader cycles
// (s_memtime) per iteration of wave 0.  mode 0 MFMA only, 1 vector only, 2 interleaved, 3 blocks ----
template <int MODE>
__global__ __launch_bounds__(256) void issue_overlap_kernel(float* out, unsigned long long* cyc, int iters) {
	tsqrmi::f32x4 acc[10];
#pragma unroll
	for (int t = 0; t < 10; t++) acc[t] = tsqrmi::f32x4{0.f, 0.f, 0.f, 0.f};
	tsqrmi::bf16x8 a, b;
	{
		tsqrmi::u32x4 ua = {threadIdx.x * 3u + 1u, threadIdx.x, 7u, 9u}, ub = {threadIdx.x * 5u + 2u, 1u, threadIdx.x, 3u};
		a = __builtin_bit_cast(tsqrmi::bf16x8, ua); b = __builtin_bit_cast(tsqrmi::bf16x8, ub);
	}
	tsqrmi::f32x2_t x[10], y[10];
	unsigned z[10];
#pragma unroll
	for (int i = 0; i < 10; i++) { x[i] = tsqrmi::f32x2_t{1.0f + threadIdx.x, 2.0f + i}; y[i] = tsqrmi::f32x2_t{1e-3f * i, 1e-4f}; z[i] = i; }
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++) {
		if (MODE != 1) {
#pragma unroll
			for (int t = 0; t < 10; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
		}
		if (MODE != 0) {
#pragma unroll
			for (int i = 0; i < 10; i++) {
				x[i] = x[i] + y[i];                                  // v_pk_add_f32
				const unsigned h = tsqrmi::cvt_pk_bf16(x[i][0], x[i][1]);    // v_cvt_pk_bf16_f32
				z[i] = (z[i] & 0xffff0000u) ^ h;                     // v_and_or / v_bfi-like
			}
		}
		if (MODE == 2) {
#pragma unroll
			for (int t = 0; t < 10; t++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
		}
		if (MODE == 3) {
			__builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
			__builtin_amdgcn_sched_group_barrier(0x002, 40, 0);
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float sum = 0.f;
#pragma unroll
	for (int t = 0; t < 10; t++) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3] + x[t][0] + x[t][1] + (float)z[t];
	if (sum == 123.456f) out[0] = sum;
	if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
extern "C" int tsqr_selftest_issue_overlap(float* out, unsigned long long* cyc, int mode, int wgs, int iters) {
	switch (mode) {
		case 0: hipLaunchKernelGGL(issue_overlap_kernel<0>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters); break;
		case 1: hipLaunchKernelGGL(issue_overlap_kernel<1>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters); break;
		case 2: hipLaunchKernelGGL(issue_overlap_kernel<2>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters); break;
		default: hipLaunchKernelGGL(issue_overlap_kernel<3>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters); break;
	}
	return (int)hipDeviceSynchronize();
}

// ---- LDS bank behaviour of ds_read_b128: lane l = 16 q + c reads 16 bytes at dword (A c + B q); run under
// rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS (tools/lds_pattern.py): which operand layouts are conflict free ----
__global__ __launch_bounds__(256) void lds_b128_pattern_kernel(float* out, int A, int B, int iters) {
	__shared__ __attribute__((aligned(16))) unsigned buf[16384];
	for (int i = threadIdx.x; i < 16384; i += 256) buf[i] = i;
	__syncthreads();
	const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
	const unsigned* p = &buf[(A * c + B * q) & 16380];
	unsigned acc = 0;
	for (int i = 0; i < iters; i++) {
		typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
		u32x4_t v;
		asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(p + ((i & 7) * 1024))) : "memory");
		acc += v[0] ^ v[1] ^ v[2] ^ v[3];
	}
	if (acc == 0x12345678u) out[0] = 1.0f;
}
extern "C" int tsqr_selftest_lds_pattern(float* out, int A, int B, int iters) {
	hipLaunchKernelGGL(lds_b128_pattern_kernel, dim3(256), dim3(256), 0, 0, out, A, B, iters);
	return (int)hipDeviceSynchronize();
}
