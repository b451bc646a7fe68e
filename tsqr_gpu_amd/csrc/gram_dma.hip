// gram_dma.hip -- the first streaming pass (G = A^T A, bf16x3-split MFMA) organised per WORKGROUP around an LDS-DMA ring.
//
// Role: the R-stack reduction of the reference (src/tsqr.cu:1064-1172) collapses, for the Gram engine, into one pass over A.
// gram_bf16_kernel reads A with per-wave (c,q) loads: every load instruction touches 16 columns x 64 B, and the pass runs at
// 48 us for 268 MB where a workgroup that moves whole 256-B column runs does 40 us (tools/seq_bench.py).  Here:
//   * a workgroup owns 64-row x NP-column blocks (interleaved over the grid); every wave-instruction is one
//     global_load_lds_dwordx4 = 1 KiB = four 256-B column runs written straight into LDS (no VGPR staging);
//   * four 16-KiB slots per workgroup, DMAs for three blocks in flight while the fourth is being consumed: one raw s_barrier per
//     block and a counted s_waitcnt vmcnt (never 0 in the steady state), two workgroups per CU -> ~96 KiB in flight per CU;
//   * the 16-B slots of a column are XOR-swizzled with (column & 15) on the SOURCE side (the DMA writes LDS linearly), the
//     ds_read_b128 operand reads apply the same involution and are bank-conflict free;
//   * the four waves split a block as (K-step of 32 rows) x (half of the upper-triangular tiles), so a wave keeps five fp64 tile
//     totals instead of ten; arithmetic identical to gram_bf16_kernel (six split products, one MFMA chain from zero per K-step,
//     fp64 totals), partials in the same format.
// Used when n is a multiple of 16, A is 16-B aligned with lda % 4 == 0; the host sends the last m % 64 rows (and everything else)
// through gram_bf16_kernel.
#pragma once

namespace tsqrmi {

template <int NT> struct GramDmaCfg {
	static constexpr int NP = 16 * NT;
	static constexpr int NTRI = (NT * (NT + 1)) / 2;
	static constexpr int SLOT_BYTES = NP * 64 * 4;
	static constexpr int NSLOT = 4;
	static constexpr int RING_BYTES = NSLOT * SLOT_BYTES;
	static constexpr int HALF = (NTRI + 1) / 2;              // tiles of set 0 (set 1 takes the rest)
	static constexpr int RED_BYTES = 2 * HALF * 256 * 8;      // final reduction: one fp64 image per tile set
	static constexpr int LDS_BYTES = RING_BYTES > RED_BYTES ? RING_BYTES : RED_BYTES;
};

template <int NT>
__global__ __launch_bounds__(256, 2) void gram_dma_kernel(const GramArgs a) {
	using C = GramDmaCfg<NT>;
	constexpr int NP = C::NP, NTRI = C::NTRI, HALF = C::HALF;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	if (a.skip_status && a.skip_status[0] != 0) return;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int c = lane & 15, q = lane >> 4;
	const int kstep = wv & 1, set = wv >> 1;
	const int nblk = a.nchunks, nwg = gridDim.x;
	const int mine = (blockIdx.x < nblk) ? (nblk - 1 - (int)blockIdx.x) / nwg + 1 : 0;      // blocks blockIdx.x + k * nwg

	// DMA instruction k of this wave: columns 4 (wv + 4k) .. +3; lane L -> column +(L >> 4), physical 16-B slot L & 15 of that column
	const int dcol = lane >> 4, dp = lane & 15;
	auto issue = [&](int i) {                            // block index i of this workgroup -> ring slot i & 3
		const size_t row0 = (size_t)(blockIdx.x + (size_t)i * nwg) * 64;
		__attribute__((address_space(3))) char* slot = (__attribute__((address_space(3))) char*)smem + (i & 3) * C::SLOT_BYTES;
#pragma unroll
		for (int k = 0; k < NT; k++) {
			const int col = 4 * (wv + 4 * k) + dcol;
			const int s = dp ^ (col & 15);                 // logical slot (rows 4s .. 4s+3) stored at physical slot dp
			const float* src = a.a + (size_t)col * a.lda + row0 + 4 * s;
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
			                                 (__attribute__((address_space(3))) void*)(slot + 4 * (wv + 4 * k) * 256), 16, 0, 0);
		}
	};

	f64x4 tot[HALF];
#pragma unroll
	for (int t = 0; t < HALF; t++) tot[t] = f64x4{0.0, 0.0, 0.0, 0.0};

	if (mine > 0) issue(0);
	if (mine > 1) issue(1);
	if (mine > 2) issue(2);
	for (int i = 0; i < mine; i++) {
		// retire this wave's DMAs of block i: at most two younger blocks (2 NT instructions) may stay in flight
		const int younger = min(2, mine - 1 - i);
		if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NT) : "memory");
		else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NT) : "memory");
		else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();                     // every wave's pieces of block i have landed; everyone is done with block i - 1
		asm volatile("" ::: "memory");
		if (i + 3 < mine) issue(i + 3);                   // into the slot block i - 1 occupied
		const char* slot = smem + (i & 3) * C::SLOT_BYTES;
#ifdef TSQR_GRAM_DMA_ABLATE                               // (experiment) the ring alone: one LDS read per block, no arithmetic
		{ const f32x4 x = *reinterpret_cast<const f32x4*>(slot + threadIdx.x * 16); tot[0][0] += (double)x[0]; continue; }
#endif
		// operands of this wave's K-step (rows 32 kstep + 8q .. +7 of every column tile): two swizzled 16-B reads per tile
		bf16x8 oh[NT], om[NT], ol[NT];
#pragma unroll
		for (int t = 0; t < NT; t++) {
			// set 0 needs every column tile (as B operand), set 1 only the tiles from its first row tile (1 for NT > 1) on
			if (NT > 1 && t == 0 && set == 1) { oh[t] = om[t] = ol[t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; continue; }
			const int col = 16 * t + c;
			const int s0 = 8 * kstep + 2 * q;
			const f32x4 x0 = *reinterpret_cast<const f32x4*>(slot + (col * 16 + (s0 ^ c)) * 16);
			const f32x4 x1 = *reinterpret_cast<const f32x4*>(slot + (col * 16 + ((s0 + 1) ^ c)) * 16);
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
		f32x4 acc[HALF];
#pragma unroll
		for (int t = 0; t < HALF; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
		// the six products mm hl lh hm mh hh (smallest first) of every tile of this wave's set, as in gram_bf16_kernel
		auto run_set = [&](auto set_c) {
			constexpr int SET = decltype(set_c)::value;
#pragma unroll
			for (int pass = 3; pass < 9; pass++) {
				int idx = 0, loc = 0;
#pragma unroll
				for (int ti = 0; ti < NT; ti++)
#pragma unroll
					for (int tj = ti; tj < NT; tj++) {
						const bool in_set = SET == 0 ? idx < HALF : idx >= HALF;
						if (in_set) {
							const bf16x8 av = (pass == 4 || pass == 6 || pass == 8) ? oh[ti] : ((pass == 3 || pass == 7) ? om[ti] : ol[ti]);
							const bf16x8 bv = (pass == 5 || pass == 7 || pass == 8) ? oh[tj] : ((pass == 3 || pass == 6) ? om[tj] : ol[tj]);
							acc[loc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[loc], 0, 0, 0);
							loc++;
						}
						idx++;
					}
			}
		};
		if (set == 0) run_set(std::integral_constant<int, 0>{});
		else run_set(std::integral_constant<int, 1>{});
#pragma unroll
		for (int t = 0; t < HALF; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) tot[t][r] += (double)acc[t][r];
	}
	// workgroup sum: the two K-step waves of a set add up (odd wave -> LDS, even wave adds and stores the set's tiles)
	__builtin_amdgcn_s_barrier();                         // the ring is free (every wave has left the loop)
	asm volatile("" ::: "memory");
	double* red = reinterpret_cast<double*>(smem) + (size_t)set * HALF * 256;
	if (kstep == 1) {
#pragma unroll
		for (int t = 0; t < HALF; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[(t * 4 + r) * 64 + lane] = tot[t][r];
	}
	__syncthreads();
	if (kstep == 0) {
		double* out = a.part + (size_t)blockIdx.x * NTRI * 256;
		const int first = set == 0 ? 0 : HALF;
		const int cnt = set == 0 ? HALF : NTRI - HALF;
#pragma unroll
		for (int t = 0; t < HALF; t++)
			if (t < cnt) {
#pragma unroll
				for (int r = 0; r < 4; r++) part_store(&out[((first + t) * 4 + r) * 64 + lane], tot[t][r] + red[(t * 4 + r) * 64 + lane]);
			}
	}
}


// ---------------------------------------------------------------------------------------------
// gram_bounce_kernel: the per-wave variant that won.  The workgroup ring above needs the four waves to share a block, which
// either duplicates the bf16 split (1.75x the vector work: 61 us, measured) or leaves too few blocks in flight.  Here every wave
// stays independent as in gram_bf16_kernel (no barrier in the loop) but its 64-row x NP chunk arrives by LDS-DMA in a 16-KiB
// region of LDS that only this wave touches:
//   * 16 global_load_lds_dwordx4 per chunk, each four 256-B column runs: full 128-B lines on the memory side instead of the
//     64-B pieces of the (c,q) register loads (48 -> 40 us for the bare pass, tools/seq_bench.py);
//   * the wave reads the chunk back into registers in MFMA-operand order (rows 32 kt + 8q .. +7 of column 16 t + c: two swizzled
//     ds_read_b128 per tile and K-step), and as soon as those reads have returned it issues the DMA of its NEXT chunk -- that
//     chunk is in flight during the whole split + MFMA phase, at no cost in VGPRs;
//   * only the issuing wave's own vmcnt orders a DMA against its ds_reads, so no workgroup barrier is needed.
// Arithmetic and partial format are those of gram_bf16_kernel.  a.nchunks counts COMPLETE chunks; the host sends the last
// m % 64 rows through gram_bf16_kernel (one more partial).
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256, 2) void gram_bounce_kernel(const GramArgs a) {
	constexpr int NTRI = (NT * (NT + 1)) / 2;
	constexpr int WAVE_BYTES = NT * 16 * 64 * 4;         // one chunk
	extern __shared__ __attribute__((aligned(16))) char smem[];      // 4 * WAVE_BYTES, aliased by the final reduction (2 * NTRI * 256 doubles)
	if (a.skip_status && a.skip_status[0] != 0) return;
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	const int gw = blockIdx.x * 4 + wv;
	const int c = lane & 15, q = lane >> 4;
	char* mybuf = smem + wv * WAVE_BYTES;
	f32x4 acc[NTRI];
	f64x4 tot[NTRI];
#pragma unroll
	for (int t = 0; t < NTRI; t++) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; tot[t] = f64x4{0.0, 0.0, 0.0, 0.0}; }
	// DMA instruction k covers columns 4k .. 4k+3: lane L -> column 4k + (L >> 4), physical 16-B slot L & 15 holding the logical slot
	// (L & 15) ^ (column & 15).  (column & 15) only depends on k & 3, so four 32-bit lane offsets + a wave-uniform base per k do.
	const int dcol = lane >> 4, dp = lane & 15;
	unsigned loff[4];
#pragma unroll
	for (int j = 0; j < 4; j++) loff[j] = (unsigned)(((size_t)dcol * a.lda + 4 * (dp ^ (4 * j + dcol))) * sizeof(float));
	auto issue = [&](int ch) {                           // DMA of chunk ch (64 full rows) into this wave's region
		const char* base = reinterpret_cast<const char*>(a.a + (size_t)ch * 64);
#pragma unroll
		for (int k = 0; k < 4 * NT; k++) {
			const char* src = base + (size_t)(4 * k) * a.lda * sizeof(float) + loff[k & 3];
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
			                                 (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)mybuf + k * 1024), 16, 0, 0);
		}
	};
	if (gw < a.nwaves) {
		const int ch_end = a.nchunks, ch_step = a.nwaves;     // complete 64-row chunks only (the host sends ragged rows elsewhere)
		int ch = gw;
		if (ch < ch_end) issue(ch);
		for (; ch < ch_end; ch += ch_step) {
			float x[NT][2][8];                            // x[t][kt][j] = A(row 64 ch + 32 kt + 8 q + j, column 16 t + c)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
			for (int t = 0; t < NT; t++)
#pragma unroll
				for (int kt = 0; kt < 2; kt++) {
					const int col = 16 * t + c, s0 = 8 * kt + 2 * q;
					const f32x4 x0 = *reinterpret_cast<const f32x4*>(mybuf + (col * 16 + (s0 ^ c)) * 16);
					const f32x4 x1 = *reinterpret_cast<const f32x4*>(mybuf + (col * 16 + ((s0 + 1) ^ c)) * 16);
#pragma unroll
					for (int j = 0; j < 4; j++) { x[t][kt][j] = x0[j]; x[t][kt][4 + j] = x1[j]; }
				}
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the region is free again: prefetch this wave's next chunk
			__builtin_amdgcn_sched_barrier(0);
			if (ch + ch_step < ch_end) issue(ch + ch_step);
#pragma unroll
			for (int kt = 0; kt < 2; kt++) {
				bf16x8 oh[NT], om[NT], ol[NT];
#pragma unroll
				for (int t = 0; t < NT; t++) {
					u32x4 hh, mm, ll;
#pragma unroll
					for (int jp = 0; jp < 4; jp++) {
						unsigned h, m, lo;
						split3_pair(x[t][kt][2 * jp], x[t][kt][2 * jp + 1], h, m, lo);
						hh[jp] = h; mm[jp] = m; ll[jp] = lo;
					}
					oh[t] = __builtin_bit_cast(bf16x8, hh);
					om[t] = __builtin_bit_cast(bf16x8, mm);
					ol[t] = __builtin_bit_cast(bf16x8, ll);
				}
#pragma unroll
				for (int pass = 3; pass < 9; pass++) {       // mm hl lh hm mh hh, smallest first (as gram_bf16_kernel)
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
#if !defined(TSQR_GRAM_FLUSH_PER_CHUNK)
#pragma unroll
				for (int t = 0; t < NTRI; t++) {
#pragma unroll
					for (int r = 0; r < 4; r++) tot[t][r] += (double)acc[t][r];
					acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
				}
#endif
			}
#if defined(TSQR_GRAM_FLUSH_PER_CHUNK)                    // (experiment) one fp64 flush per 64 rows instead of per 32
#pragma unroll
			for (int t = 0; t < NTRI; t++) {
#pragma unroll
				for (int r = 0; r < 4; r++) tot[t][r] += (double)acc[t][r];
				acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
			}
#endif
		}
	}
	// workgroup sum in fp64 (LDS aliases the bounce regions: every DMA has been waited for by its wave, the barrier orders the rest)
	__syncthreads();
	double* red = reinterpret_cast<double*>(smem);        // [2][NTRI * 256]
	if (wv >= 2) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) red[(size_t)(wv - 2) * NTRI * 256 + (t * 4 + r) * 64 + lane] = tot[t][r];
	}
	__syncthreads();
	if (wv < 2) {
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) tot[t][r] += red[(size_t)wv * NTRI * 256 + (t * 4 + r) * 64 + lane];
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
		double* out = a.part + (size_t)blockIdx.x * NTRI * 256;
#pragma unroll
		for (int t = 0; t < NTRI; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) part_store(&out[(t * 4 + r) * 64 + lane], tot[t][r] + red[(t * 4 + r) * 64 + lane]);
	}
}

}  // namespace tsqrmi
