/*
 * oracle/ref_tsqr.c -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * CPU restatement, in plain C, of the algorithm of enp1s0/tsqr-gpu for the path
 *   mtk::qr::qr<fp32_notc|fp32_tc_cor|fp32_tc_nocor|fp16_notc|fp16_tc_nocor, Reorth>()      (src/blockqr.hpp:142-175)
 * written from the source text of the reference.  Each function cites the
 * reference file:line it follows.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, no known-answer tests
 * and no published numbers, it cannot be compiled in
 * this image (CUDA + four empty submodules), and three pieces of its arithmetic
 * live in un-vendored, un-pinned dependencies (enp1s0/cutf, enp1s0/wmma_extension,
 * enp1s0/gemm_core_cuh; URLs only in .gitmodules) plus closed-source cuBLAS SGEMM
 * and the Tensor-Core accumulator.  Assumptions made for those: sign(0)=+1,
 * k-ascending fp32 FMA inside the 16x16 matmul cores, plain fp32 sums for the
 * Tensor-Core accumulate; fp32_tc_nocor: an fp16 accumulator fragment is rounded to fp16
 * once per mma_sync (K = 16), and cuBLAS SGEMM under CUBLAS_TENSOR_OP_MATH (src/blockqr.cu:64-68)
 * rounds its inputs to fp16 and accumulates in fp32; the half-typed modes (fp16_notc, fp16_tc_nocor: io and working types half,
 * src/tsqr.hpp:27-39): half FMA chains with one rounding per step inside gemm_core's matmul cores (fp16_notc), and cublasHgemm
 * with fp32 accumulation and ONE rounding of the result to half -- the assumption most favourable to the reference (an fp16
 * accumulator over 2^20 rows would be far worse).  What IS pinned, by formula, is checked in
 * tests/test_oracle.py: batch-size rule, workspace sizes, error codes, metric
 * definitions, and agreement with LAPACK (scipy) up to column signs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (tsqr_gpu_amd/) never does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

#define REF_FP16_NOTC   0   /* position in mtk::qr::compute_mode, src/blockqr.hpp:12-23 */
#define REF_FP16_TC_NOCOR 1
#define REF_FP32_NOTC   2
#define REF_FP32_TC_COR 3
#define REF_FP32_TC_NOCOR 4

/* ---- fp16 round trip (cutf::type::cast<half>(float) then back), RNE, subnormals kept ---- */
static inline float h16(float x) {
	uint32_t u; memcpy(&u, &x, 4);
	const uint32_t sign = u & 0x80000000u;
	const uint32_t a = u & 0x7fffffffu;
	if (a >= 0x7f800000u) return x;                 /* inf / nan */
	float ax; memcpy(&ax, &a, 4);
	if (ax >= 65520.0f) {                           /* rounds to inf in fp16 */
		uint32_t r = sign | 0x7f800000u; float f; memcpy(&f, &r, 4); return f;
	}
	float r;
	if (ax < 6.103515625e-05f) {                    /* fp16 subnormal range: quantum 2^-24 */
		const float q = 5.9604644775390625e-08f;    /* 2^-24 */
		r = nearbyintf(ax / q) * q;                 /* default rounding mode = RNE */
	} else {
		/* keep 11 significant bits: add/sub trick on the integer representation */
		uint32_t m = a;
		const uint32_t lsb = (m >> 13) & 1u;
		m += 0x00000fffu + lsb;
		m &= 0xffffe000u;
		memcpy(&r, &m, 4);
	}
	uint32_t ru; memcpy(&ru, &r, 4); ru |= sign; memcpy(&r, &ru, 4);
	return r;
}

/* ---- src/tsqr.cu:39-44 ---- */
size_t ref_get_batch_size_log2(size_t m) {
	unsigned c = (unsigned)ceilf(log2f((float)m));
	if (c < 5u) c = 5u;
	return (size_t)(c - 5u);
}
size_t ref_get_batch_size(size_t m) { return (size_t)1 << ref_get_batch_size_log2(m); }

/* ---- src/tsqr.cu:47-60, src/tsqr.hpp:44-46 (n here is already min(16, n), src/blockqr.cu:34-42) ---- */
size_t ref_tsqr_working_q_size(size_t m, size_t n) {
	const size_t b = ref_get_batch_size(m);
	return n * m + 2 * n * n * (b - 1);
}
size_t ref_tsqr_working_r_size(size_t m, size_t n) {
	const size_t b = ref_get_batch_size(m);
	return n * n * b + n * n * b / 2;
}
size_t ref_working_q_size(size_t m, size_t n) { return ref_tsqr_working_q_size(m, n < 16 ? n : 16); }
size_t ref_working_r_size(size_t m, size_t n) { return ref_tsqr_working_r_size(m, n < 16 ? n : 16); }
size_t ref_working_l_size(size_t m) { return ref_get_batch_size(m) + 1; }

/* ---- src/tcqr32x16.cu:71-95: sum of squares over 32 lanes, xor-butterfly 16,8,4,2,1, fp32 ---- */
static float norm2_32(const float *u, unsigned size) {
	float t[32];
	for (unsigned i = 0; i < 32; i++) t[i] = (i < size) ? u[i] * u[i] : 0.0f;
	for (unsigned mask = 16; mask > 0; mask >>= 1) {
		float s[32];
		for (unsigned i = 0; i < 32; i++) s[i] = t[i] + t[i ^ mask];
		memcpy(t, s, sizeof t);
	}
	return t[0];
}

/* C(32 x nc) = H(32x32) * X(32 x nc), fp32 FMA, k ascending  (src/tcqr32x16.cu:464-496,
 * src/matmul.hpp:21-24 -> gemm_core matmul_core16x16<32>, not in tree: order assumed) */
static void hx_notc(float *X, const float *H, unsigned nc) {
	float out[32 * 32];
	for (unsigned c = 0; c < nc; c++)
		for (unsigned r = 0; r < 32; r++) {
			float acc = 0.0f;
			for (unsigned k = 0; k < 32; k++) acc = fmaf(H[k * 32 + r], X[c * 32 + k], acc);
			out[c * 32 + r] = acc;
		}
	memcpy(X, out, sizeof(float) * 32 * nc);
}

/* fp16_notc: the same product with T = half (src/matmul.hpp:21-24 -> gemm_core matmul_core16x16<32, true>, not in tree): a half
 * FMA chain, k ascending, one rounding to half per step (assumed: __hfma) */
static void hx_h_notc(float *X, const float *H, unsigned nc) {
	float out[32 * 32];
	for (unsigned c = 0; c < nc; c++)
		for (unsigned r = 0; r < 32; r++) {
			float acc = 0.0f;
			for (unsigned k = 0; k < 32; k++) acc = h16(fmaf(H[k * 32 + r], X[c * 32 + k], acc));
			out[c * 32 + r] = acc;
		}
	memcpy(X, out, sizeof(float) * 32 * nc);
}

/* src/tcqr32x16.cu:669-819: per 16-column group, correction terms first (K as 2 x 16),
 * rescale 1.0, then the main term on top; fp32 accumulators. */
static void hx_tc_cor(float *X, const float *H, unsigned nc) {
	float out[32 * 32], Hh[32 * 32], Hl[32 * 32], Xh[32 * 32], Xl[32 * 32];
	for (unsigned i = 0; i < 32 * 32; i++) { Hh[i] = h16(H[i]); Hl[i] = h16(H[i] - Hh[i]); }
	for (unsigned i = 0; i < 32 * nc; i++) { Xh[i] = h16(X[i]); Xl[i] = h16(X[i] - Xh[i]); }
	for (unsigned c = 0; c < nc; c++)
		for (unsigned r = 0; r < 32; r++) {
			float acc = 0.0f;
			for (unsigned kb = 0; kb < 32; kb += 16) {
				float s = 0.0f;   /* one mma_sync: h_diff * x */
				for (unsigned k = kb; k < kb + 16; k++) s += Hl[k * 32 + r] * Xh[c * 32 + k];
				acc += s;
				s = 0.0f;         /* one mma_sync: h * x_diff */
				for (unsigned k = kb; k < kb + 16; k++) s += Hh[k * 32 + r] * Xl[c * 32 + k];
				acc += s;
			}
			/* acc *= 1.0f / correction_rescale (=1) */
			for (unsigned kb = 0; kb < 32; kb += 16) {
				float s = 0.0f;
				for (unsigned k = kb; k < kb + 16; k++) s += Hh[k * 32 + r] * Xh[c * 32 + k];
				acc += s;
			}
			out[c * 32 + r] = acc;
		}
	memcpy(X, out, sizeof(float) * 32 * nc);
}

/* fp32_tc_nocor, src/tcqr32x16.cu:498-614.  H is an fp16 matrix (h_mat_t = half, :39).  X enters the product rounded to fp16
 * (copy_32x16 into the half working memory, or already half when Q_T = half); one mma_sync per K = 16 half.
 * acc_half = 0: fp32 accumulator fragments (R always, :510-511/:574; Q when Q_T = float, :499-560);
 * acc_half = 1: fp16 accumulator fragments (Q when Q_T = half, :575: the tsqr working Q of this mode is half, src/tsqr.hpp:29):
 *               the running value is rounded to fp16 after each mma_sync, and so is the stored result. */
static void hx_tc_nocor(float *X, const float *H, unsigned nc, int acc_half) {
	float out[32 * 32];
	for (unsigned c = 0; c < nc; c++)
		for (unsigned r = 0; r < 32; r++) {
			float acc = 0.0f;
			for (unsigned kb = 0; kb < 32; kb += 16) {
				float s = 0.0f;
				for (unsigned k = kb; k < kb + 16; k++) s += H[k * 32 + r] * h16(X[c * 32 + k]);
				acc = acc_half ? h16(acc + s) : acc + s;
			}
			out[c * 32 + r] = acc;
		}
	memcpy(X, out, sizeof(float) * 32 * nc);
}

/*
 * One <=32 x n tile QR with explicit Householder matrices.
 * src/tcqr32x16.cu:1373-1469 (column loop), :117-137 / :228-274 (make_h),
 * :1471-1532 (tile I/O incl. the transposed Q store, src/matrix_copy.cuh:95-162).
 * a: rows x n at a[ r + c*lda ];  q_out: rows x n (ldq);  r_out: n x n (ldr), stored verbatim.
 */
static void tile_qr(int mode, float *q_out, size_t ldq, float *r_out, size_t ldr,
                    const float *a, size_t lda, unsigned rows, unsigned n, int q_half) {
	float Rt[32 * 16], Qt[32 * 32], H[32 * 32], u[32];
	const int hmode = (mode == REF_FP16_NOTC || mode == REF_FP16_TC_NOCOR);   /* Q_T = R_T = A_T = half (src/tsqr.hpp:27-39), H half (:42-43) */
	for (unsigned c = 0; c < 16; c++)
		for (unsigned r = 0; r < 32; r++)
			Rt[c * 32 + r] = (c < n && r < rows) ? (hmode ? h16(a[r + c * lda]) : a[r + c * lda]) : 0.0f;
	for (unsigned c = 0; c < 32; c++)
		for (unsigned r = 0; r < 32; r++) Qt[c * 32 + r] = (r == c) ? 1.0f : 0.0f;

	for (unsigned k = 0; k < n; k++) {
		for (unsigned i = 0; i < 32; i++) u[i] = (i >= k && i < rows) ? Rt[32 * k + i] : 0.0f;
		const float norm_u_0 = sqrtf(norm2_32(u, rows));
		if (k < 32) {
			const float sgn = (u[k] < 0.0f) ? -1.0f : 1.0f;   /* cutf::math::sign, sign(0)=+1 assumed */
			if (hmode) u[k] = h16(u[k] + h16(sgn * norm_u_0));    /* u_ptr[k] += cast<A_T>(...): a half addition (src/tcqr32x16.cu:1421-1423) */
			else u[k] += sgn * norm_u_0;
		}
		const float norm2_u_1 = norm2_32(u, rows);
		if (mode == REF_FP32_TC_COR) {
			/* src/tcqr32x16.cu:228-274 */
			const float alpha = sqrtf(2.0f / norm2_u_1);
			float uh[32], ul[32];
			for (unsigned i = 0; i < 32; i++) {
				const float uf = u[i] * alpha;
				uh[i] = h16(uf);
				ul[i] = h16(uf - uh[i]);
			}
			for (unsigned x = 0; x < 32; x++)
				for (unsigned y = 0; y < 32; y++) {
					const float acc = uh[y] * uh[x] + ul[y] * uh[x] + uh[y] * ul[x];
					H[x * 32 + y] = -acc + ((x == y) ? 1.0f : 0.0f);
				}
		} else if (mode == REF_FP16_TC_NOCOR) {
			/* src/tcqr32x16.cu:140-184: u *= sqrt(2 / |u|^2) in half, one outer product per 16 x 16 fragment with an fp16 accumulator
			 * (load_vector fragments: one product per entry), then H = I - that in half arithmetic */
			const float alpha = sqrtf(2.0f / norm2_u_1);
			float us[32];
			for (unsigned i = 0; i < 32; i++) us[i] = h16(u[i] * alpha);
			for (unsigned x = 0; x < 32; x++)
				for (unsigned y = 0; y < 32; y++)
					H[x * 32 + y] = h16(((x == y) ? 1.0f : 0.0f) - h16(us[y] * us[x]));
		} else if (mode == REF_FP32_TC_NOCOR) {
			/* src/tcqr32x16.cu:186-226: H in fp16; one outer product per 16 x 16 fragment with an fp16 accumulator:
			 * H(y, x) = half(delta - half(half(u_y * alpha) * half(u_x))), alpha = 2 / |u|^2 */
			const float alpha = 2.0f / norm2_u_1;
			for (unsigned x = 0; x < 32; x++)
				for (unsigned y = 0; y < 32; y++) {
					const float p = h16(h16(u[y] * alpha) * h16(u[x]));
					H[x * 32 + y] = h16(((x == y) ? 1.0f : 0.0f) - p);
				}
		} else {
			/* src/tcqr32x16.cu:117-137 */
			for (unsigned y = 0; y < 32; y++) {
				const float uy = 2.0f * u[y] / norm2_u_1;
				for (unsigned x = 0; x < 32; x++) {
					float tmp = (x == y) ? 1.0f : 0.0f;
					if (x < rows && y < rows) tmp -= uy * u[x];
					H[x * 32 + y] = hmode ? h16(tmp) : tmp;              /* cast<T>(tmp): T = half for fp16_notc (:42) */
				}
			}
		}
		if (mode == REF_FP16_NOTC)          { hx_h_notc(Qt, H, 32); hx_h_notc(Rt, H, 16); }                  /* :464-496 with T = half */
		else if (mode == REF_FP16_TC_NOCOR) { hx_tc_nocor(Qt, H, 32, 1); hx_tc_nocor(Rt, H, 16, 1); }        /* :617-667: half accumulator fragments for Q and R */
		else if (mode == REF_FP32_TC_COR)   { hx_tc_cor(Qt, H, 32); hx_tc_cor(Rt, H, 16); }
		else if (mode == REF_FP32_TC_NOCOR) { hx_tc_nocor(Qt, H, 32, q_half); hx_tc_nocor(Rt, H, 16, 0); }
		else                                { hx_notc(Qt, H, 32);   hx_notc(Rt, H, 16); }
	}
	for (unsigned y = 0; y < n; y++)            /* s2g32x32_16x32_t_2w: Q_out(x, y) = Qt(y, x) */
		for (unsigned x = 0; x < rows; x++) q_out[ldq * y + x] = Qt[32 * x + y];
	for (unsigned c = 0; c < n; c++)            /* s2g32x16_2w: top n x n of Rt, verbatim */
		for (unsigned r = 0; r < n; r++) r_out[ldr * c + r] = Rt[32 * c + r];
}

/* AC(2n x n, ld) <- AC * B(n x n, ldb); both 16-row halves use the same B.
 * notc: src/tsqr.cu:143-204 / :591-656.  tc_cor: :330-412 / :790-876 (rescale 1024).
 * tc_nocor: operands are the fp16 working Q (already half values); tree levels keep an fp16 accumulator and store half
 * (:206-266, out_half = 1), layer 0 accumulates in fp32 and writes the user's float Q (:724-788, out_half = 0). */
static void back_mul(int mode, float *out, size_t ldo, const float *ac, size_t ldac, unsigned rows,
                     const float *b, size_t ldb, unsigned n, int out_half) {
	float tmp[32 * 16];
	for (unsigned c = 0; c < n; c++)
		for (unsigned r = 0; r < rows; r++) {
			float acc = 0.0f;
			if (mode == REF_FP32_TC_COR) {
				const float s = 1024.0f;
				float c1 = 0.0f, c2 = 0.0f, mn = 0.0f;
				for (unsigned k = 0; k < n; k++) {
					const float av = ac[r + k * ldac], ah = h16(av), al = h16((av - ah) * s);
					const float bv = b[k + c * ldb],  bh = h16(bv), bl = h16((bv - bh) * s);
					c1 += al * bh; c2 += ah * bl; mn += ah * bh;
				}
				acc = (c1 + c2) * (1.0f / s) + mn;
			} else if (mode == REF_FP32_TC_NOCOR) {
				for (unsigned k = 0; k < n; k++) acc += h16(ac[r + k * ldac]) * h16(b[k + c * ldb]);
				if (out_half) acc = h16(acc);
			} else if (mode == REF_FP16_TC_NOCOR) {           /* src/tsqr.cu:268-328, :658-722: half operands, half accumulator, half result */
				for (unsigned k = 0; k < n; k++) acc += h16(ac[r + k * ldac]) * h16(b[k + c * ldb]);
				acc = h16(acc);
			} else if (mode == REF_FP16_NOTC) {               /* src/tsqr.cu:143-204, :591-656 with T = half: half FMA chain (assumed) */
				for (unsigned k = 0; k < n; k++) acc = h16(fmaf(ac[r + k * ldac], b[k + c * ldb], acc));
			} else {
				for (unsigned k = 0; k < n; k++) acc = fmaf(ac[r + k * ldac], b[k + c * ldb], acc);
			}
			tmp[c * 32 + r] = acc;
		}
	for (unsigned c = 0; c < n; c++)
		for (unsigned r = 0; r < rows; r++) out[r + c * ldo] = tmp[c * 32 + r];
}

/*
 * src/tsqr.cu:1064-1310  tsqr16 / tsqr16_geq32 (n <= 16).  wq/wr sized by ref_tsqr_working_*_size;
 * hl holds batch+1 row offsets (the d_wl/h_wl pair of the reference collapses to one host array).
 */
static void tsqr16(int mode, float *q, size_t ldq, float *r, size_t ldr, const float *a, size_t lda,
                   size_t m, unsigned n, float *wq, float *wr, unsigned *hl) {
	if (m <= 32) { tile_qr(mode, q, ldq, r, ldr, a, lda, (unsigned)m, n, 0); return; }   /* :1301-1309: Q_T = the user's float */
	const int qh = (mode == REF_FP32_TC_NOCOR);      /* the working Q of this mode is half: src/tsqr.hpp:29 */
	const size_t L = ref_get_batch_size_log2(m), B = (size_t)1 << L;
	float *wrs[2] = { wr, wr + (size_t)n * n * B };
	const size_t ldrs[2] = { n * B, n * B / 2 };
	hl[0] = 0;
	for (size_t i = 1; i < B; i++) hl[i] = (unsigned)(m * i / B);                    /* :1088-1092 */
	hl[B] = (unsigned)m;
	#pragma omp parallel for schedule(static)
	for (long i = 0; i < (long)B; i++)                                                /* :1102-1108 */
		tile_qr(mode, wq + hl[i], m, wrs[0] + (size_t)n * i, ldrs[0], a + hl[i], lda, hl[i + 1] - hl[i], n, qh);
	for (size_t k = L - 1; k > 0 && L >= 1; k--) {                                    /* :1121-1159 */
		const size_t lb = (size_t)1 << k;
		const size_t off = 2 * (size_t)n * n * (B - ((size_t)1 << (k + 1))) + m * n;
		const size_t idx = 1 - (L - k) % 2;
		#pragma omp parallel for schedule(static)
		for (long j = 0; j < (long)lb; j++)
			tile_qr(mode, wq + off + 2 * (size_t)n * j, 2 * n * lb,
			        wrs[1 - idx] + (size_t)n * j, ldrs[1 - idx],
			        wrs[idx] + 2 * (size_t)n * j, ldrs[idx], 2 * n, n, qh);
	}
	{                                                                                 /* root :1164-1172 */
		const size_t off = 2 * (size_t)n * n * (B - 2) + m * n;
		tile_qr(mode, wq + off, 2 * n, r, ldr, wrs[1 - (L % 2)], ldrs[1 - (L % 2)], 2 * n, n, qh);
	}
	for (size_t k = 1; k < L; k++) {                                                  /* backward :1205-1230 */
		const size_t lb = (size_t)1 << k;
		const size_t off = 2 * (size_t)n * n * (B - ((size_t)1 << (k + 1))) + m * n;
		float *ac = wq + off; const float *bb = wq + off + lb * 2 * n * n;
		const size_t ac_m = lb * 2 * n;
		#pragma omp parallel for schedule(static)
		for (long j = 0; j < (long)lb; j++)
			back_mul(mode, ac + 2 * (size_t)n * j, ac_m, ac + 2 * (size_t)n * j, ac_m, 2 * n,
			         bb + (size_t)n * j, ac_m / 2, n, 1);
	}
	#pragma omp parallel for schedule(static)
	for (long i = 0; i < (long)B; i++)                                                /* layer 0 :1232-1260 */
		back_mul(mode, q + hl[i], ldq, wq + hl[i], m, hl[i + 1] - hl[i], wq + m * n + (size_t)n * i, n * B, n, 0);
}

/* plain fp32 GEMMs standing in for cuBLAS default-math SGEMM (src/blockqr.cu:92-116, 230-332).  top = 1: the handle is in
 * CUBLAS_TENSOR_OP_MATH (fp32_tc_nocor, src/blockqr.cu:64-68, 209-213): inputs rounded to fp16, fp32 accumulation (assumed). */
static int g_top = 0, g_half = 0;                    /* g_half: half-typed modes -- cublasHgemm: half in, half out; fp32 accumulation and one
                                                         * rounding of the result assumed (see the header comment) */
static inline float gin(float x) { return (g_top || g_half) ? h16(x) : x; }
static inline float gout(float x) { return g_half ? h16(x) : x; }
static void gemm_tn(float *c, size_t ldc, const float *a, size_t lda, const float *b, size_t ldb,
                    size_t mm, size_t nn, size_t kk) {          /* C(mm x nn) = A^T(mm x kk) B(kk x nn) */
	#pragma omp parallel for collapse(2) schedule(static)
	for (long j = 0; j < (long)nn; j++)
		for (long i = 0; i < (long)mm; i++) {
			float acc = 0.0f;
			for (size_t k = 0; k < kk; k++) acc = fmaf(gin(a[k + i * lda]), gin(b[k + j * ldb]), acc);
			c[i + j * ldc] = gout(acc);
		}
}
static void gemm_nn(float *c, size_t ldc, float alpha, const float *a, size_t lda, const float *b, size_t ldb,
                    float beta, size_t mm, size_t nn, size_t kk) { /* C = alpha A B + beta C */
	#pragma omp parallel for schedule(static)
	for (long i = 0; i < (long)mm; i++)
		for (size_t j = 0; j < nn; j++) {
			float acc = 0.0f;
			for (size_t k = 0; k < kk; k++) acc = fmaf(gin(a[i + k * lda]), gin(b[k + j * ldb]), acc);
			c[i + j * ldc] = gout(alpha * acc + (beta == 0.0f ? 0.0f : beta * c[i + j * ldc]));
		}
}

/*
 * mtk::qr::qr<mode, Reorth>: src/blockqr.cu:394-433 (argument check), :45-178 (BCGS),
 * :180-390 (BCGS2).  Returns 0 / 1 like state_t; -1 for a mode this oracle does not restate.
 * a is overwritten for n > 16, r must be pre-zeroed by the caller (src/test.cu:129).
 */
int ref_qr_f32(int mode, int reorth, float *q, size_t ldq, float *r, size_t ldr, float *a, size_t lda,
               size_t m, size_t n) {
	if (n > m || m == 0 || n == 0) return 1;
	if (mode != REF_FP32_NOTC && mode != REF_FP32_TC_COR && mode != REF_FP32_TC_NOCOR && mode != REF_FP16_NOTC && mode != REF_FP16_TC_NOCOR) return -1;
	g_top = (mode == REF_FP32_TC_NOCOR);             /* (one factorisation at a time: the oracle is not re-entrant) */
	g_half = (mode == REF_FP16_NOTC || mode == REF_FP16_TC_NOCOR);
	/* half-typed modes: a, q, r are float arrays HOLDING half values (the caller rounds a; everything written is rounded here) */
	const size_t nb = 16;
	float *wq = (float *)malloc(sizeof(float) * ref_working_q_size(m, n));
	float *wr = (float *)malloc(sizeof(float) * ref_working_r_size(m, n));
	unsigned *hl = (unsigned *)malloc(sizeof(unsigned) * ref_working_l_size(m));
	float *w_reorth = reorth ? (float *)calloc(nb * nb * 2 + m * nb, sizeof(float)) : NULL;
	float *r2 = w_reorth, *s2 = w_reorth ? r2 + nb * nb : NULL, *w = w_reorth ? s2 + m * nb : NULL;
	const size_t nblk = (n + nb - 1) / nb;
	for (size_t b = 0; b < nblk; b++) {
		const size_t c = (n - b * nb < nb) ? n - b * nb : nb, P = b * nb;
		if (b != 0) {
			gemm_tn(r + ldr * P, ldr, q, ldq, a + lda * P, lda, P, c, m);
			gemm_nn(a + lda * P, lda, -1.0f, q, ldq, r + ldr * P, ldr, 1.0f, m, c, P);
		}
		if (!reorth || b == 0) {
			tsqr16(mode, q + P * ldq, ldq, r + P * ldr + P, ldr, a + P * lda, lda, m, (unsigned)c, wq, wr, hl);
		} else {
			tsqr16(mode, q + P * ldq, ldq, r2, nb, a + P * lda, lda, m, (unsigned)c, wq, wr, hl);
			gemm_tn(s2, m, q, ldq, q + P * ldq, ldq, P, c, m);
			gemm_nn(q + P * ldq, ldq, -1.0f, q, ldq, s2, m, 1.0f, m, c, P);
			tsqr16(mode, q + P * ldq, ldq, w, nb, q + P * ldq, ldq, m, (unsigned)c, wq, wr, hl);
			gemm_nn(r + ldr * P, ldr, 1.0f, s2, m, r2, nb, 1.0f, P, c, c);
			gemm_nn(r + ldr * P + P, ldr, 1.0f, w, nb, r2, nb, 0.0f, c, c, c);
		}
	}
	free(wq); free(wr); free(hl); free(w_reorth);
	return 0;
}

/* exported for unit tests of the fp16 model */
float ref_h16(float x) { return h16(x); }
