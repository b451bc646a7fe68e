// The reference's speed protocol (src/test.cu:257-343: one warm-up call, then wall clock over C = 16 blocking calls) from a C++
// caller of include/tsqr/blockqr.hpp -- no Python, no torch: what a user of the reference sees after switching.  The first mode is
// also timed as a stream of calls, two in flight (qr_submit / qr_finish).
// also timed as a stream of calls, two in flight (qr_submit / qr_finish), and as ONE mtk::qr::qr_batch call over four different
// matrices in rotation (the caller with many matrices; no call finds its A in the Infinity Cache from an earlier call).
// Prints the reference's speed CSV line plus the algorithmic TFLOP/s (F_QR = 4MN^2 - 4/3 N^3).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <tsqr/blockqr.hpp>

template <mtk::qr::compute_mode mode, bool reorth>
int speed(const std::size_t M, const std::size_t N, const unsigned C, const char* mode_name, const bool stream_of_calls = false) {
	std::vector<float> h_a(M * N);
	unsigned long long s = 88172645463325252ull;                     // xorshift64: U(-1,1)
	for (auto& v : h_a) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
	float *d_a, *d_q, *d_r;
	if (hipMalloc((void**)&d_a, sizeof(float) * M * N) != hipSuccess || hipMalloc((void**)&d_q, sizeof(float) * M * N) != hipSuccess ||
	    hipMalloc((void**)&d_r, sizeof(float) * N * N) != hipSuccess) return 1;
	(void)hipMemcpy(d_a, h_a.data(), sizeof(float) * M * N, hipMemcpyHostToDevice);
	(void)hipMemset(d_r, 0, sizeof(float) * N * N);
	mtk::qr::buffer<mode, reorth> buffer;
	buffer.allocate(M, N);
	hipStream_t stream;
	(void)hipStreamCreate(&stream);
	if (mtk::qr::qr<mode, reorth>(d_q, M, d_r, N, d_a, M, M, N, buffer, stream) != mtk::qr::success_factorization) return 1;
	const auto t0 = std::chrono::system_clock::now();
	for (unsigned c = 0; c < C; c++) mtk::qr::qr<mode, reorth>(d_q, M, d_r, N, d_a, M, M, N, buffer, stream);
	const auto t1 = std::chrono::system_clock::now();
	const double el = std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() * 1e-9 / C;
	const double fqr = 4.0 * M * N * N - 4.0 / 3.0 * N * N * N;
	std::printf("%zu,%zu,1,float,%s,%d,%e,%e,%zu\n", M, N, mode_name, (int)reorth, el, fqr / el / 1e12, buffer.get_device_memory_size());
	if (stream_of_calls) {
		// the same C calls as a stream: call c + 1 is submitted before call c is finished (mtk::qr::qr_submit / qr_finish)
		mtk::qr::ticket tk[2];
		const auto t2 = std::chrono::system_clock::now();
		mtk::qr::qr_submit<mode, reorth>(tk[0], d_q, M, d_r, N, d_a, M, M, N, buffer, stream);
		for (unsigned c = 0; c < C; c++) {
			if (c + 1 < C) mtk::qr::qr_submit<mode, reorth>(tk[(c + 1) & 1], d_q, M, d_r, N, d_a, M, M, N, buffer, stream);
			if (mtk::qr::qr_finish(tk[c & 1]) != mtk::qr::success_factorization) return 1;
		}
		const auto t3 = std::chrono::system_clock::now();
		const double el2 = std::chrono::duration_cast<std::chrono::nanoseconds>(t3 - t2).count() * 1e-9 / C;
		std::printf("%zu,%zu,1,float,%s/two_in_flight,%d,%e,%e,%zu\n", M, N, mode_name, (int)reorth, el2, fqr / el2 / 1e12, buffer.get_device_memory_size());
	}
	if (stream_of_calls) {
		// C different-matrix calls through one mtk::qr::qr_batch call: four (A, Q, R) triples in rotation
		constexpr int R = 4;
		float *ra[R], *rq[R], *rr[R];
		ra[0] = d_a; rq[0] = d_q; rr[0] = d_r;
		for (int k = 1; k < R; k++) {
			for (auto& v : h_a) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
			if (hipMalloc((void**)&ra[k], sizeof(float) * M * N) != hipSuccess || hipMalloc((void**)&rq[k], sizeof(float) * M * N) != hipSuccess ||
			    hipMalloc((void**)&rr[k], sizeof(float) * N * N) != hipSuccess) return 1;
			(void)hipMemcpy(ra[k], h_a.data(), sizeof(float) * M * N, hipMemcpyHostToDevice);
		}
		std::vector<float*> pa(C), pq(C), pr(C);
		for (unsigned c = 0; c < C; c++) { pa[c] = ra[c % R]; pq[c] = rq[c % R]; pr[c] = rr[c % R]; }
		if (mtk::qr::qr_batch<mode, reorth>(C, pq.data(), M, pr.data(), N, pa.data(), M, M, N, buffer, stream) != mtk::qr::success_factorization) return 1;   // warm-up
		const auto t4 = std::chrono::system_clock::now();
		if (mtk::qr::qr_batch<mode, reorth>(C, pq.data(), M, pr.data(), N, pa.data(), M, M, N, buffer, stream) != mtk::qr::success_factorization) return 1;
		const auto t5 = std::chrono::system_clock::now();
		const double el3 = std::chrono::duration_cast<std::chrono::nanoseconds>(t5 - t4).count() * 1e-9 / C;
		std::printf("%zu,%zu,1,float,%s/qr_batch_4_rotating_matrices,%d,%e,%e,%zu\n", M, N, mode_name, (int)reorth, el3, fqr / el3 / 1e12, buffer.get_device_memory_size());
		for (int k = 1; k < R; k++) { (void)hipFree(ra[k]); (void)hipFree(rq[k]); (void)hipFree(rr[k]); }
	}
	(void)hipFree(d_a); (void)hipFree(d_q); (void)hipFree(d_r); (void)hipStreamDestroy(stream);
	return 0;
}

int main(int argc, char** argv) {
	const std::size_t M = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : (1u << 20);
	const std::size_t N = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 64;
	const unsigned C = argc > 3 ? (unsigned)std::atoi(argv[3]) : 16;
	std::printf("m,n,rand_range,type,compute_mode,reorthogonalization,elapsed_time,tflops_fqr,working_memory_size\n");
	int rc = 0;
	rc |= speed<mtk::qr::fp32_tc_cor, false>(M, N, C, "fp32_tc_cor", /*stream_of_calls=*/true);
	rc |= speed<mtk::qr::fp32_notc, false>(M, N, C, "fp32_notc");
	rc |= speed<mtk::qr::fp32_tc_cor, true>(M, N, C, "fp32_tc_cor");
	return rc;
}
