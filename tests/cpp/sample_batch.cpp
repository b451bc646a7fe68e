// A caller of mtk::qr::qr with MANY matrices (the reference's README.md:52-87 call inside the caller's loop), switched to
// mtk::qr::qr_batch: K different 2^k x 64 matrices, first factored one blocking call after the other, then through one qr_batch call --
// the factors must agree bit for bit.  Also names mtk::qr::get_tsqr_compute_mode<> (reference src/blockqr.hpp:31-43).
// usage: sample_batch [M [N [K]]]   exit code 0 when every byte agrees
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <tsqr/blockqr.hpp>

static_assert(mtk::qr::get_tsqr_compute_mode<mtk::qr::fp32_tc_cor>() == mtk::tsqr::fp32_tc_cor, "mode mapping of reference src/blockqr.hpp:31-43");
static_assert(mtk::qr::get_tsqr_compute_mode<mtk::qr::fp16_notc>() == mtk::tsqr::fp16_notc, "mode mapping of reference src/blockqr.hpp:31-43");

// (the half-typed instantiation: mtk::qr::qr_batch<fp16_tc_nocor, false> takes arrays of half pointers and forwards to tsqr_mi_qr_f16_batch;
// compiled here, run by tests/test_gpu_f16.py through the Python mirror)
[[maybe_unused]] static mtk::qr::state_t half_batch(std::size_t K, mtk::qr::half_t* const* q, mtk::qr::half_t* const* r, mtk::qr::half_t* const* a,
                                                    std::size_t M, std::size_t N, mtk::qr::buffer<mtk::qr::fp16_tc_nocor, false>& b) {
	return mtk::qr::qr_batch<mtk::qr::fp16_tc_nocor, false>(K, q, M, r, N, a, M, M, N, b);
}

int main(int argc, char** argv) {
	constexpr auto mode = mtk::qr::fp32_tc_cor;
	const std::size_t M = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : (1u << 17);
	const std::size_t N = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 64;
	const std::size_t K = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 6;
	std::vector<float*> d_a(K), d_q(K), d_r(K), d_q1(K), d_r1(K);
	std::vector<float> h(M * N);
	unsigned long long s = 88172645463325252ull;                     // xorshift64: U(-1,1), a different matrix for every k
	for (std::size_t k = 0; k < K; k++) {
		for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
		for (float** p : {&d_a[k], &d_q[k], &d_q1[k]}) if (hipMalloc((void**)p, sizeof(float) * M * N) != hipSuccess) return 2;
		for (float** p : {&d_r[k], &d_r1[k]}) if (hipMalloc((void**)p, sizeof(float) * N * N) != hipSuccess) return 2;
		(void)hipMemcpy(d_a[k], h.data(), sizeof(float) * M * N, hipMemcpyHostToDevice);
	}
	mtk::qr::buffer<mode, false> buffer;
	buffer.allocate(M, N);
	hipStream_t stream;
	(void)hipStreamCreate(&stream);
	// the caller's loop of blocking calls ...
	const auto t0 = std::chrono::system_clock::now();
	for (std::size_t k = 0; k < K; k++)
		if (mtk::qr::qr<mode, false>(d_q1[k], M, d_r1[k], N, d_a[k], M, M, N, buffer, stream) != mtk::qr::success_factorization) return 3;
	const auto t1 = std::chrono::system_clock::now();
	// ... and the same matrices through one call
	std::vector<mtk::qr::state_t> states(K, -1);
	const auto st = mtk::qr::qr_batch<mode, false>(K, d_q.data(), M, d_r.data(), N, d_a.data(), M, M, N, buffer, stream, states.data());
	const auto t2 = std::chrono::system_clock::now();
	if (st != mtk::qr::success_factorization) return 4;
	int bad = 0;
	std::vector<float> hq(M * N), hq1(M * N), hr(N * N), hr1(N * N);
	for (std::size_t k = 0; k < K; k++) {
		(void)hipMemcpy(hq.data(), d_q[k], sizeof(float) * M * N, hipMemcpyDeviceToHost);
		(void)hipMemcpy(hq1.data(), d_q1[k], sizeof(float) * M * N, hipMemcpyDeviceToHost);
		(void)hipMemcpy(hr.data(), d_r[k], sizeof(float) * N * N, hipMemcpyDeviceToHost);
		(void)hipMemcpy(hr1.data(), d_r1[k], sizeof(float) * N * N, hipMemcpyDeviceToHost);
		if (states[k] != 0 || std::memcmp(hq.data(), hq1.data(), sizeof(float) * M * N) != 0 || std::memcmp(hr.data(), hr1.data(), sizeof(float) * N * N) != 0) bad++;
	}
	const double us_loop = std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() * 1e-3 / K;
	const double us_batch = std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count() * 1e-3 / K;
	std::printf("m=%zu n=%zu matrices=%zu  blocking loop %.1f us/matrix  qr_batch %.1f us/matrix  mismatching matrices: %d\n", M, N, K, us_loop, us_batch, bad);
	return bad ? 1 : 0;
}
