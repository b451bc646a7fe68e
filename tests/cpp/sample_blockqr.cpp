// The reference README's sample (README.md:52-87) written against include/tsqr/blockqr.hpp: shows that a caller
// of mtk::qr::qr<mode, Reorth>() / mtk::qr::buffer switches by changing the handle argument only.
// Prints residual and orthogonality; exit code 0 when both are within tolerance.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <tsqr/blockqr.hpp>

template <mtk::qr::compute_mode compute_mode, bool reorthogonalize>
int run(const std::size_t M, const std::size_t N) {
	using compute_t = typename mtk::qr::get_io_type<compute_mode>::type;      // float, or IEEE binary16 for the fp16 I/O modes
	constexpr bool f16 = sizeof(compute_t) == 2;
	std::mt19937 mt(0);
	std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
	std::vector<compute_t> h_a(M * N), h_q(M * N), h_r(N * N, (compute_t)0.0f);
	for (auto& v : h_a) v = (compute_t)dist(mt);

	compute_t *d_a, *d_r, *d_q;
	hipMalloc((void**)&d_a, sizeof(compute_t) * M * N);
	hipMalloc((void**)&d_r, sizeof(compute_t) * N * N);
	hipMalloc((void**)&d_q, sizeof(compute_t) * M * N);
	hipMemcpy(d_a, h_a.data(), sizeof(compute_t) * M * N, hipMemcpyHostToDevice);
	hipMemset(d_r, 0, sizeof(compute_t) * N * N);

	mtk::qr::buffer<compute_mode, reorthogonalize> buffer;
	buffer.allocate(M, N);
	bool threw = false;
	try { buffer.allocate(M, N); } catch (const std::runtime_error&) { threw = true; }   // reference blockqr.hpp:77-79

	hipStream_t stream;
	hipStreamCreate(&stream);
	const auto st = mtk::qr::qr<compute_mode, reorthogonalize>(d_q, M, d_r, N, d_a, M, M, N, buffer, stream);
	hipMemcpy(h_q.data(), d_q, sizeof(compute_t) * M * N, hipMemcpyDeviceToHost);
	hipMemcpy(h_r.data(), d_r, sizeof(compute_t) * N * N, hipMemcpyDeviceToHost);

	double num = 0, den = 0, orth = 0;
	for (std::size_t j = 0; j < N; j++)
		for (std::size_t i = 0; i < M; i++) {
			double s = 0;
			for (std::size_t k = 0; k <= j; k++) s += (double)h_q[i + k * M] * (double)h_r[k + j * N];
			const double d = s - (double)h_a[i + j * M];
			num += d * d; den += (double)h_a[i + j * M] * (double)h_a[i + j * M];
		}
	for (std::size_t a = 0; a < N; a++)
		for (std::size_t b = 0; b < N; b++) {
			double s = 0;
			for (std::size_t i = 0; i < M; i++) s += (double)h_q[i + a * M] * (double)h_q[i + b * M];
			s -= (a == b);
			orth += s * s;
		}
	const double residual = std::sqrt(num / den), orthogonality = std::sqrt(orth);
	std::printf("mode=%d reorth=%d state=%d residual=%e orthogonality_F=%e double_allocate_threw=%d bytes=%zu\n",
	            (int)compute_mode, (int)reorthogonalize, st, residual, orthogonality, (int)threw, buffer.get_device_memory_size());
	const auto bad = mtk::qr::qr<compute_mode, reorthogonalize>(d_q, N, d_r, M, d_a, N, N, M, buffer, stream);   // n > m
	hipFree(d_a); hipFree(d_r); hipFree(d_q); hipStreamDestroy(stream);
	return (st == mtk::qr::success_factorization && bad == mtk::qr::error_invalid_matrix_size && threw &&
	        residual < (f16 ? 1e-3 : 5e-7) && orthogonality < (f16 ? 5e-3 : 5e-6)) ? 0 : 1;      // (fp16: the rounding of Q and R to half)
}

// mtk::qr::qr_submit / qr_finish (not in the reference: include/tsqr/blockqr.hpp): two different matrices in flight through ONE buffer,
// their factors compared bit for bit with those of the blocking call
template <mtk::qr::compute_mode compute_mode>
int run_two_in_flight(const std::size_t M, const std::size_t N) {
	std::mt19937 mt(1);
	std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
	std::vector<float> h_a[2], h_q[2], h_r[2], h_qb(M * N), h_rb(N * N);
	float *d_a[2], *d_q[2], *d_r[2], *d_qb, *d_rb;
	for (int k = 0; k < 2; k++) {
		h_a[k].resize(M * N); h_q[k].resize(M * N); h_r[k].resize(N * N);
		for (auto& v : h_a[k]) v = dist(mt);
		hipMalloc((void**)&d_a[k], sizeof(float) * M * N); hipMalloc((void**)&d_q[k], sizeof(float) * M * N); hipMalloc((void**)&d_r[k], sizeof(float) * N * N);
		hipMemcpy(d_a[k], h_a[k].data(), sizeof(float) * M * N, hipMemcpyHostToDevice);
		hipMemset(d_r[k], 0, sizeof(float) * N * N);
	}
	hipMalloc((void**)&d_qb, sizeof(float) * M * N); hipMalloc((void**)&d_rb, sizeof(float) * N * N);
	mtk::qr::buffer<compute_mode, false> buffer;
	buffer.allocate(M, N);
	hipStream_t stream;
	hipStreamCreate(&stream);
	mtk::qr::ticket t[2];
	mtk::qr::qr_submit<compute_mode, false>(t[0], d_q[0], M, d_r[0], N, d_a[0], M, M, N, buffer, stream);
	mtk::qr::qr_submit<compute_mode, false>(t[1], d_q[1], M, d_r[1], N, d_a[1], M, M, N, buffer, stream);
	int bad = 0;
	for (int k = 0; k < 2; k++) bad |= (mtk::qr::qr_finish(t[k]) != mtk::qr::success_factorization);
	for (int k = 0; k < 2; k++) {
		hipMemcpy(h_q[k].data(), d_q[k], sizeof(float) * M * N, hipMemcpyDeviceToHost);
		hipMemcpy(h_r[k].data(), d_r[k], sizeof(float) * N * N, hipMemcpyDeviceToHost);
		hipMemset(d_rb, 0, sizeof(float) * N * N);
		bad |= (mtk::qr::qr<compute_mode, false>(d_qb, M, d_rb, N, d_a[k], M, M, N, buffer, stream) != mtk::qr::success_factorization);
		hipMemcpy(h_qb.data(), d_qb, sizeof(float) * M * N, hipMemcpyDeviceToHost);
		hipMemcpy(h_rb.data(), d_rb, sizeof(float) * N * N, hipMemcpyDeviceToHost);
		bad |= (std::memcmp(h_qb.data(), h_q[k].data(), sizeof(float) * M * N) != 0) | (std::memcmp(h_rb.data(), h_r[k].data(), sizeof(float) * N * N) != 0);
	}
	std::printf("mode=%d two calls in flight (qr_submit / qr_finish) %zu x %zu: %s\n", (int)compute_mode, M, N, bad ? "MISMATCH" : "the blocking call's bits");
	for (int k = 0; k < 2; k++) { hipFree(d_a[k]); hipFree(d_q[k]); hipFree(d_r[k]); }
	hipFree(d_qb); hipFree(d_rb); hipStreamDestroy(stream);
	return bad;
}

int main() {
	int rc = 0;
	rc |= run_two_in_flight<mtk::qr::compute_mode::fp32_tc_cor>(20000, 64);
	rc |= run_two_in_flight<mtk::qr::compute_mode::fp32_notc>(30000, 128);
	rc |= run<mtk::qr::compute_mode::fp32_tc_cor, false>(9211, 51);
	rc |= run<mtk::qr::compute_mode::fp32_notc, false>(9211, 51);
	rc |= run<mtk::qr::compute_mode::fp32_tc_cor, true>(2000, 100);
	rc |= run<mtk::qr::compute_mode::fp16_tc_nocor, false>(9211, 51);          // io type half (reference src/tsqr.hpp:38-39)
	rc |= run<mtk::qr::compute_mode::fp16_notc, true>(2000, 100);
	std::printf(rc == 0 ? "SAMPLE OK\n" : "SAMPLE FAILED\n");
	return rc;
}
