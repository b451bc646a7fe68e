// A caller of the reference's lower-level entry, mtk::tsqr::tsqr16<mode>() with mtk::tsqr::buffer<mode> (src/tsqr.hpp:49-140),
// written against include/tsqr/tsqr.hpp: one panel of 16 columns (the reference's limit) and one of 64 (this engine's).
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include <tsqr/blockqr.hpp>                      // pulls in tsqr.hpp, as the reference's header does

template <mtk::tsqr::compute_mode mode>
int run(const std::size_t M, const std::size_t N) {
	std::mt19937 mt(1);
	std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
	using io_t = typename mtk::tsqr::get_io_type<mode>::type;                 // float, or IEEE binary16 for the fp16 I/O modes
	constexpr bool f16 = sizeof(io_t) == 2;
	std::vector<io_t> h_a(M * N), h_q(M * N), h_r(N * N, (io_t)0.0f), h_a_after(M * N);
	for (auto& v : h_a) v = (io_t)dist(mt);
	io_t *d_a, *d_r, *d_q;
	hipMalloc((void**)&d_a, sizeof(io_t) * M * N);
	hipMalloc((void**)&d_r, sizeof(io_t) * N * N);
	hipMalloc((void**)&d_q, sizeof(io_t) * M * N);
	hipMemcpy(d_a, h_a.data(), sizeof(io_t) * M * N, hipMemcpyHostToDevice);
	hipMemset(d_r, 0, sizeof(io_t) * N * N);
	mtk::tsqr::buffer<mode> buffer;
	buffer.allocate(M, N);
	bool threw = false;
	try { buffer.allocate(M, N); } catch (const std::runtime_error&) { threw = true; }   // reference tsqr.hpp:66-68
	hipStream_t stream;
	hipStreamCreate(&stream);
	mtk::tsqr::tsqr16<mode>(d_q, M, d_r, N, d_a, M, M, N, buffer, stream);
	hipMemcpy(h_q.data(), d_q, sizeof(io_t) * M * N, hipMemcpyDeviceToHost);
	hipMemcpy(h_r.data(), d_r, sizeof(io_t) * N * N, hipMemcpyDeviceToHost);
	hipMemcpy(h_a_after.data(), d_a, sizeof(io_t) * M * N, hipMemcpyDeviceToHost);
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
	bool a_intact = true;
	for (std::size_t i = 0; i < M * N; i++) a_intact = a_intact && ((double)h_a_after[i] == (double)h_a[i]);
	const double residual = std::sqrt(num / den), orthogonality = std::sqrt(orth);
	std::printf("tsqr16 mode=%d %zux%zu residual=%e orthogonality_F=%e a_intact=%d double_allocate_threw=%d batch=%zu bytes=%zu\n",
	            (int)mode, M, N, residual, orthogonality, (int)a_intact, (int)threw, mtk::tsqr::get_batch_size(M), buffer.get_device_memory_size());
	hipFree(d_a); hipFree(d_r); hipFree(d_q); hipStreamDestroy(stream);
	return (threw && a_intact && residual < (f16 ? 1e-3 : 5e-7) && orthogonality < (f16 ? 5e-3 : 5e-6)) ? 0 : 1;   // (fp16: the rounding of Q and R to half)
}

int main() {
	int rc = 0;
	rc |= run<mtk::tsqr::compute_mode::fp32_tc_cor>(9211, 16);
	rc |= run<mtk::tsqr::compute_mode::fp32_notc>(4096, 16);
	rc |= run<mtk::tsqr::compute_mode::fp32_tc_cor>(20000, 64);
	rc |= run<mtk::tsqr::compute_mode::fp16_tc_nocor>(9211, 16);                // io type half (reference src/tsqr.hpp:38-39)
	rc |= run<mtk::tsqr::compute_mode::fp16_notc>(4096, 64);
	bool unsupported_threw = false;
	try {
		mtk::tsqr::buffer<mtk::tsqr::compute_mode::tf32_tc_cor> b;
		b.allocate(64, 16);
		float* p = nullptr;
		hipMalloc((void**)&p, sizeof(float) * 64 * 16);
		mtk::tsqr::tsqr16<mtk::tsqr::compute_mode::tf32_tc_cor>(p, 64, p, 16, p, 64, 64, 16, b, nullptr);
	} catch (const std::runtime_error&) { unsupported_threw = true; }
	rc |= unsupported_threw ? 0 : 1;
	std::printf(rc == 0 ? "TSQR16 SAMPLE OK\n" : "TSQR16 SAMPLE FAILED\n");
	return rc;
}
