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
	std::vector<float> h_a(M * N), h_q(M * N), h_r(N * N, 0.0f), h_a_after(M * N);
	for (auto& v : h_a) v = dist(mt);
	float *d_a, *d_r, *d_q;
	hipMalloc((void**)&d_a, sizeof(float) * M * N);
	hipMalloc((void**)&d_r, sizeof(float) * N * N);
	hipMalloc((void**)&d_q, sizeof(float) * M * N);
	hipMemcpy(d_a, h_a.data(), sizeof(float) * M * N, hipMemcpyHostToDevice);
	hipMemset(d_r, 0, sizeof(float) * N * N);
	mtk::tsqr::buffer<mode> buffer;
	buffer.allocate(M, N);
	bool threw = false;
	try { buffer.allocate(M, N); } catch (const std::runtime_error&) { threw = true; }   // reference tsqr.hpp:66-68
	hipStream_t stream;
	hipStreamCreate(&stream);
	mtk::tsqr::tsqr16<mode>(d_q, M, d_r, N, d_a, M, M, N, buffer, stream);
	hipMemcpy(h_q.data(), d_q, sizeof(float) * M * N, hipMemcpyDeviceToHost);
	hipMemcpy(h_r.data(), d_r, sizeof(float) * N * N, hipMemcpyDeviceToHost);
	hipMemcpy(h_a_after.data(), d_a, sizeof(float) * M * N, hipMemcpyDeviceToHost);
	double num = 0, den = 0, orth = 0;
	for (std::size_t j = 0; j < N; j++)
		for (std::size_t i = 0; i < M; i++) {
			double s = 0;
			for (std::size_t k = 0; k <= j; k++) s += (double)h_q[i + k * M] * h_r[k + j * N];
			const double d = s - h_a[i + j * M];
			num += d * d; den += (double)h_a[i + j * M] * h_a[i + j * M];
		}
	for (std::size_t a = 0; a < N; a++)
		for (std::size_t b = 0; b < N; b++) {
			double s = 0;
			for (std::size_t i = 0; i < M; i++) s += (double)h_q[i + a * M] * h_q[i + b * M];
			s -= (a == b);
			orth += s * s;
		}
	const bool a_intact = (h_a_after == h_a);
	const double residual = std::sqrt(num / den), orthogonality = std::sqrt(orth);
	std::printf("tsqr16 mode=%d %zux%zu residual=%e orthogonality_F=%e a_intact=%d double_allocate_threw=%d batch=%zu bytes=%zu\n",
	            (int)mode, M, N, residual, orthogonality, (int)a_intact, (int)threw, mtk::tsqr::get_batch_size(M), buffer.get_device_memory_size());
	hipFree(d_a); hipFree(d_r); hipFree(d_q); hipStreamDestroy(stream);
	return (threw && a_intact && residual < 5e-7 && orthogonality < 5e-6) ? 0 : 1;
}

int main() {
	int rc = 0;
	rc |= run<mtk::tsqr::compute_mode::fp32_tc_cor>(9211, 16);
	rc |= run<mtk::tsqr::compute_mode::fp32_notc>(4096, 16);
	rc |= run<mtk::tsqr::compute_mode::fp32_tc_cor>(20000, 64);
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
