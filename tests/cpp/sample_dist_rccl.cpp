// A C++ caller that LINKS RCCL and hands its own ncclComm_t to tsqr_mi_qr_f32_dist: the library then finds ncclAllReduce /
// ncclAllGather in the global symbol scope, i.e. in the very RCCL copy this program created the communicator with (include/tsqr_mi.h).
// One rank (a communicator of size 1): the Gram all-reduce (policy 0) and the Householder all-gather (policy 1) both run through RCCL.
// Exit code 0 when residual and orthogonality are within tolerance for both.
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <tsqr_mi.h>

static int run(int policy, ncclComm_t comm, hipStream_t stream) {
	const size_t M = 30011, N = 64;
	std::mt19937 mt(3);
	std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
	std::vector<float> h_a(M * N), h_q(M * N), h_r(N * N);
	for (auto& v : h_a) v = dist(mt);
	float *d_a, *d_q, *d_r, *wq, *wr, *gather;
	hipMalloc((void**)&d_a, 4 * M * N); hipMalloc((void**)&d_q, 4 * M * N); hipMalloc((void**)&d_r, 4 * N * N);
	hipMalloc((void**)&wq, 4 * tsqr_mi_working_q_size_dist(M, N, 1));
	hipMalloc((void**)&wr, 4 * tsqr_mi_working_r_size_dist(M, N, 1));
	hipMalloc((void**)&gather, 4 * N * N);
	hipMemcpy(d_a, h_a.data(), 4 * M * N, hipMemcpyHostToDevice);
	tsqr_mi_set_policy(policy);
	const int st = tsqr_mi_qr_f32_dist(TSQR_MI_FP32_TC_COR, 0, d_q, M, d_r, N, d_a, M, M, N, wq, wr, gather, comm, 1, stream);
	tsqr_mi_set_policy(0);
	if (st != 0) { std::printf("policy %d: state %d (%s)\n", policy, st, tsqr_mi_last_error()); return 1; }
	hipMemcpy(h_q.data(), d_q, 4 * M * N, hipMemcpyDeviceToHost);
	hipMemcpy(h_r.data(), d_r, 4 * N * N, hipMemcpyDeviceToHost);
	double num = 0, den = 0, orth = 0;
	for (size_t j = 0; j < N; j++)
		for (size_t i = 0; i < M; i++) {
			double s = 0;
			for (size_t k = 0; k <= j; k++) s += (double)h_q[i + k * M] * h_r[k + j * N];
			const double d = s - h_a[i + j * M];
			num += d * d; den += (double)h_a[i + j * M] * h_a[i + j * M];
		}
	for (size_t a = 0; a < N; a++)
		for (size_t b = 0; b < N; b++) {
			double s = 0;
			for (size_t i = 0; i < M; i++) s += (double)h_q[i + a * M] * h_q[i + b * M];
			s -= (a == b);
			orth += s * s;
		}
	const double residual = std::sqrt(num / den), orthogonality = std::sqrt(orth);
	std::printf("policy=%d engine=%d residual=%e orthogonality_F=%e\n", policy, tsqr_mi_last_engine(), residual, orthogonality);
	hipFree(d_a); hipFree(d_q); hipFree(d_r); hipFree(wq); hipFree(wr); hipFree(gather);
	return (residual < 5e-7 && orthogonality < 5e-6) ? 0 : 1;
}

int main() {
	ncclUniqueId id;
	ncclComm_t comm;
	if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) { std::printf("RCCL init failed\n"); return 2; }
	hipStream_t stream;
	hipStreamCreate(&stream);
	int rc = run(0, comm, stream);
	rc |= run(1, comm, stream);
	ncclCommDestroy(comm);
	std::printf(rc == 0 ? "DIST SAMPLE OK\n" : "DIST SAMPLE FAILED\n");
	return rc;
}
