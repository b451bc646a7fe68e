// blockqr.hpp -- MI355X-native replacement for the reference header of the same name.
//
// Keeps the public surface of reference src/blockqr.hpp (mtk::qr::compute_mode :12-23, tsqr_colmun_size :25,
// state_t :27-29, get_working_*_size :55-57, buffer :59-140, qr :142-175) so that callers written against
// enp1s0/tsqr-gpu compile unchanged, except for ONE argument: the reference passes a cublasHandle_t, which it
// uses only to obtain the stream (src/blockqr.cu:58-59) and to run its inter-panel GEMMs.  There is no cuBLAS on
// ROCm and no compatibility shim here; the last argument is a hipStream_t (mtk::qr::handle_t).  Overloads
// without the handle use the null stream.
//
// Header-only: everything forwards to the extern "C" ABI of libtsqr_mi.so (include/tsqr_mi.h).
// Link with -ltsqr_mi.
#ifndef __BLOCKQR_HPP__
#define __BLOCKQR_HPP__
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <type_traits>
#include "../tsqr_mi.h"
#include "tsqr.hpp"                             // mtk::tsqr (the reference's blockqr.hpp includes its tsqr.hpp as well)

namespace mtk {
namespace qr {

enum compute_mode {
	fp16_notc,
	fp16_tc_nocor,
	fp32_notc,
	fp32_tc_cor,
	fp32_tc_nocor,
	mixed_tc_cor_emu,
	tf32_tc_cor,
	tf32_tc_cor_emu,
	tf32_tc_nocor,
	tf32_tc_nocor_emu,
};

constexpr std::size_t tsqr_colmun_size = 16;

using state_t = int;
const state_t success_factorization = 0;
const state_t error_invalid_matrix_size = 1;
const state_t error_unsupported_mode = 2;      // new: compute_mode without a gfx950 implementation

using handle_t = hipStream_t;                  // takes the place of cublasHandle_t (see the header comment)

// reference src/blockqr.hpp:31-43: the one-panel engine's name for a block-QR mode (same enumerator names in both enums)
template <mtk::qr::compute_mode>
constexpr mtk::tsqr::compute_mode get_tsqr_compute_mode();
#define BQR_GET_TSQR_COMPUTE_MODE(mode) template<> constexpr mtk::tsqr::compute_mode get_tsqr_compute_mode<mtk::qr::compute_mode::mode>() {return mtk::tsqr::compute_mode::mode;}
BQR_GET_TSQR_COMPUTE_MODE(fp16_notc        );
BQR_GET_TSQR_COMPUTE_MODE(fp32_notc        );
BQR_GET_TSQR_COMPUTE_MODE(fp16_tc_nocor    );
BQR_GET_TSQR_COMPUTE_MODE(fp32_tc_nocor    );
BQR_GET_TSQR_COMPUTE_MODE(tf32_tc_nocor    );
BQR_GET_TSQR_COMPUTE_MODE(fp32_tc_cor      );
BQR_GET_TSQR_COMPUTE_MODE(tf32_tc_cor      );
BQR_GET_TSQR_COMPUTE_MODE(tf32_tc_cor_emu  );
BQR_GET_TSQR_COMPUTE_MODE(tf32_tc_nocor_emu);
BQR_GET_TSQR_COMPUTE_MODE(mixed_tc_cor_emu );

// Element types, as in reference src/tsqr.hpp:25-39 for the io type: float for the fp32 modes, IEEE binary16 (the reference's
// `half`) for fp16_notc / fp16_tc_nocor.  The WORKING types are float for every mode (the reference's are half for the fp16 modes
// and for the working Q of fp32_tc_nocor): this engine factors in fp32 and converts at the boundary.  The remaining modes are
// declared so that code naming them still compiles; calling them returns error_unsupported_mode.
using half_t = mtk::tsqr::half_t;
template <mtk::qr::compute_mode mode> struct get_working_q_type { using type = float; };
template <mtk::qr::compute_mode mode> struct get_working_r_type { using type = float; };
template <mtk::qr::compute_mode mode> struct get_io_type { using type = float; };
template <> struct get_io_type<fp16_notc> { using type = half_t; };
template <> struct get_io_type<fp16_tc_nocor> { using type = half_t; };

// get working memory size (element counts), reference src/blockqr.hpp:55-57
inline std::size_t get_working_q_size(const std::size_t m, const std::size_t n) { return tsqr_mi_working_q_size(m, n); }
inline std::size_t get_working_r_size(const std::size_t m, const std::size_t n) { return tsqr_mi_working_r_size(m, n); }
inline std::size_t get_working_l_size(const std::size_t m) { return tsqr_mi_working_l_size(m); }
// the same per mode: the fp16 I/O modes need room for the widened A, Q and R on top (what buffer<mode>::allocate uses; a caller
// that allocates by hand for an fp16 mode must use these, in elements of get_working_{q,r}_type<mode>::type = float)
template <mtk::qr::compute_mode mode> inline std::size_t get_working_q_size(const std::size_t m, const std::size_t n) {
	return (mode == fp16_notc || mode == fp16_tc_nocor) ? tsqr_mi_working_q_size_f16(m, n) : tsqr_mi_working_q_size(m, n);
}
template <mtk::qr::compute_mode mode> inline std::size_t get_working_r_size(const std::size_t m, const std::size_t n) {
	return (mode == fp16_notc || mode == fp16_tc_nocor) ? tsqr_mi_working_r_size_f16(m, n) : tsqr_mi_working_r_size(m, n);
}

namespace detail {
inline void check(hipError_t e, const char* what) {
	if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
// the C entry point of an io type
template <class IO> struct entry;
template <> struct entry<float> {
	static int call(int mode, int reorth, float* q, std::size_t ldq, float* r, std::size_t ldr, float* a, std::size_t lda, std::size_t m, std::size_t n,
	                void* wq, void* wr, float* reorth_w, unsigned* d_wl, unsigned* h_wl, hipStream_t stream) {
		return tsqr_mi_qr_f32(mode, reorth, q, ldq, r, ldr, a, lda, m, n, wq, wr, reorth_w, d_wl, h_wl, stream);
	}
};
template <> struct entry<half_t> {
	static int call(int mode, int reorth, half_t* q, std::size_t ldq, half_t* r, std::size_t ldr, half_t* a, std::size_t lda, std::size_t m, std::size_t n,
	                void* wq, void* wr, half_t* reorth_w, unsigned* d_wl, unsigned* h_wl, hipStream_t stream) {
		return tsqr_mi_qr_f16(mode, reorth, q, ldq, r, ldr, a, lda, m, n, wq, wr, reorth_w, d_wl, h_wl, stream);
	}
};
}  // namespace detail

template <mtk::qr::compute_mode mode, bool Reorthogonalize>
struct buffer {
	typename get_working_q_type<mode>::type* dwq;
	typename get_working_r_type<mode>::type* dwr;
	typename get_io_type<mode>::type* dw_reorth_r;
	unsigned* dl;
	unsigned* hl;

	std::size_t total_memory_size;

	buffer() : dwq(nullptr), dwr(nullptr), dw_reorth_r(nullptr), dl(nullptr), hl(nullptr), total_memory_size(0lu) {}
	~buffer() { destroy(); }
	buffer(const buffer&) = delete;
	buffer& operator=(const buffer&) = delete;

	void allocate(const std::size_t m, const std::size_t n) {
		if (dwq != nullptr || dwr != nullptr || dl != nullptr || hl != nullptr) {
			throw std::runtime_error("The buffer has been already allocated");
		}
		const auto wq_size = sizeof(typename get_working_q_type<mode>::type) * get_working_q_size<mode>(m, n);
		const auto wr_size = sizeof(typename get_working_r_type<mode>::type) * get_working_r_size<mode>(m, n);
		const auto l_size = sizeof(unsigned) * get_working_l_size(m);
		detail::check(hipMalloc(reinterpret_cast<void**>(&dwq), wq_size), "hipMalloc(dwq)");
		detail::check(hipMalloc(reinterpret_cast<void**>(&dwr), wr_size), "hipMalloc(dwr)");
		detail::check(hipMalloc(reinterpret_cast<void**>(&dl), l_size), "hipMalloc(dl)");
		detail::check(hipHostMalloc(reinterpret_cast<void**>(&hl), l_size), "hipHostMalloc(hl)");
		total_memory_size = wq_size + wr_size + l_size;
		if (Reorthogonalize) {
			const auto reorth_r_size = sizeof(typename get_io_type<mode>::type) * tsqr_mi_working_reorth_size(m);
			detail::check(hipMalloc(reinterpret_cast<void**>(&dw_reorth_r), reorth_r_size), "hipMalloc(dw_reorth_r)");
			total_memory_size += reorth_r_size;
		}
	}

	void destroy() {
		if (dwq) (void)hipFree(dwq);
		dwq = nullptr;
		if (dwr) (void)hipFree(dwr);
		dwr = nullptr;
		if (dw_reorth_r) (void)hipFree(dw_reorth_r);
		dw_reorth_r = nullptr;
		if (dl) (void)hipFree(dl);
		dl = nullptr;
		if (hl) (void)hipHostFree(hl);
		hl = nullptr;
	}

	// host-resident variants of the reference (src/blockqr.hpp:97-135): pinned host memory mapped to the device
	void allocate_host(const std::size_t m, const std::size_t n) {
		if (dwq != nullptr || dwr != nullptr || dl != nullptr || hl != nullptr) {
			throw std::runtime_error("The buffer has been already allocated");
		}
		const auto wq_size = sizeof(typename get_working_q_type<mode>::type) * get_working_q_size<mode>(m, n);
		const auto wr_size = sizeof(typename get_working_r_type<mode>::type) * get_working_r_size<mode>(m, n);
		const auto l_size = sizeof(unsigned) * get_working_l_size(m);
		detail::check(hipHostMalloc(reinterpret_cast<void**>(&dwq), wq_size), "hipHostMalloc(dwq)");
		detail::check(hipHostMalloc(reinterpret_cast<void**>(&dwr), wr_size), "hipHostMalloc(dwr)");
		detail::check(hipHostMalloc(reinterpret_cast<void**>(&dl), l_size), "hipHostMalloc(dl)");
		detail::check(hipHostMalloc(reinterpret_cast<void**>(&hl), l_size), "hipHostMalloc(hl)");
		total_memory_size = wq_size + wr_size + l_size;
		if (Reorthogonalize) {
			const auto reorth_r_size = sizeof(typename get_io_type<mode>::type) * tsqr_mi_working_reorth_size(m);
			detail::check(hipHostMalloc(reinterpret_cast<void**>(&dw_reorth_r), reorth_r_size), "hipHostMalloc(dw_reorth_r)");
			total_memory_size += reorth_r_size;
		}
	}

	void destroy_host() {
		if (dwq) (void)hipHostFree(dwq);
		dwq = nullptr;
		if (dwr) (void)hipHostFree(dwr);
		dwr = nullptr;
		if (dw_reorth_r) (void)hipHostFree(dw_reorth_r);
		dw_reorth_r = nullptr;
		if (dl) (void)hipHostFree(dl);
		dl = nullptr;
		if (hl) (void)hipHostFree(hl);
		hl = nullptr;
	}

	std::size_t get_device_memory_size() const { return total_memory_size; }
};

// reference src/blockqr.hpp:142-154 / src/blockqr.cu:394-433.  Blocking; returns state_t; runtime failures throw
// std::runtime_error (the reference throws from CUTF_CHECK_ERROR).
template <mtk::qr::compute_mode mode, bool Reorthogonalize>
inline state_t qr(
		typename mtk::qr::get_io_type<mode>::type* const q_ptr, const std::size_t ldq,
		typename mtk::qr::get_io_type<mode>::type* const r_ptr, const std::size_t ldr,
		typename mtk::qr::get_io_type<mode>::type* const a_ptr, const std::size_t lda,
		const std::size_t m, const std::size_t n,
		typename mtk::qr::get_working_q_type<mode>::type* const wq_ptr,
		typename mtk::qr::get_working_r_type<mode>::type* const wr_ptr,
		typename mtk::qr::get_io_type<mode>::type* const reorth_r_ptr,
		unsigned* const d_wl_ptr,
		unsigned* const h_wl_ptr,
		handle_t const stream = nullptr) {
	const int st = detail::entry<typename mtk::qr::get_io_type<mode>::type>::call(static_cast<int>(mode), Reorthogonalize ? 1 : 0,
	                              q_ptr, ldq, r_ptr, ldr, a_ptr, lda, m, n,
	                              wq_ptr, wr_ptr, reorth_r_ptr, d_wl_ptr, h_wl_ptr, stream);
	if (st < 0) throw std::runtime_error(std::string("mtk::qr::qr: ") + tsqr_mi_last_error());
	return st;
}

// reference src/blockqr.hpp:155-175
template <mtk::qr::compute_mode mode, bool Reorthogonalize>
inline state_t qr(
		typename mtk::qr::get_io_type<mode>::type* const q_ptr, const std::size_t ldq,
		typename mtk::qr::get_io_type<mode>::type* const r_ptr, const std::size_t ldr,
		typename mtk::qr::get_io_type<mode>::type* const a_ptr, const std::size_t lda,
		const std::size_t m, const std::size_t n,
		buffer<mode, Reorthogonalize>& bf,
		handle_t const stream = nullptr) {
	return qr<mode, Reorthogonalize>(
			q_ptr, ldq,
			r_ptr, ldr,
			a_ptr, lda,
			m, n,
			bf.dwq,
			bf.dwr,
			bf.dw_reorth_r,
			bf.dl,
			bf.hl,
			stream
			);
}

// Not in the reference: its qr() synchronises the stream itself (src/blockqr.cu:78, 122, 140), so one call's host round trip sits
// between two calls of a loop.  submit enqueues a call's first attempt and returns; finish waits for it (and completes the fallback
// ladder for a matrix the conditioning check rejected).  Two calls of a thread may be in flight: a loop over many matrices keeps the
// GPU busy back to back (tsqr_mi_qr_f32_submit / _finish, include/tsqr_mi.h, for the rules).  fp32 I/O modes.
using ticket = tsqr_mi_ticket;
template <mtk::qr::compute_mode mode, bool Reorthogonalize>
inline void qr_submit(
		ticket& t,
		float* const q_ptr, const std::size_t ldq,
		float* const r_ptr, const std::size_t ldr,
		float* const a_ptr, const std::size_t lda,
		const std::size_t m, const std::size_t n,
		buffer<mode, Reorthogonalize>& bf,
		handle_t const stream = nullptr) {
	static_assert(std::is_same<typename mtk::qr::get_io_type<mode>::type, float>::value, "qr_submit takes the fp32 I/O modes");
	const int st = tsqr_mi_qr_f32_submit(static_cast<int>(mode), Reorthogonalize ? 1 : 0, q_ptr, ldq, r_ptr, ldr, a_ptr, lda, m, n,
	                                     bf.dwq, bf.dwr, bf.dw_reorth_r, bf.dl, bf.hl, stream, &t);
	if (st < 0) throw std::runtime_error(std::string("mtk::qr::qr_submit: ") + tsqr_mi_last_error());
}
inline state_t qr_finish(ticket& t) {
	const int st = tsqr_mi_qr_f32_finish(&t);
	if (st < 0) throw std::runtime_error(std::string("mtk::qr::qr_finish: ") + tsqr_mi_last_error());
	return st;
}

// Not in the reference either: `count` different matrices of one shape through one call (tsqr_mi_qr_f32_batch).  What a caller's loop
//     for (i = 0; i < count; i++) mtk::qr::qr<mode, Reorth>(q[i], ldq, r[i], ldr, a[i], lda, m, n, buffer, handle);
// computes -- bit for bit, including the fallback ladder of a matrix the conditioning check rejects -- issued as a stream: the 22 us
// in which one workgroup factors the Gram matrix of matrix i are hidden in the Gram pass of matrix i + 1 (tsqr_mi.h for the rules:
// q[i] and r[i] must be clear of a[i + 1] for that; in place, q[i] == a[i], is fine).  q, r, a: host arrays of device pointers.
// states (optional): the state_t of every call.  Returns the first non-zero state_t.  All implemented modes (half-typed pointers for the fp16 I/O modes).
template <mtk::qr::compute_mode mode, bool Reorthogonalize>
inline state_t qr_batch(
		const std::size_t count,
		typename mtk::qr::get_io_type<mode>::type* const* const q_ptrs, const std::size_t ldq,
		typename mtk::qr::get_io_type<mode>::type* const* const r_ptrs, const std::size_t ldr,
		typename mtk::qr::get_io_type<mode>::type* const* const a_ptrs, const std::size_t lda,
		const std::size_t m, const std::size_t n,
		buffer<mode, Reorthogonalize>& bf,
		handle_t const stream = nullptr,
		state_t* const states = nullptr) {
	int st;
	if constexpr (std::is_same<typename mtk::qr::get_io_type<mode>::type, float>::value)
		st = tsqr_mi_qr_f32_batch(static_cast<int>(count), static_cast<int>(mode), Reorthogonalize ? 1 : 0, q_ptrs, ldq, r_ptrs, ldr, a_ptrs, lda, m, n,
		                          bf.dwq, bf.dwr, bf.dw_reorth_r, bf.dl, bf.hl, stream, states);
	else                                                  // the half-typed modes (same pointer arrays, elements are halves)
		st = tsqr_mi_qr_f16_batch(static_cast<int>(count), static_cast<int>(mode), Reorthogonalize ? 1 : 0, reinterpret_cast<void* const*>(q_ptrs), ldq,
		                          reinterpret_cast<void* const*>(r_ptrs), ldr, reinterpret_cast<const void* const*>(a_ptrs), lda, m, n,
		                          bf.dwq, bf.dwr, bf.dw_reorth_r, bf.dl, bf.hl, stream, states);
	if (st < 0) throw std::runtime_error(std::string("mtk::qr::qr_batch: ") + tsqr_mi_last_error());
	return st;
}
}  // namespace qr
}  // namespace mtk

#endif /* end of include guard */
