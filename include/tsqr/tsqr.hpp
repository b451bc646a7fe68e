// tsqr.hpp -- the mtk::tsqr surface a caller of the reference sees through blockqr.hpp (reference src/blockqr.hpp:7 includes
// src/tsqr.hpp): compute_mode (src/tsqr.hpp:9-20), the batch-size rule (:22-23), the element-type traits (:25-39), the work-space
// sizes (:42-46), buffer<mode> (:49-108) and tsqr16<mode> (:110-140) -- thin QR of one panel of at most 16 columns in the
// reference; this engine takes any n <= 64 through the same entry.
//
// Header-only over the extern "C" ABI of libtsqr_mi.so.  The stream argument is a hipStream_t where the reference has a
// cudaStream_t.  fp32_notc, fp32_tc_cor, fp32_tc_nocor (float I/O) and fp16_notc, fp16_tc_nocor (half I/O); the others throw
// std::runtime_error -- tsqr16 returns void in the reference, so there is no status to report them through.
#ifndef __TSQR_HPP__
#define __TSQR_HPP__
#include <hip/hip_runtime.h>
#include <cstddef>
#include <stdexcept>
#include <string>
#include "../tsqr_mi.h"

namespace mtk {
namespace tsqr {

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

inline std::size_t get_batch_size_log2(const std::size_t m) { return tsqr_mi_batch_size_log2(m); }
inline std::size_t get_batch_size(const std::size_t m) { return tsqr_mi_batch_size(m); }

// io type as in reference src/tsqr.hpp:36-39: float for the fp32 modes, IEEE binary16 (the reference's `half`) for fp16_notc /
// fp16_tc_nocor.  The WORKING types are float for every mode this engine implements (it factors in fp32 and converts at the
// boundary), the other modes keep their names so that code mentioning them compiles.
// One difference a hand-allocating caller must know: the reference's working types are `half` for the fp16 modes and for the
// working Q of fp32_tc_nocor (reference src/tsqr.hpp:27-34), here they are float -- size the work space in ELEMENTS of these traits'
// types (get_working_q_size<mode>(m, n) * sizeof(get_working_q_type<mode>::type), as mtk::tsqr::buffer and mtk::qr::buffer do),
// never in elements of the reference's type: a `half`-sized wq would be half the bytes this engine writes.
using half_t = _Float16;
template <compute_mode mode> struct get_working_q_type { using type = float; };
template <compute_mode mode> struct get_working_r_type { using type = float; };
template <compute_mode mode> struct get_io_type { using type = float; };
template <> struct get_io_type<fp16_notc> { using type = half_t; };
template <> struct get_io_type<fp16_tc_nocor> { using type = half_t; };

inline std::size_t get_working_q_size(const std::size_t m, const std::size_t n) { return tsqr_mi_working_q_size(m, n); }
inline std::size_t get_working_r_size(const std::size_t m, const std::size_t n) { return tsqr_mi_working_r_size(m, n); }
inline std::size_t get_working_l_size(const std::size_t m) { return tsqr_mi_working_l_size(m); }
// per mode: the fp16 I/O modes need room for the widened A, Q and R on top
template <compute_mode mode> inline std::size_t get_working_q_size(const std::size_t m, const std::size_t n) {
	return (mode == fp16_notc || mode == fp16_tc_nocor) ? tsqr_mi_working_q_size_f16(m, n) : tsqr_mi_working_q_size(m, n);
}
template <compute_mode mode> inline std::size_t get_working_r_size(const std::size_t m, const std::size_t n) {
	return (mode == fp16_notc || mode == fp16_tc_nocor) ? tsqr_mi_working_r_size_f16(m, n) : tsqr_mi_working_r_size(m, n);
}

namespace detail {
// the C entry point of an io type (reorth = 0: one panel)
inline int panel_entry(int mode, float* q, std::size_t ldq, float* r, std::size_t ldr, const float* a, std::size_t lda, std::size_t m, std::size_t n,
                       void* wq, void* wr, unsigned* d_wl, unsigned* h_wl, hipStream_t stream) {
	// (a panel of at most 64 columns is never written to by the engine: the const_cast only matches the C signature)
	return tsqr_mi_qr_f32(mode, 0, q, ldq, r, ldr, const_cast<float*>(a), lda, m, n, wq, wr, nullptr, d_wl, h_wl, stream);
}
inline int panel_entry(int mode, half_t* q, std::size_t ldq, half_t* r, std::size_t ldr, const half_t* a, std::size_t lda, std::size_t m, std::size_t n,
                       void* wq, void* wr, unsigned* d_wl, unsigned* h_wl, hipStream_t stream) {
	return tsqr_mi_qr_f16(mode, 0, q, ldq, r, ldr, a, lda, m, n, wq, wr, nullptr, d_wl, h_wl, stream);
}
enum class where { device, pinned_host };
inline void* grab(where w, std::size_t bytes, const char* what) {
	void* p = nullptr;
	const hipError_t e = (w == where::device) ? hipMalloc(&p, bytes) : hipHostMalloc(&p, bytes);
	if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
	return p;
}
template <class T> inline void drop(where w, T*& p) {
	if (p) (void)((w == where::device) ? hipFree(p) : hipHostFree(p));
	p = nullptr;
}
}  // namespace detail

// work space of one panel factorisation: dwq, dwr, dl on the device (or all in pinned host memory: allocate_host), hl pinned
template <mtk::tsqr::compute_mode mode>
struct buffer {
	typename get_working_q_type<mode>::type* dwq = nullptr;
	typename get_working_r_type<mode>::type* dwr = nullptr;
	unsigned* dl = nullptr;
	unsigned* hl = nullptr;
	std::size_t total_memory_size = 0;

	buffer() = default;
	buffer(const buffer&) = delete;
	buffer& operator=(const buffer&) = delete;
	~buffer() { destroy(); }

	void allocate(const std::size_t m, const std::size_t n) { fill(detail::where::device, m, n); }
	void allocate_host(const std::size_t m, const std::size_t n) { fill(detail::where::pinned_host, m, n); }
	void destroy() { release(detail::where::device); }
	void destroy_host() { release(detail::where::pinned_host); }
	std::size_t get_device_memory_size() const { return total_memory_size; }

private:
	void fill(detail::where w, const std::size_t m, const std::size_t n) {
		if (dwq || dwr || dl || hl) throw std::runtime_error("The buffer has been already allocated");
		const std::size_t q_bytes = sizeof(*dwq) * get_working_q_size<mode>(m, n), r_bytes = sizeof(*dwr) * get_working_r_size<mode>(m, n),
		                  l_bytes = sizeof(unsigned) * get_working_l_size(m);
		dwq = static_cast<decltype(dwq)>(detail::grab(w, q_bytes, "mtk::tsqr::buffer dwq"));
		dwr = static_cast<decltype(dwr)>(detail::grab(w, r_bytes, "mtk::tsqr::buffer dwr"));
		dl = static_cast<unsigned*>(detail::grab(w, l_bytes, "mtk::tsqr::buffer dl"));
		hl = static_cast<unsigned*>(detail::grab(detail::where::pinned_host, l_bytes, "mtk::tsqr::buffer hl"));
		total_memory_size = q_bytes + r_bytes + l_bytes;
	}
	void release(detail::where w) {
		detail::drop(w, dwq);
		detail::drop(w, dwr);
		detail::drop(w, dl);
		detail::drop(detail::where::pinned_host, hl);
	}
};

// Q (m x n, ldq), R (n x n, ldr) of the panel A (m x n, lda); column-major; blocking; A is not modified.
template <mtk::tsqr::compute_mode mode>
inline void tsqr16(
		typename get_io_type<mode>::type* const q_ptr, const std::size_t ldq,
		typename get_io_type<mode>::type* const r_ptr, const std::size_t ldr,
		const typename get_io_type<mode>::type* const a_ptr, const std::size_t lda,
		const std::size_t m, const std::size_t n,
		typename get_working_q_type<mode>::type* const working_q_ptr,
		typename get_working_r_type<mode>::type* const working_r_ptr,
		unsigned* const d_working_l_ptr,
		unsigned* const h_working_l_ptr,
		hipStream_t const stream = nullptr) {
	if (n > 64) throw std::runtime_error("mtk::tsqr::tsqr16: one panel has at most 64 columns (the reference's limit is 16)");
	const int st = detail::panel_entry(static_cast<int>(mode), q_ptr, ldq, r_ptr, ldr, a_ptr, lda, m, n,
	                                   working_q_ptr, working_r_ptr, d_working_l_ptr, h_working_l_ptr, stream);
	if (st < 0) throw std::runtime_error(std::string("mtk::tsqr::tsqr16: ") + tsqr_mi_last_error());
	if (st != 0) throw std::runtime_error(st == 2 ? "mtk::tsqr::tsqr16: compute_mode not implemented on gfx950"
	                                              : "mtk::tsqr::tsqr16: invalid matrix size");
}

template <mtk::tsqr::compute_mode mode>
inline void tsqr16(
		typename get_io_type<mode>::type* const q_ptr, const std::size_t ldq,
		typename get_io_type<mode>::type* const r_ptr, const std::size_t ldr,
		const typename get_io_type<mode>::type* const a_ptr, const std::size_t lda,
		const std::size_t m, const std::size_t n,
		mtk::tsqr::buffer<mode>& buffer,
		hipStream_t const stream) {
	mtk::tsqr::tsqr16<mode>(q_ptr, ldq, r_ptr, ldr, a_ptr, lda, m, n, buffer.dwq, buffer.dwr, buffer.dl, buffer.hl, stream);
}

}  // namespace tsqr
}  // namespace mtk
#endif /* end of include guard */
