"""tsqr_gpu_amd -- MI355X-native tall-skinny QR behind the mtk::qr::qr / mtk::qr::buffer surface.

csrc/      hand-written HIP kernels (gfx950) + the C ABI (include/tsqr_mi.h)
blockqr.py Python mirror of the reference's blockqr.hpp interface (ctypes over the C ABI)
dist.py    row-partitioned multi-GPU TSQR (one process per GPU, R factors all-gathered over RCCL)
"""
from .blockqr import (buffer, compute_mode, error_invalid_matrix_size, error_unsupported_mode,  # noqa: F401
                      get_batch_size, get_batch_size_log2, get_working_l_size, get_working_q_size,
                      get_working_r_size, qr, success_factorization, tsqr_colmun_size)
