"""GPU unit tests of the wave-level primitives and MFMA operand layouts the kernels rely on
(tsqr_gpu_amd/csrc/selftest.hip -> libtsqr_selftest.so).  These pin hardware/compiler behaviour that the big kernels
assume: DPP row_newbcast, the inline-asm v_permlane{32,16}_swap 4-lane sum (the builtin form miscompiled under
ROCm 7.2), the 16x16x4 f32 / 16x16x32 bf16 MFMA operand and C/D layouts, and the 3-way bf16 split."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def st():
    import subprocess
    import torch
    assert torch.cuda.is_available()
    so = os.path.join(ROOT, "tsqr_gpu_amd", "csrc", "libtsqr_selftest.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.dirname(so), "-s", "libtsqr_selftest.so"])
    return ctypes.CDLL(so), torch


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_dpp_broadcast_and_lane_sum(st):
    L, torch = st
    out = torch.zeros(5 * 64, device="cuda")
    assert L.tsqr_selftest_prims(_p(out)) == 0
    o = out.cpu().numpy().reshape(5, 64)
    lanes = np.arange(64)
    for row, k in ((0, 5), (1, 0), (2, 15)):
        assert np.array_equal(o[row], 16 * (lanes >> 4) + k)            # row_newbcast:k = lane 16q+k of every row of 16
    assert np.array_equal(o[3], 4 * (lanes & 15) + 96)                  # sum over lanes {c, c+16, c+32, c+48} of the lane id
    assert np.array_equal(o[4], 4369.0 * (1 + (lanes & 15)))            # 1 + 16 + 256 + 4096 weights: every group counted once


def test_mfma_layouts(st):
    L, torch = st
    rng = np.random.default_rng(0)
    a = rng.integers(-4, 5, (16, 4)).astype(np.float32)
    b = rng.integers(-4, 5, (4, 16)).astype(np.float32)                 # asymmetric on purpose (catches a transposed C write)
    d = torch.zeros(256, device="cuda")
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert L.tsqr_selftest_mfma_f32(_p(d), _p(da), _p(db)) == 0
    assert np.array_equal(d.cpu().numpy().reshape(16, 16), a @ b)
    a = rng.integers(-4, 5, (16, 32)).astype(np.float32)
    b = rng.integers(-4, 5, (32, 16)).astype(np.float32)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert L.tsqr_selftest_mfma_bf16(_p(d), _p(da), _p(db)) == 0
    assert np.array_equal(d.cpu().numpy().reshape(16, 16), a @ b)


def test_bf16_three_way_split(st):
    L, torch = st
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(4096) * s for s in (1e-20, 1e-3, 1.0, 1e6, 1e30)]).astype(np.float32)
    dx = torch.from_numpy(x).cuda()
    out = torch.zeros(3 * x.size, device="cuda")
    assert L.tsqr_selftest_split(_p(out), _p(dx), x.size) == 0
    h, m, lo = out.cpu().numpy().reshape(-1, 3).T
    for part in (h, m, lo):                                             # every part is a bf16 value (low 16 bits clear)
        assert np.all(part.view(np.uint32) & 0xFFFF == 0)
    recon = h.astype(np.float64) + m.astype(np.float64) + lo.astype(np.float64)
    rel = np.abs(recon - x.astype(np.float64)) / np.abs(x.astype(np.float64))
    assert rel.max() < 2.0 ** -23                                       # three 8-bit pieces carry the fp32 mantissa
    ref_h = ((x.view(np.uint32).astype(np.uint64) + 0x7FFF + ((x.view(np.uint32) >> 16) & 1)) >> 16 << 16).astype(np.uint32).view(np.float32)
    assert np.array_equal(h, ref_h)                                     # hi = round-to-nearest-even bf16
