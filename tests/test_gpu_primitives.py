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


def test_apply_result_does_not_depend_on_the_block_shares(st):
    """The library picks the apply pass's block shares by measurement on the box it runs on (round 4) -- legitimate only because Q = A Z is
    bit for bit the same whoever computes a block.  apply_wg_kernel's body under the share settings the library can take (equal, by
    dispatch round only, by XCD parity and round) and two others: identical Q."""
    import ctypes
    L, torch = st
    L.tsqr_selftest_apply_balance.restype = ctypes.c_int
    L.tsqr_selftest_apply_balance.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 5
    m, nwg = 1 << 20, 1024
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    a = torch.rand(64, m, generator=g, device="cuda") * 2 - 1
    z = torch.triu(torch.rand(64, 64, generator=g, device="cuda")).T.contiguous()
    q = torch.empty(64, m, device="cuda")
    qref = None
    for sh in ((0, 0, 0, 0, 0), (18, 17, 15, 14, 64), (18, 17, 15, 14, 69), (20, 17, 14, 13, 72), (16, 16, 16, 16, 60)):
        stamps = torch.zeros(4 * nwg, dtype=torch.int64, device="cuda")
        q.fill_(float("nan"))
        assert L.tsqr_selftest_apply_balance(_p(q), _p(a), m, m, _p(z), nwg, _p(stamps), 1, *sh) == 0
        assert not torch.isnan(q).any()
        if qref is None:
            qref = q.clone()
        assert torch.equal(q, qref), "Q depends on the block shares %s" % (sh,)
