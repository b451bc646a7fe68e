#!/usr/bin/env python3
"""How evenly does the static partition of the Gram pass finish?  gram_blk_kernel's body with a start and an end stamp per workgroup
and the XCD / CU it ran on (selftest library): end times per XCD, per position in the dispatch order, and what a balanced pass would take."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
L = ctypes.CDLL(os.path.join(ROOT, "tsqr_gpu_amd", "csrc", "libtsqr_selftest.so"))
L.tsqr_selftest_gram_balance.restype = ctypes.c_int
L.tsqr_selftest_gram_balance.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
if len(sys.argv) > 1 and sys.argv[1] == "apply":
    # the apply pass: 1024 workgroups (four per CU), each 16 blocks of 64 rows
    L.tsqr_selftest_apply_balance.restype = ctypes.c_int
    L.tsqr_selftest_apply_balance.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_int] * 5
    m = 1 << 20
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    a = torch.rand(64, m, generator=g, device="cuda") * 2 - 1
    q = torch.empty(64, m, device="cuda")
    z = torch.triu(torch.rand(64, 64, generator=g, device="cuda")).T.contiguous()
    qref = None
    for nwg, sh in ((1024, (0, 0, 0, 0, 0)), (1024, (16, 16, 16, 16, 64)), (1024, (16, 16, 16, 16, 69)), (1024, (20, 17, 14, 13, 69)), (1024, (18, 17, 15, 14, 69)),
                    (1024, (18, 17, 15, 14, 72)), (1024, (20, 17, 14, 13, 66)), (1024, (0, 0, 0, 0, 0)), (1024, (20, 17, 14, 13, 69))):
        st = torch.zeros(4 * nwg, dtype=torch.int64, device="cuda")
        q.fill_(float("nan"))
        assert L.tsqr_selftest_apply_balance(q.data_ptr(), a.data_ptr(), m, m, z.data_ptr(), nwg, st.data_ptr(), 30, *sh) == 0
        if qref is None:
            qref = q.clone()
        assert torch.equal(q, qref), "Q depends on the partition"
        print("shares %s:" % (sh,), end=" ")
        t = st.cpu().numpy().reshape(nwg, 4)
        t0 = t[:, 0].min()
        start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
        print("apply pass, %d workgroups: starts %.1f .. %.1f us; ends min %.1f  median %.1f  max %.1f us; mean duration %.1f us" % (
            nwg, start.min(), start.max(), end.min(), np.median(end), end.max(), (end - start).mean()))
        q4 = nwg // 4
        print("  by position in the grid (quarters of the workgroup index), mean end: " + "  ".join("%.1f" % end[i * q4:(i + 1) * q4].mean() for i in range(4)))
        xcc, hw = t[:, 2] & 0xF, t[:, 3]
        print("  per XCD (mean / max end): " + "  ".join("%d: %.1f / %.1f" % (x, end[xcc == x].mean(), end[xcc == x].max()) for x in sorted(set(xcc))))
        print("  workgroup index mod 8 == XCC_ID for %d of %d workgroups" % ((np.arange(nwg) % 8 == xcc).sum(), nwg))
        key = xcc * 4096 + (hw >> 13 & 7) * 256 + (hw >> 12 & 1) * 16 + (hw >> 8 & 0xF)
        cu_last = np.array([end[key == k].max() for k in set(key)])
        cu_mean = np.array([end[key == k].mean() for k in set(key)])
        print("  %d CUs: last end per CU min %.1f  median %.1f  max %.1f us; mean end per CU min %.1f  max %.1f us" % (
            len(cu_last), cu_last.min(), np.median(cu_last), cu_last.max(), cu_mean.min(), cu_mean.max()))
    sys.exit(0)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
g = torch.Generator(device="cuda"); g.manual_seed(5)
a = torch.rand(64, m, generator=g, device="cuda") * 2 - 1
nparts = min(m // 128, 512)
part = torch.zeros(nparts * 2560, dtype=torch.float64, device="cuda")
shares = [int(x) for x in sys.argv[2:]] or [16, 18, 19, 20, 21, 22]
for share in shares * 2:
    rep = share
    st = torch.zeros(4 * nparts, dtype=torch.int64, device="cuda")
    assert L.tsqr_selftest_gram_balance(a.data_ptr(), m, m, nparts, part.data_ptr(), st.data_ptr(), 30, share) == 0
    t = st.cpu().numpy().reshape(nparts, 4)
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
    xcc = t[:, 2] & 0xF
    hw = t[:, 3]
    cu, sh, se = hw >> 8 & 0xF, hw >> 12 & 1, hw >> 13 & 7
    dur = end - start
    print("older workgroup's share %d / 32: starts %.1f .. %.1f us; ends min %.1f  median %.1f  max %.1f us; mean duration %.1f us (a balanced pass: ~%.1f us)" % (
        rep, start.min(), start.max(), end.min(), np.median(end), end.max(), dur.mean(), dur.mean()))
    print("  per XCD (id: workgroups, mean / max end): " + "  ".join("%d: %d, %.1f / %.1f" % (x, (xcc == x).sum(), end[xcc == x].mean(), end[xcc == x].max()) for x in sorted(set(xcc))))
    q = nparts // 4
    print("  by position in the grid (quarters of the workgroup index), mean end: " + "  ".join("%.1f" % end[i * q:(i + 1) * q].mean() for i in range(4)))
    # two workgroups per CU: do CU-mates end together?
    key = xcc * 4096 + se * 256 + sh * 16 + cu
    pairs = [end[key == k] for k in set(key) if (key == k).sum() == 2]
    if pairs:
        d = np.array([abs(p[0] - p[1]) for p in pairs])
        print("  %d CUs with two workgroups: |end difference| median %.1f us; spread of the CU means %.1f .. %.1f us" % (len(pairs), np.median(d), min(p.mean() for p in pairs), max(p.mean() for p in pairs)))
