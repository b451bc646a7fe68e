#!/usr/bin/env python3
"""Experiment: the R-factor chain of call i + 1 (Gram pass, reduction, Cholesky) on a second stream while the apply pass of call i runs
on the first -- the staged entry points (tsqr_mi_gram_f32 / _chol_f32 / _apply_z_f32), two work buffers, events between the streams.
Prints the period per call against the same stages on ONE stream.  two_stream.py [calls] [m] [n] [mode]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tsqr_gpu_amd import blockqr as bq
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
mode = bq.compute_mode[sys.argv[4]] if len(sys.argv) > 4 else bq.compute_mode.fp32_tc_cor
L = bq.lib()
g = torch.Generator(device="cuda"); g.manual_seed(0)
a = torch.rand(n, m, generator=g, device="cuda") * 2 - 1
q = torch.empty(n, m, device="cuda")
rs = [torch.zeros(n, n, device="cuda") for _ in range(2)]
bfs = [bq.buffer(mode, False) for _ in range(2)]
for b in bfs:
    b.allocate(m, n)
vp = ctypes.c_void_p


def chain(i, st):          # Gram pass + reduction + Cholesky (verdict stays on the device) of call i on stream st
    b = bfs[i & 1]
    assert L.tsqr_mi_gram_f32(2, None, a.data_ptr(), m, m, n, b.dwq.data_ptr(), b.dwr.data_ptr(), st.cuda_stream) == 0
    assert L.tsqr_mi_chol_f32(2, rs[i & 1].data_ptr(), n, None, m, n, b.dwq.data_ptr(), None, st.cuda_stream) == 0


def apply(i, st):
    b = bfs[i & 1]
    assert L.tsqr_mi_apply_z_f32(int(mode), q.data_ptr(), m, a.data_ptr(), m, m, n, b.dwq.data_ptr(), st.cuda_stream) == 0


def one_stream(k):
    s = torch.cuda.current_stream()
    for i in range(k):
        chain(i, s); apply(i, s)


def two_streams(k, S, H):
    e_chol = [torch.cuda.Event() for _ in range(k + 1)]
    e_apply = [torch.cuda.Event() for _ in range(k + 1)]
    chain(0, H); e_chol[0].record(H)
    for i in range(k):
        if i + 1 < k:
            if i >= 1:
                H.wait_event(e_apply[i - 1])        # work set (i + 1) & 1 was read by apply(i - 1)
            chain(i + 1, H); e_chol[i + 1].record(H)
        S.wait_event(e_chol[i])
        apply(i, S); e_apply[i].record(S)


def free_cu_streams(k, S, H):
    """S: Gram pass + reduction of call i + 1, then the apply pass of call i; H: only the Cholesky launches -- with grids that leave ONE CU
    free (tsqr_mi_set_tuning2), so that the one-workgroup factorisation can run beside the passes instead of behind them"""
    eg = [torch.cuda.Event() for _ in range(k + 2)]
    ec = [torch.cuda.Event() for _ in range(k + 2)]
    ea = [torch.cuda.Event() for _ in range(k + 2)]
    def G(i):
        b = bfs[i & 1]
        assert L.tsqr_mi_gram_f32(2, None, a.data_ptr(), m, m, n, b.dwq.data_ptr(), b.dwr.data_ptr(), S.cuda_stream) == 0
        eg[i].record(S)
    def C(i):
        b = bfs[i & 1]
        H.wait_event(eg[i])
        if i >= 2:
            H.wait_event(ea[i - 2])                 # Z of this work set was read by apply(i - 2)
        assert L.tsqr_mi_chol_f32(2, rs[i & 1].data_ptr(), n, None, m, n, b.dwq.data_ptr(), None, H.cuda_stream) == 0
        ec[i].record(H)
    G(0); C(0)
    for i in range(k):
        if i + 1 < k:
            G(i + 1); C(i + 1)
        S.wait_event(ec[i])
        apply(i, S); ea[i].record(S)


for name in ("one stream", "two streams", "Cholesky on a second stream, one CU left free by the passes", "one stream, one CU left free"):
    S, H = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name.endswith("left free by the passes") or name.endswith("one CU left free"):
            L.tsqr_mi_set_tuning2(2040, 4080)      # 510 Gram workgroups (two per CU on 255 CUs), 1020 apply workgroups (four per CU on 255)
        if name.startswith("one stream"):
            one_stream(calls)
        elif name == "two streams":
            two_streams(calls, S, H)
        else:
            free_cu_streams(calls, S, H)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-60s %d calls of %d x %d %s: %.2f us per call (host enqueue %.2f us per call)" % (name, calls, m, n, mode.name, dt / calls * 1e6, t_host / calls * 1e6))
# sanity: Q^T Q = I
qq = q.double() @ q.double().T
print("orth %.3e" % float((qq - torch.eye(n, device="cuda", dtype=torch.float64)).norm()))
