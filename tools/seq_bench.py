"""What the two streaming passes of one call can reach when they alternate over the same A (skeleton kernels, no arithmetic):
sequences of {load-only / copy} passes with forward / backward block order, each pass timed with HIP events."""
import ctypes, sys, torch
L = ctypes.CDLL('tsqr_gpu_amd/csrc/libtsqr_selftest.so')
n = 64
lm = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = 1 << lm
a = torch.rand(n, m, device='cuda'); q = torch.zeros(n, m, device='cuda')
NAMES = {0: 'copy', 1: 'load', 2: 'store', 3: 'load(c,q)'}
def run(seq, nwg=768, reps=10):
    k = len(seq)
    modes = (ctypes.c_int * k)(*[s[0] for s in seq]); dirs = (ctypes.c_int * k)(*[s[1] for s in seq]); ntl = (ctypes.c_int * k)(*[s[2] for s in seq])
    out = (ctypes.c_float * k)()
    rc = L.tsqr_selftest_seq(ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(a.data_ptr()), ctypes.c_size_t(m), ctypes.c_size_t(m), nwg, k, modes, dirs, ntl, reps, out)
    assert rc == 0, rc
    desc = ' | '.join('%s %s%s %6.1f us' % (NAMES[s[0]], 'bwd' if s[1] else 'fwd', ' ntl' if s[2] else '', out[i]) for i, s in enumerate(seq))
    print('nwg %4d: %s  || sum %.1f' % (nwg, desc, sum(out)), flush=True)
NAMES[4] = 'idle'
for nwg in (768,):
    run([(1, 0, 0), (0, 1, 0)], nwg)
    run([(1, 0, 0), (4, 25, 0), (0, 1, 0)], nwg)     # 25 us of a one-wave kernel between the passes (what the Cholesky step is to the chip)
    run([(1, 0, 0), (4, 10, 0), (0, 1, 0)], nwg)
    run([(1, 0, 0), (4, 5, 0), (0, 1, 0)], nwg)
    run([(4, 25, 0), (1, 0, 0), (4, 25, 0), (0, 1, 0)], nwg)
    run([(3, 0, 0), (4, 25, 0), (0, 1, 0)], nwg)
