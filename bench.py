#!/usr/bin/env python3
"""bench.py -- headline benchmark of the tall-skinny QR hot path (BASELINE.json: TSQR GFLOP/s and ||Q^T Q - I||_F,
M = 2^20 x N = 64 per GPU, fp32_tc_cor).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A bare `python bench.py --gpus N` (N > 1, no launcher in front: WORLD_SIZE unset) starts its own N ranks: before anything touches
the GPU it runs the torch.distributed.run command above as a child process, forwards its output and exits with its status.

A "step" is one blocking mtk::qr::qr call (C ABI tsqr_mi_qr_f32, or the row-partitioned driver for N > 1) on a
synthetic matrix already resident in HBM.  Weak scaling: every rank owns 2^20 rows (N = 8 is BASELINE's C4,
2^23 x 64).  For n <= 64 the engine does not modify A, so no restore is needed between steps.
--workload selects the BASELINE.json configuration: c2 (default, the headline: 2^20 x 64 fp32_tc_cor U(-1,1)), c3 (2^20 x 128,
--mode fp32_tc_cor | fp32_notc), c5 (latms cond 1e8, 2^20 x 64, Reorthogonalize = true); c4 is `--gpus 8` of c2's per-GPU shape.
Prints ONE JSON line on rank 0 (contract in the task prompt) including `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def usable_cpus():
    """Host threads the CPU baseline may use: affinity mask, cgroup CPU quota, and at most 16 (the CPU share of a one-GPU
    box on this pool -- os.cpu_count() reports the whole 256-thread host there, and oversubscribing slows OpenMP down)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


CPU_THREADS = usable_cpus()
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):   # before numpy / the oracle's libgomp load
    os.environ.setdefault(_v, str(CPU_THREADS))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3     # MI355X fp32 MFMA/vector peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_TBS = 8.0                 # HBM3E spec


def f_qr(m, n):
    """Algorithmic flops of geqrf + explicit thin Q: 4MN^2 - 4/3 N^3 (SURVEY.md 8d / BASELINE.md section 2)."""
    return 4.0 * m * n * n - 4.0 / 3.0 * n ** 3


def f_r(m, n):
    return 2.0 * m * n * n - 2.0 / 3.0 * n ** 3


def synth_block(m_local, n, m_global, row0, seed, device):
    """U(-1,1) entries keyed by (seed, global row, column) with a splitmix64-style integer hash, so that 1/2/4/8-GPU
    runs see the identical global matrix.  Returned as an (n, m_local) tensor = column-major m_local x n."""
    rows = torch.arange(row0, row0 + m_local, device=device, dtype=torch.int64)
    out = torch.empty(n, m_local, device=device, dtype=torch.float32)
    for j in range(n):
        x = rows + (j * m_global + seed * 1000003 + 0x632BE5AB)     # int64 arithmetic wraps; only the bit mixing matters
        x = (x ^ (x >> 30)) * 0x1CE4E5B9
        x = (x ^ (x >> 27)) * 0x133111EB
        x = x ^ (x >> 31)
        u = (x & ((1 << 24) - 1)).to(torch.float32) * (1.0 / (1 << 24))
        out[j] = u * 2.0 - 1.0
    return out


def cpu_baseline(n, mode_name, sample_rows):
    """The reported CPU baseline (north star): host LAPACK sgeqrf + sorgqr (explicit thin Q, F_QR flops -- like for like) on the box's
    host cores, on a bounded sample of the same workload (sample_rows x n, U(-1,1)); sgeqrf alone (F_R) and the CPU oracle
    (restatement of the reference algorithm, oracle/ref_tsqr.c, OpenMP over leaves) are nested extras."""
    from oracle import ref_oracle as ro
    a = ro.uniform_matrix(sample_rows, n, seed=0)
    cores = int(os.environ.get("OMP_NUM_THREADS", CPU_THREADS))
    out = None
    try:
        from scipy.linalg import lapack
        af = np.asfortranarray(a)
        lapack.sgeqrf(np.asfortranarray(a[:4096]))
        t = time.time(); qr_, tau, _, info = lapack.sgeqrf(af); t_geqrf = time.time() - t
        t = time.time(); qq, _, info = lapack.sorgqr(qr_[:, :n], tau); t_orgqr = time.time() - t
        out = {"value": f_qr(sample_rows, n) / (t_geqrf + t_orgqr) / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "lapack",
               "sample": "scipy.linalg.lapack (OpenBLAS) sgeqrf + sorgqr on %d x %d U(-1,1), %.2f s on %d threads; F_QR = 4MN^2 - 4/3 N^3" % (
                   sample_rows, n, t_geqrf + t_orgqr, cores),
               "sgeqrf_only_gflops": f_r(sample_rows, n) / t_geqrf / 1e9, "orth_fro": ro.orthogonality_fro(qq)}
    except Exception as e:  # no LAPACK on the box: the oracle port becomes the baseline
        out = None
        lapack_error = str(e)
    md = ro.FP32_NOTC if mode_name == "fp32_notc" else ro.FP32_TC_COR   # the oracle models the two north-star modes
    ro.qr(a[:4096], md, False)
    t = time.time()
    st, q, r = ro.qr(a, md, False)
    dt = time.time() - t
    port = {"value": f_qr(sample_rows, n) / dt / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": "oracle/ref_tsqr.c (%s, reference algorithm, OpenMP) on %d x %d U(-1,1), %.2f s" % (mode_name, sample_rows, n, dt),
            "orth_fro": ro.orthogonality_fro(q), "residual": ro.residual(a, q, r)}
    if out is None:
        port["lapack_error"] = lapack_error
        return port
    out["oracle_port"] = port
    return out


def pmc_traffic(kernel_class, m, n, mode):
    """HBM bytes per launch of the dominant kernel.  NOT measured in this run: replayed from the newest committed rocprofv3 PMC
    summary (profiles/rNN_pmc_hbm_traffic.json: FETCH_SIZE x2 for gfx950, WRITE_SIZE exact, separate --pmc passes), valid only
    for the workload those passes were taken on (2^20 x 64, fp32_tc_cor).  Returns (bytes or None, source label)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not files or m != 1 << 20 or n != 64 or mode != "fp32_tc_cor":
        return None, None
    try:
        data = json.load(open(files[-1]))
        for k in data["kernels"].values():
            if k.get("class") == kernel_class:
                return k["hbm_bytes"], "static: %s (rocprofv3 --pmc passes of an earlier run, not this one)" % os.path.relpath(files[-1], ROOT)
    except Exception:
        pass
    return None, None


WORKLOADS = {
    # BASELINE.json configs[1], [2], [4] ([3] = `--gpus 8` of c2's per-GPU shape; [0] is the CPU-runnable LAPACK case, tests/test_gpu_configs.py)
    "c2": dict(m=1 << 20, n=64, mode="fp32_tc_cor", reorth=0, input="uniform", cpu_rows=1 << 20),
    "c3": dict(m=1 << 20, n=128, mode="fp32_tc_cor", reorth=0, input="uniform", cpu_rows=1 << 19),
    "c5": dict(m=1 << 20, n=64, mode="fp32_tc_cor", reorth=1, input="latms_cond1e8", cpu_rows=1 << 20),
    # not a BASELINE configuration: c2's shape through the half-typed modes (io type half, reference src/tsqr.hpp:38-39)
    "c2h": dict(m=1 << 20, n=64, mode="fp16_tc_nocor", reorth=0, input="uniform", cpu_rows=1 << 20),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)       # reference protocol: 1 warm-up + C = 16 calls (src/test.cu:289-309)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS), help="BASELINE.json configuration (see the module docstring)")
    ap.add_argument("--m", type=int, default=None, help="rows per GPU (default: the workload's)")
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--mode", default=None, choices=["fp32_tc_cor", "fp32_notc", "fp32_tc_nocor", "fp16_notc", "fp16_tc_nocor"])
    ap.add_argument("--reorth", type=int, default=None)
    ap.add_argument("--cpu-sample-rows", type=int, default=None)   # c2: the whole headline matrix, ~15 s of CPU work on 16 host threads
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-first-window", action="store_true", help="skip the extra timing window at process start (profiling runs)")
    ap.add_argument("--gram-waves", type=int, default=0)
    ap.add_argument("--apply-waves", type=int, default=0)
    ap.add_argument("--policy", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true", help="use the row-partitioned driver even on one GPU")
    ap.add_argument("--dist-comm", default="auto", choices=["auto", "rccl", "callbacks"],
                    help="transport of the row-partitioned driver's two exchanges: raw RCCL communicator (C calls ncclAllReduce itself, "
                         "no Python between kernels) or torch.distributed callbacks; auto = rccl when every rank can create it")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only to rehearse the multi-rank logic with several ranks on ONE GPU (or none)")
    ap.add_argument("--rehearse", action="store_true",
                    help="NO GPU: the per-rank executor is the numpy test double of tests/dist_double.py.  Exercises the launcher, the "
                         "collectives and the JSON plumbing on CPU; the numbers say nothing about the product and are labelled so")
    ap.add_argument("--ld-pad", type=int, default=0, help="leading dimension = m + pad (experiments on DRAM channel mapping)")
    args = ap.parse_args(argv)
    w = WORKLOADS[args.workload]
    for k in ("m", "n", "mode", "reorth"):
        if getattr(args, k) is None:
            setattr(args, k, w[k])
    if args.cpu_sample_rows is None:
        args.cpu_sample_rows = min(w["cpu_rows"], args.m)
    args.input = w["input"]
    return args


def spawn_own_ranks(argv, gpus):
    """`python bench.py --gpus N` without a launcher: run N ranks of this file under torch.distributed.run as ONE child process
    (this process has not touched the GPU and never will), forward its output, return its exit status."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    # the ranks' own arguments travel in the environment: the launcher's argument parser chokes on options of the script that
    # happen to abbreviate one of its own (--m is "ambiguous" to it even behind the script name)
    env = dict(os.environ, TSQR_BENCH_ARGV=json.dumps(list(argv)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    return subprocess.call(cmd, env=env)


def make_input(args, m, n, m_glob, rank, dev):
    if args.input == "uniform" or args.rehearse:
        return synth_block(m, n, m_glob, rank * m, 0, dev)
    # C5: latms-style A = U diag(s) V^T with the reference's singular-value draw (src/test_cond.cu:31-50), cond 1e8, fixed seed
    from tsqr_gpu_amd import harness
    return harness.get_rand_matrix_with_cond_number(m, n, 1e8, seed=5, device=dev)


def main():
    argv = sys.argv[1:]
    if not argv and os.environ.get("TSQR_BENCH_ARGV"):       # a rank started by spawn_own_ranks
        argv = json.loads(os.environ["TSQR_BENCH_ARGV"])
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:    # bare multi-GPU invocation: become the launcher, before any GPU call
        sys.exit(spawn_own_ranks(argv, args.gpus))

    from tsqr_gpu_amd import blockqr as bq
    if not args.rehearse:
        bq.lib()                                            # fail loudly if the HIP library is missing
        bq.lib().tsqr_mi_set_tuning2(args.gram_waves, args.apply_waves)
        bq.set_policy(args.policy)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus)
    gpu = not args.rehearse
    if gpu:
        assert torch.cuda.is_available(), "bench.py needs a GPU (--rehearse runs the launcher/collective plumbing on a CPU double)"
        if args.backend == "gloo":                           # rehearsal: ranks may share a device
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    else:
        assert args.backend == "gloo", "--rehearse runs on CPU: use --backend gloo"
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local_rank) if gpu else torch.device("cpu")
    if world > 1:                                           # communicator set-up (lazy in RCCL) must never land in the timed region
        warm = torch.zeros(2560, dtype=torch.float64, device=dev)
        dist.all_reduce(warm)
        if gpu:
            torch.cuda.synchronize()
    m, n = args.m, args.n
    m_glob = m * world
    assert world == 1 or n <= 64, "the row-partitioned path factors one 64-wide panel (c3 is a one-GPU workload)"
    assert world == 1 or args.input == "uniform", "c5 is a one-GPU workload"
    mode = bq.compute_mode[args.mode]

    ld = m + args.ld_pad
    io_half = (not args.rehearse) and mode in bq.FP16_MODES   # the half-typed modes: fp16 in, fp16 out (one-GPU workloads)
    assert not io_half or (world == 1 and not args.force_dist), "the fp16 I/O modes are single-GPU entry points"
    io_dt = torch.float16 if io_half else torch.float32
    d_a = make_input(args, m, n, m_glob, rank, dev).to(io_dt)
    d_q = torch.empty(n, ld, dtype=io_dt, device=dev)[:, :m]
    if args.ld_pad:
        a_pad = torch.zeros(n, ld, dtype=io_dt, device=dev)
        a_pad[:, :m] = d_a
        d_a = a_pad[:, :m]
    a_keep = d_a.clone() if n > 64 else None                 # n > 64: the engine may overwrite A (it does not on the one-panel path)
    d_r = torch.zeros(n, n, dtype=io_dt, device=dev)
    eng = None
    if args.rehearse:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from dist_double import NumpyRowBackend
        from tsqr_gpu_amd import dist as tdist
        eng = tdist.RowPartitionedQR(mode, m, n, backend=NumpyRowBackend(n, tdist.TorchCollectives()))

        def run_steps(k):
            for _ in range(k):
                assert eng.qr(d_q, ld, d_r, d_a, ld, reorthogonalize=bool(args.reorth)) == 0
    elif world == 1 and not args.force_dist:
        bf = bq.buffer(mode, bool(args.reorth), device=dev)
        bf.allocate(m, n)
        loop = bq.bind_loop(d_q, ld, d_r, n, d_a, ld, m, n, bf)  # K calls issued by ONE C loop (tsqr_mi_qr_f32_loop): what a C++ caller's
        # loop costs (the reference's speed protocol is such a loop, src/test.cu:299-309), no interpreter time between calls.  The loop
        # keeps the stream fed: call i + 1 is submitted before call i is finished (tsqr_mi_qr_f32_submit / _finish); every call runs
        # all of its kernels and has its verdict looked at.  The same K calls as plain blocking calls are timed below as well.

        def run_steps(k):
            st = loop(k)
            assert st == 0, st
    else:
        from tsqr_gpu_amd import dist as tdist
        eng = tdist.RowPartitionedQR(mode, m, n, comm=args.dist_comm)    # K steps = one C loop (two calls in flight); RCCL called from C on this stream
        dloop = eng.bind_loop(d_q, ld, d_r, d_a, ld, reorthogonalize=bool(args.reorth))

        def run_steps(k):
            st = dloop(k)
            assert st == 0, st                              # (complete on return, like the single-GPU loop)

    def barrier():
        if world > 1:
            dist.barrier()
        if gpu:
            torch.cuda.synchronize()

    def timed_window():
        """The contract's measurement: W untimed warm-up steps, then exactly K steps between barrier + synchronize, MAX over ranks."""
        run_steps(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt / args.steps * 1e3

    # 1. one step, checked: accuracy of the factorisation evaluated in fp64 (global Q: all-reduce of Q_p^T Q_p), A untouched
    run_steps(1)
    barrier()
    q64 = d_q.double()
    gram = q64 @ q64.T
    r64 = d_r.double().T.contiguous()
    a64 = (a_keep if a_keep is not None else d_a).double()
    res_num = ((r64.T @ q64) - a64).pow(2).sum().reshape(1)
    res_den = a64.pow(2).sum().reshape(1)
    if world > 1:
        dist.all_reduce(gram); dist.all_reduce(res_num); dist.all_reduce(res_den)
    orth_fro = float((gram - torch.eye(n, device=dev, dtype=torch.float64)).norm().item())
    residual = float(torch.sqrt(res_num / res_den).item())
    a_untouched = True if a_keep is None else bool(torch.equal(a_keep, d_a))
    if not a_untouched:                                      # panel path for n > 64 (ill-conditioned input): every step needs a fresh A
        raise SystemExit("bench.py: the engine overwrote A on this workload; timing repeated calls on it would not measure the workload")
    del q64, a64

    # 2. a timing window at process start (reported as `first_window`, not as `value`): the first ~30 calls of a process run
    #    5-8 % slower than all later ones (GPU clock / power settling after idle: tools/ramp.py, profiles/r02_experiment_log.md)
    first_ms = None if (args.no_first_window or args.rehearse) else timed_window()

    # 3. per-kernel-class timing with HIP events on the engine's stream, K steps (the roofline leg)
    prof = None
    if gpu:
        bq.profile_enable(True)
        run_steps(args.steps)
        barrier()
        prof = bq.profile_read()
        bq.profile_enable(False)

    # 4. the same window as 5. with the loop entry degraded to plain blocking calls (one host round trip per call): reported, not `value`
    blocking_ms = None
    if gpu and not args.rehearse:                             # (every rank alike: the loop depth is a per-process setting)
        bq.set_loop_depth(1)
        try:
            blocking_ms = timed_window()
        finally:
            bq.set_loop_depth(3)
    # 5. the contract's window
    ms_per_step = timed_window()
    flops = f_qr(m_glob, n)
    gflops = flops / (ms_per_step * 1e-3) / 1e9

    if rank == 0:
        engine_name = "numpy-double (rehearsal)" if args.rehearse else bq.ENGINE_NAMES.get(eng.last_engine if eng is not None else bq.last_engine(), "?")
        roofline = None
        if prof is not None:
            dom = max(prof, key=lambda k: prof[k][0])
            dom_ms, dom_launches = prof[dom]
            per_launch_s = dom_ms * 1e-3 / max(dom_launches, 1)
            # algorithmic work of one launch of the dominant kernel
            esz = 2.0 if io_half else 4.0                       # bytes per element of A and Q
            if dom == "apply":                                  # Q = A * inverse(R): read A, write Q
                bound, alg_bytes, alg_flops = "hbm", 2 * esz * m * n, 2.0 * m * n * n
            elif dom == "gram":                                 # G = A^T A: read A once
                bound, alg_bytes, alg_flops = "hbm", esz * m * n, 2.0 * m * n * n
            else:                                               # Householder fold: R factor of the local block
                bound, alg_bytes, alg_flops = "mfma", 4.0 * m * n, f_r(m, n)
            if bound == "hbm":
                ach, peak, unit = alg_bytes / per_launch_s / 1e9, PEAK_HBM_TBS * 1e3, "GB/s"
            else:
                ach, peak, unit = alg_flops / per_launch_s / 1e12, PEAK_F32_MATRIX_TFLOPS, "TFLOP/s"
            traffic, traffic_source = pmc_traffic(dom, m, n, args.mode) if args.workload == "c2" else (None, None)
            roofline = {"kernel": dom, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                        "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_source,
                        "note": "HBM fractions are fabric-side: the same A is factored every step (the reference's own protocol, "
                                "src/test.cu:299-309) and a 256 MiB A stays largely resident in the 256 MiB Infinity Cache between the two "
                                "passes and between steps; FETCH_SIZE counts those hits as memory-side requests",
                        "avg_launch_us": per_launch_s * 1e6, "launches": dom_launches,
                        "algorithmic_flops_per_launch": alg_flops, "algorithmic_bytes_per_launch": alg_bytes,
                        "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items() if v[1]},
                        "r_factor_engine": engine_name,
                        "whole_path": {"tflops": gflops / 1e3 / world, "peak_tflops_f32_matrix": PEAK_F32_MATRIX_TFLOPS,
                                       "frac_f32_matrix_peak": gflops / 1e3 / world / PEAK_F32_MATRIX_TFLOPS,
                                       "algorithmic_gbs": 4.0 * (2 * m * n + n * n) / (ms_per_step * 1e-3) / 1e9,
                                       "frac_hbm_peak": 4.0 * (2 * m * n + n * n) / (ms_per_step * 1e-3) / 1e12 / PEAK_HBM_TBS}}
        inp = "U(-1,1)" if args.input == "uniform" else "latms cond 1e8 (src/test_cond.cu:31-50 spectrum, seed 5)"
        out = {"metric": "tsqr_gflops", "value": gflops, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": ("f16 in / out, f32 accumulation" if io_half else "f32"),
               "data": "synthetic",
               "config": {"workload": "%s: M=2^%d x N=%d per GPU, %s, reorth=%d, %s; global %d x %d; F_QR=4MN^2-4/3N^3; R-factor engine: %s" % (
                   args.workload, int(np.log2(m)) if m & (m - 1) == 0 else -1, n, args.mode, args.reorth, inp, m_glob, n, engine_name),
                   "engine": engine_name, "m_per_gpu": m, "n": n, "mode": args.mode, "reorthogonalize": bool(args.reorth),
                   "parallelism": "row-partitioned x%d" % world,
                   "dist_transport": (eng.transport if eng is not None else None)},
               "orth_fro": orth_fro, "orth_ref_metric": orth_fro / np.sqrt(n), "residual": residual,
               "window_order": "1 checked step, first_window (W + K), K steps under HIP events, blocking_calls (W + K), then the W + K window `value` is taken from",
               "roofline": roofline}
        if args.rehearse:
            out["rehearsal"] = True
            out["data"] = "synthetic (CPU rehearsal with the numpy test double: NOT a measurement of the product)"
        if blocking_ms is not None:
            out["call_protocol"] = ("the K calls of the window are ONE call of the C loop entry (tsqr_mi_qr_f32_loop / _dist_*_loop), issued as a stream: call i + 1 "
                                    "is submitted before call i is finished (two in flight); for full 64-column matrices of <= 2^20 rows on one GPU the "
                                    "R-factor chain of call i (reduction, Cholesky, verdict) runs inside the Gram launch of call i + 1.  Every call runs "
                                    "all of its kernels, every conditioning verdict is read, Q and R are bit for bit those of the blocking call; "
                                    "`blocking_calls` is the same window with one blocking call after the other (tsqr_mi_set_loop_depth(1))")
            out["blocking_calls"] = {"ms_per_step": blocking_ms, "value": flops / (blocking_ms * 1e-3) / 1e9, "unit": "GFLOP/s"}
        if first_ms is not None:
            out["first_window"] = {"ms_per_step": first_ms, "value": flops / (first_ms * 1e-3) / 1e9, "unit": "GFLOP/s",
                                   "note": "the same W + K window taken at process start, inside the GPU's clock / power transient after idle"}
        if world == 1 and not args.no_cpu_baseline and not args.rehearse and not io_half:
            out["cpu_baseline"] = cpu_baseline(n, args.mode, args.cpu_sample_rows)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
