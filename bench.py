#!/usr/bin/env python3
"""bench.py -- headline benchmark of the tall-skinny QR hot path (BASELINE.json: TSQR GFLOP/s and ||Q^T Q - I||_F,
M = 2^20 x N = 64 per GPU, fp32_tc_cor).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one blocking mtk::qr::qr call (C ABI tsqr_mi_qr_f32, or the row-partitioned driver for N > 1) on a
synthetic U(-1,1) matrix already resident in HBM.  Weak scaling: every rank owns 2^20 rows (N = 8 is BASELINE's C4,
2^23 x 64).  For n <= 64 the engine does not modify A, so no restore is needed between steps.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)       # reference protocol: 1 warm-up + C = 16 calls (src/test.cu:289-309); a few more
    # untimed calls let the clocks settle (0.193 vs 0.20 ms per call)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--m", type=int, default=1 << 20, help="rows per GPU")
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--mode", default="fp32_tc_cor", choices=["fp32_tc_cor", "fp32_notc", "fp32_tc_nocor"])
    ap.add_argument("--reorth", type=int, default=0)
    ap.add_argument("--cpu-sample-rows", type=int, default=1 << 20)   # the whole headline matrix: ~15 s of CPU work on 16 host threads
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steady-after", type=int, default=300, help="untimed calls before the extra steady-state timing leg (0 = skip it)")
    ap.add_argument("--gram-waves", type=int, default=0)
    ap.add_argument("--apply-waves", type=int, default=0)
    ap.add_argument("--policy", type=int, default=0)
    ap.add_argument("--force-dist", action="store_true", help="use the row-partitioned driver even on one GPU")
    ap.add_argument("--dist-comm", default="auto", choices=["auto", "rccl", "callbacks"],
                    help="transport of the row-partitioned driver's two exchanges: raw RCCL communicator (C calls ncclAllReduce itself, "
                         "no Python between kernels) or torch.distributed callbacks; auto = rccl when every rank can create it")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only to rehearse the multi-rank logic with several ranks on ONE GPU")
    ap.add_argument("--ld-pad", type=int, default=0, help="leading dimension = m + pad (experiments on DRAM channel mapping)")
    args = ap.parse_args()

    from tsqr_gpu_amd import blockqr as bq
    bq.lib()                                                # fail loudly if the HIP library is missing
    bq.lib().tsqr_mi_set_tuning2(args.gram_waves, args.apply_waves)
    bq.set_policy(args.policy)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.backend == "gloo":                               # rehearsal: ranks may share a device
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local_rank)
    if world > 1:                                           # communicator set-up (lazy in RCCL) must never land in the timed region
        warm = torch.zeros(2560, dtype=torch.float64, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
    m, n = args.m, args.n
    m_glob = m * world
    mode = bq.compute_mode[args.mode]

    ld = m + args.ld_pad
    d_a = synth_block(m, n, m_glob, rank * m, 0, dev)
    d_q = torch.empty(n, ld, dtype=torch.float32, device=dev)[:, :m]
    if args.ld_pad:
        a_pad = torch.zeros(n, ld, dtype=torch.float32, device=dev)
        a_pad[:, :m] = d_a
        d_a = a_pad[:, :m]
    d_r = torch.zeros(n, n, dtype=torch.float32, device=dev)
    eng = None
    if world == 1 and not args.force_dist:
        bf = bq.buffer(mode, bool(args.reorth), device=dev)
        bf.allocate(m, n)

        call = bq.bind(d_q, ld, d_r, n, d_a, ld, m, n, bf)   # arguments marshalled once, as in a C++ caller's loop

        def step():
            st = call()
            assert st == 0, st
    else:
        from tsqr_gpu_amd import dist as tdist
        eng = tdist.RowPartitionedQR(mode, m, n, comm=args.dist_comm)    # one C call per step; RCCL called from C on this stream

        dcall = eng.bind(d_q, ld, d_r, d_a, ld, reorthogonalize=bool(args.reorth))   # arguments marshalled once here too

        def step():
            st = dcall()
            assert st == 0, st                              # (blocking like the single-GPU call: complete on return)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    flops = f_qr(m_glob, n)
    gflops = flops / (ms_per_step * 1e-3) / 1e9

    # accuracy of the last step, evaluated on the device in fp64 (global Q: all-reduce of Q_p^T Q_p)
    q64 = d_q.double()
    gram = q64 @ q64.T
    r64 = d_r.double().T.contiguous()
    res_num = ((r64.T @ q64) - d_a.double()).pow(2).sum().reshape(1)
    res_den = d_a.double().pow(2).sum().reshape(1)
    if world > 1:
        dist.all_reduce(gram); dist.all_reduce(res_num); dist.all_reduce(res_den)
    orth_fro = float((gram - torch.eye(n, device=dev, dtype=torch.float64)).norm().item())
    residual = float(torch.sqrt(res_num / res_den).item())
    del q64

    # per-kernel-class timing with HIP events on the engine's stream, same number of steps
    prof = None
    bq.profile_enable(True)
    for _ in range(args.steps):
        step()
    barrier()
    prof = bq.profile_read()
    bq.profile_enable(False)

    # Steady state, reported BESIDE the contract's number (never instead of it): the first ~30 calls of a process run 5-8 % slower
    # than the rest (clock / power settling: tools/ramp.py shows 177-180 us for calls 10..30 and 165 us from call ~30 on at the
    # headline size), and a run with 5 warm-up + 20 timed steps sits exactly there.  The same K steps are timed once more after
    # `--steady-after` further untimed calls (the same count on every rank: the steps are collective).
    steady_ms = None
    if args.steady_after > 0:
        for _ in range(args.steady_after):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dts = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dts], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = float(tt.item())
        steady_ms = dts / args.steps * 1e3

    if rank == 0:
        dom = max(prof, key=lambda k: prof[k][0])
        dom_ms, dom_launches = prof[dom]
        per_launch_s = dom_ms * 1e-3 / max(dom_launches, 1)
        # algorithmic work of one launch of the dominant kernel
        if dom == "apply":                                  # Q = A * inverse(R): read A, write Q
            bound, alg_bytes, alg_flops = "hbm", 8.0 * m * n, 2.0 * m * n * n
        elif dom == "gram":                                 # G = A^T A: read A once
            bound, alg_bytes, alg_flops = "hbm", 4.0 * m * n, 2.0 * m * n * n
        else:                                               # Householder fold: R factor of the local block
            bound, alg_bytes, alg_flops = "mfma", 4.0 * m * n, f_r(m, n)
        if bound == "hbm":
            ach, peak, unit = alg_bytes / per_launch_s / 1e9, PEAK_HBM_TBS * 1e3, "GB/s"
        else:
            ach, peak, unit = alg_flops / per_launch_s / 1e12, PEAK_F32_MATRIX_TFLOPS, "TFLOP/s"
        traffic, traffic_source = pmc_traffic(dom, m, n, args.mode)
        engine_name = bq.ENGINE_NAMES.get(eng.last_engine if eng is not None else bq.last_engine(), "?")
        roofline = {"kernel": dom, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                    "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_source,
                    "note": "HBM fractions are fabric-side: the same A is factored every step (the reference's own protocol, "
                            "src/test.cu:299-309) and stays largely resident in the 256 MiB Infinity Cache between the two passes "
                            "and between steps; FETCH_SIZE counts those hits as memory-side requests",
                    "avg_launch_us": per_launch_s * 1e6, "launches": dom_launches,
                    "algorithmic_flops_per_launch": alg_flops, "algorithmic_bytes_per_launch": alg_bytes,
                    "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items() if v[1]},
                    "r_factor_engine": engine_name,
                    "whole_path": {"tflops": gflops / 1e3 / world, "peak_tflops_f32_matrix": PEAK_F32_MATRIX_TFLOPS,
                                   "frac_f32_matrix_peak": gflops / 1e3 / world / PEAK_F32_MATRIX_TFLOPS,
                                   "algorithmic_gbs": 4.0 * (2 * m * n + n * n) / (ms_per_step * 1e-3) / 1e9,
                                   "frac_hbm_peak": 4.0 * (2 * m * n + n * n) / (ms_per_step * 1e-3) / 1e12 / PEAK_HBM_TBS}}
        out = {"metric": "tsqr_gflops", "value": gflops, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": "M=2^%d x N=%d per GPU, %s, reorth=%d, U(-1,1); global %d x %d; F_QR=4MN^2-4/3N^3; R-factor engine: %s" % (
                   int(np.log2(m)) if m & (m - 1) == 0 else -1, n, args.mode, args.reorth, m_glob, n, engine_name),
                   "engine": engine_name, "m_per_gpu": m, "n": n, "mode": args.mode, "reorthogonalize": bool(args.reorth),
                   "parallelism": "row-partitioned x%d" % world,
                   "dist_transport": (eng.transport if eng is not None else None)},
               "orth_fro": orth_fro, "orth_ref_metric": orth_fro / np.sqrt(n), "residual": residual,
               "roofline": roofline}
        if steady_ms is not None:
            out["steady_state"] = {"ms_per_step": steady_ms, "value": flops / (steady_ms * 1e-3) / 1e9, "unit": "GFLOP/s",
                                   "after_untimed_calls": args.warmup + 2 * args.steps + 8 + args.steady_after,
                                   "note": "same K steps timed again later in the process; `value` above is the contract's number"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, args.mode, args.cpu_sample_rows)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
