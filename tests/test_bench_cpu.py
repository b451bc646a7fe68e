"""CPU checks of bench.py's launcher and of the row-partitioned driver's argument validation (no GPU).

  * `python bench.py --gpus 2 --backend gloo --rehearse` with NO launcher in front must start its own two ranks
    (torch.distributed.run as a child process, before anything touches a GPU), run the W + K protocol with the numpy test double
    of tests/dist_double.py as the per-rank executor, and print exactly one JSON line -- the shape the driver's SCALE runs parse.
  * RowPartitionedQR refuses an empty row block on EVERY rank together (a rank that bailed out alone would leave the others
    waiting in the first exchange)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):   # a bare invocation: no launcher variables
        env.pop(k, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bare_multi_gpu_invocation_starts_its_own_ranks():
    d = _bench("--gpus", "2", "--backend", "gloo", "--rehearse", "--m", "4096", "--steps", "3", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["rehearsal"] is True
    assert d["metric"] == "tsqr_gflops" and d["unit"] == "GFLOP/s" and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["parallelism"] == "row-partitioned x2" and d["config"]["m_per_gpu"] == 4096 and d["config"]["n"] == 64
    assert "c2" in d["config"]["workload"] and "8192 x 64" in d["config"]["workload"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert d["orth_fro"] < 1e-5 and d["residual"] < 1e-6        # the double's LAPACK arithmetic on the global 8192 x 64 matrix


def test_single_rank_rehearsal_and_workload_names():
    d = _bench("--gpus", "1", "--backend", "gloo", "--rehearse", "--m", "2048", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 1 and d["rehearsal"] is True and d["config"]["parallelism"] == "row-partitioned x1"
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--workload", "c3"])
    assert (a.m, a.n, a.mode, a.reorth) == (1 << 20, 128, "fp32_tc_cor", 0)
    a = bench.parse_args(["--workload", "c5", "--mode", "fp32_notc"])
    assert (a.m, a.n, a.mode, a.reorth, a.input) == (1 << 20, 64, "fp32_notc", 1, "latms_cond1e8")
    a = bench.parse_args([])
    assert (a.workload, a.m, a.n, a.mode, a.reorth, a.gpus) == ("c2", 1 << 20, 64, "fp32_tc_cor", 0, 1)


def test_world_size_8_rehearsal_weak_and_strong():
    """the N = 8 launch (BASELINE's C4 shape of ranks) rehearsed on CPU -- eight gloo ranks, the numpy double as the
    per-rank executor -- in both scaling modes.  weak: --m rows per rank; strong: --m is the GLOBAL row count, split over the ranks."""
    d = _bench("--gpus", "8", "--backend", "gloo", "--rehearse", "--m", "1024", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["m_per_gpu"] == 1024 and "8192 x 64" in d["config"]["workload"]
    assert d["config"]["parallelism"] == "row-partitioned x8" and d["config"]["dist_exchange"].startswith("gram_allreduce")
    assert d["orth_fro"] < 1e-5 and d["residual"] < 1e-6
    d = _bench("--gpus", "8", "--backend", "gloo", "--rehearse", "--scaling", "strong", "--m", "8192", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["m_per_gpu"] == 1024 and "8192 x 64" in d["config"]["workload"]
    assert d["orth_fro"] < 1e-5 and d["residual"] < 1e-6
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--scaling", "strong", "--gpus", "4"])
    assert a.m == (1 << 23) // 4                                 # default global matrix of the strong mode: 2^23 x 64
    a = bench.parse_args(["--scaling", "strong", "--gpus", "1"])
    assert a.m == 1 << 23


def test_launcher_failure_is_reported():
    """a rank that fails makes the bare invocation exit non-zero (the children's status is forwarded)"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl", "--rehearse", "--m", "256"],
                         capture_output=True, text=True, timeout=600, env=env)      # --rehearse refuses the nccl backend
    assert out.returncode != 0


def _empty_block_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_double import NumpyRowBackend
    from tsqr_gpu_amd import dist as tdist
    n = 8
    raised = []
    try:
        tdist.RowPartitionedQR(3, 0 if rank == 1 else 100, n, backend=NumpyRowBackend(n, tdist.TorchCollectives()))
        raised.append(False)
    except ValueError:
        raised.append(True)
    drv = tdist.RowPartitionedQR(3, 100, n, backend=NumpyRowBackend(n, tdist.TorchCollectives()))
    a = torch.from_numpy(np.random.default_rng(rank).uniform(-1, 1, (n, 100)).astype(np.float32))
    q = torch.zeros(n, 100); r = torch.zeros(n, n)
    try:                                                        # per-call override: rank 0 passes an empty block
        drv.qr(q, 100, r, a, 100, m_local=(0 if rank == 0 else 50))
        raised.append(False)
    except ValueError:
        raised.append(True)
    st = drv.qr(q, 100, r, a, 100)                              # and the engine is still usable afterwards, on both ranks
    out.put((rank, raised, st))
    dist.destroy_process_group()


def test_empty_row_block_raises_on_every_rank():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_empty_block_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        got = sorted(out.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    assert got == [(0, [True, True], 0), (1, [True, True], 0)]
