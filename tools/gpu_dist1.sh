#!/bin/bash
# the row-partitioned driver on ONE rank: fixed cost of the dist path (raw RCCL communicator vs torch.distributed callbacks vs direct call)
cd $GRAFT_REPO_ROOT
python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py direct
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --no-cpu-baseline --force-dist --dist-comm rccl 2>gpurun_out/dist1_rccl.err | python tools/bench_line.py dist_rccl_1rank
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 30 --no-cpu-baseline --force-dist --dist-comm callbacks 2>gpurun_out/dist1_cb.err | python tools/bench_line.py dist_callbacks_1rank
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --backend gloo --dist-comm callbacks 2>gpurun_out/dist2_gloo.err | python tools/bench_line.py dist_gloo_2ranks_one_gpu
tail -3 gpurun_out/dist1_rccl.err
