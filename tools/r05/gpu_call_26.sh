#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05/c26
timeout 600 python bench.py --no-cpu-baseline --force-dp --steps 5 --warmup 2 2> gpurun_out/r05/c26/force_dp.err | head -c 300; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r05/c26/torchrun1.err | head -c 300; echo
tail -3 gpurun_out/r05/c26/force_dp.err | cut -c1-200
