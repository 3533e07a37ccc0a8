#!/bin/bash
# round 5, call 2: the 16-bit modes on own kernels — kernel tests, end-to-end parity, first bench lines
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_06_x16_kernels.py -x -q -m gpu > gpurun_out/r05/pytest_x16_kernels_1.log 2>&1
tail -25 gpurun_out/r05/pytest_x16_kernels_1.log
timeout 1200 python -m pytest tests/test_gpu_04_x16_parity.py -q -m gpu > gpurun_out/r05/pytest_x16_parity_1.log 2>&1
tail -40 gpurun_out/r05/pytest_x16_parity_1.log
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype bf16 > gpurun_out/r05/bench_bf16_own_1.json 2> gpurun_out/r05/bench_bf16_own_1.err
tail -c 400 gpurun_out/r05/bench_bf16_own_1.err; head -c 300 gpurun_out/r05/bench_bf16_own_1.json
