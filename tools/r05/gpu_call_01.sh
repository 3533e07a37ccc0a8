#!/bin/bash
# round 5, call 1: where the 16-bit modes spend their step today (MIOpen under autocast), and the start-of-round lines
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r05/bench_f32_start.json 2> gpurun_out/r05/bench_f32_start.err
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype bf16 > gpurun_out/r05/bench_bf16_start.json 2> gpurun_out/r05/bench_bf16_start.err
python bench.py --no-cpu-baseline --steps 6 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 > gpurun_out/r05/bench_f16_T10_B32_start.json 2> gpurun_out/r05/bench_f16_T10_B32_start.err
bash profiles/run_profile.sh r05_bf16_start --steps 6 --warmup 2 --dtype bf16
bash profiles/run_profile.sh r05_f16_T10_start --steps 4 --warmup 2 --dtype f16 --T 10 --batch 32 --count-rates 1
tail -c 600 gpurun_out/r05/bench_bf16_start.json
