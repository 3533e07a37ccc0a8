#!/bin/bash
# round 5, call 4: config 5's share with deconv1 on the box kernels at NB = 320 and the step-wise low-rank backward; GEMM-algorithm record for the 16-bit modes' shapes
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
python bench.py --no-cpu-baseline --steps 6 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 2> gpurun_out/r05/bench_f16_T10_B32_own_2.err | tee gpurun_out/r05/bench_f16_T10_B32_own_2.json | cut -c1-220
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype bf16 2> gpurun_out/r05/bench_bf16_own_2.err | tee gpurun_out/r05/bench_bf16_own_2.json | cut -c1-220
bash tools/tune_gemms.sh 2>&1 | tail -5
cp gpurun_out/tunableop_results.csv gpurun_out/r05/tunableop_results_r05.csv
cp gpurun_out/tunableop_results.csv stereospike_amd/tunableop/tunableop_results.csv
python bench.py --no-cpu-baseline --steps 6 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 2> gpurun_out/r05/bench_f16_T10_B32_own_3.err | tee gpurun_out/r05/bench_f16_T10_B32_own_3.json | cut -c1-220
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype bf16 2> gpurun_out/r05/bench_bf16_own_3.err | tee gpurun_out/r05/bench_bf16_own_3.json | cut -c1-220
python bench.py --no-cpu-baseline --steps 10 --warmup 3 2> gpurun_out/r05/bench_f32_3.err | tee gpurun_out/r05/bench_f32_3.json | cut -c1-220
