#!/bin/bash
# round 5, call 5: one box plane in the bf16 mode as well — kernel tests, parity, bench; the full-size config 2 / 5 tests incl. the restored GraphedTrainer assertion
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_06_x16_kernels.py -x -q -m gpu 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_04_x16_parity.py -q -m gpu 2>&1 | tail -6
cp gpurun_out/parity_report_x16.json gpurun_out/r05/parity_report_x16_own_2.json
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype bf16 2> gpurun_out/r05/bench_bf16_own_4.err | tee gpurun_out/r05/bench_bf16_own_4.json | cut -c1-200
python bench.py --no-cpu-baseline --steps 6 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 2> gpurun_out/r05/bench_f16_T10_B32_own_4.err | tee gpurun_out/r05/bench_f16_T10_B32_own_4.json | cut -c1-200
timeout 1500 python -m pytest tests/test_gpu_05_full_size.py -q -m gpu -k "config5 or config2" 2>&1 | tail -15
