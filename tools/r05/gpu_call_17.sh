#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/c17; mkdir -p $O
python bench.py --no-cpu-baseline > $O/bench_f32.json 2>/dev/null; head -c 220 $O/bench_f32.json; echo
python bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2>/dev/null; head -c 220 $O/bench_bf16.json; echo
python bench.py --no-cpu-baseline --steps 6 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 > $O/bench_f16_T10_B32_rates.json 2>/dev/null; head -c 220 $O/bench_f16_T10_B32_rates.json; echo
timeout 2400 python -m pytest tests/test_gpu_00_default_path.py tests/test_gpu_01_kernels.py tests/test_gpu_04_x16_parity.py tests/test_gpu_06_x16_kernels.py -q -m gpu -x 2>&1 | tail -5 | cut -c1-300
