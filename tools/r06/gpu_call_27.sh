#!/bin/bash
# round 6, call 27: im2col with 8 patch rows per workgroup: parity tests, isolated timing, the NG = 1 box geometries, bench lines
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_06_x16_kernels.py tests/test_gpu_01_kernels.py -x -q -m gpu -k "im2col or box" > $O/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_sel.log | cut -c1-300
timeout 600 python tools/r06/bench_im2col.py > $O/im2col_rows8.log 2>&1; grep -v amdgpu.ids $O/im2col_rows8.log
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 5 > $O/bench_f16_T10_B32_rates.json 2> $O/bench_f16_T10.err; head -c 200 $O/bench_f16_T10_B32_rates.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --dtype bf16 --sustained-seconds 5 > $O/bench_bf16.json 2> $O/bench_bf16.err; head -c 200 $O/bench_bf16.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --sustained-seconds 5 > $O/bench_f32.json 2> $O/bench_f32.err; head -c 200 $O/bench_f32.json; echo
