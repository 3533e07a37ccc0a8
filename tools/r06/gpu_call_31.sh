#!/bin/bash
# round 6, call 31: prediction heads: union-scan adjoint + frames-per-thread gather: parity (C oracle, bit for bit), timing per head (A/B of the gather), bench lines
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c31; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "upconv" > $O/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_sel.log | cut -c1-300
SS_HEAD_GATHER_OLD=1 timeout 600 python tools/r06/bench_heads.py > $O/heads_gather_per_frame.log 2>&1; grep -v amdgpu.ids $O/heads_gather_per_frame.log
timeout 600 python tools/r06/bench_heads.py > $O/heads_after.log 2>&1; grep -v amdgpu.ids $O/heads_after.log
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 5 > $O/bench_f16_T10_B32_rates.json 2> $O/bench_f16_T10.err; head -c 200 $O/bench_f16_T10_B32_rates.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --dtype bf16 --sustained-seconds 5 > $O/bench_bf16.json 2> $O/bench_bf16.err; head -c 200 $O/bench_bf16.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --sustained-seconds 5 > $O/bench_f32.json 2> $O/bench_f32.err; head -c 200 $O/bench_f32.json; echo
