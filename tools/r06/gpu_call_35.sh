#!/bin/bash
# round 6, call 35: adjoint row prefetch (16-bit input) + fp16 operands in the fp16 mode's g_P forms: kernel / parity suites of the 16-bit modes, bench lines
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c35; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_01_kernels.py tests/test_gpu_04_x16_parity.py tests/test_gpu_06_x16_kernels.py -x -q -m gpu -k "upconv or x16 or parity" > $O/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_sel.log | cut -c1-300
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 5 > $O/bench_f16_T10_B32_rates.json 2> $O/bench_f16_T10.err; head -c 200 $O/bench_f16_T10_B32_rates.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --dtype bf16 --sustained-seconds 5 > $O/bench_bf16.json 2> $O/bench_bf16.err; head -c 200 $O/bench_bf16.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --dtype f16 --sustained-seconds 5 > $O/bench_f16.json 2> $O/bench_f16.err; head -c 200 $O/bench_f16.json; echo
python - <<'PY'
import json
for n in ('bench_f16_T10_B32_rates','bench_bf16','bench_f16'):
    d=json.loads(open('gpurun_out/r06/c35/%s.json' % n).read().strip().splitlines()[-1]); print(n, d['value'], d['ms_per_step'], d['upconv_by_stage_ms_per_step'])
PY
