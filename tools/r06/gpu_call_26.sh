#!/bin/bash
# round 6, call 26: after the chunk-group box kernels: x16 suites + the two 16-bit bench lines
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c26; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_04_x16_parity.py tests/test_gpu_06_x16_kernels.py tests/test_gpu_05_x16_own.py -x -q -m gpu > $O/pytest_x16.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_x16.log | cut -c1-300
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 5 > $O/bench_f16_T10_B32_rates.json 2> $O/bench_f16_T10.err; head -c 200 $O/bench_f16_T10_B32_rates.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --dtype bf16 --sustained-seconds 5 > $O/bench_bf16.json 2> $O/bench_bf16.err; head -c 200 $O/bench_bf16.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --dtype bf16 --model PLIFNetMono --T 1 --batch 8 --graph 1 --sustained-seconds 3 > $O/bench_config1_mono_graph.json 2> $O/bench_config1_mono_graph.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/c26/bench_config1_mono_graph.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['kernel'][:50], d['roofline']['frac'], d['roofline']['launches'], d['roofline_fwd']['frac'], d['roofline_fwd']['all_launches_of_this_instantiation'])
PY
tail -3 $O/bench_config1_mono_graph.err | cut -c1-300
