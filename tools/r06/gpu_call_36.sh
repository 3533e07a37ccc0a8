#!/bin/bash
# round 6, call 36: which GPU tests skip (reasons), and the bench lines with the closing batch's PMC files installed
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c36; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_06_x16_kernels.py -q -m gpu -rs -k "box" 2>&1 | tail -8 | cut -c1-250
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 200 $O/bench_default.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 5 > $O/bench_f16_T10_B32_rates.json 2> $O/bench_f16_T10.err; head -c 200 $O/bench_f16_T10_B32_rates.json; echo
python - <<'PY'
import json
for f in ('bench_default','bench_f16_T10_B32_rates'):
    d=json.loads(open(f'gpurun_out/r06/c36/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_source',{}).get('matches_loaded_library'))
PY
