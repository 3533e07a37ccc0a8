#!/bin/bash
# round 6, call 32: the bench lines again with the closing batch's PMC files installed (profiles/pmc_traffic*.json now carry the loaded library's hash: `traffic_source.matches_loaded_library`)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c32; mkdir -p $O
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 200 $O/bench_default.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 5 > $O/bench_f16_T10_B32_rates.json 2> $O/bench_f16_T10.err; head -c 200 $O/bench_f16_T10_B32_rates.json; echo
python - <<'PY'
import json
for f in ('bench_default','bench_f16_T10_B32_rates'):
    d=json.loads(open(f'gpurun_out/r06/c32/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_source'))
PY
