#!/bin/bash
# round 6, call 1: machine facts for the 16-bit neuron kernels + start-of-round bench lines
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c01; mkdir -p $O
timeout 300 tools/ubench/_build/valu_facts.out > $O/valu_facts.log 2>&1; echo "valu_facts rc $?"
tail -60 $O/valu_facts.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f32_start.json 2> $O/bench_f32_start.err; echo "bench f32 rc $?"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f16 --T 10 --batch 32 --count-rates 1 > $O/bench_f16_T10_start.json 2> $O/bench_f16_T10_start.err; echo "bench f16 rc $?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 > $O/bench_bf16_start.json 2> $O/bench_bf16_start.err; echo "bench bf16 rc $?"
python - <<'PY'
import json
for f in ('bench_f32_start','bench_f16_T10_start','bench_bf16_start'):
    try:
        j=json.loads(open(f'gpurun_out/r06/c01/{f}.json').read().strip().splitlines()[-1])
        print(f, j['value'], j['ms_per_step'], 'roofline', j['roofline'].get('frac'), j.get('roofline_fwd',{}).get('frac'))
    except Exception as e: print(f,'FAILED',e)
PY
