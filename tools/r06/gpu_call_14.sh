#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c14; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_04_x16_parity.py -x -q > $O/pytest_x16_parity.log 2>&1; echo "rc $?"; tail -4 $O/pytest_x16_parity.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 0 > $O/bench_f16_T10.json 2> $O/bench_f16_T10.err; echo "bench f16 rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 --sustained-seconds 0 > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc $?"
python - <<'PY'
import json
for f in ('bench_f16_T10','bench_bf16'):
    j=json.loads(open(f'gpurun_out/r06/c14/{f}.json').read().strip().splitlines()[-1])
    print(f, j['value'], j['ms_per_step'])
PY
