#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c15; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc $?"; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc $?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/c15/bench_f32.json').read().strip().splitlines()[-1])
print('f32', j['value'], j['ms_per_step'])
PY
