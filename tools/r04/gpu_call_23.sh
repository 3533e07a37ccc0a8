#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_04_x16_parity.py -q -m gpu > gpurun_out/r04/pytest_x16_own.log 2>&1
tail -15 gpurun_out/r04/pytest_x16_own.log
for dt in bf16 f16; do for own in 1 0; do
SS_X16_OWN_CONVS=$own timeout 900 python bench.py --no-cpu-baseline --dtype $dt > gpurun_out/r04/bench_${dt}_own$own.json 2> gpurun_out/r04/bench_${dt}_own$own.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r04/bench_${dt}_own$own.json').read().strip().splitlines()[-1]); print('$dt own=$own', d['value'], d['ms_per_step'], d.get('peak_mem_GB'))
except Exception as e: print('$dt own=$own failed', e)
PY
done; done
