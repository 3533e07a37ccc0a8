#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_00_default_path.py tests/test_gpu_03_model.py -x -q -m gpu 2>&1 | tail -4
for k in 1 2; do
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_tall_$k.json 2> gpurun_out/r04/bench_tall.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench_tall_$k.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d['upconv_by_stage_ms_per_step'])
PY
done
