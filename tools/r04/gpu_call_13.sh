#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 2700 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04/pytest_gpu_full2.log 2>&1
tail -5 gpurun_out/r04/pytest_gpu_full2.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_sub.json 2> gpurun_out/r04/bench_sub.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_sub.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d['other_fused_kernels_ms_per_step']); print(d['upconv_by_stage_ms_per_step']); print(d['roofline_upconv'])
PY
