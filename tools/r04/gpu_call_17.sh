#!/bin/bash
# closing batch: the driver's GPU test command, the default bench (with CPU legs), rocprofv3 kernel stats of the bench command, the PMC traffic passes
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 2700 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04/pytest_gpu_final4.log 2>&1
tail -4 gpurun_out/r04/pytest_gpu_final4.log
timeout 1200 python bench.py > gpurun_out/r04/bench_default_v4.json 2> gpurun_out/r04/bench_default_v4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_default_v4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d['other_fused_kernels_ms_per_step']); print(d['upconv_by_stage_ms_per_step'])
PY
bash profiles/run_profile.sh r04d --steps 5 --warmup 2 | tail -3
bash profiles/collect_pmc.sh r04d rc | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
