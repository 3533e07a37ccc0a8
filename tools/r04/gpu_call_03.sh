#!/bin/bash
# round 4, GPU call 3: the whole GPU suite after the EngineConfig refactor + the box-sum backward, then the bench line
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r04/pytest_gpu_call03.log 2>&1
tail -15 gpurun_out/r04/pytest_gpu_call03.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_call03.json 2> gpurun_out/r04/bench_call03.err
tail -2 gpurun_out/r04/bench_call03.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04/bench_call03.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['other_fused_kernels_ms_per_step'])
for k, v in d['plan'].items():
    print(k, v)
PY
