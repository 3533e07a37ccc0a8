#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -k "spike_conv_fwd or wgrad_reduce3 or decoder_stage" 2>&1 | tail -4
timeout 600 python tools/r04/ab_conv34.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/conv34_ab.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_reduce3.json 2> gpurun_out/r04/bench_reduce3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_reduce3.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
PY
