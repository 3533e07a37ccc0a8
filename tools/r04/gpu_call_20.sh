#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "upconv_box or decoder_stage or spike_conv_fwd" 2>&1 | tail -3
ONLY=deconv1,deconv2 timeout 600 python tools/bench_upconv_bwd.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04/bench_box_bwd_v10.log
grep -E "box:" gpurun_out/r04/bench_box_bwd_v10.log
timeout 300 python tools/r04/ab_conv34.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/conv34_ab_v2.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_f4_epilogues.json 2> gpurun_out/r04/bench_f4_epilogues.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_f4_epilogues.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d['other_fused_kernels_ms_per_step']); print(d['upconv_by_stage_ms_per_step'])
PY
