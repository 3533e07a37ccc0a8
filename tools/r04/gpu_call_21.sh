#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "conv_s2_dgrad or dense_conv_s1 or spike_conv_fwd" 2>&1 | tail -3
for k in 1 2; do
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_f4_epilogues_v2_$k.json 2> gpurun_out/r04/bench_f4_epilogues_v2.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r04/bench_f4_epilogues_v2_$k.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print(d['other_fused_kernels_ms_per_step'])
PY
done
