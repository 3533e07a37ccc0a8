#!/bin/bash
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -k "upconv_fused_mfma_forward or exact_bf16x3_projection or upconv_sub" 2>&1 | tail -3
SS_WINOGRAD_GEMM6=1 timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_wino_gemm6_on.json 2> gpurun_out/r04/bench_wino_gemm6_on.err
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_wino_gemm6_off.json 2> gpurun_out/r04/bench_wino_gemm6_off.err
SS_WINOGRAD_GEMM6=1 timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_wino_gemm6_on2.json 2>> gpurun_out/r04/bench_wino_gemm6_on.err
python - <<'PY'
import json
for f in ('on','off','on2'):
    d=json.loads(open(f'gpurun_out/r04/bench_wino_gemm6_{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['plan']['bottleneck.0.conv1']['synapse_bwd'])
PY
