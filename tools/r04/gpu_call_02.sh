#!/bin/bash
# round 4, GPU call 2: the box-sum backward kernels — correctness, micro-benchmark, then the step
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -k "upconv_box" -x > gpurun_out/r04/pytest_box.log 2>&1
tail -5 gpurun_out/r04/pytest_box.log
timeout 600 python tools/bench_upconv_bwd.py > gpurun_out/r04/bench_box_bwd.log 2>&1
tail -60 gpurun_out/r04/bench_box_bwd.log
timeout 900 python -m pytest tests/test_gpu_00_default_path.py -q -m gpu -k "penalize or T5" -x > gpurun_out/r04/pytest_default_T5.log 2>&1
tail -5 gpurun_out/r04/pytest_default_T5.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_call02.json 2> gpurun_out/r04/bench_call02.err
SS_BOX_BWD=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_call02_boxoff.json 2> gpurun_out/r04/bench_call02_boxoff.err
python - <<'PY'
import json
for f in ('bench_call02', 'bench_call02_boxoff'):
    try:
        d = json.loads(open(f'gpurun_out/r04/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['other_fused_kernels_ms_per_step'], d['upconv_by_stage_ms_per_step'])
    except Exception as e:
        print(f, 'failed', e)
PY
