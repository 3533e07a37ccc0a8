#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c24; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_04_x16_parity.py tests/test_gpu_05_full_size.py tests/test_gpu_00_default_path.py tests/test_gpu_03_model.py -x -q > $O/pytest.log 2>&1; echo "rc $?"; tail -6 $O/pytest.log | cut -c1-300
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 0 > $O/bench_f16_T10.json 2> $O/bench_f16_T10.err; echo "bench f16 T10 rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 --sustained-seconds 0 > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc $?"
python - <<'PY'
import json
for f in ('bench_f16_T10','bench_bf16'):
    j=json.loads(open(f'gpurun_out/r06/c24/{f}.json').read().strip().splitlines()[-1])
    print(f, j['value'], j['ms_per_step'], 'bwd', j['roofline_bwd']['frac'], 'neuron ms', j['neuron_kernels_all_layers']['ms_per_step'], j['plan'].get('predict_depth3'), j['plan'].get('deconv3',{}).get('neuron_bwd'))
PY
