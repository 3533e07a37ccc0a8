#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c22; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_04_x16_parity.py tests/test_gpu_05_full_size.py -x -q > $O/pytest_x16.log 2>&1; echo "rc $?"; tail -4 $O/pytest_x16.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 0 > $O/bench_f16_T10.json 2> $O/bench_f16_T10.err; echo "bench f16 T10 rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype f16 --sustained-seconds 0 > $O/bench_f16.json 2> $O/bench_f16.err; echo "bench f16 rc $?"
python - <<'PY'
import json
for f in ('bench_f16_T10','bench_f16'):
    j=json.loads(open(f'gpurun_out/r06/c22/{f}.json').read().strip().splitlines()[-1])
    print(f, j['value'], j['ms_per_step'], j['plan'].get('deconv4'), j['plan'].get('predict_depth4'))
PY
cat gpurun_out/parity_report_x16.json | python -c "
import json,sys; j=json.load(sys.stdin)
for k,v in j.items(): print(k, {a:b for a,b in v.items() if a in ('max_flips_per_layer','worst_weight_rel','loss_rel','depth_rel')} if isinstance(v,dict) else v)" 2>/dev/null | head -12
