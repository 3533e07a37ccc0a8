#!/bin/bash
# round 6, call 33: A/B of the box-sum backward's geometry floor in the 16-bit modes (deconv3 / deconv4 on the box kernels with two chunks per window)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c33; mkdir -p $O
for fl in 2048 1024 0; do
  timeout -k 10 600 python tools/r06/ab_box_floor.py $fl --no-cpu-baseline --dtype bf16 --sustained-seconds 0 > $O/bf16_floor$fl.json 2> $O/bf16_floor$fl.err
  timeout -k 10 600 python tools/r06/ab_box_floor.py $fl --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 0 > $O/f16T10_floor$fl.json 2> $O/f16T10_floor$fl.err
  timeout -k 10 600 python tools/r06/ab_box_floor.py $fl --no-cpu-baseline --sustained-seconds 0 > $O/f32_floor$fl.json 2> $O/f32_floor$fl.err
  python - <<PY
import json
for n in ('bf16','f16T10','f32'):
    try:
        d=json.loads(open('$O/%s_floor$fl.json' % n).read().strip().splitlines()[-1]); print('floor $fl', n, d['value'], d['ms_per_step'], d['upconv_by_stage_ms_per_step'])
    except Exception as e: print('floor $fl', n, 'ERR', e)
PY
done
