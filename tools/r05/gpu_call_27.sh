#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/c27; mkdir -p $O
for mode in f32 bf16; do
  extra=""; [ $mode = bf16 ] && extra="--dtype bf16"
  for cfg in "- -" "0 -" "- 0" "0 0"; do
    tag=$(echo $cfg | tr ' ' '_')
    timeout 600 python tools/r05/ab_stage_floors.py $cfg $extra > $O/${mode}_$tag.json 2> $O/${mode}_$tag.err
    python - "$O/${mode}_$tag.json" "$mode box/sub floors: $cfg" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print(sys.argv[2], j['value'], j['ms_per_step'], {k: v for k, v in j.get('upconv_by_stage_ms_per_step', {}).items()} if isinstance(j.get('upconv_by_stage_ms_per_step'), dict) else '')
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
  done
done 2>&1 | tee $O/summary.log | cut -c1-400
