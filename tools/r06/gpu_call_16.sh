#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
export SS_GIT_HEAD=$(cat tools/r06/.git_head 2>/dev/null || echo unknown)
O=gpurun_out/r06/c16; mkdir -p $O
timeout -k 10 600 bash profiles/collect_pmc.sh r06_c16_rc rc > $O/pmc_rc.log 2>&1; echo "rc $?"
for i in 1 2 3; do
  timeout -k 10 600 bash profiles/collect_pmc.sh r06_c16_x16c5_$i x16c5 > $O/pmc_x16c5_$i.log 2>&1; echo "rc $?"
done
python - <<'PY'
import json
j=json.load(open('gpurun_out/pmc_r06_c16_rc/pmc_traffic.json'))
print('rc', {k:(v['hbm_bytes_per_launch'], v['ratio_to_algorithmic']) for k,v in j.items() if isinstance(v,dict) and 'ratio_to_algorithmic' in v and 'neuron' in k})
for i in (1,2,3):
    j=json.load(open(f'gpurun_out/pmc_r06_c16_x16c5_{i}/pmc_traffic.json'))
    print('x16c5', i, {k:(v['FETCH_SIZE_KiB_raw'], v['WRITE_SIZE_KiB_raw'], v['ratio_to_algorithmic']) for k,v in j.items() if isinstance(v,dict) and 'ratio_to_algorithmic' in v})
PY
