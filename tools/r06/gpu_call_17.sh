#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c17; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_07_round6.py tests/test_gpu_06_x16_kernels.py -q -k "neuron or packed_spike or low_rank or round6 or membrane or fwd16 or lr_x16 or lazy" > $O/pytest_neuron16.log 2>&1; echo "rc $?"; tail -3 $O/pytest_neuron16.log
timeout 600 tools/ubench/_build/neuron16_ab.out 10 32 1 > $O/ab_f16_T10.log 2>&1; echo "rc $?"; grep -E "^bwd lr" $O/ab_f16_T10.log
timeout 600 tools/ubench/_build/neuron16_ab.out 5 16 2 > $O/ab_bf16_T5.log 2>&1; echo "rc $?"; grep -E "^bwd lr" $O/ab_bf16_T5.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 0 > $O/bench_f16_T10.json 2> $O/bench_f16_T10.err; echo "bench f16 rc $?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06/c17/bench_f16_T10.json').read().strip().splitlines()[-1])
print('f16 T10', j['value'], j['ms_per_step'], 'roofline bwd', j['roofline_bwd']['frac'], 'fwd', j['roofline_fwd']['frac'], 'neuron ms', j['neuron_kernels_all_layers']['ms_per_step'])
PY
export SS_GIT_HEAD=$(cat tools/r06/.git_head 2>/dev/null || echo unknown)
timeout -k 10 600 bash profiles/collect_pmc.sh r06_c17_x16c5 x16c5 > $O/pmc_x16c5.log 2>&1; echo "rc $?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/pmc_r06_c17_x16c5/pmc_traffic.json'))
print('x16c5', {k:(v['hbm_bytes_per_launch'], v['ratio_to_algorithmic']) for k,v in j.items() if isinstance(v,dict) and 'ratio_to_algorithmic' in v})
PY
