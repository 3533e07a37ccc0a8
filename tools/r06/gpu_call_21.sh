#!/bin/bash
# A/B: grid caps of the fp32 neuron kernels (variant libraries swapped in on the GPU box's scratch copy)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c21; mkdir -p $O
cp stereospike_amd/lib/libss_neuron.so /tmp/lib_default.so
for v in default fg4k bg8k fbg default; do
  if [ $v = default ]; then cp /tmp/lib_default.so stereospike_amd/lib/libss_neuron.so; else cp stereospike_amd/lib/libss_neuron_$v.so stereospike_amd/lib/libss_neuron.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $O/bench_f32_$v.json 2> $O/bench_f32_$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
j=json.loads(open(f'gpurun_out/r06/c21/bench_f32_{v}.json').read().strip().splitlines()[-1])
print(v, j['value'], j['ms_per_step'], 'fwd', j['roofline_fwd']['frac'], j['roofline_fwd']['avg_launch_us'], 'bwd', j['roofline_bwd']['frac'], j['roofline_bwd']['avg_launch_us'], 'neuron ms', j['neuron_kernels_all_layers']['ms_per_step'])
PY
done
cp /tmp/lib_default.so stereospike_amd/lib/libss_neuron.so
