python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
for d in f32 bf16 f16; do python bench.py --no-cpu-baseline --dtype $d > gpurun_out/bench_rc_$d.json 2>/dev/null; done
python bench.py --no-cpu-baseline --dtype bf16 --recompute-h 0 > gpurun_out/bench_rc0_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --dtype f16 --T 10 --batch 32 > gpurun_out/bench_rc_f16_T10_B32.json 2>/dev/null
for f in rc_f32 rc_bf16 rc_f16 rc0_bf16 rc_f16_T10_B32; do python - <<PY
import json
d=json.load(open('gpurun_out/bench_$f.json'))
print('$f', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d['roofline_bwd']['achieved'], d['roofline_bwd']['avg_launch_us'], d['neuron_kernels_all_layers'])
PY
done
