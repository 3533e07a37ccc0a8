# Round-3 closing measurement batch, run on the GPU box:  bash tools/measure_r03.sh   (outputs under gpurun_out/r03h/)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
SS_CONV_DGRAD_MFMA=0 SS_PACKED_HEAD=0 SS_PACKED_DECONV2=0 SS_CONV_S1_WGRAD_MFMA=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_second_session_kernels_only.json 2>/dev/null
SS_CONV_FWD_MFMA=0 SS_CONV_S1_MFMA=0 SS_FUSED_DGRAD=0 SS_CONV_DGRAD_MFMA=0 SS_PACKED_HEAD=0 SS_CONV_S1_WGRAD_MFMA=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_round2_kernels_only.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype bf16 > $O/bench_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 > $O/bench_f16.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --model PLIFNet > $O/bench_plif.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --count-rates 1 > $O/bench_count_rates.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 6 --warmup 2 --dtype f16 --T 10 --batch 32 --count-rates 1 > $O/bench_f16_T10_B32_rates.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --force-dp > $O/bench_force_dp.json 2>/dev/null
python tools/profile_step.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" | cut -c1-230 | head -70 > $O/profile_step.log
bash profiles/run_profile.sh r03h --steps 13 --warmup 3 > /dev/null 2>&1
bash profiles/collect_pmc.sh r03h rc > /dev/null 2>&1
for f in bench_default bench_second_session_kernels_only bench_round2_kernels_only bench_bf16 bench_f16 bench_plif bench_count_rates bench_f16_T10_B32_rates bench_force_dp; do python - "$O/$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
    print(sys.argv[1], d['value'], d['ms_per_step'], r['frac'], r.get('frac_by_12B_per_update_definition'), r['avg_launch_us'], d['roofline_fwd']['frac'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
