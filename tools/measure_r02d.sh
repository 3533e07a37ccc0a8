# Round-2 closing measurement batch (low-rank head gradients), run on the GPU box:  bash tools/measure_r02d.sh
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
SS_LOWRANK_HEAD_GRAD=0 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_lowrank_off.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype bf16 > $O/bench_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 > $O/bench_f16.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --model PLIFNet > $O/bench_plif.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --count-rates 1 > $O/bench_count_rates.json 2>/dev/null
python tools/profile_step.py > $O/profile_step.log 2>&1
bash profiles/run_profile.sh r02d --steps 10 --warmup 3 > /dev/null 2>&1
bash profiles/collect_pmc.sh r02d rc > /dev/null 2>&1
for f in bench_default bench_lowrank_off bench_bf16 bench_f16 bench_plif bench_count_rates; do python - "$O/$f.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['roofline']
print(sys.argv[1], d['value'], d['ms_per_step'], r['frac'], r.get('frac_by_12B_per_update_definition'), r['avg_launch_us'])
PY
done
