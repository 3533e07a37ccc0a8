# Round-2 final measurement batch, run on the GPU box:  bash tools/measure_r02c.sh
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype bf16 > $O/bench_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 > $O/bench_f16.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --model PLIFNet > $O/bench_plif.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --count-rates 1 > $O/bench_count_rates.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 5 --warmup 2 --dtype f16 --T 10 --batch 32 --count-rates 1 > $O/bench_f16_T10_B32_rates.json 2>/dev/null
SS_GEMM6_CIN= SS_FUSED_BWD_CIN= SS_WGRAD_MFMA_CIN= python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_no_backward_kernels.json 2>/dev/null
ROUNDS=5 python tools/bench_fused_upconv.py > $O/decoder_kernels.log 2>&1
python tools/profile_step.py > $O/profile_step.log 2>&1
bash profiles/run_profile.sh r02c --steps 10 --warmup 3 > /dev/null 2>&1
bash profiles/collect_pmc.sh r02c rc > /dev/null 2>&1
for f in bench_default bench_bf16 bench_f16 bench_plif bench_count_rates bench_f16_T10_B32_rates bench_no_backward_kernels; do python - "$O/$f.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'])
PY
done
