set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
python bench.py --steps 10 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --count-rates 1 > $O/bench_count_rates.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --pack-spikes 0 > $O/bench_nopack.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype bf16 > $O/bench_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 > $O/bench_f16.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 5 --warmup 2 --dtype f16 --T 10 --batch 32 --count-rates 1 > $O/bench_f16_T10_B32_rates.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --model PLIFNet > $O/bench_plif.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --force-dp > $O/bench_forcedp.json 2>/dev/null
bash profiles/run_profile.sh r02final --steps 10 --warmup 3 > /dev/null 2>&1
bash profiles/collect_pmc.sh r02 rc > /dev/null 2>&1
