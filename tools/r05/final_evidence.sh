#!/bin/bash
# (every step under its own timeout: a rocprofv3 --pmc pass that met a GPU memory fault once sat in its signal handler until gpurun's limit, 60 GPU-minutes)
# round 5, closing batch on the final tree (one gpurun call; SS_GIT_HEAD is passed in by the caller: the GPU box holds no .git):
#   the driver's own commands (bench, pytest -m gpu -x -q, smoke), rocprofv3 kernel stats of the bench command in the three modes, the two --pmc passes,
#   the 16-bit modes' and config 2's lines, the dispatch plans of all BASELINE configurations
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/final
mkdir -p $O
echo "git head ${SS_GIT_HEAD:-unknown}; lib source hash $(python -c 'from stereospike_amd import _lib; print(_lib.source_hash(), _lib.tree_source_hash())')" | tee $O/source.txt
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err; head -c 260 $O/bench_default.json; echo
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout -k 10 600 python bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err; head -c 220 $O/bench_bf16.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --dtype f16 > $O/bench_f16.json 2> $O/bench_f16.err; head -c 220 $O/bench_f16.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 6 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 > $O/bench_f16_T10_B32_rates.json 2> $O/bench_f16_T10_B32_rates.err; head -c 220 $O/bench_f16_T10_B32_rates.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --dtype bf16 --model PLIFNet --T 1 --batch 8 --graph 1 > $O/bench_config2_graph.json 2> $O/bench_config2_graph.err; head -c 220 $O/bench_config2_graph.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --model PLIFNet > $O/bench_plif.json 2> $O/bench_plif.err; head -c 220 $O/bench_plif.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --dtype bf16 --x16-own 0 > $O/bench_bf16_legacy_miopen.json 2> $O/bench_bf16_legacy_miopen.err; head -c 220 $O/bench_bf16_legacy_miopen.json; echo
python tools/dump_plans.py > $O/plans.json 2> $O/plans.err
timeout -k 10 600 bash profiles/run_profile.sh r05_final_f32 --steps 10 --warmup 2
timeout -k 10 600 bash profiles/run_profile.sh r05_final_bf16 --steps 10 --warmup 2 --dtype bf16
timeout -k 10 600 bash profiles/run_profile.sh r05_final_f16_T10 --steps 6 --warmup 2 --dtype f16 --T 10 --batch 32 --count-rates 1
timeout -k 10 600 bash profiles/collect_pmc.sh r05_final rc > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-200
timeout -k 10 600 bash profiles/collect_pmc.sh r05_final_x16 x16 > $O/pmc_x16.log 2>&1; tail -3 $O/pmc_x16.log | cut -c1-200
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log | cut -c1-200
