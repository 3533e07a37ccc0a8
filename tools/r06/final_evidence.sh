#!/bin/bash
# round 6, closing batch on the final tree (one gpurun call; the git head is read from tools/r06/.git_head, written by the caller: the GPU box holds no .git):
#   the driver's own commands (bench, pytest -m gpu -x -q, smoke), the other BASELINE configurations' lines (labels derived from the arguments), the dispatch plans,
#   rocprofv3 kernel stats of the bench command in three modes, the --pmc passes (fp32 layer of config 3; fp16 T = 10 layer of config 5; bf16 config-3 shapes).
#   Every step under its own timeout.
set -u
cd "$GRAFT_REPO_ROOT"
export SS_GIT_HEAD=$(cat tools/r06/.git_head 2>/dev/null || echo unknown)
O=gpurun_out/r06/final
mkdir -p $O
echo "git head $SS_GIT_HEAD; lib source hash $(python -c 'from stereospike_amd import _lib; print(_lib.source_hash(), _lib.tree_source_hash())')" | tee $O/source.txt
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err; head -c 260 $O/bench_default.json; echo
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout -k 10 600 python bench.py --no-cpu-baseline --dtype bf16 --sustained-seconds 5 > $O/bench_bf16.json 2> $O/bench_bf16.err; head -c 220 $O/bench_bf16.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --dtype f16 --sustained-seconds 5 > $O/bench_f16.json 2> $O/bench_f16.err; head -c 220 $O/bench_f16.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 5 > $O/bench_f16_T10_B32_rates.json 2> $O/bench_f16_T10_B32_rates.err; head -c 220 $O/bench_f16_T10_B32_rates.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --dtype bf16 --model PLIFNetMono --T 1 --batch 8 --graph 1 --sustained-seconds 5 > $O/bench_config1_mono_graph.json 2> $O/bench_config1_mono_graph.err; head -c 220 $O/bench_config1_mono_graph.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --dtype bf16 --model PLIFNetMono --T 1 --batch 8 --sustained-seconds 5 > $O/bench_config1_mono_eager.json 2> $O/bench_config1_mono_eager.err; head -c 220 $O/bench_config1_mono_eager.json; echo
timeout -k 10 600 python bench.py --no-cpu-baseline --model PLIFNet --sustained-seconds 0 > $O/bench_plif.json 2> $O/bench_plif.err; head -c 220 $O/bench_plif.json; echo
python tools/dump_plans.py > $O/plans.json 2> $O/plans.err
timeout -k 10 600 bash profiles/run_profile.sh r06_final_f32 --steps 10 --warmup 2 --sustained-seconds 0
timeout -k 10 600 bash profiles/run_profile.sh r06_final_bf16 --steps 10 --warmup 2 --dtype bf16 --sustained-seconds 0
timeout -k 10 600 bash profiles/run_profile.sh r06_final_f16_T10 --steps 6 --warmup 2 --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 0
timeout -k 10 600 bash profiles/collect_pmc.sh r06_final rc > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-200
timeout -k 10 600 bash profiles/collect_pmc.sh r06_final_x16c5 x16c5 > $O/pmc_x16c5.log 2>&1; tail -3 $O/pmc_x16c5.log | cut -c1-200
timeout -k 10 600 bash profiles/collect_pmc.sh r06_final_x16 x16 > $O/pmc_x16.log 2>&1; tail -3 $O/pmc_x16.log | cut -c1-200
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log | cut -c1-200
