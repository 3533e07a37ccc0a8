#!/bin/bash
# round 5, call 3: where the own-kernel 16-bit modes spend their step; config 5's per-GPU share; config 2; the plans of all configs
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
python bench.py --no-cpu-baseline --steps 6 --warmup 3 --dtype f16 --T 10 --batch 32 --count-rates 1 > gpurun_out/r05/bench_f16_T10_B32_own_1.json 2> gpurun_out/r05/bench_f16_T10_B32_own_1.err
tail -c 300 gpurun_out/r05/bench_f16_T10_B32_own_1.err; head -c 250 gpurun_out/r05/bench_f16_T10_B32_own_1.json; echo
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --dtype f16 > gpurun_out/r05/bench_f16_own_1.json 2> gpurun_out/r05/bench_f16_own_1.err
head -c 250 gpurun_out/r05/bench_f16_own_1.json; echo
bash profiles/run_profile.sh r05_bf16_own1 --steps 8 --warmup 2 --dtype bf16
bash profiles/run_profile.sh r05_f16_T10_own1 --steps 4 --warmup 2 --dtype f16 --T 10 --batch 32 --count-rates 1
python tools/dump_plans.py > gpurun_out/r05/plans.json 2> gpurun_out/r05/plans.err; tail -3 gpurun_out/r05/plans.err
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --dtype bf16 --model PLIFNet --T 1 --batch 8 --graph 1 > gpurun_out/r05/bench_config2_graph_1.json 2> gpurun_out/r05/bench_config2_graph_1.err
tail -c 600 gpurun_out/r05/bench_config2_graph_1.err; head -c 250 gpurun_out/r05/bench_config2_graph_1.json; echo
