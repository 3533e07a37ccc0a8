#!/bin/bash
# round 6, call 7: bench lines after the 16-bit neuron kernels + lazy membrane; kernel stats of the f16 T10 step; PMC pass at config 5's layer shape
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c07; mkdir -p $O
export SS_GIT_HEAD=$(cat tools/r06/.git_head 2>/dev/null || echo unknown)
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc $?"
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 5 > $O/bench_f16_T10.json 2> $O/bench_f16_T10.err; echo "bench f16 rc $?"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 --sustained-seconds 5 > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc $?"
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --dtype bf16 --model PLIFNetMono --T 1 --batch 8 --graph 1 --sustained-seconds 5 > $O/bench_config2_mono_graph.json 2> $O/bench_config2.err; echo "bench config2 rc $?"
python - <<'PY'
import json
for f in ('bench_f32','bench_f16_T10','bench_bf16','bench_config2_mono_graph'):
    try:
        j=json.loads(open(f'gpurun_out/r06/c07/{f}.json').read().strip().splitlines()[-1])
        print(f, j['value'], j['ms_per_step'], 'sustained', j.get('sustained_frames_per_s'), j.get('sustained_over_value'), 'roofline', j['roofline'].get('frac'), j.get('roofline_fwd',{}).get('frac'), j.get('roofline_bwd',{}).get('frac'), 'neuron ms', j.get('neuron_kernels_all_layers',{}).get('ms_per_step'))
        print('   ', j['config']['workload'][:200])
    except Exception as e: print(f,'FAILED',e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_f16_T10 -o ks -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f16 --T 10 --batch 32 --count-rates 1 --sustained-seconds 0 > $GRAFT_REPO_ROOT/$O/stats_f16_T10_bench.log 2>&1; echo "rocprof rc $?"
cd $GRAFT_REPO_ROOT
find $O/stats_f16_T10 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_f16_T10.csv
head -12 $O/kernel_stats_f16_T10.csv | cut -c1-200
timeout 900 bash profiles/collect_pmc.sh r06_x16c5 x16c5 > $O/pmc_x16c5.log 2>&1; echo "pmc rc $?"; tail -40 $O/pmc_x16c5.log | head -60
cp gpurun_out/pmc_r06_x16c5/pmc_traffic.json $O/pmc_traffic_x16c5.json 2>/dev/null
