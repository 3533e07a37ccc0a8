#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/c15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_06_x16_kernels.py -q -m gpu -x 2>&1 | tail -3 | cut -c1-250
python bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16_a.json 2>/dev/null; head -c 220 $O/bench_bf16_a.json; echo
bash profiles/run_profile.sh r05_c15_bf16 --steps 10 --warmup 2 --dtype bf16 > /dev/null 2>&1
grep -E "conv_s2_dgrad_kernel|spike_conv_fwd_kernel|upconv_box" gpurun_out/prof_r05_c15_bf16/trace_kernel_stats.csv | cut -c1-75,140-270
