#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/c22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_01_kernels.py tests/test_gpu_06_x16_kernels.py -q -m gpu -x -k "spike_conv or upconv_sub or sub_fwd or packed" 2>&1 | tail -3 | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $O/bench_f32.json 2>/dev/null; head -c 220 $O/bench_f32.json; echo
timeout 600 python bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2>/dev/null; head -c 220 $O/bench_bf16.json; echo
timeout 600 python tools/r05/ab_spike_conv_wgrad.py 2>/dev/null | tee $O/spike_conv_wgrad_fast_unpack.log
timeout -k 10 600 bash profiles/run_profile.sh r05_c22_f32 --steps 10 --warmup 2 > /dev/null 2>&1
grep -E "upconv_sub_fwd|spike_conv_fwd_kernel|spike_conv_wgrad_tr" gpurun_out/prof_r05_c22_f32/trace_kernel_stats.csv | cut -c1-75,140-270
