#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/c18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -x -k "wgrad or upconv" 2>&1 | tail -4 | cut -c1-300
python bench.py --no-cpu-baseline > $O/bench_f32.json 2>/dev/null; head -c 220 $O/bench_f32.json; echo
bash profiles/run_profile.sh r05_c18_f32 --steps 10 --warmup 2 > /dev/null 2>&1
grep -E "spike_wgrad_kernel|gemm6|spike_conv_wgrad" gpurun_out/prof_r05_c18_f32/trace_kernel_stats.csv | cut -c1-75,140-270
