#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/c14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_06_x16_kernels.py tests/test_gpu_01_kernels.py -q -m gpu -x -k "conv or dgrad or spike" 2>&1 | tail -4 | cut -c1-250
for i in 1 2; do
python tools/_abl_dgrad.py 2>/dev/null | tee -a $O/dgrad_pre0_ab.log
SS_LIB=stereospike_amd/lib/libss_neuron_pre0.so python tools/_abl_dgrad.py 2>/dev/null | tee -a $O/dgrad_pre0_ab.log
done
python bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16_a.json 2>/dev/null; head -c 220 $O/bench_bf16_a.json; echo
python bench.py --no-cpu-baseline > $O/bench_f32_a.json 2>/dev/null; head -c 220 $O/bench_f32_a.json; echo
bash profiles/run_profile.sh r05_c14_bf16 --steps 10 --warmup 2 --dtype bf16 > /dev/null 2>&1
bash profiles/run_profile.sh r05_c14_f32 --steps 10 --warmup 2 > /dev/null 2>&1
grep -E "conv_s2_dgrad_kernel|spike_conv_fwd_kernel" gpurun_out/prof_r05_c14_bf16/trace_kernel_stats.csv | cut -c1-75,140-270
grep -E "conv_s2_dgrad_kernel|spike_conv_fwd_kernel" gpurun_out/prof_r05_c14_f32/trace_kernel_stats.csv | cut -c1-75,140-270
