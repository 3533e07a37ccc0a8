#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "upconv_sub or upconv_box" 2>&1 | tail -3
ONLY=deconv1,deconv2 timeout 600 python tools/r04/bench_sub_fwd.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/bench_sub_fwd_v8.log
for n in "" bshu4; do
  lib=stereospike_amd/lib/libss_neuron${n:+_$n}.so
  echo "== $lib" >> gpurun_out/r04/boxsum_variants.log
  SS_LIB=$lib ONLY=deconv1,deconv2 timeout 300 python tools/bench_upconv_bwd.py 2>&1 | grep -E "box-sum planes" | tee -a gpurun_out/r04/boxsum_variants.log
done
