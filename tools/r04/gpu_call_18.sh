#!/bin/bash
mkdir -p gpurun_out/r04
SS_LIB=stereospike_amd/lib/libss_neuron_sbt.so timeout 300 python tools/r04/trace_sub.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/sub_trace_v2.log
tail -22 gpurun_out/r04/sub_trace_v2.log
: > gpurun_out/r04/boxsum_variants.log
for n in "" bsjs4 bsjs16 bshu bshu16; do
  lib=stereospike_amd/lib/libss_neuron${n:+_$n}.so
  echo "== $lib" >> gpurun_out/r04/boxsum_variants.log
  SS_LIB=$lib ONLY=deconv1,deconv2 timeout 300 python tools/bench_upconv_bwd.py 2>&1 | grep -E "box-sum planes" >> gpurun_out/r04/boxsum_variants.log
done
cat gpurun_out/r04/boxsum_variants.log
