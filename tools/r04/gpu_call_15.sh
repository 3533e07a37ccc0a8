#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -k "upconv_fused_mfma_forward or exact_bf16x3_projection or upconv_sub" 2>&1 | tail -3
: > gpurun_out/r04/sub_fwd_ablations_v2.log
echo "== shipped build" >> gpurun_out/r04/sub_fwd_ablations_v2.log
ONLY=deconv1,deconv2 ROUNDS=3 timeout 300 python tools/r04/bench_sub_fwd.py 2>&1 | grep -E "sub-pixel, packed" >> gpurun_out/r04/sub_fwd_ablations_v2.log
for v in 1 2 4 8 3 11; do
  echo "== SS_SB_ABLATE=$v (1 no window traffic, 2 no weight stream, 4 no MFMA, 8 no stage barriers)" >> gpurun_out/r04/sub_fwd_ablations_v2.log
  SS_LIB=stereospike_amd/lib/libss_neuron_sb$v.so ONLY=deconv1,deconv2 ROUNDS=3 timeout 300 python tools/r04/bench_sub_fwd.py 2>&1 | grep -E "sub-pixel, packed" >> gpurun_out/r04/sub_fwd_ablations_v2.log
done
cat gpurun_out/r04/sub_fwd_ablations_v2.log
