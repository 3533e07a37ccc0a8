#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "upconv_sub" > gpurun_out/r04/pytest_sub2.log 2>&1
tail -5 gpurun_out/r04/pytest_sub2.log
ONLY=deconv1,deconv2 timeout 600 python tools/r04/bench_sub_fwd.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/bench_sub_fwd_v2.log
: > gpurun_out/r04/sub_fwd_ablations.log
for v in 1 2 4 8 3 11; do
  echo "== SS_SB_ABLATE=$v (1 no window traffic, 2 no weight stream, 4 no MFMA, 8 no stage barriers)" >> gpurun_out/r04/sub_fwd_ablations.log
  SS_LIB=stereospike_amd/lib/libss_neuron_sb$v.so ONLY=deconv1 ROUNDS=3 timeout 300 python tools/r04/bench_sub_fwd.py 2>&1 | grep -E "sub-pixel, packed" >> gpurun_out/r04/sub_fwd_ablations.log
done
cat gpurun_out/r04/sub_fwd_ablations.log
