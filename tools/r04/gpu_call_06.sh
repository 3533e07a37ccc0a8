#!/bin/bash
# box kernels: tests on the shipped build, micro-benchmark, then the data-gradient kernel's ablation builds (timing only; results are wrong by construction)
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -k "upconv_box" > gpurun_out/r04/pytest_box3.log 2>&1
tail -3 gpurun_out/r04/pytest_box3.log
ONLY=deconv1,deconv2 timeout 600 python tools/bench_upconv_bwd.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04/bench_box_bwd_v7.log
grep -E "^deconv|box:" gpurun_out/r04/bench_box_bwd_v7.log
: > gpurun_out/r04/box_dgrad_ablations.log
for v in 1 2 4 8 3 11; do
  echo "== SS_BX_ABLATE=$v (1 no window traffic, 2 no weight stream, 4 no MFMA, 8 no stage barriers)" >> gpurun_out/r04/box_dgrad_ablations.log
  SS_LIB=stereospike_amd/lib/libss_neuron_bx$v.so ONLY=deconv1,deconv2 timeout 300 python tools/bench_upconv_bwd.py 2>&1 | grep -E "^deconv|box: dgrad" >> gpurun_out/r04/box_dgrad_ablations.log
done
cat gpurun_out/r04/box_dgrad_ablations.log
