#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "upconv_box or decoder_stage" 2>&1 | tail -3
ONLY=deconv1,deconv2 timeout 600 python tools/bench_upconv_bwd.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04/bench_box_bwd_v11.log
grep -E "box:" gpurun_out/r04/bench_box_bwd_v11.log
