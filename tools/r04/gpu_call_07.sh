#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -k "upconv_box" > gpurun_out/r04/pytest_box4.log 2>&1
tail -3 gpurun_out/r04/pytest_box4.log
ONLY=deconv1,deconv2 timeout 600 python tools/bench_upconv_bwd.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04/bench_box_bwd_v8.log
grep -E "^deconv|box:" gpurun_out/r04/bench_box_bwd_v8.log
