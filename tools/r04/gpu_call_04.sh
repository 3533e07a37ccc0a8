#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -k "upconv_box" > gpurun_out/r04/pytest_box2.log 2>&1
tail -8 gpurun_out/r04/pytest_box2.log
timeout 600 python tools/bench_upconv_bwd.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r04/bench_box_bwd_v2.log
grep -E "^deconv|box:|gemm6|spike_wgrad|fused, NO" gpurun_out/r04/bench_box_bwd_v2.log
