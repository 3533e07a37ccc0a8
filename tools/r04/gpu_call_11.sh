#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "upconv_sub" > gpurun_out/r04/pytest_sub1.log 2>&1
tail -15 gpurun_out/r04/pytest_sub1.log
timeout 600 python tools/r04/bench_sub_fwd.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/bench_sub_fwd_v1.log
