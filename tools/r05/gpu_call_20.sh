#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/c20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_01_kernels.py tests/test_gpu_06_x16_kernels.py -q -m gpu -x -k "s1" 2>&1 | tail -12 | cut -c1-300
for i in 1 2; do
timeout 300 python tools/r05/ab_s1_wgrad.py 2>/dev/null | tee -a $O/s1_wgrad_ab.log
SS_S1_WGRAD_TR=0 timeout 300 python tools/r05/ab_s1_wgrad.py 2>/dev/null | tee -a $O/s1_wgrad_ab.log
done
