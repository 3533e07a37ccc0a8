#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/c19; mkdir -p $O
SS_DGRAD_MB2=1 timeout 900 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -x -k "conv_s2_dgrad" 2>&1 | tail -3 | cut -c1-300
for i in 1 2; do
python tools/_abl_dgrad.py 2>/dev/null | tee -a $O/dgrad_mb2_ab.log
SS_DGRAD_MB2=1 python tools/_abl_dgrad.py 2>/dev/null | sed 's/^default/MB2/' | tee -a $O/dgrad_mb2_ab.log
done
