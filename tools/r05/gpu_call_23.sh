#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -x -k "test_spike_conv_wgrad_mfma" 2>&1 | grep -E "^E|assert|Error|passed|failed" | head -30 | cut -c1-300
