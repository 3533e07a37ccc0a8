#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 900 python -X faulthandler -m pytest tests/test_gpu_05_full_size.py -q -m gpu -k "config5" -x > gpurun_out/r05/full_size_config5.log 2>&1; echo "config5 rc $?"
tail -5 gpurun_out/r05/full_size_config5.log | cut -c1-300
timeout 900 python -X faulthandler -m pytest tests/test_gpu_05_full_size.py -q -m gpu -k "config2" -x > gpurun_out/r05/full_size_config2.log 2>&1; echo "config2 rc $?"
grep -n "Fatal\|Segmentation\|Error\|error\|File \"/root\|rootrepo\|stereospike_amd" gpurun_out/r05/full_size_config2.log | head -40 | cut -c1-300
tail -5 gpurun_out/r05/full_size_config2.log | cut -c1-300
