#!/bin/bash
# round 5, call 7: the full GPU suite on the pruned tree; the full-size config 2 / 5 tests on their own first (a core dump was seen in call 5)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 900 python -X faulthandler -m pytest tests/test_gpu_05_full_size.py -q -m gpu -k "config5" -x > gpurun_out/r05/full_size_config5.log 2>&1; echo "config5 rc $?"
tail -4 gpurun_out/r05/full_size_config5.log | cut -c1-300
timeout 900 python -X faulthandler -m pytest tests/test_gpu_05_full_size.py -q -m gpu -k "config2" -x > gpurun_out/r05/full_size_config2.log 2>&1; echo "config2 rc $?"
grep -n "Fatal\|Segmentation\|Error\|rror:\|stereospike_amd/" gpurun_out/r05/full_size_config2.log | head -30 | cut -c1-250
tail -4 gpurun_out/r05/full_size_config2.log | cut -c1-300
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_05_full_size.py::test_config2_mono_plif_T1_bf16_B8_full_resolution > gpurun_out/r05/pytest_gpu_pruned_1.log 2>&1; echo "suite rc $?"
tail -30 gpurun_out/r05/pytest_gpu_pruned_1.log | cut -c1-300
