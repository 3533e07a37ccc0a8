#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 1500 python tools/r05/repro_graph.py > gpurun_out/r05/repro_graph_2.log 2>&1
cat gpurun_out/r05/repro_graph_2.log | cut -c1-250
timeout 2700 python -m pytest tests/test_gpu_04_x16_parity.py tests/test_gpu_05_full_size.py tests/test_gpu_06_x16_kernels.py tests/test_gpu_zz_layouts.py -q -m gpu -x --deselect tests/test_gpu_05_full_size.py::test_config2_mono_plif_T1_bf16_B8_full_resolution > gpurun_out/r05/pytest_gpu_pruned_3.log 2>&1; echo "suite rc $?"
tail -8 gpurun_out/r05/pytest_gpu_pruned_3.log | cut -c1-300
