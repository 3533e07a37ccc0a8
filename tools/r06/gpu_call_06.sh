#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_07_round6.py tests/test_gpu_06_x16_kernels.py -q -k "neuron or packed_spike or low_rank or round6 or membrane or fwd16 or lr_x16 or lazy" > $O/pytest_neuron16.log 2>&1; echo "rc $?"; tail -15 $O/pytest_neuron16.log

