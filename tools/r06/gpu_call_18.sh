#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c18; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_07_round6.py tests/test_gpu_06_x16_kernels.py -q -k "neuron or packed_spike or low_rank or round6 or membrane or fwd16 or lr_x16 or lazy" > $O/pytest_neuron16.log 2>&1; echo "rc $?"; tail -3 $O/pytest_neuron16.log
SS_LIB=stereospike_amd/lib/libss_neuron.so timeout 600 tools/ubench/_build/neuron16_ab.out 10 32 1 > $O/ab_f16_T10.log 2>&1; echo "rc $?"; grep -E "^fwd" $O/ab_f16_T10.log
timeout 600 tools/ubench/_build/neuron16_ab.out 5 16 2 > $O/ab_bf16_T5.log 2>&1; echo "rc $?"; grep -E "^fwd" $O/ab_bf16_T5.log
