#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c04; mkdir -p $O
timeout 600 tools/ubench/_build/neuron16_ab.out 10 32 1 > $O/ab_f16_T10_B32.log 2>&1; echo "rc $?"; cat $O/ab_f16_T10_B32.log
timeout 600 tools/ubench/_build/neuron16_ab.out 5 16 2 > $O/ab_bf16_T5_B16.log 2>&1; echo "rc $?"; cat $O/ab_bf16_T5_B16.log
