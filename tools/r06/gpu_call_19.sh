#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c19; mkdir -p $O
for cap in 1048576 16384 8192 4096 2048 1024; do
  SS_AB_FWD_GRID=$cap timeout 600 tools/ubench/_build/neuron16_ab.out 10 32 1 > $O/ab_f16_T10_grid$cap.log 2>&1; echo "cap $cap"; grep -E "^fwd" $O/ab_f16_T10_grid$cap.log | cut -c1-150
done
for cap in 1048576 8192 2048; do
  SS_AB_FWD_GRID=$cap timeout 600 tools/ubench/_build/neuron16_ab.out 5 16 2 > $O/ab_bf16_T5_grid$cap.log 2>&1; echo "cap $cap"; grep -E "^fwd" $O/ab_bf16_T5_grid$cap.log | cut -c1-150
done
