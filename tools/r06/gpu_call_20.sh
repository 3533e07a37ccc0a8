#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c20; mkdir -p $O
for cap in 1048576 16384 8192 4096 2048; do
  SS_AB_BWD_GRID=$cap timeout 600 tools/ubench/_build/neuron16_ab.out 10 32 1 > $O/ab_f16_T10_bgrid$cap.log 2>&1; echo "bwd cap $cap"; grep -E "^bwd (lr  V4 S2 w3|lr\+sum|fork V4 S2 w1|rc  V4 S2 w4|fwd)|^fwd" $O/ab_f16_T10_bgrid$cap.log | cut -c1-150
done
for cap in 1048576 8192 2048; do
  SS_AB_BWD_GRID=$cap timeout 600 tools/ubench/_build/neuron16_ab.out 5 16 2 > $O/ab_bf16_T5_bgrid$cap.log 2>&1; echo "bwd cap $cap"; grep -E "^bwd (lr  V4 S1 w4|lr\+sum V4|fork V4 S1|rc  V4 S1)|^fwd" $O/ab_bf16_T5_bgrid$cap.log | cut -c1-150
done
timeout 1500 python -m pytest tests/test_gpu_07_round6.py tests/test_gpu_06_x16_kernels.py -q -k "neuron or packed_spike or low_rank or round6 or membrane or fwd16 or lr_x16 or lazy" > $O/pytest_neuron16.log 2>&1; echo "rc $?"; tail -3 $O/pytest_neuron16.log
