#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c09; mkdir -p $O
for P in 0 1 2; do
  timeout 600 tools/ubench/_build/neuron16_ab_pipe$P.out 10 32 1 > $O/ab_pipe${P}_f16_T10.log 2>&1; echo "pipe $P f16 T10 rc $?"; grep -E "^bwd lr" $O/ab_pipe${P}_f16_T10.log
  timeout 600 tools/ubench/_build/neuron16_ab_pipe$P.out 5 16 2 > $O/ab_pipe${P}_bf16_T5.log 2>&1; echo "pipe $P bf16 T5 rc $?"; grep -E "^bwd lr" $O/ab_pipe${P}_bf16_T5.log
done
bash tools/r06/gpu_call_08.sh
