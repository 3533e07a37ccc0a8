#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c23; mkdir -p $O
MIN_US=20 DTYPE=f16 T=10 B=32 RATES=1 timeout 900 python tools/profile_ops.py > $O/ops_f16_T10_shapes.log 2>&1; echo "rc $?"; grep -E "aten::(copy_|sum|cat|fill_|add|mul|div|to|contiguous|clone|_to_copy)" $O/ops_f16_T10_shapes.log | cut -c1-200 | head -40
MIN_US=20 DTYPE=bf16 timeout 900 python tools/profile_ops.py > $O/ops_bf16_shapes.log 2>&1; echo "rc $?"; grep -E "aten::(copy_|sum|cat|fill_|add|mul|div|to|contiguous|clone|_to_copy)" $O/ops_bf16_shapes.log | cut -c1-200 | head -30
