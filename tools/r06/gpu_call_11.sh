#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c11; mkdir -p $O
STACK=1 MIN_US=25 DTYPE=f16 T=10 B=32 RATES=1 timeout 900 python tools/profile_ops.py > $O/ops_f16_T10.log 2>&1; echo "rc $?"; tail -45 $O/ops_f16_T10.log | cut -c1-420
STACK=1 MIN_US=15 timeout 900 python tools/profile_ops.py > $O/ops_f32.log 2>&1; echo "rc $?"; tail -45 $O/ops_f32.log | cut -c1-420
