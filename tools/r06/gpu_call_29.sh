#!/bin/bash
# round 6, call 29: the heads' gather / adjoint per head, as shipped
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c29; mkdir -p $O
timeout 600 python tools/r06/bench_heads.py > $O/heads_before.log 2>&1; grep -v amdgpu.ids $O/heads_before.log
