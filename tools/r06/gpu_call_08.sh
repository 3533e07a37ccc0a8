#!/bin/bash
# round 6, call 8: ADVICE r05 medium — guard-band stress of the x16 entry points (caching allocator, then one mapping per tensor)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c08; mkdir -p $O
timeout 900 python tools/r06/guard_x16.py > $O/guard_x16.log 2>&1; echo "rc $?"; cat $O/guard_x16.log | cut -c1-300
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 900 python tools/r06/guard_x16.py > $O/guard_x16_nocache.log 2>&1; echo "rc $?"; cat $O/guard_x16_nocache.log | cut -c1-300
