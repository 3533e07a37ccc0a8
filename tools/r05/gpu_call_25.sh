#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05/c25
timeout 300 python tools/r05/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/c25/overlap_probe.log
