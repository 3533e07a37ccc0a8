#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c10; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc $?"; tail -6 $O/pytest_gpu.log
