#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench command (run on the GPU box through gpurun).
# usage: profiles/run_profile.sh <out-tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o trace -- python "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
echo "rocprofv3 exit $?" >> "$OUT/bench.log"
ls -R "$OUT" | head -30
