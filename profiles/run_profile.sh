#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench command (run on the GPU box through gpurun).
# usage: profiles/run_profile.sh <out-tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
echo "rocprofv3 exit $?" >> "$OUT/bench.log"
# keep only the summaries (the raw trace is tens of MB; gpurun_out must stay under 64 MiB)
find "$OUT" -type f ! -name '*stats*.csv' ! -name 'bench.log' ! -name '*agent_info*.csv' -delete
find "$OUT" -type d -empty -delete
ls -R "$OUT" | head -30
