#!/bin/bash
# HBM traffic of the fused kernels from hardware counters, collected exactly as MI355X_MICROARCH.md §HBM prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), no trace domains combined with --pmc;
# gfx950 correction: FETCH_SIZE counts 64 B per 128-B request => double it; units are KiB.
# usage (GPU box): profiles/collect_pmc.sh <tag> [rc|saveh]   ->  gpurun_out/pmc_<tag>/{fetch,write}/..., gpurun_out/pmc_<tag>/pmc_traffic.json
set -u
TAG=${1:-r01}
MODE=${2:-rc}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o pmc -- python "$REPO/tools/pmc_target.py" $MODE > "$OUT/fetch.log" 2>&1
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o pmc -- python "$REPO/tools/pmc_target.py" $MODE > "$OUT/write.log" 2>&1
python "$REPO/profiles/parse_pmc.py" "$OUT" $MODE | tee "$OUT/pmc_traffic.json"
