#!/bin/bash
# Regenerates the TunableOp record of the decoder's GEMM shapes on an MI355X (run through gpurun): times every hipBLASLt / rocBLAS
# solution per shape for the configurations below and merges the winners into gpurun_out/tunableop_results.csv; copy that file to
# stereospike_amd/tunableop/tunableop_results.csv to ship it.
set -u
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=${TUNE_MS:-100} PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python bench.py --no-cpu-baseline --gemm-tuning 2 --warmup 2 --steps 3 2>/dev/null | cut -c1-160
python bench.py --no-cpu-baseline --gemm-tuning 2 --warmup 2 --steps 3 --dtype f16 --T 10 --batch 32 2>/dev/null | cut -c1-160
python bench.py --no-cpu-baseline --gemm-tuning 2 --warmup 2 --steps 3 --dtype bf16 2>/dev/null | cut -c1-160
wc -l gpurun_out/tunableop_results.csv
