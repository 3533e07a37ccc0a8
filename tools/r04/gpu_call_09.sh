#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command + the two PMC traffic passes (separate runs, no trace domains with --pmc) + the decoder-stage test
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "decoder_stage" 2>&1 | tail -3
bash profiles/run_profile.sh r04a --steps 5 --warmup 2 | tail -5
bash profiles/collect_pmc.sh r04a rc | tail -40
