#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 900 python tools/r05/repro_graph.py > gpurun_out/r05/repro_graph_5_aliases.log 2>&1
cat gpurun_out/r05/repro_graph_5_aliases.log | cut -c1-250
timeout 1200 python -m pytest tests/test_gpu_05_full_size.py tests/test_gpu_03_model.py tests/test_gpu_00_default_path.py -q -m gpu -x 2>&1 | tail -8 | cut -c1-250
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --dtype bf16 --model PLIFNet --T 1 --batch 8 --graph 1 2>/dev/null | cut -c1-200
