#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
timeout 900 python tools/r05/repro_graph.py > gpurun_out/r05/repro_graph_3.log 2>&1
cat gpurun_out/r05/repro_graph_3.log | cut -c1-250
