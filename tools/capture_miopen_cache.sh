#!/bin/bash
# After a find-mode run on the GPU box: pack the in-tree MIOpen user db / kernel cache into gpurun_out/ so it can be
# brought back and travel with later snapshots (stereospike_amd/lib/miopen_cache/, git-ignored like the built .so).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO/stereospike_amd/lib" && du -sh miopen_cache && tar czf "$REPO/gpurun_out/miopen_cache.tgz" miopen_cache && ls -la "$REPO/gpurun_out/miopen_cache.tgz"
