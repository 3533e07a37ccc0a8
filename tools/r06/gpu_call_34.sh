#!/bin/bash
# round 6, call 34: the adjoint gather with the next window row prefetched: parity tests + A/B against the variant build without
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c34; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_01_kernels.py -x -q -m gpu -k "upconv" > $O/pytest_sel.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_sel.log | cut -c1-300
SS_LIB=stereospike_amd/lib/libss_neuron_nopf.so timeout 600 python tools/r06/bench_adjoint.py > $O/adjoint_nopf.log 2>&1; grep -v amdgpu.ids $O/adjoint_nopf.log
timeout 600 python tools/r06/bench_adjoint.py > $O/adjoint_pf.log 2>&1; grep -v amdgpu.ids $O/adjoint_pf.log
