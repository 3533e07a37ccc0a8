#!/bin/bash
# round 6, call 25: 16-bit box kernels, chunk groups as window planes: parity (tests/test_gpu_06_x16_kernels.py box tests) + A/B timing
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06/c25; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_06_x16_kernels.py -x -q -m gpu -k "box" > $O/pytest_box.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_box.log | cut -c1-300
SS_BOX_X16_NG1=1 timeout 600 python tools/r06/bench_box_x16.py > $O/box_x16_ng1.log 2>&1; cat $O/box_x16_ng1.log | grep -v amdgpu.ids
SS_BOX_X16_NG1=0 timeout 600 python tools/r06/bench_box_x16.py > $O/box_x16_ng2.log 2>&1; cat $O/box_x16_ng2.log | grep -v amdgpu.ids
