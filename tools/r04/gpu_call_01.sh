#!/bin/bash
# round 4, GPU call 1: probes + the new tests + the new bench line
mkdir -p gpurun_out/r04
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w tools/r04/probe_tr16.hip -o /tmp/tr16 && /tmp/tr16 > gpurun_out/r04/tr16.log 2>&1
python tools/r04/probe_determinism.py > gpurun_out/r04/determinism.log 2>&1
python -m pytest tests/test_gpu_00_default_path.py -q -m gpu -k "penalize" > gpurun_out/r04/pytest_penalize.log 2>&1
python -m pytest tests/test_gpu_05_full_size.py -q -m gpu > gpurun_out/r04/pytest_full_size.log 2>&1
python -m pytest tests/test_gpu_01_kernels.py -q -m gpu -k "spike_conv_as_exact" > gpurun_out/r04/pytest_spikeconv.log 2>&1
python bench.py > gpurun_out/r04/bench_call01.json 2> gpurun_out/r04/bench_call01.err
tail -3 gpurun_out/r04/*.log
