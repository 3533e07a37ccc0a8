#!/bin/bash
# full GPU suite (the driver's own command line), the default bench, and the rocprofv3 kernel-trace summary of the same bench command
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04/pytest_gpu_full.log 2>&1
tail -5 gpurun_out/r04/pytest_gpu_full.log
timeout 900 python bench.py > gpurun_out/r04/bench_r04_a.json 2> gpurun_out/r04/bench_r04_a.err
tail -c 600 gpurun_out/r04/bench_r04_a.json | head -c 300; echo
rm -rf gpurun_out/r04/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/r04/prof_bench -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04/bench_under_rocprof.json 2> gpurun_out/r04/bench_under_rocprof.err
find gpurun_out/r04/prof_bench -name "*kernel_stats.csv" | head; find gpurun_out/r04/prof_bench -name "*kernel_trace.csv" -size +20M -delete
find gpurun_out/r04/prof_bench -name "*.csv" -size +20M -delete
f=$(find gpurun_out/r04/prof_bench -name "*kernel_stats.csv" | head -1); head -40 "$f"
