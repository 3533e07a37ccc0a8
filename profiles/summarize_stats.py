#!/usr/bin/env python3
"""Per-category summary of a `rocprofv3 --kernel-trace --stats` CSV of bench.py.
usage: summarize_stats.py <trace_kernel_stats.csv> <iterations = warmup + steps>
Kernels launched by MIOpen's find search (start-up only) are removed by keeping, for every kernel name, only
floor(Calls / iterations) * iterations launches (steady-state kernels run the same number of times every iteration)."""
import csv
import sys

path, iters = sys.argv[1], int(sys.argv[2])


def cat(n):
    if 'upconv' in n and 'bwd' in n: return 'up-conv gather adjoint (ours)'
    if 'upconv' in n and 'fwd' in n: return 'up-conv gather (ours)'
    if 'neuron_' in n: return 'fused neuron kernels (ours)'
    if 'ipool' in n or 'gk_finish' in n: return 'I-pool (ours)'
    if n.startswith('Cijk'): return 'rocBLAS / hipBLASLt GEMM (decoder projections)'
    if 'transpose' in n: return 'MIOpen layout transposes'
    if 'igemm' in n or 'conv' in n.lower() or 'miopen' in n.lower(): return 'MIOpen conv (encoder, bottleneck)'
    if 'elementwise' in n or 'reduce_kernel' in n or 'ill' in n or 'copy' in n.lower(): return 'torch element-wise / reduce / copy'
    if 'adam' in n.lower() or 'multi_tensor' in n: return 'Adam (fused)'
    return 'other'


tot, cats, dropped = 0.0, {}, 0.0
for r in csv.DictReader(open(path)):
    calls, avg = int(r['Calls']), float(r['AverageNs'])
    steady = (calls // iters) * iters
    dropped += (calls - steady) * avg
    t = steady * avg
    tot += t
    cats[cat(r['Name'])] = cats.get(cat(r['Name']), 0.0) + t
print(f'| category | ms/step | share |\n|---|---|---|')
for k, v in sorted(cats.items(), key=lambda kv: -kv[1]):
    print(f'| {k} | {v / iters / 1e6:.2f} | {100 * v / tot:.1f} % |')
print(f'| **sum of steady-state kernel time** | **{tot / iters / 1e6:.2f}** | |')
print(f'\n(start-up-only kernels removed: {dropped / 1e9:.1f} s in total)')
