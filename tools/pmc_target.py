#!/usr/bin/env python3
"""Target of the rocprofv3 --pmc passes: a few launches of the fused forward (train) and backward kernels at the
config-3 bottom-layer size (B16 x T5 x 32x260x346 = 2.3e8 updates).  Mode (argv[1]):
  rc    (default, what training runs for the dominant layer): forward without h_seq (8 B/update = 1.84 GB) + backward recomputing h
          from x with the second consumer's gradient added on load (ss_neuron_bwd_fork_f32: 16 B/update = 3.68 GB)
  saveh : forward writing h_seq (12 B/update) + backward reading it (12 B/update)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereospike_amd import _lib
dev = 'cuda:0'
T, N = 5, 16 * 32 * 260 * 346
x = torch.randn(T, N, device=dev) * 0.15
out, h, g, gx = torch.empty_like(x), torch.empty_like(x), torch.randn(T, N, device=dev), torch.empty_like(x)
v = torch.empty(N, device=dev)
g2 = torch.randn(T, N, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else 'rc'
for _ in range(5):
    if mode == 'rc':
        _lib.neuron_fwd(x, None, None, out, None, v, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)
        _lib.neuron_bwd_fork(g, g2, None, None, None, x, None, gx, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)
    else:
        _lib.neuron_fwd(x, None, None, out, h, v, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)
        _lib.neuron_bwd(g, None, h, None, gx, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)
torch.cuda.synchronize()
print('mode', mode, 'algorithmic bytes per launch: fwd', (8 if mode == 'rc' else 12) * T * N, 'bwd', (16 if mode == 'rc' else 12) * T * N)
