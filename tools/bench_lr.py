#!/usr/bin/env python3
"""Microbenchmark of ss_neuron_bwd_fork_lr_f32 against ss_neuron_bwd_fork_f32 at the config-3 bottom-layer size (B16 x T5 x 32x260x346),
IF and PLIF, HIP-event timed: us per launch and TB/s over the bytes each form moves."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereospike_amd import _lib
dev = 'cuda:0'
T, C = 5, 32
N = 16 * C * 260 * 346
x = torch.randn(T, N, device=dev) * 0.15
g, g2, gx = torch.randn(T, N, device=dev), torch.randn(T, N, device=dev), torch.empty(T, N, device=dev)
lr_p, lr_w = torch.randn(T, N // C, 9, device=dev), torch.randn(9, C, device=dev)
kk = torch.tensor([0.4], device=dev)
ws = torch.empty(_lib.gk_ws_floats(), device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for kind, name in ((0, 'IF'), (2, 'PLIF')):
    k = kk if kind == 2 else None
    gk = torch.zeros(1, device=dev) if kind == 2 else None
    args = (T, N, 10.0, kind, 2.0, k, 1.0, 0.0, 0, 2.0, True)
    t_fork = timeit(lambda: _lib.neuron_bwd_fork(g, g2, None, None, None, x, None, gx, None, gk, ws if kind == 2 else None, *args))
    t_lr = timeit(lambda: _lib.neuron_bwd_fork_lr(g, lr_p, lr_w, None, None, x, None, gx, None, gk, ws if kind == 2 else None, *args))
    t_only = timeit(lambda: _lib.neuron_bwd_fork_lr(None, lr_p, lr_w, None, None, x, None, gx, None, gk, ws if kind == 2 else None, *args))
    b = T * N
    print(f'{name}: fork 16 B {t_fork:7.1f} us {16 * b / t_fork / 1e6:5.2f} TB/s | lr 13.1 B {t_lr:7.1f} us {13.125 * b / t_lr / 1e6:5.2f} TB/s '
          f'(12 B definition {12 * b / t_lr / 1e6 / 8:.3f} of peak) | lr-only 9.1 B {t_only:7.1f} us {9.125 * b / t_only / 1e6:5.2f} TB/s')
