#!/usr/bin/env python3
"""GPU diagnostic: would a Winograd F(2x2, 3x3) data gradient of the bottleneck convs (512 -> 512, 3x3, 17x22 maps, NB = 80) beat
MIOpen's fp32 implicit GEMM (1.14 - 1.2 ms)?  Times the batched transform-domain GEMM [16][7920 x 512] @ [16][512 x 512] (fp32) and a
torch-op version of the two transforms (upper bound of what hand-written transform kernels would cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = 'cuda:0'


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


NB, C, H, W = 80, 512, 17, 22
th, tw = (H + 1) // 2, (W + 1) // 2
T = NB * th * tw
V = torch.randn(16, T, C, device=dev)
U = torch.randn(16, C, C, device=dev)
M = torch.empty(16, T, C, device=dev)
t = timeit(lambda: torch.bmm(V, U, out=M))
print(f'bmm [16][{T} x {C}] @ [16][{C} x {C}] fp32: {t:.3f} ms  ({2 * 16 * T * C * C / t / 1e9:.1f} TFLOP/s)')
V2 = V.permute(1, 0, 2).reshape(T, 16 * C)
for name, fn in (('single GEMM per matrix, loop of 16 mm', lambda: [torch.mm(V[i], U[i], out=M[i]) for i in range(16)]),):
    t = timeit(fn)
    print(f'{name}: {t:.3f} ms')
# copy cost of the transforms (read 61 MB write 260 MB and back), as plain device copies
a = torch.empty(NB, H, W, C, device=dev); b = torch.empty(16, T, C, device=dev)
t = timeit(lambda: (b.copy_(V), a.copy_(a)))
print(f'traffic stand-in for the two transforms: {t:.3f} ms')
