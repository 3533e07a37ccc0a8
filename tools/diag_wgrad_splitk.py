#!/usr/bin/env python3
"""Probe: split-K for the bf16x3 weight-gradient GEMM of the spike convs  g_W[K, 3N] = A^T[K, M] @ g3[M, 3N]  (M = NB*ho*wo rows)."""
import os, sys, torch
dev = 'cuda:0'
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, M, K, N in (('bottleneck', 29920, 4608, 512), ('conv4', 29920, 6400, 512), ('conv3', 116160, 3200, 256), ('conv2', 452400, 1600, 128)):
    A = (torch.rand(M, K, device=dev) < 0.2).to(torch.bfloat16)
    g3 = torch.randn(M, 3 * N, device=dev).to(torch.bfloat16)
    W3 = torch.randn(K, 3 * N, device=dev).to(torch.bfloat16)
    line = f'{name:10s} M={M} K={K} 3N={3 * N}: fwd {timeit(lambda: torch.mm(A, W3, out_dtype=torch.float32)):.3f} ms | wgrad'
    for S in (1, 2, 4, 8, 16, 32):
        if M % S:
            continue
        L = M // S
        if S == 1:
            fn = lambda: torch.mm(A.t(), g3, out_dtype=torch.float32)
        else:
            fn = lambda: torch.bmm(A.view(S, L, K).transpose(1, 2), g3.view(S, L, 3 * N), out_dtype=torch.float32).sum(0)
        line += f'  S={S}: {timeit(fn):.3f}'
    print(line, flush=True)
