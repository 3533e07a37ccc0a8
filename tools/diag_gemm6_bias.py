#!/usr/bin/env python3
"""GPU diagnostic: signed error statistics of ss_gemm6_f32 vs the library fp32 GEMM against float64 (bias shows up in cancelling sums)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereospike_amd import _lib
dev = 'cuda:0'
torch.manual_seed(0)
for R, K, N in ((480, 3200, 256), (120, 6400, 512), (4000, 3200, 256)):
    A = torch.randn(R, K, device=dev) * 1e-6 * torch.exp(torch.randn(R, 1, device=dev))
    B = torch.randn(K, N, device=dev) * 0.02
    ref = A.double() @ B.double()
    mag = A.double().abs() @ B.double().abs()
    C6 = torch.empty(R, N, device=dev); _lib.gemm6(A, B, C6, R, K, N)
    CL = A @ B
    for tag, C in (('gemm6', C6), ('lib32', CL)):
        e = (C.double() - ref) / mag
        print(f'{R}x{K}x{N} {tag}: mean signed err / mag {float(e.mean()):+.3e}   rms {float(e.pow(2).mean().sqrt()):.3e}   max {float(e.abs().max()):.3e}   '
              f'total-sum rel err {float(((C.double() - ref).sum() / ref.sum()).abs()):.3e}')
