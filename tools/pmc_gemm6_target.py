#!/usr/bin/env python3
"""Target of an SQ-counter pass: a few launches of ss_gemm6_f32 at the deconv3 data-gradient shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereospike_amd import _lib
dev = 'cuda:0'
R, K, N = 80 * 33 * 44, 3200, 256
A = torch.randn(R, K, device=dev); B = torch.randn(K, N, device=dev) * 0.05; C = torch.empty(R, N, device=dev)
for _ in range(4):
    _lib.gemm6(A, B, C, R, K, N)
torch.cuda.synchronize()
