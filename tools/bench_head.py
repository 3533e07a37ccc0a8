#!/usr/bin/env python3
"""GPU micro-benchmark of the full-resolution prediction head on packed spikes (ss_head_proj_packed_f32 / ss_head_wgrad_packed_f32) against the
library GEMMs on the dense tensor at the config-3 geometry (80 frames x 260 x 346 pixels, 32 channels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from stereospike_amd import _lib
from oracle import np_pack
dev = 'cuda:0'
rows, C = 80 * 260 * 346, 32
x = (torch.rand(rows, C, device=dev) < 0.3).float() + (torch.rand(rows, C, device=dev) < 0.1).float()
xp = torch.from_numpy(np_pack.pack(x.cpu().numpy().reshape(-1)).view(np.int32)).to(dev)
Wt = torch.randn(C, 9, device=dev) * 0.1
g = torch.randn(rows, 9, device=dev) * 1e-5
P, gW = torch.empty(rows, 9, device=dev), torch.empty(C, 9, device=dev)
S = rows // 8192
cases = {'library GEMM  P = x @ Wt (dense x)': lambda: torch.mm(x, Wt, out=P),
         'head_proj_packed': lambda: _lib.head_proj_packed(xp, Wt, P, rows, C),
         'library split-K g_W = x^T @ g_P': lambda: torch.bmm(x[:S * 8192].view(S, 8192, C).transpose(1, 2), g[:S * 8192].view(S, 8192, 9)).sum(0),
         'head_wgrad_packed': lambda: _lib.head_wgrad_packed(xp, g, gW, rows, C)}
for f in cases.values():
    f()
torch.cuda.synchronize()
for k, f in cases.items():
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(4):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    print(f'{k:40s} {best:7.3f} ms', flush=True)
