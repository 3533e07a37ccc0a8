#!/usr/bin/env python3
"""Timing-only run of ss_conv_s2_dgrad_f32 at the config-3 geometries for a (possibly ablated) build: SS_LIB=... python tools/_abl_dgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereospike_amd import _lib
if os.environ.get('SS_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SS_LIB'])
dev = 'cuda:0'
NB = 80
out = []
for name, Cin, (h, w) in [('conv1', 32, (260, 346)), ('conv2', 64, (130, 173)), ('conv3', 128, (65, 87)), ('conv4', 256, (33, 44))]:
    Cout = 2 * Cin
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    g = torch.randn(NB, ho, wo, Cout, device=dev) * 1e-3
    wt = torch.randn(Cout, Cin, 5, 5, device=dev) * 0.05
    gx = torch.empty(NB, h, w, Cin, device=dev)
    for _ in range(2):
        _lib.conv_s2_dgrad(g, wt, gx, NB, Cin, Cout, h, w)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(4):
            _lib.conv_s2_dgrad(g, wt, gx, NB, Cin, Cout, h, w)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    out.append(f'{name} {best:6.3f}')
print(os.environ.get('SS_LIB', 'default'), ' | '.join(out), flush=True)
