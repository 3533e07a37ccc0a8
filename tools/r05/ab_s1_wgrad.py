#!/usr/bin/env python3
"""Timing of the first layer's weight gradient (C_in = 4, BASELINE config 3: NB = 80, 260 x 346), fp32 and bf16 modes.  SS_S1_WGRAD_TR=0 selects the first
form, default the window / transposed-read form: run once per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stereospike_amd import _lib
dev = 'cuda:0'
NB, h, w, Cin, Cout = 80, 260, 346, 4, 32
x = torch.poisson(torch.full((NB, h, w, Cin), 0.3, device=dev))
gw = torch.empty(Cout, Cin, 5, 5, device=dev)
out = []
for mode in ('f32', 'bf16'):
    g = (torch.randn(NB, h, w, Cout, device=dev) * 1e-3).to(torch.float32 if mode == 'f32' else torch.bfloat16)
    fn = (lambda: _lib.dense_conv_s1_wgrad(g, x, gw, NB, Cin, Cout, h, w)) if mode == 'f32' else (lambda: _lib.dense_conv_s1_wgrad_x16(g, x, gw, NB, Cin, Cout, h, w))
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    out.append(f'{mode} {best:6.3f} ms')
print('SS_S1_WGRAD_TR=' + os.environ.get('SS_S1_WGRAD_TR', '1 (default)'), ' | '.join(out), flush=True)
