#!/usr/bin/env python3
"""Timing of the stride-2 encoder weight gradient on PACKED spikes at BASELINE config 3's conv1 / conv2 geometries (NB = 80), fp32 and bf16 modes.
SS_SPIKE_WGRAD_TR=0 selects the first form (xprep + gprep + global-memory fragments), default the window / transposed-read form: run once per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from stereospike_amd import _lib
dev = 'cuda:0'
NB = 80
out = []
for name, Cin, (h, w) in [('conv1', 32, (260, 346)), ('conv2', 64, (130, 173))]:
    Cout = 2 * Cin
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    xp = torch.randint(-2 ** 31, 2 ** 31 - 1, (NB * h * w * Cin // 16,), dtype=torch.int32, device=dev)
    gw = torch.empty(Cout, Cin, 5, 5, device=dev)
    for mode in ('f32', 'bf16'):
        g = (torch.randn(NB, ho, wo, Cout, device=dev) * 1e-3).to(torch.float32 if mode == 'f32' else torch.bfloat16)
        fn = (lambda: _lib.spike_conv_wgrad(g, None, gw, NB, Cin, Cout, h, w, x_packed=xp)) if mode == 'f32' else \
             (lambda: _lib.spike_conv_wgrad_x16(g, None, gw, NB, Cin, Cout, h, w, x_packed=xp))
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
        out.append(f'{name} {mode} {best:6.3f} ms')
print('SS_SPIKE_WGRAD_TR=' + os.environ.get('SS_SPIKE_WGRAD_TR', '1 (default)'), ' | '.join(out), flush=True)
