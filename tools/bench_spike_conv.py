#!/usr/bin/env python3
"""GPU micro-benchmark of the exact MFMA forward of conv1 / conv2 (ss_spike_conv_fwd_f32) against MIOpen's fp32 convolution at the config-3
geometries (80 frames): HIP-event time per launch, interleaved rounds; useful bf16 FLOPs = 3 exact products per MAC."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stereospike_amd import _lib
if os.environ.get('SS_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SS_LIB'])
from oracle import np_pack
import numpy as np
dev = 'cuda:0'
torch.backends.cudnn.benchmark = True
NB = int(os.environ.get('NB', 80))
for name, Cin, Cout, (h, w) in [('conv1', 32, 64, (260, 346)), ('conv2', 64, 128, (130, 173))]:
    torch.manual_seed(0)
    x = (torch.rand(NB, h, w, Cin, device=dev) < 0.3).float()
    wt = torch.randn(Cout, Cin, 5, 5, device=dev) * 0.05
    xp = torch.from_numpy(np_pack.pack(x.cpu().numpy().reshape(-1)).view(np.int32)).to(dev)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty(NB, ho, wo, Cout, device=dev)
    w_cl = wt.contiguous(memory_format=torch.channels_last)
    cases = {'MIOpen fp32 conv (NHWC)': lambda: F.conv2d(x.permute(0, 3, 1, 2), w_cl, None, 2, 2),
             'spike_conv_fwd, dense fp32 input': lambda: _lib.spike_conv_fwd(x, None, wt, y, NB, Cin, Cout, h, w),
             'spike_conv_fwd, packed input': lambda: _lib.spike_conv_fwd(None, xp, wt, y, NB, Cin, Cout, h, w)}
    for f in cases.values():
        f()
    torch.cuda.synchronize()
    ref = cases['MIOpen fp32 conv (NHWC)']().permute(0, 2, 3, 1)
    print(name, 'max |diff| / max vs MIOpen', float((ref - y).abs().max() / ref.abs().max()), flush=True)
    best = {k: 1e9 for k in cases}
    for _ in range(4):
        for k, f in cases.items():
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(4):
                f()
            e1.record()
            torch.cuda.synchronize()
            best[k] = min(best[k], e0.elapsed_time(e1) / 4)
    macs = NB * ho * wo * 25 * Cin * Cout
    for k, ms in best.items():
        t = 3 if 'spike' in k else 1
        print(f'   {name} {k:34s} {ms:7.3f} ms   {2 * t * macs / ms / 1e9:7.1f} TFLOP/s {"bf16 (%.3f of MFMA peak)" % (2 * t * macs / ms / 1e9 / 2500) if t == 3 else "fp32"}', flush=True)

for name, Cin, (h, w) in [('bottom', 4, (260, 346))]:
    x = torch.poisson(torch.full((NB, h, w, Cin), 0.05, device=dev))
    wt = torch.randn(32, Cin, 5, 5, device=dev) * 0.1
    y = torch.empty(NB, h, w, 32, device=dev)
    w_cl = wt.contiguous(memory_format=torch.channels_last)
    cases = {'MIOpen fp32 conv (NHWC)': lambda: F.conv2d(x.permute(0, 3, 1, 2), w_cl, None, 1, 2),
             'dense_conv_s1_fwd (six-term MFMA)': lambda: _lib.dense_conv_s1_fwd(x, wt, y, NB, Cin, 32, h, w)}
    for f in cases.values():
        f()
    torch.cuda.synchronize()
    ref = cases['MIOpen fp32 conv (NHWC)']().permute(0, 2, 3, 1)
    print(name, 'max |diff| / max vs MIOpen', float((ref - y).abs().max() / ref.abs().max()), flush=True)
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
        print(f'   {name} {k:34s} {best:7.3f} ms   output {4 * y.numel() / best / 1e6:7.1f} GB/s', flush=True)

    g = torch.randn(NB, h, w, 32, device=dev) * 1e-4
    gw = torch.empty(32, Cin, 5, 5, device=dev)
    cases = {'MIOpen fp32 weight gradient (NHWC)': lambda: torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w_cl, None, [1, 1], [2, 2], [1, 1],
                                                                                               False, [0, 0], 1, [False, True, False])[1],
             'dense_conv_s1_wgrad (six-term MFMA)': lambda: _lib.dense_conv_s1_wgrad(g, x, gw, NB, Cin, 32, h, w)}
    for f in cases.values():
        f()
    torch.cuda.synchronize()
    ref = cases['MIOpen fp32 weight gradient (NHWC)']()
    print(name, 'weight gradient max |diff| / max vs MIOpen', float((ref - gw).abs().max() / ref.abs().max()), flush=True)
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
        print(f'   {name} {k:38s} {best:7.3f} ms', flush=True)
