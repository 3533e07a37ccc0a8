#!/usr/bin/env python3
"""Decoder stage forward at config 3's shapes (80 frames): the sub-pixel (merged tap) implicit GEMM (ss_upconv_sub_fwd_f32) against the forms the network
ran before (fused projection + gather for deconv1 / deconv2, exact-split GEMM + gather for deconv3 / deconv4), interleaved rounds, HIP events."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get('SS_LIB'):
    from stereospike_amd import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ['SS_LIB'])
from stereospike_amd import _lib, fused, config
from stereospike_amd.network.blocks import NNConvUpsampling
from oracle import np_pack            # (tools/ script: the packer only builds the input)
dev = 'cuda:0'
ROUNDS, REPS = int(os.environ.get('ROUNDS', 5)), int(os.environ.get('REPS', 5))
only = os.environ.get('ONLY', 'deconv1,deconv2,deconv3').split(',')
geo = {'deconv1': (64, 32, (130, 173), (260, 346)), 'deconv2': (128, 64, (65, 87), (130, 173)), 'deconv3': (256, 128, (33, 44), (65, 87)),
       'deconv4': (512, 256, (17, 22), (33, 44))}
NB = 80
for name in only:
    Cin, Cout, (h, w), (H, W) = geo[name]
    up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(dev)
    tabs = up._tables(h, w, torch.device(dev))
    st = fused.sub_tables(tabs, H, W)
    g = torch.Generator(device=dev).manual_seed(1)
    x = (torch.rand(NB, h, w, Cin, device=dev, generator=g) < 0.2).float()
    xp = torch.from_numpy(np_pack.pack(x.cpu().numpy().reshape(-1)).view(np.int32)).to(dev)
    wt = up.up[1].weight.detach().contiguous()
    y = torch.empty(NB, H, W, Cout, device=dev)

    def sub_dense():
        wm = _lib.upconv_sub_prep(wt, st, Cin, Cout)
        _lib.upconv_sub_fwd(x, None, wm, st, y, NB, Cin, Cout, h, w)

    def sub_packed():
        wm = _lib.upconv_sub_prep(wt, st, Cin, Cout)
        _lib.upconv_sub_fwd(None, xp, wm, st, y, NB, Cin, Cout, h, w)

    def old():
        with torch.no_grad(), config.engine_config(SUB_FWD=False):
            return up.forward_projected_cl(x, spikes_in=True)
    yo = old(); sub_dense(); torch.cuda.synchronize()
    err = float((y - yo).abs().max() / yo.abs().max())
    cases = {'projected form (before)': old, 'sub-pixel, dense fp32 input': sub_dense, 'sub-pixel, packed input': sub_packed}
    times = {k: [] for k in cases}
    for _ in range(ROUNDS):
        for label, fn in cases.items():
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                fn()
            e1.record(); torch.cuda.synchronize()
            times[label].append(e0.elapsed_time(e1) / REPS)
    macs_min = NB * h * w * 25 * Cin * Cout
    macs_sub = NB * H * W * 9 * Cin * Cout
    print(f'{name}: C_in {Cin} C_out {Cout} {h}x{w} -> {H}x{W} x {NB} frames; minimal (projection) {macs_min / 1e9:.1f} G MACs, merged taps {macs_sub / 1e9:.1f} G MACs (x 3 bf16 terms); '
          f'tiles {NB * st["NVB"] * st["NHB"]}; max |sub - before| / max = {err:.2e}')
    for label, ts in times.items():
        t = float(np.median(ts))
        print(f'   {label:34s} {t:7.3f} ms   issued-MFMA rate {2 * 3 * macs_sub / t / 1e9 / 2500:.3f} of peak (merged), useful {2 * 3 * macs_min / t / 1e9 / 2500:.3f} (minimal MACs)')
