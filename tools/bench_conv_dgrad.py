#!/usr/bin/env python3
"""GPU micro-benchmark of the six-term MFMA data gradient of conv1 .. conv4 (ss_conv_s2_dgrad_f32) against MIOpen's fp32 data gradient at the
config-3 geometries (80 frames): HIP-event time per launch, interleaved rounds; useful bf16 FLOPs = 6 cross terms per MAC.  Also a float64
check on a small case per layer shape."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from stereospike_amd import _lib
if os.environ.get('SS_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SS_LIB'])
dev = 'cuda:0'
torch.backends.cudnn.benchmark = True
NB = int(os.environ.get('NB', 80))
LAYERS = [('conv1', 32, (260, 346)), ('conv2', 64, (130, 173)), ('conv3', 128, (65, 87)), ('conv4', 256, (33, 44))]


def ref_dgrad(g, wt, h, w, dtype):
    NBq, Cin = g.shape[0], wt.shape[1]
    return F.conv_transpose2d(g.permute(0, 3, 1, 2).to(dtype), wt.to(dtype), None, 2, 2,
                              output_padding=((h - 1) % 2, (w - 1) % 2)).permute(0, 2, 3, 1)


def check(Cin, h, w, nb=2):
    Cout = 2 * Cin
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    gen = torch.Generator(device=dev).manual_seed(Cin + h)
    g = torch.randn(nb, ho, wo, Cout, device=dev, generator=gen) * torch.exp(2 * torch.randn(nb, ho, wo, 1, device=dev, generator=gen))
    wt = torch.randn(Cout, Cin, 5, 5, device=dev, generator=gen) * 0.05
    for cb in ('8', '32'):
        os.environ['SS_DGRAD_CB'] = cb
        gx = torch.full((nb, h, w, Cin), float('nan'), device=dev)
        _lib.conv_s2_dgrad(g, wt, gx, nb, Cin, Cout, h, w)
        ref = ref_dgrad(g, wt, h, w, torch.float64)
        mag = ref_dgrad(g.abs(), wt.abs(), h, w, torch.float64)
        err = (gx.double() - ref).abs()
        ok = bool(torch.isfinite(gx).all())
        print(f'   check C_in {Cin} {h}x{w} CB {cb}: finite {ok}, max err / (2^-21 mag) = {float((err / (mag * 2.0 ** -21 + 1e-300)).max()):.3f}, '
              f'rel L2 {float(err.pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()):.2e}', flush=True)
    os.environ.pop('SS_DGRAD_CB')


for name, Cin, (h, w) in LAYERS:
    check(Cin, 2 * (h // 8) + 1, 2 * (w // 8) + (1 if Cin == 64 else 0))
    Cout = 2 * Cin
    torch.manual_seed(0)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    g = torch.randn(NB, ho, wo, Cout, device=dev) * 1e-3
    wt = torch.randn(Cout, Cin, 5, 5, device=dev) * 0.05
    gx = torch.empty(NB, h, w, Cin, device=dev)
    w_cl = wt.contiguous(memory_format=torch.channels_last)
    x_meta = torch.empty((NB, Cin, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)

    def miopen():
        return torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), x_meta, w_cl, None, [2, 2], [2, 2], [1, 1], False, [0, 0], 1,
                                                   [True, False, False])[0]

    def own(cb):
        def f():
            os.environ['SS_DGRAD_CB'] = cb
            _lib.conv_s2_dgrad(g, wt, gx, NB, Cin, Cout, h, w)
        return f
    cases = {'MIOpen fp32 dgrad (NHWC)': miopen, 'conv_s2_dgrad CB 8': own('8'), 'conv_s2_dgrad CB 32': own('32')}
    for f in cases.values():
        f()
    torch.cuda.synchronize()
    ref = miopen().permute(0, 2, 3, 1)
    own('8')()
    print(name, 'max |diff| / max vs MIOpen', float((ref - gx).abs().max() / ref.abs().max()), flush=True)
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
        t = 6 if 'conv_s2' in k else 1
        print(f'   {name} {k:28s} {ms:7.3f} ms   {2 * t * macs / ms / 1e9:7.1f} TFLOP/s {"bf16 (%.3f of MFMA peak)" % (2 * t * macs / ms / 1e9 / 2500) if t == 6 else "fp32"}', flush=True)
