#!/usr/bin/env python3
"""GPU micro-benchmark of the decoder backward of one stage at the config-3 geometries (80 frames): the pieces of the two-kernel form
(adjoint gather -> g_P in HBM; data gradient = fp32 GEMM or ss_gemm6_f32 on g_P; weight gradient = ss_spike_wgrad_f32 on g_P, or the fused
adjoint + weight gradient kernel) against the forms that keep g_P on chip (ss_upconv_bwd_dgrad_f32; ss_upconv_bwd_fused_f32 without its
g_P store).  HIP-event time per launch, interleaved rounds; roofline figures: algorithmic bytes of the stage's backward = g_y + x + g_x
(+ weights), useful bf16 FLOPs = 6 (data gradient) / 3 (weight gradient) products per MAC.  ONLY=deconv1,deconv2 restricts the geometry."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereospike_amd import _lib, fused
if os.environ.get('SS_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['SS_LIB'])
from stereospike_amd.network.blocks import NNConvUpsampling
dev = 'cuda:0'
NB = int(os.environ.get('NB', 80))
rounds, reps = int(os.environ.get('ROUNDS', 4)), int(os.environ.get('REPS', 4))
geoms = [('deconv1', 64, 32, (130, 173), (260, 346)), ('deconv2', 128, 64, (65, 87), (130, 173)),
         ('deconv3', 256, 128, (33, 44), (65, 87)), ('deconv4', 512, 256, (17, 22), (33, 44))]
only = os.environ.get('ONLY')
for name, Cin, Cout, (h, w), (H, W) in geoms:
    if only and name not in only.split(','):
        continue
    torch.manual_seed(0)
    up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(dev)
    tabs = up._tables(h, w, torch.device(dev))
    ext = fused.adjoint_extents(tabs)
    wt = up.up[1].weight.detach().contiguous()
    x = ((torch.rand(NB, h, w, Cin, device=dev) < 0.35).float() + (torch.rand(NB, h, w, Cin, device=dev) < 0.1).float()).contiguous()
    g = torch.randn(NB, H, W, Cout, device=dev)
    R, N = NB * h * w, 25 * Cout
    gP = torch.empty(R, N, device=dev)
    W2 = wt.permute(2, 3, 0, 1).reshape(N, Cin).contiguous()
    gx_a, gx_b = torch.empty(NB, h, w, Cin, device=dev), torch.empty(NB, h, w, Cin, device=dev)
    gw_a, gw_b = torch.empty(Cin, N, device=dev), torch.empty(Cin, N, device=dev)
    t = (tabs[1], tabs[2], tabs[4], tabs[5])
    cases = {'adjoint -> g_P (HBM)': lambda: _lib.upconv_cl_bwd(g, *t, gP, NB, 5, Cout, h, w, H, W),
             'dgrad: fp32 GEMM on g_P': lambda: torch.mm(gP, W2, out=gx_a.view(R, Cin))}
    if _lib.gemm6_supported(N, Cin):
        cases['dgrad: gemm6 on g_P'] = lambda: _lib.gemm6(gP, W2, gx_a.view(R, Cin), R, N, Cin)
    if _lib.spike_wgrad_supported(Cin, N):
        cases['wgrad: spike_wgrad on g_P'] = lambda: _lib.spike_wgrad(gP, x.view(R, Cin), gw_a, R, Cin, N)
    if _lib.upconv_bwd_fused_supported(Cin, Cout, 5, ext):
        cases['adjoint + wgrad fused, g_P stored'] = lambda: _lib.upconv_bwd_fused(g, x, *t, gP, gw_a, NB, Cin, Cout, h, w, H, W)
        cases['adjoint + wgrad fused, NO g_P'] = lambda: _lib.upconv_bwd_fused(g, x, *t, None, gw_b, NB, Cin, Cout, h, w, H, W)
    if _lib.upconv_bwd_dgrad_supported(Cin, Cout, 5, ext):
        cases['adjoint + dgrad fused, NO g_P'] = lambda: _lib.upconv_bwd_dgrad(g, wt, *t, gx_b, NB, Cin, Cout, h, w, H, W)
    # round 4: the box-sum form (ss_upconv_box.hip)
    bt = fused.box_tables(tabs, H, W)
    if _lib.upconv_box_dgrad_supported(Cin, Cout, 5, bt) and _lib.upconv_box_wgrad_supported(Cin, Cout, 5, bt):
        box = _lib.upconv_boxsum(g, bt, NB, Cout, H, W)
        gx_c, gw_c = torch.empty(NB, h, w, Cin, device=dev), torch.empty(Cout, Cin, 5, 5, device=dev)
        cases['box: box-sum planes (HBM)'] = lambda: _lib.upconv_boxsum(g, bt, NB, Cout, H, W)
        cases['box: dgrad on the planes'] = lambda: _lib.upconv_box_dgrad(box, wt, bt, gx_c, NB, Cin, Cout, h, w)
        cases['box: wgrad on the planes'] = lambda: _lib.upconv_box_wgrad(box, x, None, bt, gw_c, NB, Cin, Cout, h, w)
    for f in cases.values():
        f()
    torch.cuda.synchronize()
    if 'box: dgrad on the planes' in cases:
        cases['dgrad: fp32 GEMM on g_P']()
        if 'wgrad: spike_wgrad on g_P' in cases:
            cases['wgrad: spike_wgrad on g_P']()
        torch.cuda.synchronize()
        print(name, 'box dgrad vs fp32 GEMM on g_P: max |diff| / max', float((gx_a - gx_c).abs().max() / gx_a.abs().max()),
              ' box wgrad vs spike_wgrad:', float((gw_a.view(Cin, 5, 5, Cout).permute(3, 0, 1, 2) - gw_c).abs().max() / gw_c.abs().max()), flush=True)
    if 'adjoint + dgrad fused, NO g_P' in cases:
        cases['dgrad: fp32 GEMM on g_P']()
        torch.cuda.synchronize()
        print(name, 'fused dgrad vs fp32 GEMM on g_P: max |diff| / max', float((gx_a - gx_b).abs().max() / gx_a.abs().max()), flush=True)
    best = {k: 1e9 for k in cases}
    for _ in range(rounds):
        for k, f in cases.items():
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            best[k] = min(best[k], e0.elapsed_time(e1) / reps)
    io = 4 * (g.numel() + x.numel() + gx_a.numel())
    macs = R * N * Cin
    print(f'{name}: C_in {Cin} C_out {Cout} rows {R}  g_P {4 * R * N / 1e9:.2f} GB  stage I/O {io / 1e9:.2f} GB  MACs per contraction {macs / 1e9:.1f} G', flush=True)
    for k, ms in best.items():
        terms = 6 if 'dgrad' in k else (3 if 'wgrad' in k else 0)
        if 'box-sum planes' in k:
            print(f'   {k:36s} {ms:7.3f} ms  {(4 * g.numel() + 6 * NB * Cout * bt["NVR"] * bt["NHR"]) / ms / 1e6:7.1f} GB/s (g_y read + planes written)', flush=True)
            continue
        extra = f'  {2 * terms * macs / ms / 1e9:7.1f} TFLOP/s bf16 ({2 * terms * macs / ms / 1e9 / 2500:.3f} of MFMA peak)' if terms and 'fp32' not in k else ''
        print(f'   {k:36s} {ms:7.3f} ms{extra}', flush=True)
