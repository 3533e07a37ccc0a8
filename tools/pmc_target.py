#!/usr/bin/env python3
"""Target of the rocprofv3 --pmc passes: a few launches of the fused forward (train) and backward kernels at the
config-3 bottom-layer size (B16 x T5 x 32x260x346 = 2.3e8 updates).  Mode (argv[1]):
  rc    (default, what training runs for the dominant layer): forward without h_seq (8 B/update = 1.84 GB) + backward recomputing h
          from x with the second consumer's gradient added on load (ss_neuron_bwd_fork_f32: 16 B/update = 3.68 GB)
  saveh : forward writing h_seq (12 B/update) + backward reading it (12 B/update)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereospike_amd import _lib
dev = 'cuda:0'
T, N = 5, 16 * 32 * 260 * 346
x = torch.randn(T, N, device=dev) * 0.15
out, h, g, gx = torch.empty_like(x), torch.empty_like(x), torch.randn(T, N, device=dev), torch.empty_like(x)
v = torch.empty(N, device=dev)
g2 = torch.randn(T, N, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else 'rc'
if mode == 'x16c5':
    # round 6: the fused neuron kernels at BASELINE config 5's per-GPU layer shape — fp16 activations, T = 10, batch 32, the 32 x 260 x 346 layer (9.2e8 updates):
    # packed-only forward without / with a packed skip (membrane unwritten, as the training step runs it: EngineConfig.LAZY_MEMBRANE) and the low-rank backward.
    # Counters of this mode: profiles/parse_pmc.py <dir> x16c5 -> profiles/pmc_traffic_x16.json
    del x, out, h, g, gx, v, g2
    T, N = 10, 32 * 32 * 260 * 346
    dt = torch.float16
    x16 = (torch.randn(T, N, device=dev) * 0.06).to(dt)
    g16 = (torch.randn(T, N, device=dev) * 1e-3).to(dt)
    gx16 = torch.empty(T, N, dtype=dt, device=dev)
    pk, skp = torch.empty(T, N // 16, dtype=torch.int32, device=dev), torch.randint(0, 2 ** 31 - 1, (T, N // 16), dtype=torch.int32, device=dev) & 0x55555555
    lr_p, lr_w = torch.randn(T, N // 32, 9, device=dev) * 1e-3, torch.randn(9, 32, device=dev) * 0.3
    for _ in range(5):
        _lib.neuron_fwd_ex(x16, None, None, None, None, pk, None, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)
        _lib.neuron_fwd_ex(x16, None, None, skp, None, pk, None, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)
        _lib.neuron_bwd_fork_lr_x16(g16, lr_p, lr_w, None, None, x16, None, gx16, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)
    torch.cuda.synchronize()
    sys.exit(0)
if mode == 'x16':
    # round 5: the 16-bit activation modes' own kernels (bf16) at the same config-3 shapes — packed-only neuron forward, low-rank forked backward, first layer,
    # conv1 forward / data gradient, deconv1 sub-pixel forward and box-sum backward.  Counters of this mode: profiles/parse_pmc.py <dir> x16
    import numpy as np
    from oracle import np_pack
    from stereospike_amd import fused
    from stereospike_amd.network.blocks import NNConvUpsampling
    dt = torch.bfloat16
    x16, gx16, g16 = x.to(dt), torch.empty(T, N, dtype=dt, device=dev), g.to(dt)
    pk = torch.empty(T, N // 16, dtype=torch.int32, device=dev)
    lr_p, lr_w = torch.randn(T, N // 32, 9, device=dev), torch.randn(9, 32, device=dev)
    NB, h, w, H, W, Cin, Cout = 80, 130, 173, 260, 346, 64, 32
    up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(dev)
    tabs = up._tables(h, w, torch.device(dev))
    st, bt = fused.sub_tables(tabs, H, W), fused.box_tables(tabs, H, W)
    wt = up.up[1].weight.detach().contiguous()
    xs = (torch.rand(NB, h, w, Cin, device=dev) < 0.4).float()
    xsp = torch.from_numpy(np_pack.pack(xs.cpu().numpy().reshape(-1)).view(np.int32)).to(dev)
    o16 = torch.empty(NB, H, W, Cout, dtype=dt, device=dev)
    gy16 = (torch.randn(NB, H, W, Cout, device=dev) * 1e-3).to(dt)
    gxs16, gw = torch.empty(NB, h, w, Cin, dtype=dt, device=dev), torch.empty(Cout, Cin, 5, 5, device=dev)
    xb = (torch.rand(NB, H, W, 32, device=dev) < 0.3).float()
    xbp = torch.from_numpy(np_pack.pack(xb.cpu().numpy().reshape(-1)).view(np.int32)).to(dev)
    w1 = torch.randn(64, 32, 5, 5, device=dev) * 0.05
    y1 = torch.empty(NB, h, w, 64, dtype=dt, device=dev)
    g1 = (torch.randn(NB, h, w, 64, device=dev) * 1e-3).to(dt)
    gx1 = torch.empty(NB, H, W, 32, dtype=dt, device=dev)
    xv = torch.poisson(torch.full((NB, H, W, 4), 0.05, device=dev))
    w0 = torch.randn(32, 4, 5, 5, device=dev) * 0.1
    y0 = torch.empty(NB, H, W, 32, dtype=dt, device=dev)
    g0 = (torch.randn(NB, H, W, 32, device=dev) * 1e-3).to(dt)
    gw0 = torch.empty(32, 4, 5, 5, device=dev)
    for _ in range(5):
        _lib.neuron_fwd_ex(x16, None, None, None, None, pk, None, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)      # (round 6: membrane unwritten, as the step runs it)
        _lib.neuron_bwd_fork_lr_x16(g16, lr_p, lr_w, None, None, x16, None, gx16, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)
        _lib.dense_conv_s1_fwd_x16(xv, w0, y0, NB, 4, 32, H, W)
        _lib.dense_conv_s1_wgrad_x16(g0, xv, gw0, NB, 4, 32, H, W)
        _lib.spike_conv_fwd_x16(None, xbp, w1, y1, NB, 32, 64, H, W)
        _lib.conv_s2_dgrad_x16(g1, w1, gx1, NB, 32, 64, H, W)
        wm = _lib.upconv_sub_prep_x16(wt, st, Cin, Cout, dt)
        _lib.upconv_sub_fwd_x16(None, xsp, wm, st, o16, NB, Cin, Cout, h, w)
        box = _lib.upconv_boxsum_x16(gy16, bt, NB, Cout, H, W)
        _lib.upconv_box_dgrad_x16(box, wt, bt, gxs16, NB, Cin, Cout, h, w)
        _lib.upconv_box_wgrad_x16(box, None, xsp, bt, gw, NB, Cin, Cout, h, w)
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(5):
    if mode == 'rc':
        _lib.neuron_fwd(x, None, None, out, None, v, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)
        _lib.neuron_bwd_fork(g, g2, None, None, None, x, None, gx, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)
    else:
        _lib.neuron_fwd(x, None, None, out, h, v, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)
        _lib.neuron_bwd(g, None, h, None, gx, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)
if mode == 'rc':
    # round 2: the packed-only forward (x 4 B + packed 0.25 B per update); the decoder kernels below run at the deconv1 geometry of config 3
    # (80 frames, 130x173x64 -> 260x346x32)
    pk = torch.empty(T, N // 16, dtype=torch.int32, device=dev)
    NB, h, w, H, W, Cin, Cout = 80, 130, 173, 260, 346, 64, 32
    from stereospike_amd.network.blocks import NNConvUpsampling
    up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(dev)
    xs = (torch.rand(NB, h, w, Cin, device=dev) < 0.4).float()
    tabs = up._tables(h, w, torch.device(dev))
    o = torch.empty(NB, H, W, Cout, device=dev)
    # round 2, last: the backward whose second gradient is a prediction head's rank-9 pair (ss_neuron_bwd_fork_lr_f32; 13.125 B/update at C = 32)
    lr_p, lr_w = torch.randn(T, N // 32, 9, device=dev), torch.randn(9, 32, device=dev)
    for _ in range(5):
        _lib.neuron_bwd_fork_lr(g, lr_p, lr_w, None, None, x, None, gx, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)
    for _ in range(5):
        _lib.neuron_fwd_ex(x, None, None, None, None, pk, None, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)      # (round 6: membrane unwritten, as the training step runs it)
    gy = torch.randn(NB, H, W, Cout, device=dev)
    gxs = torch.empty(NB, h, w, Cin, device=dev)
    wt = up.up[1].weight.detach().contiguous()
    # round 3: conv1's forward as the exact MFMA implicit GEMM on the packed spikes (ss_spike_conv_fwd_f32), the first layer's forward (ss_dense_conv_s1_fwd_f32)
    xb = (torch.rand(NB, H, W, 32, device=dev) < 0.3).float()
    import numpy as np
    from oracle import np_pack
    xbp = torch.from_numpy(np_pack.pack(xb.cpu().numpy().reshape(-1)).view(np.int32)).to(dev)
    w1 = torch.randn(64, 32, 5, 5, device=dev) * 0.05
    y1 = torch.empty(NB, h, w, 64, device=dev)
    xv = torch.poisson(torch.full((NB, H, W, 4), 0.05, device=dev))
    w0 = torch.randn(32, 4, 5, 5, device=dev) * 0.1
    y0 = torch.empty(NB, H, W, 32, device=dev)
    for _ in range(5):
        _lib.spike_conv_fwd(None, xbp, w1, y1, NB, 32, 64, H, W)
        _lib.dense_conv_s1_fwd(xv, w0, y0, NB, 4, 32, H, W)
    # round 3, third session: conv1's data gradient (ss_conv_s2_dgrad_f32), the first layer's weight gradient (ss_dense_conv_s1_wgrad_f32), the
    # full-resolution head on packed spikes (ss_head_proj_packed_f32 / ss_head_wgrad_packed_f32), the packed-only forward with a packed skip
    g1 = torch.randn(NB, h, w, 64, device=dev) * 1e-4
    gx1 = torch.empty(NB, H, W, 32, device=dev)
    g0 = torch.randn(NB, H, W, 32, device=dev) * 1e-4
    gw0 = torch.empty(32, 4, 5, 5, device=dev)
    rows = NB * H * W
    Wh = torch.randn(32, 9, device=dev) * 0.1
    Ph, gPh, gWh = torch.empty(rows, 9, device=dev), torch.randn(rows, 9, device=dev) * 1e-5, torch.empty(32, 9, device=dev)
    for _ in range(5):
        _lib.conv_s2_dgrad(g1, w1, gx1, NB, 32, 64, H, W)
        _lib.dense_conv_s1_wgrad(g0, xv, gw0, NB, 4, 32, H, W)
        _lib.head_proj_packed(xbp, Wh, Ph, rows, 32)
        _lib.head_wgrad_packed(xbp, gPh, gWh, rows, 32)
        _lib.neuron_fwd_ex(x, None, None, xbp.view(T, -1), None, pk, None, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)
if mode == 'rc':
    # round 4: the decoder backward of deconv1 on the box-sum image (ss_upconv_boxsum_f32 -> ss_upconv_box_dgrad_f32 + ss_upconv_box_wgrad_f32), what the default path runs
    from stereospike_amd import fused
    bt = fused.box_tables(tabs, H, W)
    gxs2, gw2 = torch.empty(NB, h, w, Cin, device=dev), torch.empty(Cout, Cin, 5, 5, device=dev)
    for _ in range(5):
        box = _lib.upconv_boxsum(gy, bt, NB, Cout, H, W)
        _lib.upconv_box_dgrad(box, wt, bt, gxs2, NB, Cin, Cout, h, w)
        _lib.upconv_box_wgrad(box, xs, None, bt, gw2, NB, Cin, Cout, h, w)
    print('box planes', tuple(box.shape), box.dtype, 'bytes', box.numel() * box.element_size())
    # round 4: deconv1's forward as the sub-pixel (merged tap) implicit GEMM on the packed spikes (ss_upconv_sub_fwd_f32)
    st = fused.sub_tables(tabs, H, W)
    xsp = torch.from_numpy(np_pack.pack(xs.cpu().numpy().reshape(-1)).view(np.int32)).to(dev)
    for _ in range(5):
        wm = _lib.upconv_sub_prep(wt, st, Cin, Cout)
        _lib.upconv_sub_fwd(None, xsp, wm, st, o, NB, Cin, Cout, h, w)
torch.cuda.synchronize()
print('mode', mode, 'algorithmic bytes per launch: fwd', (8 if mode == 'rc' else 12) * T * N, 'bwd', (16 if mode == 'rc' else 12) * T * N)
