#!/usr/bin/env python3
"""A/B for VERDICT r03 item 5: conv3 / conv4 forward at config 3's shapes (80 frames; /root/reference/network/SNN_models.py:91-101) as
  (a) what the network runs: ss_im2col_cl_bf16_packed + the library GEMM on [A] x [Wh | Wm | Wl] + the three-term sum,
  (b) ss_spike_conv_fwd_f32's implicit GEMM reading the packed spikes directly (no patch matrix), output-channel slices of 128 per workgroup.
Interleaved rounds, HIP events, both results compared with each other (exact products either way: fp32 summation order only)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stereospike_amd import _lib, fused
from oracle import np_pack            # (tools/ script: the packer is only used to build the input)
dev = 'cuda:0'
ROUNDS, REPS = int(os.environ.get('ROUNDS', 5)), int(os.environ.get('REPS', 5))
for name, (NB, h, w, Cin, Cout) in (('conv3', (80, 65, 87, 128, 256)), ('conv4', (80, 33, 44, 256, 512))):
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    M, K = NB * ho * wo, 25 * Cin
    g = torch.Generator(device=dev).manual_seed(3)
    x = (torch.rand(NB, h, w, Cin, device=dev, generator=g) < 0.15).float()
    xp = torch.from_numpy(np_pack.pack(x.cpu().numpy().reshape(-1)).view(np.int32)).to(dev)
    W = torch.randn(Cout, Cin, 5, 5, device=dev, generator=g) * 0.05
    A = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
    y_b = torch.empty(NB, ho, wo, Cout, device=dev)

    def lib_path():
        _lib.im2col_cl_bf16_packed(xp, A, NB, h, w, Cin, 5, 2, 2, ho, wo)
        Wt = W.permute(2, 3, 1, 0).reshape(K, Cout)
        y3 = torch.mm(A, fused._split3_cols(Wt.float()), out_dtype=torch.float32)
        return y3.view(M, 3, Cout).sum(1).view(NB, ho, wo, Cout)

    def own_path():
        _lib.spike_conv_fwd(None, xp, W, y_b, NB, Cin, Cout, h, w)
        return y_b
    ya = lib_path(); yb = own_path().clone()
    torch.cuda.synchronize()
    err = float((ya - yb).abs().max() / ya.abs().max())
    times = {'library (im2col + GEMM + sum)': [], 'implicit GEMM (ss_spike_conv_fwd_f32, wide)': []}
    for _ in range(ROUNDS):
        for label, fn in zip(times, (lib_path, own_path)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                fn()
            e1.record(); torch.cuda.synchronize()
            times[label].append(e0.elapsed_time(e1) / REPS)
    macs = M * K * Cout
    print(f'{name}: C_in {Cin} C_out {Cout} out {ho}x{wo} x {NB} frames  {macs / 1e9:.1f} G MACs (x 3 bf16 terms)  max |a - b| / max |a| = {err:.2e}')
    for label, ts in times.items():
        t = float(np.median(ts))
        print(f'   {label:48s} {t:7.3f} ms   {2 * 3 * macs / t / 1e9:7.1f} TFLOP/s bf16 ({2 * 3 * macs / t / 1e9 / 2500:.3f} of the MFMA peak)  rounds {["%.3f" % v for v in ts]}')
