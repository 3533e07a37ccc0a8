#!/usr/bin/env python3
"""Feasibility probe: encoder / bottleneck convs on spike inputs as explicit-im2col bf16 GEMMs with the exact bf16x3 weight split,
against MIOpen's fp32 NHWC conv (find mode).  Prints ms per call."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereospike_amd import miopen_cache
miopen_cache.enable(skip_naive_solvers=True)
torch.backends.cudnn.benchmark = True
dev = 'cuda:0'

def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

NB = 80
for name, Cin, Cout, k, s, pad, (h, w) in (('bottleneck', 512, 512, 3, 1, 1, (17, 22)), ('conv4', 256, 512, 5, 2, 2, (33, 44)),
                                            ('conv3', 128, 256, 5, 2, 2, (65, 87)), ('conv2', 64, 128, 5, 2, 2, (130, 173))):
    x = (torch.rand(NB, h, w, Cin, device=dev) < 0.2).float()                 # NHWC array
    W = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    xc = x.permute(0, 3, 1, 2)                                                 # logical NCHW, channels_last memory
    Wc = W.contiguous(memory_format=torch.channels_last)
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    M, K = NB * ho * wo, k * k * Cin
    t_conv = timeit(lambda: F.conv2d(xc, Wc, stride=s, padding=pad))
    y = F.conv2d(xc, Wc, stride=s, padding=pad)
    g = torch.randn_like(y)
    xr = xc.detach().requires_grad_(); Wr = Wc.detach().requires_grad_()
    def bwd():
        yy = F.conv2d(xr, Wr, stride=s, padding=pad)
        return torch.autograd.grad(yy, (xr, Wr), g)
    t_all = timeit(bwd)
    # explicit im2col in bf16
    def im2col():
        xp = F.pad(x.to(torch.bfloat16), (0, 0, pad, pad, pad, pad))          # [NB, h+2p, w+2p, Cin]
        sn, sh, sw, sc = xp.stride()
        v = xp.as_strided((NB, ho, wo, k, k, Cin), (sn, sh * s, sw * s, sh, sw, sc))
        return v.reshape(M, K)
    A = im2col()
    t_i2c = timeit(im2col)
    Wt = W.permute(2, 3, 1, 0).reshape(K, Cout)
    Wh = Wt.to(torch.bfloat16); r = Wt - Wh.float(); Wm = r.to(torch.bfloat16); Wl = (r - Wm.float()).to(torch.bfloat16)
    W3n = torch.cat((Wh, Wm, Wl), 1).contiguous()                              # [K, 3N]
    def fwd_gemm():
        o = torch.mm(A, W3n, out_dtype=torch.float32)                          # [M, 3N]
        return o[:, :Cout] + o[:, Cout:2 * Cout] + o[:, 2 * Cout:]
    t_fg = timeit(fwd_gemm)
    yg = fwd_gemm().view(NB, ho, wo, Cout).permute(0, 3, 1, 2)
    err = float((yg - y).abs().max() / y.abs().max())
    g2 = g.permute(0, 2, 3, 1).reshape(M, Cout)
    def wgrad_gemm():
        gh = g2.to(torch.bfloat16); r1 = g2 - gh.float(); gm = r1.to(torch.bfloat16); gl = (r1 - gm.float()).to(torch.bfloat16)
        g3 = torch.cat((gh, gm, gl), 1)                                        # [M, 3N]
        o = torch.mm(A.t(), g3, out_dtype=torch.float32)                       # [K, 3N]
        return o[:, :Cout] + o[:, Cout:2 * Cout] + o[:, 2 * Cout:]
    t_wg = timeit(wgrad_gemm)
    gw_ref = bwd()[1].permute(2, 3, 1, 0).reshape(K, Cout)
    errw = float((wgrad_gemm() - gw_ref).abs().max() / gw_ref.abs().max())
    fl = 2 * M * K * Cout / 1e9
    print(f'{name:10s} M={M} K={K} N={Cout} ({fl:.0f} GF): MIOpen fwd {t_conv:.2f} ms, fwd+dgrad+wgrad {t_all:.2f} ms | im2col(bf16) {t_i2c:.2f} ms, '
          f'fwd bf16x3 GEMM {t_fg:.2f} ms (err {err:.1e}), wgrad bf16x3 GEMM {t_wg:.2f} ms (err {errw:.1e})', flush=True)
