#!/usr/bin/env python3
"""GPU experiment: projection GEMMs of the decoder stages — fp32 (rocBLAS) vs exact 3-way bf16 split on the MFMA
bf16 path with fp32 accumulation (spike inputs are small integers, exactly representable in bf16; fp32 weights /
gradients split into hi+mid+lo bf16 pieces => every product is exact, only the summation order differs)."""
import os, sys, time
import torch
dev = 'cuda:0'
torch.manual_seed(0)
layers = [('deconv4', 512, 256, 17 * 22), ('deconv3', 256, 128, 33 * 44), ('deconv2', 128, 64, 65 * 87), ('deconv1', 64, 32, 130 * 173)]
CH = 96 << 20


def split3(a):
    hi = a.to(torch.bfloat16)
    r = a - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print('bmm out_dtype supported:', end=' ')
try:
    a = torch.randn(2, 4, 8, device=dev).bfloat16(); b = torch.randn(2, 8, 4, device=dev).bfloat16()
    print(torch.bmm(a, b, out_dtype=torch.float32).dtype)
except Exception as e:
    print('NO', repr(e)[:200])
for name, Cin, Cout, hw in layers:
    M, K = Cout * 25, Cin
    n = max(1, min(80, CH // (M * hw * 4)))
    W2 = torch.randn(M, K, device=dev) * 0.05
    x = torch.randint(0, 3, (n, K, hw), device=dev).float()
    flops = 2.0 * M * K * hw * n
    # ---- forward  P = W2 @ x
    ref = torch.matmul(W2.double(), x.double())
    t32 = timeit(lambda: torch.matmul(W2, x))
    e32 = float((torch.matmul(W2, x).double() - ref).abs().max() / ref.abs().max())
    Wh, Wm, Wl = split3(W2)
    W3 = torch.cat([Wh, Wm, Wl], dim=1).contiguous()                    # [M, 3K]
    xb = x.to(torch.bfloat16)
    x3 = torch.cat([xb, xb, xb], dim=1).contiguous()                    # [n, 3K, hw]
    res = {}
    try:
        f = lambda: torch.bmm(W3.unsqueeze(0).expand(n, M, 3 * K), x3, out_dtype=torch.float32)
        tb = timeit(f)
        eb = float((f().double() - ref).abs().max() / ref.abs().max())
        res['bmm_expand'] = (tb, eb)
    except Exception as e:
        res['bmm_expand'] = repr(e)[:100]
    try:
        f = lambda: torch.stack([torch.mm(W3, x3[i], out_dtype=torch.float32) for i in range(n)])
        tb = timeit(f)
        eb = float((f().double() - ref).abs().max() / ref.abs().max())
        res['mm_per_frame'] = (tb, eb)
    except Exception as e:
        res['mm_per_frame'] = repr(e)[:100]
    try:   # three separate accumulating GEMMs instead of K-concatenation
        def f3():
            o = torch.bmm(Wh.unsqueeze(0).expand(n, M, K), xb, out_dtype=torch.float32)
            o += torch.bmm(Wm.unsqueeze(0).expand(n, M, K), xb, out_dtype=torch.float32)
            o += torch.bmm(Wl.unsqueeze(0).expand(n, M, K), xb, out_dtype=torch.float32)
            return o
        tb = timeit(f3)
        eb = float((f3().double() - ref).abs().max() / ref.abs().max())
        res['bmm_3pass'] = (tb, eb)
    except Exception as e:
        res['bmm_3pass'] = repr(e)[:100]
    print(f'{name} fwd  n={n} M={M} K={K} N={hw}: fp32 {t32:.3f} ms ({flops / t32 / 1e9:.0f} TF, err {e32:.1e})  ' +
          '  '.join(f'{k}: {v[0]:.3f} ms ({flops / v[0] / 1e9:.0f} TF-equiv, err {v[1]:.1e})' if isinstance(v, tuple) else f'{k}: {v}' for k, v in res.items()), flush=True)
    # ---- wgrad  gW2 = sum_n gP[n] @ x[n]^T
    gP = torch.randn(n, M, hw, device=dev)
    refw = torch.bmm(gP.double(), x.double().transpose(1, 2)).sum(0)
    fw32 = lambda: torch.bmm(gP, x.transpose(1, 2)).sum(0)
    tw32 = timeit(fw32)
    ew32 = float((fw32().double() - refw).abs().max() / refw.abs().max())
    xbt = xb.transpose(1, 2)
    def fwb():
        gh, gm, gl = split3(gP)
        o = torch.bmm(gh, xbt, out_dtype=torch.float32)
        o += torch.bmm(gm, xbt, out_dtype=torch.float32)
        o += torch.bmm(gl, xbt, out_dtype=torch.float32)
        return o.sum(0)
    try:
        twb = timeit(fwb)
        ewb = float((fwb().double() - refw).abs().max() / refw.abs().max())
        tsp = timeit(lambda: split3(gP))
        print(f'{name} wgrad: fp32 {tw32:.3f} ms ({flops / tw32 / 1e9:.0f} TF, err {ew32:.1e})  bf16x3 {twb:.3f} ms incl. split {tsp:.3f} ms '
              f'({flops / twb / 1e9:.0f} TF-equiv, err {ewb:.1e})', flush=True)
    except Exception as e:
        print(name, 'wgrad bf16 failed', repr(e)[:200])
    # ---- dgrad (fp32 only): g_x = W2^T @ gP
    W2t = W2.t().contiguous()
    td = timeit(lambda: torch.matmul(W2t, gP))
    print(f'{name} dgrad fp32 {td:.3f} ms ({flops / td / 1e9:.0f} TF)', flush=True)
