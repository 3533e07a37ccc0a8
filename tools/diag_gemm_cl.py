#!/usr/bin/env python3
"""GPU experiment: the NHWC decoder's three GEMMs per stage at config-3 size — fp32 (rocBLAS/hipBLASLt) vs exact 3-way bf16
split on the bf16 MFMA path with fp32 accumulation (x holds small integers: exact in bf16)."""
import torch
dev = 'cuda:0'
torch.manual_seed(0)
layers = [('deconv4', 512, 256, 17 * 22), ('deconv3', 256, 128, 33 * 44), ('deconv2', 128, 64, 65 * 87), ('deconv1', 64, 32, 130 * 173)]
NB = 80


def split3(a):
    hi = a.to(torch.bfloat16)
    r = a - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def timeit(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, Cin, Cout, hw in layers:
    M, K, N = NB * hw, Cin, 25 * Cout
    x = torch.randint(0, 3, (M, K), device=dev).float()
    Wt = torch.randn(K, N, device=dev) * 0.05
    flops = 2.0 * M * K * N
    t32 = timeit(lambda: torch.mm(x, Wt))
    ref = torch.mm(x[:4096].double(), Wt.double())
    e32 = float((torch.mm(x[:4096], Wt).double() - ref).abs().max() / ref.abs().max())
    Wh, Wm, Wl = split3(Wt)
    W3 = torch.cat([Wh, Wm, Wl], dim=0).contiguous()          # [3K, N]
    def fwd_b():
        xb = x.to(torch.bfloat16)
        x3 = torch.cat([xb, xb, xb], dim=1)                   # [M, 3K]
        return torch.mm(x3, W3, out_dtype=torch.float32)
    tb = timeit(fwd_b)
    eb = float((fwd_b()[:4096].double() - ref).abs().max() / ref.abs().max())
    xb = x.to(torch.bfloat16)
    def fwd_b3():
        o = torch.mm(xb, Wh, out_dtype=torch.float32)
        o.addmm_(xb.float()[:0], Wt[:0])  # no-op keep signature
        return o
    print(f'{name} fwd M={M} K={K} N={N}: fp32 {t32:.2f} ms ({flops / t32 / 1e9:.0f} TF, err {e32:.1e})   bf16x3(concat, incl. cast) {tb:.2f} ms '
          f'({flops / tb / 1e9:.0f} TF-equiv, err {eb:.1e})', flush=True)
    # dgrad g_x = g_P @ W2 (fp32 only)
    gP = torch.randn(M, N, device=dev)
    W2 = Wt.t().contiguous()
    td = timeit(lambda: torch.mm(gP, W2))
    # wgrad g_Wt = x^T @ g_P : split-K bmm fp32 vs single mm vs bf16x3
    S = max(1, M // 16384); L = M // S
    tw_split = timeit(lambda: torch.bmm(x[:S * L].view(S, L, K).transpose(1, 2), gP[:S * L].view(S, L, N)).sum(0))
    tw_mm = timeit(lambda: torch.mm(x.t(), gP))
    refw = torch.mm(x.t().double(), gP.double())
    def wg_b():
        gh, gm, gl = split3(gP)
        xt = xb.t()
        o = torch.mm(xt, gh, out_dtype=torch.float32)
        o += torch.mm(xt, gm, out_dtype=torch.float32)
        o += torch.mm(xt, gl, out_dtype=torch.float32)
        return o
    try:
        twb = timeit(wg_b)
        ewb = float((wg_b().double() - refw).abs().max() / refw.abs().max())
    except Exception as e:
        twb, ewb = float('nan'), repr(e)[:80]
    ew = float((torch.mm(x.t(), gP).double() - refw).abs().max() / refw.abs().max())
    print(f'{name} dgrad fp32 {td:.2f} ms ({flops / td / 1e9:.0f} TF)   wgrad: split-K bmm {tw_split:.2f} ms ({flops / tw_split / 1e9:.0f} TF)  '
          f'single mm {tw_mm:.2f} ms ({flops / tw_mm / 1e9:.0f} TF, err {ew:.1e})  bf16x3 {twb:.2f} ms (err {ewb})', flush=True)
    del x, gP
    torch.cuda.empty_cache()
