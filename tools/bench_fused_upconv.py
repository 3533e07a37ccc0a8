#!/usr/bin/env python3
"""GPU micro-benchmark of the fused projection + gather MFMA kernels (ss_upconv_fused_fwd_f32 / ss_upconv_fused2_fwd_f32) at the
config-3 geometries of deconv1 / deconv2 (80 frames): interleaved rounds of both forms, HIP-event time per launch, equality check,
roofline figures (algorithmic bytes = x + out; useful bf16 FLOPs = 3 exact products per MAC of the minimal projection).
FORMS=1,2 (default) selects the forms; ROUNDS / REPS control the timing; ONLY=deconv1 restricts the geometry (for rocprofv3 passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stereospike_amd import _lib
if os.environ.get('SS_LIB'):                       # a `make variant` build (timing experiments)
    _lib.LIB_PATH = os.path.abspath(os.environ['SS_LIB'])
from stereospike_amd.network.blocks import NNConvUpsampling
dev = 'cuda:0'
NB = int(os.environ.get('NB', 80))
forms = [int(f) for f in os.environ.get('FORMS', '1,2').split(',')]
rounds, reps = int(os.environ.get('ROUNDS', 5)), int(os.environ.get('REPS', 5))
geoms = [('deconv1', 64, 32, (130, 173), (260, 346)), ('deconv2', 128, 64, (65, 87), (130, 173))]
only = os.environ.get('ONLY')
for name, Cin, Cout, (h, w), (H, W) in geoms:
    if only and name not in only.split(','):
        continue
    torch.manual_seed(0)
    up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(dev)
    x = ((torch.rand(NB, h, w, Cin, device=dev) < 0.35).float() + (torch.rand(NB, h, w, Cin, device=dev) < 0.1).float()).contiguous()
    tabs = up._tables(h, w, torch.device(dev))
    win = up.max_tile_window(h, w)
    wt = up.up[1].weight.detach().contiguous()
    outs, Wfs = {}, {}
    for f in forms:
        Wfs[f] = torch.empty(_lib.upconv_fused_wf_elems(Cin, Cout, f), dtype=torch.bfloat16, device=dev)
        _lib.upconv_fused_prep_w(wt, Wfs[f], Cin, Cout, f)
        outs[f] = torch.full((NB, H, W, Cout), float('nan'), device=dev)

    def run(f):
        _lib.upconv_fused_fwd(x, None, Wfs[f], tabs[0], tabs[3], outs[f], NB, Cin, Cout, h, w, H, W, win, f)
    for f in forms:
        run(f)
    torch.cuda.synchronize()
    if len(forms) > 1:
        print(name, 'forms equal bit for bit:', bool(torch.equal(outs[forms[0]], outs[forms[1]])),
              'max |diff|', float((outs[forms[0]] - outs[forms[1]]).abs().max()), flush=True)
    best = {f: 1e9 for f in forms}
    for _ in range(rounds):
        for f in forms:
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(reps):
                run(f)
            e1.record(); torch.cuda.synchronize()
            best[f] = min(best[f], e0.elapsed_time(e1) / reps)
    alg = 4 * (x.numel() + outs[forms[0]].numel())
    flops = 2 * 3 * NB * h * w * Cin * 25 * Cout
    for f in forms:
        ms = best[f]
        print(f'{name} form {f}: {ms:7.3f} ms/launch   {alg / ms / 1e6:7.1f} GB/s algorithmic ({alg / ms / 1e6 / 8000:.3f} of HBM peak)   '
              f'{flops / ms / 1e9:7.1f} TFLOP/s useful bf16 ({flops / ms / 1e9 / 2500:.3f} of MFMA peak)   window {win}', flush=True)
    if os.environ.get('TRACE') and 2 in forms:      # a -DSS_F2_TRACE=1 build: s_memtime stamps of workgroup 0's first 64 steps
        import ctypes, numpy as np
        run(2); torch.cuda.synchronize()
        buf = np.zeros((2, 64, 4), dtype=np.uint64)
        L = _lib.lib()
        L.ss_debug_f2_trace.argtypes = [ctypes.c_void_p]
        assert L.ss_debug_f2_trace(buf.ctypes.data) == 0
        t0 = int(buf[0, 0, 0])
        print(name, 'trace (cycles rel. to producer step 0): step | producer start, mfma done, stores done, barrier out | consumer start, gather done, step done, barrier out')
        for st in range(0, 34):
            pr = [int(v) - t0 if v else -1 for v in buf[0, st]]
            co = [int(v) - t0 if v else -1 for v in buf[1, st]]
            print(f'  {st:3d} | {pr[0]:7d} {pr[1]:7d} {pr[2]:7d} {pr[3]:7d} | {co[0]:7d} {co[1]:7d} {co[2]:7d} {co[3]:7d}')

if os.environ.get('WGRAD', '1') == '1':       # exact bf16x3 MFMA weight gradient vs the library's fp32 GEMM (split-K as in fused.py)
    from stereospike_amd import fused
    for name, Cin, Cout, (h, w), _ in geoms + [('deconv3', 256, 128, (33, 44), None), ('deconv4', 512, 256, (17, 22), None)]:
        R, N = NB * h * w, 25 * Cout
        x = (torch.rand(R, Cin, device=dev) < 0.35).float()
        g = torch.randn(R, N, device=dev)
        out = torch.empty(Cin, N, device=dev)
        S = max(1, R // fused.WGRAD_SPLIT_ROWS); L = R // S

        def lib():
            return torch.bmm(x[:S * L].view(S, L, Cin).transpose(1, 2), g[:S * L].view(S, L, N)).sum(0)

        def own():
            _lib.spike_wgrad(g, x, out, R, Cin, N)
        res = {}
        for tag, fn in (('library fp32 split-K bmm', lib), ('ss_spike_wgrad_f32', own)):
            fn(); fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[tag] = e0.elapsed_time(e1) / 5
        print(f'{name} wgrad [{Cin} x {R}] @ [{R} x {N}]: ' + ', '.join(f'{k} {v:.3f} ms' for k, v in res.items()) +
              f'  (g read once at HBM rate: {4 * R * N / 5.5e9:.3f} ms)', flush=True)

if os.environ.get('BWD', '1') == '1':         # fused adjoint + weight gradient vs adjoint kernel + ss_spike_wgrad_f32
    from stereospike_amd import fused
    for name, Cin, Cout, (h, w), (H, W) in geoms:
        up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(dev)
        tabs = up._tables(h, w, torch.device(dev))
        R, N = NB * h * w, 25 * Cout
        g = torch.randn(NB, H, W, Cout, device=dev)
        x = (torch.rand(NB, h, w, Cin, device=dev) < 0.35).float()
        gP = torch.empty(R, N, device=dev); gw = torch.empty(Cin, N, device=dev)

        def two():
            _lib.upconv_cl_bwd(g, tabs[1], tabs[2], tabs[4], tabs[5], gP, NB, 5, Cout, h, w, H, W)
            _lib.spike_wgrad(gP, x.view(R, Cin), gw, R, Cin, N)

        def one():
            _lib.upconv_bwd_fused(g, x, tabs[1], tabs[2], tabs[4], tabs[5], gP, gw, NB, Cin, Cout, h, w, H, W)
        res = {}
        for tag, fn in (('adjoint + wgrad kernels', two), ('fused', one)):
            fn(); fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[tag] = e0.elapsed_time(e1) / 5
        print(f'{name} backward (adjoint + weight gradient): ' + ', '.join(f'{k} {v:.3f} ms' for k, v in res.items()) +
              f'  (g_P written once at HBM rate: {4 * R * N / 5.5e9:.3f} ms)', flush=True)

if os.environ.get('GEMM6', '1') == '1':       # decoder data gradient: library fp32 GEMM vs the six-term bf16 MFMA GEMM
    for name, R, K, N in (('deconv1', 80 * 130 * 173, 800, 64), ('deconv2', 80 * 65 * 87, 1600, 128), ('deconv3', 80 * 33 * 44, 3200, 256),
                          ('deconv4', 80 * 17 * 22, 6400, 512)):
        A = torch.randn(R, K, device=dev); B = torch.randn(K, N, device=dev) * 0.05; C = torch.empty(R, N, device=dev)
        res = {}
        for tag, fn in (('library fp32', lambda: torch.mm(A, B, out=C)), ('ss_gemm6_f32', lambda: _lib.gemm6(A, B, C, R, K, N))):
            fn(); fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[tag] = e0.elapsed_time(e1) / 5
        print(f'{name} dgrad [{R} x {K}] @ [{K} x {N}]: ' + ', '.join(f'{k} {v:.3f} ms' for k, v in res.items()) +
              f'  (A read once at HBM rate: {4 * R * K / 5.5e9:.3f} ms; 6 x bf16 MFMA at peak: {12.0 * R * K * N / 2.5e15 * 1e3:.3f} ms)', flush=True)
