#!/usr/bin/env python3
"""A/B of the 16-bit recompute-backward vector width (build variants of libss_neuron*.so) and of saved-h vs recompute, in ONE process,
interleaved rounds: median launch time per variant at the config-2/5 bottom-layer size (bf16)."""
import ctypes as C, glob, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = 'cuda:0'
libs = {os.path.basename(p)[len('libss_neuron'):-3] or 'default': C.CDLL(p) for p in sorted(glob.glob('stereospike_amd/lib/libss_neuron*.so'))}
p, i32, i64, f32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float
for L in libs.values():
    L.ss_neuron_fwd_x16.argtypes = [p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, i32, p]
    L.ss_neuron_bwd_x16.argtypes = [p, p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, i32, f32, i32, i32, p]
    L.ss_neuron_bwd_rc_x16.argtypes = L.ss_neuron_bwd_x16.argtypes
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for T, B in ((5, 16), (10, 8)):
    N = B * 32 * 260 * 346
    x = (torch.randn(T, N, device=dev) * 0.15).bfloat16()
    out, g, gx = torch.empty_like(x), torch.randn(T, N, device=dev).bfloat16(), torch.empty_like(x)
    h = torch.empty(T, N, device=dev)
    v = torch.empty(N, device=dev)
    fns = {
        'fwd_saveh': (lambda L: L.ss_neuron_fwd_x16(P(x), None, None, P(out), P(h), P(v), None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 2, st), 8),
        'fwd_noh': (lambda L: L.ss_neuron_fwd_x16(P(x), None, None, P(out), None, P(v), None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 2, st), 4),
        'bwd_saveh': (lambda L: L.ss_neuron_bwd_x16(P(g), None, P(h), None, P(gx), None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, 1, 2, st), 8),
        'bwd_rc': (lambda L: L.ss_neuron_bwd_rc_x16(P(g), None, P(x), None, P(gx), None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, 1, 2, st), 6),
    }
    res = {}
    for rnd in range(10):
        for name, L in libs.items():
            for tag, (fn, bpu) in fns.items():
                if name != 'default' and tag != 'bwd_rc':
                    continue
                assert fn(L) == 0; torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                e0.record()
                for _ in range(5): fn(L)
                e1.record(); torch.cuda.synchronize()
                res.setdefault((name, tag, bpu), []).append(e0.elapsed_time(e1) / 5 * 1e3)
    for (name, tag, bpu), v_ in sorted(res.items()):
        us = statistics.median(v_)
        print(f'T={T:2d} B={B:2d} {name:8s} {tag:10s} median {us:7.1f} us  = {bpu * T * N / us / 1e3:7.1f} GB/s ({bpu} B/update)')
