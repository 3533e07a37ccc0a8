#!/usr/bin/env python3
"""A/B of build variants of the DEFAULT training forms of the neuron kernels — forward without h_seq (8 B/update) and the recompute
backward with a second gradient (16 B/update) — in ONE process, interleaved rounds: median GB/s per variant (libss_neuron*.so)."""
import ctypes as C, glob, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = 'cuda:0'
T, N = 5, 16 * 32 * 260 * 346
libs = {os.path.basename(p)[len('libss_neuron'):-3] or 'default': C.CDLL(p) for p in sorted(glob.glob('stereospike_amd/lib/libss_neuron*.so'))}
p, i32, i64, f32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float
for L in libs.values():
    L.ss_neuron_fwd_f32.argtypes = [p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, p]
    L.ss_neuron_bwd_fork_f32.argtypes = [p, p, p, p, p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, i32, f32, i32, p]
x = torch.randn(T, N, device=dev) * 0.15
out, g, g2, gx = torch.empty_like(x), torch.randn(T, N, device=dev), torch.randn(T, N, device=dev), torch.empty_like(x)
v = torch.empty(N, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def fwd(L): assert L.ss_neuron_fwd_f32(P(x), None, None, P(out), None, P(v), None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, st) == 0
def bwd(L): assert L.ss_neuron_bwd_fork_f32(P(g), P(g2), None, None, None, P(x), None, P(gx), None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, 1, st) == 0
def copy(L): out.copy_(x)
res = {}
for rnd in range(12):
    for name, L in libs.items():
        for tag, fn, nbytes in (('fwd', fwd, (8 * T + 4) * N), ('bwd', bwd, 16 * T * N), ('copy', copy, 8 * T * N)):
            if tag == 'copy' and name != 'default': continue
            fn(L); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(5): fn(L)
            e1.record(); torch.cuda.synchronize()
            res.setdefault((name, tag), []).append(nbytes * 5 / e0.elapsed_time(e1) / 1e6)
for (name, tag), v_ in sorted(res.items()):
    print(f'{name:12s} {tag:5s} median {statistics.median(v_):7.1f} GB/s   min {min(v_):7.1f}  max {max(v_):7.1f}   (fwd counts the v_last write)')
