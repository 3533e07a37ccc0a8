#!/usr/bin/env python3
"""Do an MFMA-bound kernel and an HBM-bound kernel overlap when launched on two streams?  (conv_s2_dgrad at conv2's geometry, ~1.0 ms, 2 workgroups per CU with 256
registers per lane; neuron_bwd on the 32 x 260 x 346 layer, ~0.5 ms.)  Prints the two times alone, back to back on one stream, and on two streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stereospike_amd import _lib
dev = 'cuda:0'
NB, Cin, h, w = 80, 64, 130, 173
Cout, ho, wo = 128, 65, 87
g = torch.randn(NB, ho, wo, Cout, device=dev) * 1e-3
wt = torch.randn(Cout, Cin, 5, 5, device=dev) * 0.05
gx = torch.empty(NB, h, w, Cin, device=dev)
T, N = 5, 16 * 32 * 260 * 346
x = torch.randn(T, N, device=dev)
gs, g2, gxn = torch.randn(T, N, device=dev), torch.randn(T, N, device=dev), torch.empty(T, N, device=dev)
xp = torch.randint(-2 ** 31, 2 ** 31 - 1, (NB * 260 * 346 * 32 // 16,), dtype=torch.int32, device=dev)
gw = torch.empty(64, 32, 5, 5, device=dev)
gc1 = torch.randn(NB, 130, 173, 64, device=dev) * 1e-3
def a(): _lib.conv_s2_dgrad(g, wt, gx, NB, Cin, Cout, h, w)
def b(): _lib.neuron_bwd_fork(gs, g2, None, None, None, x, None, gxn, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)
def c(): _lib.spike_conv_wgrad(gc1, None, gw, NB, 32, 64, 260, 346, x_packed=xp)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
def both(f1, f2):
    def run():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): f1()
        with torch.cuda.stream(s2): f2()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return run
for n1, f1, n2, f2 in (('conv_s2_dgrad', a, 'neuron_bwd', b), ('spike_conv_wgrad_tr', c, 'neuron_bwd', b), ('conv_s2_dgrad', a, 'spike_conv_wgrad_tr', c)):
    t1, t2 = timed(f1), timed(f2)
    ts = timed(lambda: (f1(), f2()))
    tp = timed(both(f1, f2))
    print(f'{n1} {t1:.3f} ms | {n2} {t2:.3f} ms | one stream {ts:.3f} | two streams {tp:.3f}', flush=True)
