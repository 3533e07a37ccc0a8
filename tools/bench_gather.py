#!/usr/bin/env python3
"""(Round-1 result: a 2x2-outputs-per-lane forward with conditional operand sharing was bit-identical but 4-5x SLOWER — the
data-dependent branches serialise the 25 independent tap loads; profiles/r01/gather_2x2_blocking_rejected.log.  The shipped
one-pixel-per-lane kernels run at 3.5-3.95 TB/s of P + out.)
A/B of build variants of the channels-last gather kernels (ss_upconv_cl_fwd_f32 / _bwd_f32) at the four decoder shapes of config 3
(NB = 80), interleaved rounds in ONE process; also checks the variants agree bit for bit."""
import ctypes as C, glob, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereospike_amd.fused import nearest_tables
dev = 'cuda:0'
libs = {os.path.basename(p)[len('libss_neuron'):-3] or 'default': C.CDLL(p) for p in sorted(glob.glob('stereospike_amd/lib/libss_neuron*.so'))}
p, i32, i64 = C.c_void_p, C.c_int, C.c_longlong
for L in libs.values():
    L.ss_upconv_cl_fwd_f32.argtypes = [p, p, p, p, p, i64, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_cl_bwd_f32.argtypes = [p, p, p, p, p, p, i64, i32, i32, i32, i32, i32, i32, p]
P_ = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
NB, K = int(os.environ.get('NB', 80)), 5
for name, Cout, (h, w), (H, W) in (('deconv4', 256, (17, 22), (33, 44)), ('deconv3', 128, (33, 44), (65, 87)),
                                   ('deconv2', 64, (65, 87), (130, 173)), ('deconv1', 32, (130, 173), (260, 346))):
    sy, ylo, yhi = (t.to(dev) for t in nearest_tables(h, H + K - 1))
    sx, xlo, xhi = (t.to(dev) for t in nearest_tables(w, W + K - 1))
    Pm = torch.randn(NB, h, w, K * K * Cout, device=dev)
    g = torch.randn(NB, H, W, Cout, device=dev)
    outs, gps = {}, {}
    res = {}
    for rnd in range(6):
        for lname, L in libs.items():
            out = torch.empty(NB, H, W, Cout, device=dev)
            gP = torch.empty_like(Pm)
            fwd = lambda: L.ss_upconv_cl_fwd_f32(P_(Pm), P_(sy), P_(sx), None, P_(out), NB, K, Cout, h, w, H, W, st)
            bwd = lambda: L.ss_upconv_cl_bwd_f32(P_(g), P_(ylo), P_(yhi), P_(xlo), P_(xhi), P_(gP), NB, K, Cout, h, w, H, W, st)
            for tag, fn in (('fwd', fwd), ('bwd', bwd)):
                assert fn() == 0; torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                e0.record()
                for _ in range(3): fn()
                e1.record(); torch.cuda.synchronize()
                res.setdefault((lname, tag), []).append(e0.elapsed_time(e1) / 3 * 1e3)
            if rnd == 0:
                outs[lname], gps[lname] = out, gP
    big = torch.empty(Pm.numel(), device=dev)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); big.fill_(1.0); e0.record()
    for _ in range(3): big.fill_(2.0)
    e1.record(); torch.cuda.synchronize()
    print(f'{name} pure fill of a P-sized buffer ({big.numel() * 4 / 1e9:.2f} GB): {big.numel() * 4 * 3 / e0.elapsed_time(e1) / 1e6:.1f} GB/s (write-only ceiling)')
    del big
    ref = next(iter(outs))
    same = all(torch.equal(outs[ref], o) for o in outs.values()) and all(torch.equal(gps[ref], o) for o in gps.values())
    nbytes = 4 * (Pm.numel() + g.numel())
    for (lname, tag), v in sorted(res.items()):
        us = statistics.median(v)
        print(f'{name} {lname:8s} {tag} median {us:8.1f} us = {nbytes / us / 1e3:7.1f} GB/s of P + out ({nbytes / 1e9:.2f} GB)   variants bit-identical: {same}')
