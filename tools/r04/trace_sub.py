#!/usr/bin/env python3
"""s_memtime stamps of wavefront 0 of workgroup 0 over its first (cot, g) iterations of ss_upconv_sub_fwd_f32 (library built with -DSS_SB_TRACE=1):
slot 0 top of the iteration, 1 after the top barrier, 2 + 2 r after stage r's MFMAs (accumulators read), 3 + 2 r after its commit + barrier."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stereospike_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ['SS_LIB'])
from stereospike_amd import fused
from stereospike_amd.network.blocks import NNConvUpsampling
from oracle import np_pack
dev = 'cuda:0'
Cin, Cout, (h, w), (H, W), NB = 64, 32, (130, 173), (260, 346), 80
up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(dev)
tabs = up._tables(h, w, torch.device(dev))
st = fused.sub_tables(tabs, H, W)
x = (torch.rand(NB, h, w, Cin, device=dev) < 0.2).float()
xp = torch.from_numpy(np_pack.pack(x.cpu().numpy().reshape(-1)).view(np.int32)).to(dev)
wt = up.up[1].weight.detach().contiguous()
y = torch.empty(NB, H, W, Cout, device=dev)
wm = _lib.upconv_sub_prep(wt, st, Cin, Cout)
for _ in range(3):
    _lib.upconv_sub_fwd(None, xp, wm, st, y, NB, Cin, Cout, h, w)
torch.cuda.synchronize()
buf = np.zeros((96, 8), np.uint64)
L = _lib.lib()
L.ss_debug_sub_trace.argtypes = [ctypes.c_void_p]
assert L.ss_debug_sub_trace(buf.ctypes.data) == 0
t0 = int(buf[0, 0])
print('iter  tile cot g inst |  top  +bar |  r0:mfma +commit | r1:mfma +commit | r2:mfma +end   | next top   (clock64 ticks = 100 MHz? see ratio below)')
for i in range(60):
    b = [int(v) for v in buf[i]]
    info = b[7]
    tile, cg, inst = (info >> 32) & 0xffff, info & 0xffff, info >> 48
    d = lambda a, c: (b[c] - b[a]) if b[a] and b[c] else -1
    nxt = int(buf[i + 1, 0]) - b[0] if buf[i + 1, 0] else -1
    print(f'{i:3d} {tile:6d} {cg >> 8:3d} {cg & 255:2d} {inst:4d} | {b[0] - t0:8d} {d(0, 1):5d} | {d(1, 2):7d} {d(2, 3):6d} | {d(3, 4):7d} {d(4, 5):6d} | {d(5, 6):7d} {d(6, 7) if False else 0:6d} | {nxt:7d}')

buf2 = np.zeros((32, 8), np.uint64)
L.ss_debug_sub_trace2.argtypes = [ctypes.c_void_p]
assert L.ss_debug_sub_trace2(buf2.ctypes.data) == 0
print('tile boundary stamps (cycles): records loaded | per-lane setup | first window + stage committed | (cot, g) loops | epilogue issued | to the next tile top')
for i in range(20):
    b = [int(v) for v in buf2[i]]
    nxt = int(buf2[i + 1, 0]) - b[5] if buf2[i + 1, 0] else -1
    print(f'{i:3d} | {b[1] - b[0]:7d} | {b[2] - b[1]:7d} | {b[3] - b[2]:7d} | {b[4] - b[3]:8d} | {b[5] - b[4]:7d} | {nxt:7d}')
