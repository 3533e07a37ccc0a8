#!/usr/bin/env python3
"""s_memtime stamps of thread 0 of workgroup 0 over its first items of ss_upconv_box_dgrad_f32 (library built with -DSS_BX_TRACE=1), deconv1 geometry.
slots: 0 item top, 1 before the top barrier (sign flip / per-tile addresses done), 2 after it, 3 stage 0's MFMAs done, 4 stage 0 committed + barrier,
5 last stage's MFMAs done, 6 its commit done, 7 after the end barrier, 8 window stored, 9 epilogue issued (last chunk of a tile), 10 chunk index."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stereospike_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ['SS_LIB'])
from stereospike_amd import fused
from stereospike_amd.network.blocks import NNConvUpsampling
dev = 'cuda:0'
Cin, Cout, (h, w), (H, W), NB = 64, 32, (130, 173), (260, 346), 80
up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(dev)
tabs = up._tables(h, w, torch.device(dev))
bt = fused.box_tables(tabs, H, W)
gy = torch.randn(NB, H, W, Cout, device=dev)
wt = up.up[1].weight.detach().contiguous()
gx = torch.empty(NB, h, w, Cin, device=dev)
box = _lib.upconv_boxsum(gy, bt, NB, Cout, H, W)
for _ in range(3):
    _lib.upconv_box_dgrad(box, wt, bt, gx, NB, Cin, Cout, h, w)
torch.cuda.synchronize()
buf = np.zeros((64, 16), np.uint64)
L = _lib.lib()
L.ss_debug_box_trace.argtypes = [ctypes.c_void_p]
assert L.ss_debug_box_trace(buf.ctypes.data) == 0
print('item chunk | setup  top-bar | stage0 mfma  commit+bar | stages 1..last | last commit | end-bar | win store | epilogue | item total')
for i in range(40):
    b = [int(v) for v in buf[i]]
    nxt = int(buf[i + 1, 0]) - b[0] if buf[i + 1, 0] else -1
    print(f'{i:3d} {b[10]:3d} | {b[1]-b[0]:6d} {b[2]-b[1]:6d} | {b[3]-b[2]:7d} {b[4]-b[3]:7d} | {b[5]-b[4]:8d} | {b[6]-b[5]:6d} | {b[7]-b[6]:6d} | {b[8]-b[7]:6d} | {b[9]-b[8]:6d} | {nxt:7d}')
