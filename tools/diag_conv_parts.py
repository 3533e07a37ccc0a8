#!/usr/bin/env python3
"""GPU diagnostic: forward / data-gradient / weight-gradient time of every library convolution that remains in the step
(NHWC, MIOpen find mode, config-3 size NB = 80), each part on its own, with the HBM-bound floor of the part beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereospike_amd import miopen_cache
miopen_cache.enable(skip_naive_solvers=True)
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = 'cuda:0'
NB = int(os.environ.get('NB', 80))
layers = [('bottom', 4, 32, 5, 1, 2, (260, 346)), ('conv1', 32, 64, 5, 2, 2, (260, 346)), ('conv2', 64, 128, 5, 2, 2, (130, 173)),
          ('conv3', 128, 256, 5, 2, 2, (65, 87)), ('conv4', 256, 512, 5, 2, 2, (33, 44)), ('res', 512, 512, 3, 1, 1, (17, 22))]


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, ci, co, k, s, p, (h, w) in layers:
    x = (torch.rand(NB, ci, h, w, device=dev) < 0.3).float().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, k, k, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    y = F.conv2d(x, wt, stride=s, padding=p)
    g = torch.randn_like(y)
    ho, wo = y.shape[2:]
    macs = NB * co * ci * k * k * ho * wo
    bx, by, bw = x.numel() * 4, y.numel() * 4, wt.numel() * 4

    def bwd(mask):
        return torch.ops.aten.convolution_backward(g, x, wt, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, mask)
    tf = timeit(lambda: F.conv2d(x, wt, stride=s, padding=p))
    td = timeit(lambda: bwd([True, False, False]))
    tw = timeit(lambda: bwd([False, True, False]))
    fl = (bx + by) / 5.5e9
    print(f'{name:7s} {2 * macs / 1e9:7.1f} GF | fwd {tf:6.3f} ms ({2 * macs / tf / 1e9:6.1f} TF) | dgrad {td:6.3f} ms ({2 * macs / td / 1e9:6.1f} TF) | '
          f'wgrad {tw:6.3f} ms ({2 * macs / tw / 1e9:6.1f} TF) | HBM floor x+y {fl:5.3f} ms', flush=True)
