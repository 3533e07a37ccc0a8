#!/usr/bin/env python3
"""GPU diagnostic: per-layer time of the MIOpen convs that remain (encoder 5x5 s2, bottleneck 3x3) in NHWC, find mode,
config-3 size (NB = 80): forward, and forward+backward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereospike_amd import miopen_cache
miopen_cache.enable(skip_naive_solvers=True)
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = bool(int(os.environ.get('FIND', '1')))
dev = 'cuda:0'
NB = 80
layers = [('bottom', 4, 32, 5, 1, 2, (260, 346), False), ('conv1', 32, 64, 5, 2, 2, (260, 346), True), ('conv2', 64, 128, 5, 2, 2, (130, 173), True),
          ('conv3', 128, 256, 5, 2, 2, (65, 87), True), ('conv4', 256, 512, 5, 2, 2, (33, 44), True), ('res(x4)', 512, 512, 3, 1, 1, (17, 22), True)]
tot = 0
for name, ci, co, k, s, p, (h, w), need_dx in layers:
    x = (torch.rand(NB, ci, h, w, device=dev) < 0.3).float().contiguous(memory_format=torch.channels_last).requires_grad_(need_dx)
    wt = (torch.randn(co, ci, k, k, device=dev) * 0.02).contiguous(memory_format=torch.channels_last).requires_grad_()
    def fwd(): return F.conv2d(x, wt, stride=s, padding=p)
    def both():
        y = fwd(); y.backward(torch.ones_like(y))
    both(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    with torch.no_grad():
        fwd(); e0.record()
        for _ in range(5): fwd()
        e1.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / 5
    e0.record()
    for _ in range(5): both()
    e1.record(); torch.cuda.synchronize()
    tb = e0.elapsed_time(e1) / 5
    macs = NB * co * ci * k * k * ((h + 2 * p - k) // s + 1) * ((w + 2 * p - k) // s + 1)
    nb = 3 if need_dx else 2
    print(f'{name:8s} fwd {tf:6.2f} ms ({2 * macs / tf / 1e9:6.1f} TF)   fwd+bwd {tb:6.2f} ms ({nb * 2 * macs / tb / 1e9:6.1f} TF)', flush=True)
    tot += tb * (4 if name.startswith('res') else 1)
print(f'encoder + bottleneck total fwd+bwd: {tot:.1f} ms')
