#!/usr/bin/env python3
"""GPU diagnostic (not part of the product): MIOpen fp32 conv numerics vs CPU and per-layer timing at config-3 size."""
import os, sys, time, json
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = 'cuda:0'
print('torch', torch.__version__, 'allow_tf32 cudnn', torch.backends.cudnn.allow_tf32, 'matmul', torch.backends.cuda.matmul.allow_tf32,
      'benchmark', torch.backends.cudnn.benchmark, flush=True)
print(torch.cuda.get_device_name(0), flush=True)

def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())

torch.manual_seed(0)
x = torch.poisson(torch.full((2, 4, 64, 80), 0.1))
w = torch.randn(32, 4, 5, 5) * 0.1
ref = F.conv2d(x, w, padding=2)
for tf32 in (True, False):
    torch.backends.cudnn.allow_tf32 = tf32
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        y = F.conv2d(x.to(dev), w.to(dev), padding=2).cpu()
        print(f'conv 4->32 k5 allow_tf32={tf32} benchmark={bench}: rel err vs CPU {relerr(y, ref):.3e}', flush=True)
x2 = (torch.rand(6, 256, 9, 10) < 0.3).float()
w2 = torch.randn(512, 256, 5, 5) * 0.02
ref2 = F.conv2d(x2, w2, stride=2, padding=2)
for tf32 in (True, False):
    torch.backends.cudnn.allow_tf32 = tf32
    y = F.conv2d(x2.to(dev), w2.to(dev), stride=2, padding=2).cpu()
    print(f'conv 256->512 k5 s2 allow_tf32={tf32}: rel err {relerr(y, ref2):.3e}', flush=True)

# timing per layer at B*T = 80
torch.backends.cudnn.allow_tf32 = False
layers = [('bottom', 4, 32, 5, 1, 2, (260, 346)), ('conv1', 32, 64, 5, 2, 2, (260, 346)), ('conv2', 64, 128, 5, 2, 2, (130, 173)),
          ('conv3', 128, 256, 5, 2, 2, (65, 87)), ('conv4', 256, 512, 5, 2, 2, (33, 44)), ('res', 512, 512, 3, 1, 1, (17, 22)),
          ('deconv4', 512, 256, 5, 1, 0, (37, 48)), ('deconv3', 256, 128, 5, 1, 0, (69, 91)), ('deconv2', 128, 64, 5, 1, 0, (134, 177)),
          ('deconv1', 64, 32, 5, 1, 0, (264, 350)), ('pd4', 256, 1, 3, 1, 0, (262, 348)), ('pd3', 128, 1, 3, 1, 0, (262, 348)),
          ('pd2', 64, 1, 3, 1, 0, (262, 348)), ('pd1', 32, 1, 3, 1, 0, (262, 348))]
NB = int(os.environ.get('NB', '80'))
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    tot = 0.0
    for name, ci, co, k, s, p, (h, wd) in layers:
        xi = (torch.rand(NB, ci, h, wd, device=dev) < 0.3).float().requires_grad_()
        wt = (torch.randn(co, ci, k, k, device=dev) * 0.02).requires_grad_()
        def run():
            y = F.conv2d(xi, wt, stride=s, padding=p)
            y.backward(torch.ones_like(y))
        t0 = time.time(); run(); torch.cuda.synchronize(); first = time.time() - t0
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(3): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        macs = NB * co * ci * k * k * ((h + 2 * p - k) // s + 1) * ((wd + 2 * p - k) // s + 1)
        tot += ms
        print(f'benchmark={bench} {name:8s} fwd+bwd {ms:8.2f} ms  ({3 * 2 * macs / ms / 1e9:7.1f} TFLOP/s)  first call {first:6.1f} s', flush=True)
        del xi, wt
    print(f'benchmark={bench} total conv fwd+bwd per iteration (res x4 not multiplied): {tot:.1f} ms', flush=True)
