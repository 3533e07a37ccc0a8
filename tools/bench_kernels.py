#!/usr/bin/env python3
"""GPU micro-benchmark of the fused neuron kernels alone (HIP events, config-3 layer sizes): achieved algorithmic GB/s
vs the HBM roofline, next to a plain device copy of the same byte count.  SS_LIB=<path> selects a library variant."""
import ctypes as C
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereospike_amd import _lib
if os.environ.get('SS_LIB'):
    _lib.LIB_PATH = os.environ['SS_LIB']
dev = 'cuda:0'
B, T = int(os.environ.get('B', 16)), int(os.environ.get('T', 5))
layers = {'bottom/deconv1 32x260x346': 32 * 260 * 346, 'conv1/deconv2 64x130x173': 64 * 130 * 173,
          'conv2/deconv3 128x65x87': 128 * 65 * 87, 'conv3/deconv4 256x33x44': 256 * 33 * 44, 'conv4/res 512x17x22': 512 * 17 * 22}
reps = int(os.environ.get('REPS', 20))


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
for name, n in layers.items():
    N = B * n
    x = torch.randn(T, N, device=dev) * 0.15
    skip = (torch.rand(T, N, device=dev) < 0.3).float()
    out, h, g = torch.empty_like(x), torch.empty_like(x), torch.randn(T, N, device=dev)
    gx = torch.empty_like(x)
    v = torch.empty(N, device=dev)
    nnz = torch.zeros(2, dtype=torch.int64, device=dev)
    k = torch.tensor([1 / 3.], device=dev)
    r = {}
    ms = timeit(lambda: _lib.neuron_fwd(x, None, None, out, h, v, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0))
    r['fwd_train_IF'] = (12 * T * N / ms / 1e6, ms)
    ms = timeit(lambda: _lib.neuron_fwd(x, None, skip, out, h, v, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0))
    r['fwd_train_IF_skip'] = (16 * T * N / ms / 1e6, ms)
    ms = timeit(lambda: _lib.neuron_fwd(x, None, None, out, h, v, nnz, T, N, 10.0, 0, 2.0, None, 1.0, 0.0))
    r['fwd_train_IF_count'] = (12 * T * N / ms / 1e6, ms)
    ms = timeit(lambda: _lib.neuron_fwd(x, None, None, out, None, v, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0))
    r['fwd_infer_IF'] = (8 * T * N / ms / 1e6, ms)
    ms = timeit(lambda: _lib.neuron_fwd(x, None, None, out, h, v, None, T, N, 10.0, 2, 2.0, k, 1.0, 0.0))
    r['fwd_train_PLIF'] = (12 * T * N / ms / 1e6, ms)
    ms = timeit(lambda: _lib.neuron_bwd(g, None, h, None, gx, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True))
    r['bwd_IF_atan'] = (12 * T * N / ms / 1e6, ms)
    ms = timeit(lambda: _lib.neuron_bwd(g, None, h, None, gx, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 1, 4.0, True))
    r['bwd_IF_sigmoid'] = (12 * T * N / ms / 1e6, ms)
    gk, ws = torch.zeros(1, device=dev), torch.empty(_lib.gk_ws_floats(), device=dev)
    ms = timeit(lambda: _lib.neuron_bwd(g, None, h, None, gx, None, gk, ws, T, N, 10.0, 2, 2.0, k, 1.0, 0.0, 1, 4.0, True))
    r['bwd_PLIF_sigmoid_gk'] = (12 * T * N / ms / 1e6, ms)
    ms = timeit(lambda: out.copy_(x))
    r['torch_copy(8B/elt)'] = (8 * T * N / ms / 1e6, ms)
    res[name] = r
    print(name, f'N={N} T={T}')
    for kk, (gbs, ms) in r.items():
        print(f'   {kk:24s} {gbs:8.1f} GB/s  {ms * 1e3:9.1f} us  ({gbs / 8000:.1%} of 8 TB/s)')
    del x, skip, out, h, g, gx
    torch.cuda.empty_cache()
json.dump(res, open(os.path.join('gpurun_out', 'bench_kernels.json'), 'w'), indent=1)
