#!/usr/bin/env python3
"""Which aten-level ops (with input shapes) own the element-wise / reduction / copy kernel time of the default training step — to find
avoidable passes.  torch.profiler with record_shapes, after warm-up; prints ops whose device time is >= 30 us per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereospike_amd import miopen_cache
miopen_cache.enable(skip_naive_solvers=True)
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = [sys.argv[0]]
import bench
torch.backends.cudnn.benchmark = True
from stereospike_amd import gemm_tuning
gemm_tuning.enable(0)
dev = torch.device('cuda:0')
from stereospike_amd.engine import Trainer, synthetic_batch
# round 6: DTYPE=f16|bf16|f32, T, B, RATES=1 select the configuration (default: the headline); STACK=1 groups by the Python call site instead of input shapes
DT = {'f16': torch.float16, 'bf16': torch.bfloat16}.get(os.environ.get('DTYPE', 'f32'))
Tn, Bn = int(os.environ.get('T', 5)), int(os.environ.get('B', 16))
net = bench.build_net('StereoSpike', dev)
tr = Trainer(net, amp_dtype=DT, count_rates=bool(int(os.environ.get('RATES', 0))))
x, gt = synthetic_batch(Bn, Tn, seed=2021, device=dev)
for _ in range(3):
    tr.step(x, gt)
torch.cuda.synchronize()
STEPS = 2
STACK = bool(int(os.environ.get('STACK', 0)))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True, with_stack=STACK) as prof:
    for _ in range(STEPS):
        tr.step(x, gt)
    torch.cuda.synchronize()
rows = []
for e in (prof.key_averages(group_by_stack_n=8) if STACK else prof.key_averages(group_by_input_shape=True)):
    t = getattr(e, 'self_device_time_total', None)
    if t is None:
        t = getattr(e, 'self_cuda_time_total', 0)
    if t and t / STEPS >= float(os.environ.get('MIN_US', 30)):
        where = ' <- '.join(f.split('/')[-1] for f in (e.stack or []) if 'stereospike_amd' in f or 'bench' in f)[:260] if STACK else str(e.input_shapes)[:150]
        rows.append((t / STEPS, e.count / STEPS, e.key, where))
rows.sort(key=lambda r: -r[0])
skip = ('mm', 'bmm', 'convolution', 'miopen', '_FusedNeuron', 'UpConv', 'SpikeConv', 'IPool', 'ScaleLoss')
tot = 0
for t, c, k, sh in rows:
    if any(s in k for s in skip) or (os.environ.get('ATEN_ONLY') and not k.startswith('aten::')):
        continue
    tot += t
    print(f'{t / 1e3:7.3f} ms/step x{c:5.1f}  {k:40s} {sh}')
print(f'listed total {tot / 1e3:.2f} ms/step')
