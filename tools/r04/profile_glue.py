#!/usr/bin/env python3
"""Where the torch-native (non-HIP-extension) device time of the default training step comes from: aten ops grouped by the python source line that
issued them (torch.profiler with_stack), steady state, config 3."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
sys.argv = [sys.argv[0]]
import bench
dev = torch.device('cuda:0')
from stereospike_amd.engine import Trainer, synthetic_batch
net = bench.build_net('StereoSpike', dev)
tr = Trainer(net)
x, gt = synthetic_batch(16, 5, seed=2021, device=dev)
for _ in range(3):
    tr.step(x, gt)
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(STEPS):
        tr.step(x, gt)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if not e.name.startswith('aten::') or e.cpu_parent is not None and e.cpu_parent.name.startswith('aten::'):
        continue                                        # top-level aten ops only
    t = getattr(e, 'device_time_total', 0) or 0
    if t <= 0:
        continue
    frame = next((s for s in (e.stack or []) if '/repo/' in s or 'stereospike_amd' in s or 'bench.py' in s), (e.stack or ['?'])[0] if e.stack else '?')
    agg[(e.name, frame[-110:])][0] += 1
    agg[(e.name, frame[-110:])][1] += t
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print(f'top-level aten ops with device time: {tot / STEPS / 1e3:.3f} ms/step')
for (name, frame), (c, t) in rows[:70]:
    print(f'{t / STEPS / 1e3:7.3f} ms/step x{c / STEPS:5.1f}  {name:28s} {frame}')
