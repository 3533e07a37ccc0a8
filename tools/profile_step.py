#!/usr/bin/env python3
"""Steady-state kernel breakdown of the DEFAULT (find-mode) training step with torch.profiler, started AFTER the warm-up
steps so MIOpen's solver choice is the un-profiled one (rocprofv3 around the whole process perturbs the find search)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereospike_amd import miopen_cache
miopen_cache.enable(skip_naive_solvers=True)
import torch
from torch.profiler import profile, ProfilerActivity
_B, _T, _DT = int(os.environ.get('PB', 16)), int(os.environ.get('PT', 5)), os.environ.get('PDT', 'f32')   # PB / PT / PDT: batch, T, f32|bf16|f16
sys.argv = [sys.argv[0]]
import bench
torch.backends.cudnn.benchmark = True
from stereospike_amd import gemm_tuning
gemm_tuning.enable(0)
dev = torch.device('cuda:0')
from stereospike_amd.engine import Trainer, synthetic_batch
net = bench.build_net('StereoSpike', dev)
tr = Trainer(net)
x, gt = synthetic_batch(_B, _T, seed=2021, device=dev)
if _DT != 'f32':
    _plain = tr.step
    _amp = torch.bfloat16 if _DT == 'bf16' else torch.float16

    def _amp_step(x_, gt_):
        with torch.autocast('cuda', dtype=_amp):
            return _plain(x_, gt_)
    tr.step = _amp_step
for _ in range(3):
    tr.step(x, gt)
torch.cuda.synchronize()
STEPS = 4
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(STEPS):
        tr.step(x, gt)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, 'device_time_total', None)
    if t is None:
        t = getattr(e, 'cuda_time_total', 0)
    if t and e.device_type is not None and 'cuda' in str(e.device_type).lower():
        rows.append((e.key, e.count, t))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print(f'total device kernel time per step: {tot / STEPS / 1e3:.2f} ms')
for k, c, t in rows[:45]:
    print(f'{t / STEPS / 1e3:8.3f} ms/step  x{c / STEPS:6.1f}  {k[:130]}')
