#!/usr/bin/env python3
"""Inference throughput of the fused path (the reference's test.py loop: reset -> T-step forward -> MDE), frames/s on one MI355X.
Not the headline metric (BASELINE.json quotes TRAIN frames/s); reported beside it in profiles/README.md."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stereospike_amd import miopen_cache
miopen_cache.enable(skip_naive_solvers=True)
import torch
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16); ap.add_argument('--T', type=int, default=5)
ap.add_argument('--graph', type=int, default=0, help='1: replay a captured HIP graph (engine.GraphedInference)')
ap.add_argument('--steps', type=int, default=20); ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'f16'])
a = ap.parse_args()
torch.backends.cudnn.benchmark = True
from stereospike_amd import gemm_tuning
gemm_tuning.enable(0)
from stereospike_amd.clock_driven import functional, surrogate
from stereospike_amd.engine import synthetic_batch
from stereospike_amd.network.SNN_models import StereoSpike
from stereospike_amd.network.metrics import MeanDepthError
dev = torch.device('cuda', 0)
torch.manual_seed(2021)
net = StereoSpike(surrogate_function=surrogate.ATan(), multiply_factor=10.).to(dev).eval()
x, gt = synthetic_batch(a.batch, a.T, seed=2021, device=dev)
amp = dict(device_type='cuda', dtype={'bf16': torch.bfloat16, 'f16': torch.float16}.get(a.dtype, torch.float32), enabled=a.dtype != 'f32')
def step():
    functional.reset_net(net)
    with torch.no_grad(), torch.autocast(**amp):
        d, _ = net.forward_sequence(x)
    return MeanDepthError(d[0].float(), gt)
if a.graph:
    from stereospike_amd.engine import GraphedInference
    d_eager = step.__globals__  # noqa
    functional.reset_net(net)
    with torch.no_grad(), torch.autocast(**amp):
        ref, _ = net.forward_sequence(x)
    ref = [t.clone() for t in ref]
    gi = GraphedInference(net, x, amp_dtype=None if a.dtype == 'f32' else amp['dtype'])
    out = gi(x)[0]
    torch.cuda.synchronize()
    same = all(torch.equal(p, q) for p, q in zip(out, ref))
    print('graph replay == eager, bit for bit:', same, file=sys.stderr)

    def step():
        d, _ = gi(x)
        return MeanDepthError(d[0].float(), gt)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): m = step()
torch.cuda.synchronize(); el = time.perf_counter() - t0
print(json.dumps({'metric': 'inference frames/sec (260x346xT stereo voxels)', 'value': round(a.batch * a.steps / el, 2), 'ms_per_step': round(1e3 * el / a.steps, 3),
                  'batch': a.batch, 'T': a.T, 'hip_graph': bool(a.graph), 'dtype': a.dtype, 'mde_m': round(float(m), 5), 'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 1e9, 2)}))
