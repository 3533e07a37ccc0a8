#!/usr/bin/env python3
"""bench.py's eval-MDE leg (fresh seed-2021 StereoSpike, B = 1, T = 5, no_grad): the product against the CPU oracle, tensor by tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = [sys.argv[0]]
import bench
from stereospike_amd import fused
from stereospike_amd.clock_driven import functional as F_
from stereospike_amd.engine import synthetic_batch
from stereospike_amd.network.metrics import MeanDepthError
from oracle import ref_network as rn, sj_clock_driven as sj
dev = torch.device('cuda:0')
x0, gt0 = synthetic_batch(1, 5, seed=2021)
torch.manual_seed(2021)
orc = rn.build('StereoSpike', multiply_factor=10., surrogate_function=sj.ATan())
with torch.no_grad():
    d_ref, s_ref = rn.run_sequence(orc, x0)
with torch.no_grad(), rn.float64_convs(orc):
    d64, s64 = rn.run_sequence(orc, x0)
print('oracle with float64 convolutions: eval MDE', float(rn.mean_depth_error(d64[0], gt0)), ' spikes differing vs the fp32 eager oracle',
      [f'{float((a != b).float().mean()):.2e}' for a, b in zip(s64, s_ref)], flush=True)
print('oracle eval MDE', float(rn.mean_depth_error(d_ref[0], gt0)), 'spike tensors', [tuple(s.shape) for s in s_ref], [round(float(s.mean()), 5) for s in s_ref], flush=True)
for name, sw in [('default', {}), ('all off', {'SPIKE_CONV_FWD_MFMA': False, 'DENSE_CONV_S1_MFMA': False, 'PACKED_HEAD': False, 'PACK_SPIKES': False, 'EXACT_SPLIT_GEMM': False})]:
    from stereospike_amd.config import EngineConfig
    net0 = bench.build_net('StereoSpike', dev, config=EngineConfig.default().replace(**sw))      # (knobs are fields of the network's configuration since round 4)
    fused.TIMER.enabled = True
    fused.TIMER.clear()
    with torch.no_grad():
        F_.reset_net(net0)
        d0, s0 = net0.forward_sequence(x0.to(dev))
    torch.cuda.synchronize()
    print(name, {k: v['launches'] for k, v in fused.TIMER.summary().items()})
    fused.TIMER.enabled = False
    print(name, 'eval MDE', float(MeanDepthError(d0[0], gt0.to(dev))))
    for i, (a, b) in enumerate(zip(s0, s_ref)):
        a = a.cpu()
        print(f'   spikes[{i}] {tuple(a.shape)} mean {float(a.mean()):.5f} vs {float(b.mean()):.5f}  differing {float((a != b).float().mean()):.3e}   vs float64-conv oracle {float((a != s64[i]).float().mean()):.3e}')
    for i, (a, b) in enumerate(zip(d0, d_ref)):
        a = a.cpu()
        print(f'   depth[{i}] max |diff| / max {float((a - b).abs().max() / b.abs().max()):.3e}  mean {float(a.mean()):.5f} vs {float(b.mean()):.5f}   vs float64-conv oracle {float((a - d64[i]).abs().max() / d64[i].abs().max()):.3e}')
    # the same forward step by step (the reference's protocol: reset, then T single-step calls)
    with torch.no_grad():
        F_.reset_net(net0)
        for t in range(5):
            d1, s1 = net0(x0[:, t:t + 1].to(dev))
    print(name, 'stepwise eval MDE', float(MeanDepthError(d1[0], gt0.to(dev))), ' spikes differing vs oracle', [f'{float((a.cpu() != b).float().mean()):.2e}' for a, b in zip(s1, s_ref)], flush=True)
    del net0
