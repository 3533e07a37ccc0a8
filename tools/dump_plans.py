#!/usr/bin/env python3
"""Dispatch plans (`net.plan()`: the kernel form every layer took, recorded at the dispatch sites) of the five BASELINE.json configurations, at full resolution
with a small batch — which kernel forms and EngineConfig knobs any shipped configuration can still reach.
usage (GPU box): python tools/dump_plans.py > gpurun_out/r05/plans.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stereospike_amd.clock_driven import functional, surrogate  # noqa: E402
from stereospike_amd.engine import synthetic_batch  # noqa: E402
from stereospike_amd.network import SNN_models as S  # noqa: E402
from stereospike_amd.network.loss import Total_Loss  # noqa: E402
from stereospike_amd import fused  # noqa: E402

dev = torch.device('cuda:0')
CONFIGS = {
    'config2: monocular PLIF, T=1, bf16, B=8': dict(model='mono', T=1, B=2, dt=torch.bfloat16),
    'config3/4: StereoSpike, T=5, fp32, B=16 per GPU': dict(model='stereo', T=5, B=2, dt=None),
    'config5: StereoSpike, T=10, fp16, B=32 per GPU, counters': dict(model='stereo', T=10, B=2, dt=torch.float16, rates=True),
    'PLIF binocular (the paper\'s model), T=5, fp32': dict(model='plif', T=5, B=2, dt=None),
}
out = {}
for name, c in CONFIGS.items():
    torch.manual_seed(2021)
    if c['model'] == 'stereo':
        net = S.StereoSpike(surrogate_function=surrogate.ATan(), multiply_factor=10.)
    elif c['model'] == 'plif':
        net = S.fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike(tau=3., use_plif=True, multiply_factor=30.)
    else:
        net = S.fromZero_feedforward_multiscale_tempo_monocular_SpikeFlowNetLike(tau=3., use_plif=True, multiply_factor=30.)
    net = net.to(dev)
    x, gt = synthetic_batch(c['B'], c['T'], seed=2021)
    if c['model'] == 'mono':
        x = x[:, :, :2]
    x, gt = x.to(dev), gt.to(dev)
    functional.reset_net(net)
    fused.TIMER.enabled = True
    fused.TIMER.clear()
    import contextlib
    with (torch.autocast('cuda', dtype=c['dt']) if c['dt'] is not None else contextlib.nullcontext()):
        rates = {} if c.get('rates') else None
        res = net.forward_sequence(x, rates) if rates is not None else net.forward_sequence(x)
        d, s = res if isinstance(res, tuple) else (res, None)
        L = Total_Loss()(d, gt, s)
    (L * (16.0 if c['dt'] == torch.float16 else 1.0)).backward()
    torch.cuda.synchronize()
    out[name] = dict(plan=net.plan(), launch_tags={k: v['launches'] for k, v in fused.TIMER.summary().items()})
    fused.TIMER.enabled = False
    del net
print(json.dumps(out, indent=1))
