#!/usr/bin/env python3
"""Which ingredient of tests/test_gpu_05_full_size.py::test_config2_* makes GraphedTrainer's capture_end crash: every variant in its own process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = ['mono_fresh', 'mono_after_plain_passes', 'mono_after_plain_passes_fp32', 'mono_after_eager_trainer', 'mono_after_plain_passes_64x80']

if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    from stereospike_amd import miopen_cache
    miopen_cache.enable()
    import torch
    from stereospike_amd.clock_driven import functional
    from stereospike_amd.engine import GraphedTrainer, Trainer, synthetic_batch
    from stereospike_amd.network import SNN_models as S
    v = sys.argv[1]
    dev = 'cuda:0'
    torch.manual_seed(2021)
    size = (64, 80) if '64x80' in v else (260, 346)
    cls = S.fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike if v.startswith('bino') else S.fromZero_feedforward_multiscale_tempo_monocular_SpikeFlowNetLike
    net = cls(tau=3., use_plif=True, multiply_factor=30., input_size=size).to(dev)
    C = 4 if v.startswith('bino') else 2
    x, gt = synthetic_batch(8, 1, C=C, seed=2021, device=dev, lam=0.12)
    if size != (260, 346):
        x, gt = x[..., :size[0], :size[1]].contiguous(), gt[..., :size[0], :size[1]].contiguous()
    amp = None if 'fp32' in v else torch.bfloat16
    if 'legacy' in v:
        net.config = net.config.replace(X16_OWN_KERNELS=False)
    if 'after_eager_trainer' in v:
        state = {k: t.clone() for k, t in net.state_dict().items()}
        Trainer(net, amp_dtype=amp).step(x, gt)
        net.load_state_dict(state)
        functional.reset_net(net)
    if 'after_plain_passes' in v:
        from stereospike_amd.network.loss import Total_Loss
        if 'nolr' in v:
            net.config = net.config.replace(LOWRANK_HEAD_GRAD=False)
        for _ in range(1 if '_one' in v else 2):
            functional.reset_net(net)
            net.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=amp):
                d = net.forward_sequence(x)
                L = Total_Loss()(d, gt, None)
            L.backward()
        if 'zerograd' in v:
            net.zero_grad(set_to_none=True)
        if 'detach' in v:
            net.detach()
        if 'keepalive_emptycache' in v:
            net.zero_grad(set_to_none=True)
            functional.reset_net(net)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        elif 'del_only' in v:
            del d, L
        elif 'del_zerograd' in v:
            del d, L
            net.zero_grad(set_to_none=True)
        elif 'del_gc' in v:
            del d, L
            net.zero_grad(set_to_none=True)
            import gc
            gc.collect()
            torch.cuda.synchronize()
        elif 'emptycache' in v:
            del d, L
            net.zero_grad(set_to_none=True)
            functional.reset_net(net)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        functional.reset_net(net)
    tr = GraphedTrainer(net, amp_dtype=amp, warmup=2)
    out = tr.step(x, gt)
    torch.cuda.synchronize()
    print(v, 'OK loss', float(out[0]), flush=True)
    out = tr.step(x, gt)
    torch.cuda.synchronize()
    print(v, 'OK second replay', float(out[0]), flush=True)
    sys.exit(0)

for v in VARIANTS:
    r = subprocess.run([sys.executable, '-X', 'faulthandler', os.path.abspath(__file__), v], capture_output=True, text=True, timeout=600)
    tail = [ln for ln in (r.stdout + r.stderr).splitlines() if ('OK' in ln or 'Error' in ln or 'Fatal' in ln or 'File "/' in ln and 'stereospike' in ln)]
    print(f'== {v}: rc {r.returncode}', *tail[-6:], sep='\n   ', flush=True)
