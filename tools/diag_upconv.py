#!/usr/bin/env python3
"""GPU diagnostic: per-layer fwd+bwd time of the decoder / head synapses at config-3 size (NB = B*T = 80):
projected (1x1 projection + fused gather, this repo) vs the reference's two-op form on MIOpen; plus where MIOpen
keeps its compiled kernels (cold-start cost)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereospike_amd import miopen_cache
miopen_cache.enable()
import torch
from stereospike_amd.network.blocks import NNConvUpsampling
dev = 'cuda:0'
NB = int(os.environ.get('NB', '80'))
layers = [('deconv4', 512, 256, 5, (17, 22), (33, 44)), ('deconv3', 256, 128, 5, (33, 44), (65, 87)),
          ('deconv2', 128, 64, 5, (65, 87), (130, 173)), ('deconv1', 64, 32, 5, (130, 173), (260, 346)),
          ('pd4', 256, 1, 3, (33, 44), (260, 346)), ('pd3', 128, 1, 3, (65, 87), (260, 346)),
          ('pd2', 64, 1, 3, (130, 173), (260, 346)), ('pd1', 32, 1, 3, (260, 346), (260, 346))]
which = os.environ.get('WHICH', 'projected,miopen').split(',')
from stereospike_amd import fused
from stereospike_amd import config as _config
_cm = _config.engine_config()   # (the PROJECTION_IMPL / P_CHUNK_BYTES knobs of rounds 2 - 4 left EngineConfig in round 5: the chunk size is the private fused._P_CHUNK_BYTES)
_cm.__enter__()                # a one-off diagnostic script: the ambient configuration for the rest of the process
only = os.environ.get('LAYERS')
if only:
    layers = [l for l in layers if l[0] in only.split(',')]
print('projection chunk MB', fused._P_CHUNK_BYTES >> 20)
for name, ci, co, k, insz, up in layers:
    m = NNConvUpsampling(ci, co, k, up, bias=(co == 1)).to(dev)
    x = (torch.rand(NB, ci, *insz, device=dev) < 0.3).float().requires_grad_()
    for mode in which:
        if mode == 'miopen' and name not in os.environ.get('MIOPEN_LAYERS', 'deconv4,pd4').split(','):
            continue
        fn = m.forward_projected if mode == 'projected' else m.forward_projected_cl if mode == 'cl' else m.forward
        xin = x
        if mode == 'cl':
            xin = x.detach().permute(0, 2, 3, 1).contiguous().requires_grad_()
        def run():
            y = fn(xin)
            y.backward(torch.ones_like(y))
        t0 = time.time(); run(); torch.cuda.synchronize(); first = time.time() - t0
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(3): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        # forward only
        with torch.no_grad():
            fn(xin); torch.cuda.synchronize()
            e0.record()
            for _ in range(3): fn(xin)
            e1.record(); torch.cuda.synchronize()
        fms = e0.elapsed_time(e1) / 3
        print(f'{name:8s} {mode:10s} fwd+bwd {ms:8.2f} ms   fwd {fms:7.2f} ms   first call {first:6.1f} s   '
              f'peak mem {torch.cuda.max_memory_allocated() / 1e9:.1f} GB', flush=True)
    del m, x
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
if os.environ.get('QUIET'):
    sys.exit(0)
for d in ('~/.cache/miopen', '~/.config/miopen', miopen_cache.CACHE_DIR, '/tmp'):
    p = os.path.expanduser(d)
    if os.path.exists(p):
        print(subprocess.run(f'du -sh {p}; find {p} -maxdepth 3 | head -20', shell=True, capture_output=True, text=True).stdout)
print({k: v for k, v in os.environ.items() if 'MIOPEN' in k})
