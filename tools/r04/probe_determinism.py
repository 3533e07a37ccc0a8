#!/usr/bin/env python3
"""Is the default fp32 training step bit-reproducible run to run (no MIOpen, but hipBLASLt GEMMs)?  Two Trainers from the same seed, two steps each,
parameters compared bit for bit — decides how tight tests/test_gpu_03_model.py::test_dp_reducer_on_rccl_single_rank can be."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from stereospike_amd import miopen_cache  # noqa: E402
miopen_cache.enable_hermetic()
import torch  # noqa: E402
from stereospike_amd import gemm_tuning  # noqa: E402
gemm_tuning.enable(0)
from _models import product  # noqa: E402
from stereospike_amd.engine import Trainer, synthetic_batch  # noqa: E402

DEV = 'cuda:0'
for name, H, W, B in (('PLIFNet', 64, 80, 2), ('StereoSpike', 64, 80, 2), ('StereoSpike', 260, 346, 2)):
    x, gt = synthetic_batch(B, 5, C=4, H=H, W=W, seed=5, device=DEV, lam=0.08)
    runs = []
    for r in range(3):
        torch.manual_seed(7)
        net = product(name, input_size=(H, W)).to(DEV)
        tr = Trainer(net)
        losses = [float(tr.step(x, gt)[0]) for _ in range(2)]
        runs.append((losses, [p.detach().clone() for p in net.parameters()]))
    for r in (1, 2):
        same = [torch.equal(a, b) for a, b in zip(runs[0][1], runs[r][1])]
        md = max(float((a - b).abs().max()) for a, b in zip(runs[0][1], runs[r][1]))
        print(name, H, W, 'run', r, 'losses', runs[0][0], runs[r][0], 'params bit-equal:', all(same), f'({sum(same)}/{len(same)})', 'max abs diff', md, flush=True)
