"""Shared helpers for the tests (oracle access lives here: only tests/ may use oracle/)."""
import json
import os

import numpy as np
import torch

from oracle import c_oracle, ref_network, sj_clock_driven as sj  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def kat_cases():
    z = load_npz('neuron_kat.npz')
    for i in range(int(z['n_cases'])):
        p = f'c{i:03d}_'
        cfg = json.loads(str(z[p + 'cfg']))
        case = {k[len(p):]: z[k] for k in z.files if k.startswith(p) and k != p + 'cfg'}
        yield i, cfg, case


def bits(a):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.ascontiguousarray(a, dtype=np.float32).view(np.int32)


def bit_equal(a, b):
    return np.array_equal(bits(a), bits(b))


def rel_err(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def synth_input(B, T, C, seed, H=260, W=346, lam=0.05):
    g = torch.Generator().manual_seed(seed)
    return torch.poisson(torch.full((B, T, C, H, W), lam), generator=g)


def synth_label(B, seed, H=260, W=346):
    g = torch.Generator().manual_seed(seed)
    gt = 0.5 + 9.5 * torch.rand(B, 1, H, W, generator=g)
    gt[torch.rand(B, 1, H, W, generator=g) < 0.25] = float('nan')
    return gt
