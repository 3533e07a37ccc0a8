#!/usr/bin/env python3
"""Pin-when-possible hook for the neuron oracle.

The neuron / surrogate arithmetic of the hot path lives in the third-party `spikingjelly.clock_driven` package
(/root/reference/requirements.txt:3 — un-pinned; imported at /root/reference/network/blocks.py:8, SNN_models.py:6, train.py:12-13),
which is NOT installed in the build image and cannot be fetched (no network): the oracle restates its published algorithm
(oracle/sj_clock_driven.py) and the known-answer vectors in tests/golden/neuron_kat.npz are outputs of that restatement —
"PARITY UNPINNED".

The day a real wheel (clock_driven namespace: PyPI 0.0.0.0.4 .. 0.0.0.0.12) is importable, this tool converts that status:
it replays every KAT case through the REAL package's single-step nodes (same op sequence as tests/golden/make_golden.py::eager_neuron:
MultiplyBy -> neuronal_charge -> neuronal_fire -> neuronal_reset (+ skip add), loss = <out, g_out> + <v_last, g_v_last>, backward) and
diffs spikes, h, v_last, g_x, g_v_init and dL/dw against the committed vectors BIT FOR BIT.

    python tools/pin_spikingjelly.py            exit 0: pinned (all cases bit-identical)   1: mismatch (printed)   3: package absent

`check(provider)` is also run by tests/test_pin_spikingjelly.py against the oracle's own module (must reproduce the KATs: validates
this tool) and, when importable, against the real package (else the test skips with the PARITY UNPINNED message)."""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = os.path.join(ROOT, 'tests', 'golden', 'neuron_kat.npz')


def real_provider():
    """(neuron module, surrogate module, version string) of an installed spikingjelly, or None."""
    try:
        neuron = importlib.import_module('spikingjelly.clock_driven.neuron')
        surrogate = importlib.import_module('spikingjelly.clock_driven.surrogate')
    except Exception:
        return None
    if 'stereospike_amd' in (getattr(neuron, '__file__', '') or '') or neuron.__name__.startswith('stereospike_amd'):
        return None                                    # install_dropin() alias of this repo, not the real package
    try:
        from importlib.metadata import version
        ver = version('spikingjelly')
    except Exception:
        ver = 'unknown'
    return neuron, surrogate, ver


def replay(neuron, surrogate, cfg, case):
    T = int(cfg['T'])
    x = torch.tensor(case['x'], requires_grad=True)
    sgf = surrogate.ATan(alpha=cfg['alpha']) if cfg['surrogate'] == 'ATan' else surrogate.Sigmoid(alpha=cfg['alpha'])
    kw = dict(v_threshold=cfg['v_th'], v_reset=cfg['v_reset'], surrogate_function=sgf, detach_reset=True)
    if cfg['kind'] == 'IF':
        node = neuron.IFNode(**kw)
    elif cfg['kind'] == 'LIF':
        node = neuron.LIFNode(tau=cfg['tau'], **kw)
    else:
        node = neuron.ParametricLIFNode(init_tau=cfg['tau'], **kw)
    vi = None
    if 'v_init' in case:
        vi = torch.tensor(case['v_init'], requires_grad=True)
        node.v = vi
    outs, hs = [], []
    for t in range(T):
        xs = torch.mul(x[t], cfg['scale'])
        node.neuronal_charge(xs)
        hs.append(node.v.detach().clone())
        node.neuronal_fire()
        node.neuronal_reset()
        o = node.spike
        if 'skip' in case:
            o = o + torch.tensor(case['skip'][t].astype(np.float32))
        outs.append(o)
    out = torch.stack(outs)
    ((out * torch.tensor(case['g_out'])).sum() + (node.v * torch.tensor(case['g_v_last'])).sum()).backward()
    got = dict(out=out.detach().numpy().astype(np.uint8), h=torch.stack(hs).numpy(), v_last=node.v.detach().numpy(), g_x=x.grad.numpy())
    if vi is not None:
        got['g_v_init'] = vi.grad.numpy()
    if cfg['kind'] == 'PLIF':
        got['g_w'] = np.float32(node.w.grad.item())
    return got


def check(neuron, surrogate, max_report=5):
    """Returns (n_cases, mismatches) — mismatches is a list of (case index, field, max abs difference)."""
    z = np.load(KAT, allow_pickle=False)
    bad = []
    n = int(z['n_cases'])
    for i in range(n):
        p = f'c{i:03d}_'
        cfg = json.loads(str(z[p + 'cfg']))
        case = {k[len(p):]: z[k] for k in z.files if k.startswith(p) and k != p + 'cfg'}
        got = replay(neuron, surrogate, cfg, case)
        for f, g in got.items():
            want = case[f]
            same = np.array_equal(np.asarray(g).view(np.uint8 if np.asarray(g).dtype == np.uint8 else np.int32),
                                  np.asarray(want, dtype=np.asarray(g).dtype).view(np.uint8 if np.asarray(g).dtype == np.uint8 else np.int32))
            if not same and len(bad) < max_report:
                bad.append((i, f, float(np.abs(np.asarray(g, np.float64) - np.asarray(want, np.float64)).max())))
    return n, bad


def main():
    prov = real_provider()
    if prov is None:
        print('PARITY UNPINNED: `spikingjelly.clock_driven` is not importable here (un-vendored, un-pinned dependency of the reference, '
              'requirements.txt:3); the neuron KATs remain outputs of oracle/sj_clock_driven.py')
        return 3
    neuron, surrogate, ver = prov
    n, bad = check(neuron, surrogate)
    meta = json.loads(str(np.load(KAT)['meta']))
    print(f'spikingjelly {ver}: default Sigmoid alpha = {surrogate.Sigmoid().alpha} (fixtures assume {meta["sigmoid_alpha"]}), '
          f'default ATan alpha = {surrogate.ATan().alpha} (fixtures assume {meta["atan_alpha"]})')
    if bad:
        print(f'MISMATCH against the real package in {len(bad)}+ fields of {n} cases:')
        for i, f, d in bad:
            print(f'  case {i:03d} field {f}: max |diff| = {d:.3e}')
        return 1
    print(f'PINNED: all {n} neuron KAT cases are bit-identical to spikingjelly {ver}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
