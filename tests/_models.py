"""Model pairs for the GPU parity tests: the oracle network (oracle/ref_network.py, pinned to the reference's own network/*.py by
tests/golden/make_golden.py) and the product network with identical weights."""
import hashlib

import torch

from _util import ref_network as rn, sj

DEV = 'cuda:0'
PLIF = dict(tau=3., v_threshold=1.0, v_reset=0.0, multiply_factor=30.)   # gain 10 leaves the tau = 3 nets silent


def product(name, **kw):
    from stereospike_amd.clock_driven import surrogate
    from stereospike_amd.network import SNN_models as S, ANN_models as A
    if name == 'StereoSpike':
        return S.StereoSpike(surrogate_function=surrogate.ATan(), detach_reset=True, v_threshold=1.0, v_reset=0.,
                             multiply_factor=10., **kw)
    if name == 'PLIFNet':
        return S.fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike(use_plif=True, **PLIF, **kw)
    if name == 'LIFNet':
        return S.fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike(use_plif=False, **PLIF, **kw)
    if name == 'PLIFNetMono':
        return S.fromZero_feedforward_multiscale_tempo_monocular_SpikeFlowNetLike(use_plif=True, **PLIF, **kw)
    if name == 'ANN':
        return A.StereoSpike_equivalentANN(**kw)
    raise ValueError(name)


def oracle(name, **kw):
    if name == 'StereoSpike':
        return rn.build('StereoSpike', multiply_factor=10., surrogate_function=sj.ATan(), **kw)
    if name in ('PLIFNet', 'LIFNet'):
        return rn.build('PLIFNet', tau=3., use_plif=(name == 'PLIFNet'), multiply_factor=30., **kw)
    if name == 'PLIFNetMono':
        return rn.build('PLIFNetMono', tau=3., use_plif=True, multiply_factor=30., **kw)
    return rn.build('ANN', **kw)


def pair(name, H, W, seed=2021, device=DEV):
    """(oracle on CPU, product on `device`) with the default init of torch.manual_seed(seed) (train.py:53) in both."""
    torch.manual_seed(seed)
    orc = oracle(name, input_size=(H, W))
    net = product(name, input_size=(H, W))
    assert list(net.state_dict().keys()) == list(orc.state_dict().keys())
    net.load_state_dict(orc.state_dict())
    return orc, net.to(device)


def state_sha(net):
    """sha256 over (key, tensor bytes) of the state_dict — the `state_sha` field of tests/golden/model_*.npz."""
    h = hashlib.sha256()
    for k, v in net.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()
