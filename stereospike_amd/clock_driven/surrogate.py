"""`surrogate.ATan` / `surrogate.Sigmoid` (train.py:118, test.py:80, blocks.py:142,175, SNN_models.py:12...338).

Inside a neuron node these objects are *descriptors*: the node reads their type and `alpha` and the fused HIP
backward kernel evaluates the surrogate derivative (include/ss_neuron.h SS_SG_*).  Called directly on a tensor
(the reference does so only in the unused 'OR' connect function, blocks.py:175) they behave like upstream:
Heaviside forward, surrogate gradient backward — as a T = 1 IF step of the same fused kernel with v = 0,
threshold 0 (h = x, z = (x >= 0)).
"""
import math

import torch
import torch.nn as nn

from .. import _lib
from ..fused import NeuronCfg, fused_neuron


def heaviside(x: torch.Tensor) -> torch.Tensor:
    return (x >= 0).to(x)


class SurrogateFunctionBase(nn.Module):
    sg_id = None

    def __init__(self, alpha: float, spiking: bool = True):
        super().__init__()
        self.alpha = alpha
        self.spiking = spiking

    def extra_repr(self):
        return f'alpha={self.alpha}, spiking={self.spiking}'

    def primitive(self, x):
        raise NotImplementedError

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.spiking:
            return self.primitive(x)
        cfg = NeuronCfg(kind=_lib.KIND_IF, scale=1.0, v_th=0.0, v_reset=0.0, surrogate=self.sg_id,
                        alpha=float(self.alpha), detach_reset=True)
        out, _ = fused_neuron(x.contiguous().unsqueeze(0), cfg)
        return out[0]


class ATan(SurrogateFunctionBase):
    """g'(x) = alpha / 2 / (1 + (pi/2 * alpha * x)^2)"""
    sg_id = _lib.SG_ATAN

    def __init__(self, alpha: float = 2.0, spiking: bool = True):
        super().__init__(alpha, spiking)

    def primitive(self, x):
        return (math.pi / 2 * self.alpha * x).atan() / math.pi + 0.5


class Sigmoid(SurrogateFunctionBase):
    """g'(x) = alpha * s * (1 - s), s = sigmoid(alpha * x).  alpha = 4.0 is the later clock_driven default
    (the earliest releases used 1.0 — pass it explicitly to reproduce those; see oracle/README.md)."""
    sg_id = _lib.SG_SIGMOID

    def __init__(self, alpha: float = 4.0, spiking: bool = True):
        super().__init__(alpha, spiking)

    def primitive(self, x):
        return (x * self.alpha).sigmoid()
