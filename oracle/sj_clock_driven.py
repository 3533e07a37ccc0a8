"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path (`stereospike_amd/`).

Torch-eager, op-by-op restatement of the part of the third-party package
``spikingjelly.clock_driven`` that the reference reaches:

    /root/reference/network/blocks.py:8        from spikingjelly.clock_driven import functional, surrogate, neuron, layer, rnn
    /root/reference/network/SNN_models.py:6    from spikingjelly.clock_driven import neuron, layer, surrogate
    /root/reference/network/ANN_models.py:4    from spikingjelly.clock_driven import functional, neuron, layer, surrogate
    /root/reference/train.py:12-13             functional.reset_net, surrogate.ATan

PARITY UNPINNED for the neuron / surrogate arithmetic: ``spikingjelly`` is an un-vendored,
un-pinned dependency (/root/reference/requirements.txt:3 is the bare word ``spikingjelly``), it is
not installed in this image and there is no network.  The ``clock_driven`` namespace together with the
``detach_reset=`` kwarg and ``ParametricLIFNode(init_tau=)`` bound it to the PyPI line
0.0.0.0.4 … 0.0.0.0.12.  What follows restates that line's *published* single-step algorithm
 and is anchored on the reference's own call sites:

    IFNode(v_threshold, v_reset, surrogate_function, detach_reset)   SNN_models.py:78,85,90,95,100,113,118,123,128,150
    LIFNode(tau, v_threshold, v_reset, surrogate_function, detach_reset)          SNN_models.py:266 ... 316
    ParametricLIFNode(init_tau, v_threshold, v_reset[, surrogate_function], detach_reset)   SNN_models.py:266, blocks.py:150,157
    m.v / m.v.detach_() / hasattr(m,'reset') / isinstance(m, neuron.BaseNode|IFNode)      SNN_models.py:22-48
    surrogate.ATan(), surrogate.Sigmoid(), surrogate.ATan(spiking=True)                    train.py:118, blocks.py:142,175
    functional.reset_net(net)                                                              train.py:221,308

Two upstream details that differ between releases are explicit parameters here and are recorded in
every fixture: ``Sigmoid.alpha`` (4.0 in the later clock_driven releases, 1.0 in the earliest) and the
charge statement (``self.v = self.v + x`` — rebinding, later releases; restated here — versus the
earliest ``self.v += x``, which aliases the I-neuron read-outs of SNN_models.py:173-188).

Every binary op below is a separate eager torch op (one fp32 rounding each), in upstream's order; that
is the arithmetic the HIP kernel has to reproduce bit-for-bit in the forward direction.
"""
import math
import sys
import types

import torch
import torch.nn as nn


# --------------------------------------------------------------------------------------------------
# surrogate
# --------------------------------------------------------------------------------------------------
def heaviside(x: torch.Tensor) -> torch.Tensor:
    # upstream: (x >= 0).to(x)   — 1.0 at exactly 0
    return (x >= 0).to(x)


class _atan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        if x.requires_grad:
            ctx.save_for_backward(x)
            ctx.alpha = alpha
        return heaviside(x)

    @staticmethod
    def backward(ctx, grad_output):
        grad_x = None
        if ctx.needs_input_grad[0]:
            x = ctx.saved_tensors[0]
            # upstream: alpha / 2 / (1 + (pi / 2 * alpha * x).pow_(2)) * grad_output
            grad_x = ctx.alpha / 2 / (1 + (math.pi / 2 * ctx.alpha * x).pow_(2)) * grad_output
        return grad_x, None


class _sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        if x.requires_grad:
            ctx.save_for_backward(x)
            ctx.alpha = alpha
        return heaviside(x)

    @staticmethod
    def backward(ctx, grad_output):
        grad_x = None
        if ctx.needs_input_grad[0]:
            sgax = (ctx.saved_tensors[0] * ctx.alpha).sigmoid_()
            grad_x = grad_output * (1. - sgax) * sgax * ctx.alpha
        return grad_x, None


class _SurrogateBase(nn.Module):
    def __init__(self, alpha, spiking=True):
        super().__init__()
        self.alpha = alpha
        self.spiking = spiking

    def extra_repr(self):
        return f'alpha={self.alpha}, spiking={self.spiking}'


class ATan(_SurrogateBase):
    def __init__(self, alpha=2.0, spiking=True):
        super().__init__(alpha, spiking)

    def forward(self, x):
        if self.spiking:
            return _atan.apply(x, self.alpha)
        return (math.pi / 2 * self.alpha * x).atan_() / math.pi + 0.5


class Sigmoid(_SurrogateBase):
    # alpha: see the module docstring (4.0 assumed; recorded in fixtures)
    def __init__(self, alpha=4.0, spiking=True):
        super().__init__(alpha, spiking)

    def forward(self, x):
        if self.spiking:
            return _sigmoid.apply(x, self.alpha)
        return (x * self.alpha).sigmoid()


# --------------------------------------------------------------------------------------------------
# neuron
# --------------------------------------------------------------------------------------------------
class BaseNode(nn.Module):
    def __init__(self, v_threshold: float = 1., v_reset: float = 0.,
                 surrogate_function=None, detach_reset: bool = False):
        super().__init__()
        assert isinstance(v_reset, float) or v_reset is None
        assert isinstance(v_threshold, float)
        assert isinstance(detach_reset, bool)
        self._v_init = 0. if v_reset is None else v_reset
        self.v = self._v_init
        self.v_threshold = v_threshold
        self.v_reset = v_reset
        self.detach_reset = detach_reset
        self.surrogate_function = Sigmoid() if surrogate_function is None else surrogate_function

    def neuronal_charge(self, x):
        raise NotImplementedError

    def neuronal_fire(self):
        self.spike = self.surrogate_function(self.v - self.v_threshold)

    def neuronal_reset(self):
        spike = self.spike.detach() if self.detach_reset else self.spike
        if self.v_reset is None:
            self.v = self.v - spike * self.v_threshold
        else:
            self.v = (1. - spike) * self.v + spike * self.v_reset

    def forward(self, x):
        self.neuronal_charge(x)
        self.neuronal_fire()
        self.neuronal_reset()
        return self.spike

    def reset(self):
        self.v = self._v_init
        self.spike = 0.

    def extra_repr(self):
        return f'v_threshold={self.v_threshold}, v_reset={self.v_reset}, detach_reset={self.detach_reset}'


class IFNode(BaseNode):
    def neuronal_charge(self, x):
        self.v = self.v + x


class LIFNode(BaseNode):
    def __init__(self, tau: float = 2., v_threshold: float = 1., v_reset: float = 0.,
                 surrogate_function=None, detach_reset: bool = False):
        assert isinstance(tau, float) and tau > 1.
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset)
        self.tau = tau

    def neuronal_charge(self, x):
        if self.v_reset is None or self.v_reset == 0.:
            self.v = self.v + (x - self.v) / self.tau
        else:
            self.v = self.v + (x - (self.v - self.v_reset)) / self.tau


class ParametricLIFNode(BaseNode):
    def __init__(self, init_tau: float = 2.0, v_threshold: float = 1., v_reset: float = 0.,
                 surrogate_function=None, detach_reset: bool = False):
        assert isinstance(init_tau, float) and init_tau > 1.
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset)
        init_w = - math.log(init_tau - 1.)
        self.w = nn.Parameter(torch.as_tensor(init_w, dtype=torch.float32))

    def neuronal_charge(self, x):
        if self.v_reset is None or self.v_reset == 0.:
            self.v = self.v + (x - self.v) * self.w.sigmoid()
        else:
            self.v = self.v + (x - (self.v - self.v_reset)) * self.w.sigmoid()


# --------------------------------------------------------------------------------------------------
# functional / layer / rnn
# --------------------------------------------------------------------------------------------------
def reset_net(net: nn.Module):
    for m in net.modules():
        if hasattr(m, 'reset'):
            m.reset()


class Dropout(nn.Module):
    """Only type-checked by the reference (SNN_models.py:26); never instantiated by it."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p
        self.mask = None

    def forward(self, x):
        if not self.training:
            return x
        if self.mask is None:
            self.mask = (torch.rand_like(x) > self.p).to(x) / (1. - self.p)
        return x * self.mask

    def reset(self):
        self.mask = None


def as_modules(prefix: str = 'spikingjelly'):
    """Build module objects ``<prefix>``, ``<prefix>.clock_driven`` and its five submodules carrying the
    names above.  Used by tests/golden/make_golden.py to let the reference's own ``network/*.py`` import
    (the real package is absent) and by the CPU-baseline restatement in oracle/ref_network.py."""
    root = types.ModuleType(prefix)
    cd = types.ModuleType(prefix + '.clock_driven')
    mods = {}
    for name, members in (
            ('neuron', dict(BaseNode=BaseNode, IFNode=IFNode, LIFNode=LIFNode, ParametricLIFNode=ParametricLIFNode)),
            ('surrogate', dict(heaviside=heaviside, ATan=ATan, Sigmoid=Sigmoid)),
            ('functional', dict(reset_net=reset_net)),
            ('layer', dict(Dropout=Dropout)),
            ('rnn', dict())):
        m = types.ModuleType(f'{prefix}.clock_driven.{name}')
        for k, v in members.items():
            setattr(m, k, v)
        setattr(cd, name, m)
        mods[m.__name__] = m
    root.clock_driven = cd
    mods[root.__name__] = root
    mods[cd.__name__] = cd
    return mods


_self = sys.modules[__name__]
neuron = types.SimpleNamespace(BaseNode=BaseNode, IFNode=IFNode, LIFNode=LIFNode, ParametricLIFNode=ParametricLIFNode)
surrogate = types.SimpleNamespace(heaviside=heaviside, ATan=ATan, Sigmoid=Sigmoid)
functional = types.SimpleNamespace(reset_net=reset_net)
layer = types.SimpleNamespace(Dropout=Dropout)
