"""`neuron.BaseNode / IFNode / LIFNode / ParametricLIFNode` with the constructor signatures the reference uses
(SNN_models.py:78...150, 266...338, 451...523; blocks.py:150,157; ANN_models.py:111) and the protocol its helpers
rely on (SNN_models.py:22-48: `isinstance(m, neuron.BaseNode)`, `m.v`, `m.v.detach_()`, `hasattr(m, 'reset')`,
`isinstance(m, neuron.IFNode)`).

Every step — single (`node(x)`, the drop-in call) or a whole `[T, ...]` sequence (`node.forward_sequence`) —
is one launch of the fused HIP kernel behind include/ss_neuron.h; the membrane `v` is carried between calls
exactly like upstream's stateful single-step nodes until `reset()`.
"""
import math
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from ..fused import NeuronCfg, fused_neuron, membrane_after, _cfg as _engine_cfg
from . import surrogate as _sg


class _LazyMembrane:
    """The membrane after a fused T-step training pass whose kernel did not write it (EngineConfig.LAZY_MEMBRANE): everything needed to recompute it —
    the layer input (kept for the recompute backward anyway), the membrane the pass started from and the node's parameters at that time."""
    __slots__ = ('x_seq', 'v_init', 'cfg', 'k', 'channels_last', 'shape', 'dtype', 'device')

    def __init__(self, x_seq, v_init, cfg, k, channels_last):
        self.x_seq, self.v_init, self.cfg, self.channels_last = x_seq.detach(), (None if v_init is None else v_init.detach()), cfg, channels_last
        self.k = None if k is None else k.detach().clone()
        self.shape, self.dtype, self.device = x_seq.shape[1:], torch.float32, x_seq.device

    def materialize(self) -> torch.Tensor:
        v = membrane_after(self.x_seq, self.cfg, self.v_init, self.k)
        return v.permute(0, 3, 1, 2) if self.channels_last else v


class BaseNode(nn.Module):
    _kind = None

    def __init__(self, v_threshold: float = 1., v_reset: float = 0., surrogate_function=None,
                 detach_reset: bool = False):
        super().__init__()
        assert isinstance(v_threshold, float)
        assert isinstance(detach_reset, bool)
        if v_reset is None:
            raise NotImplementedError('soft reset (v_reset=None) is not used by the reference and not implemented')
        assert isinstance(v_reset, float)
        self.v_threshold = v_threshold
        self.v_reset = v_reset
        self.detach_reset = detach_reset
        self.surrogate_function = _sg.Sigmoid() if surrogate_function is None else surrogate_function
        self.v = v_reset          # python float until the first charge, then a tensor (as upstream)
        self.last_numel = 0       # neurons of the last pass (firing-rate denominators need no membrane)

    # ---- state protocol -------------------------------------------------------------------------
    @property
    def v(self):
        """The membrane: python float until the first charge, then a tensor (as upstream).  After a training pass that left it unwritten it is recomputed
        here, once, on first access (_LazyMembrane)."""
        v = self._v
        if isinstance(v, _LazyMembrane):
            v = self._v = v.materialize()
        return v

    @v.setter
    def v(self, value):
        object.__setattr__(self, '_v', value)

    def reset(self):
        self.v = self.v_reset

    def detach(self):
        if isinstance(self._v, torch.Tensor):   # (an unwritten membrane has no autograd history to cut)
            self.v = self._v.detach()     # (not detach_(): the I-pool membrane is a view of the read-out buffer)

    def extra_repr(self):
        return f'v_threshold={self.v_threshold}, v_reset={self.v_reset}, detach_reset={self.detach_reset}'

    # ---- fused evaluation ---------------------------------------------------------------------------
    def _tau(self) -> float:
        return 2.0

    def _k(self) -> Optional[torch.Tensor]:
        return None

    def _cfg(self, scale: float) -> NeuronCfg:
        sg = self.surrogate_function
        sg_id = getattr(sg, 'sg_id', None)
        if sg_id is None or not getattr(sg, 'spiking', True):
            raise TypeError(f'{type(self).__name__}: surrogate_function must be a spiking surrogate.ATan or '
                            f'surrogate.Sigmoid (got {sg!r}); the fused HIP backward implements these two')
        return NeuronCfg(kind=self._kind, scale=float(scale), tau=float(self._tau()), v_th=float(self.v_threshold),
                         v_reset=float(self.v_reset), surrogate=sg_id, alpha=float(sg.alpha),
                         detach_reset=bool(self.detach_reset))

    def _v_init(self, like: torch.Tensor, channels_last: bool = False) -> Optional[torch.Tensor]:
        v = self.v
        if isinstance(v, torch.Tensor) and channels_last:
            v = v.permute(0, 2, 3, 1)          # logical [B, C, H, W] -> the NHWC array the kernel walks
        if isinstance(v, torch.Tensor):
            if v.shape != like.shape:
                raise _lib.SSNeuronError(f'membrane shape {tuple(v.shape)} != input shape {tuple(like.shape)}; '
                                         f'call functional.reset_net(net) when the batch shape changes')
            return v
        if float(v) == self.v_reset:
            return None                      # the kernel starts from the v_reset constant
        return torch.full_like(like, float(v))

    def forward_sequence(self, x_seq: torch.Tensor, scale: float = 1., skip_seq: Optional[torch.Tensor] = None,
                         nnz: Optional[torch.Tensor] = None, channels_last: bool = False, fork: bool = False,
                         pack: int = 0, skip_packed: Optional[torch.Tensor] = None):
        """x_seq [T, ...]: T steps from the current membrane; returns out_seq [T, ...] and carries v forward.
        The kernel is element-wise, so any memory layout works as long as x, skip and v share it; with
        channels_last=True x_seq is an NHWC array [T, B, H, W, C] and `self.v` keeps its logical [B, C, H, W] shape
        (a permuted view of the NHWC membrane).  fork=True returns (out_seq, out_seq') — two handles on the same spikes for an output
        with two consumers (next synapse + skip / head); their gradients are summed inside the backward kernel.
        pack (1: dense + packed, 2: packed only — the returned tensor is then a data-less anchor of the logical shape) also writes the
        output as a 2-bit packed spike tensor, left in `self.last_packed` (None when the packed kernel form does not apply: the output
        is then dense); skip_packed: the skip operand in that form."""
        cfg, v_init, k = self._cfg(scale), self._v_init(x_seq[0], channels_last), self._k()
        # a multi-step pass under autograd: the membrane after it is read by nobody before the next reset_net (train.py:221) — leave it unwritten where the
        # kernel form allows and recompute on demand (`v` property); single steps (the drop-in `net(x[:, t])` loop) and inference carry it as a tensor
        lazy = bool(_engine_cfg().LAZY_MEMBRANE) and x_seq.shape[0] > 1 and torch.is_grad_enabled() and x_seq.requires_grad
        res = fused_neuron(x_seq, cfg, v_init=v_init, skip_seq=skip_seq, k=k, nnz=nnz, fork=fork, pack=pack, skip_packed=skip_packed, want_v=not lazy)
        self.last_packed = res[-1] if pack else None
        self.last_numel = x_seq[0].numel()
        v_last = res[1]
        if v_last is None:
            self.v = _LazyMembrane(x_seq, v_init, cfg, k, channels_last)
        else:
            self.v = v_last.permute(0, 3, 1, 2) if channels_last else v_last
        return (res[0], res[2]) if fork else res[0]

    def forward_fused(self, x: torch.Tensor, scale: float = 1., skip: Optional[torch.Tensor] = None,
                      nnz: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One step with the preceding MultiplyBy gain and an optional following add folded in."""
        return self.forward_sequence(x.unsqueeze(0), scale, None if skip is None else skip.unsqueeze(0), nnz)[0]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_fused(x)


class IFNode(BaseNode):
    """h = v + x"""
    _kind = _lib.KIND_IF


class LIFNode(BaseNode):
    """h = v + (x - (v - v_reset)) / tau   (true division)"""
    _kind = _lib.KIND_LIF

    def __init__(self, tau: float = 2., v_threshold: float = 1., v_reset: float = 0., surrogate_function=None,
                 detach_reset: bool = False):
        assert isinstance(tau, float) and tau > 1.
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset)
        self.tau = tau

    def _tau(self):
        return self.tau

    def extra_repr(self):
        return super().extra_repr() + f', tau={self.tau}'


class ParametricLIFNode(BaseNode):
    """h = v + (x - (v - v_reset)) * sigmoid(w), w learnable 0-dim, initialised to -log(init_tau - 1)."""
    _kind = _lib.KIND_PLIF

    def __init__(self, init_tau: float = 2.0, v_threshold: float = 1., v_reset: float = 0., surrogate_function=None,
                 detach_reset: bool = False):
        assert isinstance(init_tau, float) and init_tau > 1.
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset)
        self.w = nn.Parameter(torch.as_tensor(-math.log(init_tau - 1.), dtype=torch.float32))

    def _k(self):
        # stays on the device and in the autograd graph: dL/dw = dL/dk * k * (1 - k) is torch's sigmoid backward
        return self.w.sigmoid()

    def extra_repr(self):
        with torch.no_grad():
            return super().extra_repr() + f', tau={1. / self.w.sigmoid().item():.4g}'
