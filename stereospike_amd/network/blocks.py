"""Building blocks with the reference's names, signatures and state_dict layout
(/root/reference/network/blocks.py): MultiplyBy (:90-107), NNConvUpsampling (:110-132), SEWResBlock (:135-181),
plus BilinConvUpsampling (:15-37) and ResBlock (:40-83) for the ANN twin.

What is different is *how* they run on the MI355X: wherever the reference chains
conv -> MultiplyBy -> neuron (-> add), the conv stays a PyTorch-ROCm op and everything after it is one fused
HIP launch (stereospike_amd/csrc/ss_neuron.hip), for a single step or for a whole [T, B, C, H, W] sequence.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import config as _config
from ..clock_driven import neuron, surrogate
from ..config import current as _cfg
from ..fused import nearest_tables, register_box_tables, register_sub_tables, upconv_projected, upconv_projected_cl


# Execution layout (fields of config.EngineConfig; reads of `blocks.<NAME>` answer with the configuration in effect, assignments are refused):
#   FUSE_UPCONV            the sequence fast path evaluates NNConvUpsampling through forward_projected (no up-sampled tensor); False = the reference's
#                          two-op form on MIOpen (bench.py --fuse-upconv 0 for A/B measurements)
#   DECODER_CHANNELS_LAST  decoder (deconv4..1, their skip adds, the predict_depth heads) in NHWC memory: the projection is then ONE row-major GEMM per
#                          stage and the gather kernels read / write 16-B channel vectors; the neuron kernels are layout-agnostic
#   ENCODER_CHANNELS_LAST  encoder / bottleneck activations in NHWC memory as well: no layout copies anywhere
#   FORK_OUTPUTS           outputs with two consumers are handed out as two handles; the neuron backward adds the two gradients on load


class MultiplyBy(nn.Module):
    """y = x * scale_value (a python float, or a 1-element Parameter when learnable).  In a SpikingStage /
    SEWResBlock a non-learnable gain is folded into the neuron kernel's prologue instead of costing a pass."""

    def __init__(self, scale_value: float = 5., learnable: bool = False) -> None:
        super().__init__()
        self.scale_value = nn.Parameter(torch.tensor([float(scale_value)])) if learnable else scale_value

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return torch.mul(input, self.scale_value)

    def extra_repr(self):
        return f'scale_value={float(self.scale_value)}'


def _fold_gain(mul: MultiplyBy, y: torch.Tensor):
    """(tensor, scale) such that the neuron kernel computes (tensor * scale): fold a constant gain, apply a
    learnable one with torch so autograd reaches its Parameter."""
    if isinstance(mul.scale_value, torch.Tensor):
        return mul(y), 1.0
    return y, float(mul.scale_value)


class _UpConv(nn.Module):
    """resize to (up_size + k - 1) then a valid k x k conv => output is exactly up_size; `.up[1]` is the Conv2d."""
    _mode = None

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, up_size: tuple, bias: bool = False):
        super().__init__()
        size = (up_size[0] + (kernel_size - 1), up_size[1] + (kernel_size - 1))
        resize = nn.UpsamplingNearest2d(size=size) if self._mode == 'nearest' else nn.UpsamplingBilinear2d(size=size)
        self.up = nn.Sequential(
            resize,
            nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=1, padding=0, bias=bias),
        )

    def forward(self, x):
        return self.up(x)


class NNConvUpsampling(_UpConv):
    """Nearest-neighbour resize + conv (checkerboard-free 'deconvolution'; integer spike counts stay integer).

    `forward` is the reference's two-op form (kept for API compatibility).  `forward_projected` computes the same map
    without materialising the (H+k-1) x (W+k-1) x C up-sampled tensor: per-tap 1x1 projections at LOW resolution + one
    fused gather launch (include/ss_neuron.h ss_upconv1_*): ~4.3x fewer MACs for the decoder stages, ~60x fewer MACs and
    HBM bytes for predict_depth4.  The models' fast path uses it for deconv4..1 and the four predict_depth heads."""
    _mode = 'nearest'

    def _tables(self, h, w, device):
        cache = self.__dict__.setdefault('_tbl_cache', {})
        key = (h, w, str(device))
        if key not in cache:
            Hu, Wu = self.up[0].size
            ty, tx = nearest_tables(h, Hu), nearest_tables(w, Wu)
            cache[key] = tuple(t.to(device) for t in (ty + tx))
            k = self.up[1].kernel_size[0]
            if k == 5:                                         # decoder stages (host-side, from the CPU copies: backward never reads a table back): the index tables of the box-sum backward (ss_upconv_box.hip)
                register_box_tables(cache[key], ty + tx, Hu - k + 1, Wu - k + 1)
                register_sub_tables(cache[key], ty + tx, Hu - k + 1, Wu - k + 1)      # ... and of the sub-pixel forward (ss_upconv_sub.hip)
        return cache[key]

    def forward_projected(self, x: torch.Tensor) -> torch.Tensor:
        conv = self.up[1]
        k = conv.kernel_size[0]
        Hu, Wu = self.up[0].size
        return upconv_projected(x, conv.weight, conv.bias, self._tables(x.shape[-2], x.shape[-1], x.device), k,
                                Hu - k + 1, Wu - k + 1)


    def forward_projected_cl(self, x_cl: torch.Tensor, spikes_in: bool = False, lowrank_grad: bool = False,
                             x_packed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_cl [NB, h, w, C_in] (contiguous NHWC array) -> [NB, H, W, C_out] (NHWC array).  spikes_in: x_cl is a spike tensor.
        lowrank_grad: x_cl is a forked handle of a fused neuron layer's output with no other consumer — a one-channel 3 x 3 head may then
        hand its input gradient over as the pair (g_P, W2) for ss_neuron_bwd_fork_lr_f32 (fused.lowrank_anchor).
        x_packed: the input as a 2-bit packed spike tensor (x_cl may then be a data-less anchor; one-channel 3 x 3 heads in fp32 mode only)."""
        conv = self.up[1]
        k = conv.kernel_size[0]
        Hu, Wu = self.up[0].size
        return upconv_projected_cl(x_cl, conv.weight, conv.bias, self._tables(x_cl.shape[1], x_cl.shape[2], x_cl.device),
                                   k, Hu - k + 1, Wu - k + 1, spikes_in, lowrank_grad, x_packed)


class BilinConvUpsampling(_UpConv):
    _mode = 'bilinear'


class SpikingStage(nn.Sequential):
    """Sequential(conv | NNConvUpsampling, MultiplyBy, Node) — the layout of every spiking stage of the reference
    (SNN_models.py:75-129), so parameters keep their names ('bottom.0.weight', 'deconv4.0.up.1.weight',
    'bottom.2.w').  Calling it runs the conv in PyTorch-ROCm and gain + charge + fire + reset (+ skip add,
    + firing-rate counters) as one fused launch."""

    def forward(self, x: torch.Tensor, skip: Optional[torch.Tensor] = None, nnz=None) -> torch.Tensor:
        y, scale = _fold_gain(self[1], self[0](x))
        return self[2].forward_fused(y, scale, skip, nnz)

    def forward_sequence(self, x_seq: torch.Tensor, skip_seq: Optional[torch.Tensor] = None, nnz=None):
        """x_seq [T, B, C, H, W] -> [T, B, C', H', W']: the conv sees one [T*B] batch (time steps are independent
        for a feed-forward synapse), the neuron kernel then walks t = 0..T-1 with v in registers."""
        T, B = x_seq.shape[:2]
        syn = self[0]
        y = syn.forward_projected(x_seq.flatten(0, 1)) if (_cfg().FUSE_UPCONV and isinstance(syn, NNConvUpsampling)) \
            else syn(x_seq.flatten(0, 1))
        y, scale = _fold_gain(self[1], y)
        return self[2].forward_sequence(y.view(T, B, *y.shape[1:]), scale, skip_seq, nnz)


    def forward_sequence_conv_cl(self, x_seq: torch.Tensor, nnz=None, spikes_in: bool = False, fork: bool = False,
                                 x_packed: Optional[torch.Tensor] = None, pack: int = 0):
        """Encoder stage on NHWC arrays: x_seq [T, B, h, w, C] -> [T, B, h', w', C'].  The Conv2d sees a logical-NCHW
        view with channels_last strides (no copy) and returns channels_last memory, i.e. again an NHWC array.
        spikes_in: x_seq is the output of a spiking layer.  x_packed: the same input as a 2-bit packed spike tensor (read by the
        exact-split conv's im2col; x_seq may then be a data-less anchor).  pack: see BaseNode.forward_sequence — the packed output is
        left in `self[2].last_packed`."""
        T, B = x_seq.shape[:2]
        y = _conv_cl(self[0], x_seq.flatten(0, 1), spikes_in, x_packed)
        y, scale = _fold_gain(self[1], y)
        return self[2].forward_sequence(y.view(T, B, *y.shape[1:]), scale, None, nnz, channels_last=True, fork=fork, pack=pack)

    def forward_sequence_cl(self, x_seq: torch.Tensor, skip_seq: Optional[torch.Tensor] = None, nnz=None, spikes_in: bool = False,
                            fork: bool = False, skip_packed: Optional[torch.Tensor] = None, pack: int = 0, x_packed: Optional[torch.Tensor] = None):
        """Channels-last decoder stage: x_seq [T, B, h, w, C] -> [T, B, H, W, C'] (NHWC arrays); the synapse must be an
        NNConvUpsampling.  spikes_in: x_seq is the output of a spiking layer (+ spike skip adds).  skip_packed: the skip operand as a
        packed spike tensor (skip_seq then carries the autograd edge only).  pack: see BaseNode.forward_sequence (`self[2].last_packed`)."""
        T, B = x_seq.shape[:2]
        y = self[0].forward_projected_cl(x_seq.flatten(0, 1), spikes_in, x_packed=x_packed)     # x_packed: x_seq as packed spikes (x_seq may be an anchor)
        y, scale = _fold_gain(self[1], y)
        return self[2].forward_sequence(y.view(T, B, *y.shape[1:]), scale, skip_seq, nnz, channels_last=True, fork=fork,
                                        skip_packed=skip_packed, pack=pack)


class ResBlock(nn.Module):
    """Residual block of the ANN twin (conv -> activation -> BatchNorm, twice, then the connect function)."""

    def __init__(self, in_channels: int, connect_function='ADD', kernel_size: int = 3, bias: bool = False,
                 activation_function: nn.Module = nn.Tanh()):
        super().__init__()
        pad = (kernel_size - 1) // 2
        for name in ('conv1', 'conv2'):
            setattr(self, name, nn.Sequential(
                nn.Conv2d(in_channels, in_channels, kernel_size=kernel_size, stride=1, padding=pad, bias=bias),
                activation_function,
                nn.BatchNorm2d(in_channels)))
        self.connect_function = connect_function

    def forward(self, x):
        return _connect(self.connect_function, self.conv2(self.conv1(x)), x, spiking=False)


def _conv_cl(conv: nn.Module, x_arr: torch.Tensor, spikes_in: bool = False, x_packed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x_arr [NB, h, w, C] (NHWC array) -> conv -> [NB, h', w', C'] NHWC array, without layout copies when MIOpen returns
    channels_last memory (it does for channels_last inputs).  The filter is handed to MIOpen as a channels_last copy
    (72 MB for the whole network, ~0.03 ms per step); the Parameter itself keeps its standard layout, so optimisers,
    the DP gradient buckets and state_dict never see a layout change.
    spikes_in: x_arr is the output of a spiking layer (small integers) — wide layers then run as exact bf16x3 GEMMs
    (fused.spike_conv_cl) instead of MIOpen's fp32 convolution."""
    if x_arr.is_cuda and isinstance(conv, nn.Conv2d):
        from ..fused import conv_cl16, x16_mode
        adt = x16_mode(x_arr.device)
        if adt is not None:
            # 16-bit activation modes (round 5): the synapse on the engine's own single-term kernels with 16-bit I/O — spikes are exact in fp16 /
            # bf16 (and travel packed), the fp32 master weight is rounded once to the format inside the kernel's weight preparation (autocast's semantics),
            # accumulation and the weight gradient are fp32; nothing is cast by torch
            y = conv_cl16(conv, x_arr, spikes_in, x_packed, adt)
            if y is not None:
                return y
            if x_packed is not None and x_arr.stride(-1) == 0:
                raise RuntimeError('packed-only spike tensor handed to a convolution that reads dense activations')
    if spikes_in and isinstance(conv, nn.Conv2d):
        from ..fused import spike_conv_cl
        y = spike_conv_cl(x_arr, conv, x_packed)
        if y is not None:
            return y
    if spikes_in and isinstance(conv, nn.Conv2d):
        from ..fused import spike_conv_wgrad_cl
        y = spike_conv_wgrad_cl(x_arr, conv, x_packed)    # conv1 / conv2: exact MFMA forward (implicit GEMM on the packed spikes) and weight
        if y is not None:                                 # gradient, MIOpen data gradient
            return y
    if x_packed is not None and x_arr.stride(-1) == 0:
        raise RuntimeError('packed-only spike tensor handed to a convolution that reads dense activations')
    if isinstance(conv, nn.Conv2d):
        from ..fused import dense_conv_s1_cl
        y = dense_conv_s1_cl(x_arr, conv)                   # the first encoder layer: six-term MFMA implicit GEMM (any input values)
        if y is not None:
            return y
        w = conv.weight.contiguous(memory_format=torch.channels_last)
        y = torch.nn.functional.conv2d(x_arr.permute(0, 3, 1, 2), w, conv.bias, conv.stride, conv.padding,
                                       conv.dilation, conv.groups)
        _config.note('synapse_fwd', 'miopen')
        _config.note('synapse_bwd', 'g_x: miopen; g_w: miopen')
    else:
        y = conv(x_arr.permute(0, 3, 1, 2))
        _config.note('synapse_fwd', 'torch module (MIOpen)')
    y = y.permute(0, 2, 3, 1)
    return y if y.is_contiguous() else y.contiguous()


def _connect(fn: str, out: torch.Tensor, identity: torch.Tensor, spiking: bool):
    if fn == 'ADD':
        out += identity
        return out
    if fn in ('MUL', 'AND'):
        out *= identity
        return out
    if fn == 'NMUL':
        return identity * (1. - out)
    if fn == 'OR' and spiking:
        return surrogate.ATan(spiking=True)(out + identity)
    raise NotImplementedError(fn)


class SEWResBlock(nn.Module):
    """Spike-Element-Wise residual block (arXiv:2102.04159): conv-gain-neuron twice, then g(out, identity).
    With 'ADD' (the only connect function the reference uses, SNN_models.py:105-106) the residual add is the
    epilogue of the second neuron kernel; the other connect functions run unfused."""

    def __init__(self, in_channels: int, connect_function='ADD', v_threshold=1., v_reset=0.,
                 surrogate_function=None, use_plif=False, tau=2., multiply_factor=1.):
        super().__init__()
        if surrogate_function is None:
            surrogate_function = surrogate.Sigmoid()

        def node():
            if use_plif:
                return neuron.ParametricLIFNode(init_tau=tau, v_threshold=v_threshold, v_reset=v_reset,
                                                surrogate_function=surrogate_function, detach_reset=True)
            return neuron.IFNode(v_threshold=v_threshold, v_reset=v_reset, surrogate_function=surrogate_function,
                                 detach_reset=True)

        def conv():
            return nn.Sequential(nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1, bias=False),
                                 MultiplyBy(multiply_factor))

        self.conv1, self.sn1 = conv(), node()
        self.conv2, self.sn2 = conv(), node()
        self.connect_function = connect_function

    def forward_sequence(self, x_seq: torch.Tensor, nnz=None) -> torch.Tensor:
        T, B = x_seq.shape[:2]

        def half(conv, sn, inp, skip, cnt):
            with _config.sublayer('.conv1' if conv is self.conv1 else '.conv2'):
                y, scale = _fold_gain(conv[1], conv[0](inp.flatten(0, 1)))
                _config.note('synapse_fwd', 'torch module (MIOpen)')
                return sn.forward_sequence(y.view(T, B, *y.shape[1:]), scale, skip, cnt)

        out = half(self.conv1, self.sn1, x_seq, None, None)
        if self.connect_function == 'ADD':
            return half(self.conv2, self.sn2, out, x_seq, nnz)
        out = half(self.conv2, self.sn2, out, None, nnz)
        return _connect(self.connect_function, out, x_seq, spiking=True)

    def forward_sequence_cl(self, x_seq: torch.Tensor, nnz=None, spikes_in: bool = False, x_packed: Optional[torch.Tensor] = None,
                            pack_out: int = 0) -> torch.Tensor:
        """NHWC-array form of forward_sequence ('ADD' connect function only).  spikes_in: x_seq is a spike tensor (the inner
        activation always is).  x_packed: x_seq as a 2-bit packed spike tensor; pack_out: pack mode of the block output (left in
        `self.sn2.last_packed`).  The inner activation travels packed-only whenever conv2 is an exact-split conv."""
        if self.connect_function != 'ADD':
            raise NotImplementedError('channels-last SEW block supports the ADD connect function only')
        from .. import fused
        T, B = x_seq.shape[:2]

        def half(conv, sn, inp, skip, cnt, spk, inp_packed, skip_packed, pack):
            with _config.sublayer('.conv1' if conv is self.conv1 else '.conv2'):           # the block's two halves are two layers of the dispatch plan
                y, scale = _fold_gain(conv[1], _conv_cl(conv[0], inp.flatten(0, 1), spk, inp_packed))
                return sn.forward_sequence(y.view(T, B, *y.shape[1:]), scale, skip, cnt, channels_last=True, pack=pack, skip_packed=skip_packed)

        exact1 = fused.spike_conv_applies(self.conv1[0], x_seq.device, x_seq.dtype)
        exact2 = _cfg().PACK_SPIKES and fused.spike_conv_applies(self.conv2[0], x_seq.device, x_seq.dtype)
        out = half(self.conv1, self.sn1, x_seq, None, None, spikes_in, x_packed if (exact1 and spikes_in) else None, None, 2 if exact2 else 0)
        return half(self.conv2, self.sn2, out, x_seq, nnz, True, self.sn1.last_packed, x_packed, pack_out)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_sequence(x.unsqueeze(0))[0]


_config.guard_module(__name__, 'blocks')
