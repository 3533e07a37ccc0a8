"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement (eager PyTorch, unfused, fp32) of the reference's
network graph, loss and metric, built on oracle/sj_clock_driven.py.  It is the checker for the GPU parity
tests and the "port" timed by bench.py's cpu_baseline leg; nothing under stereospike_amd/ imports it.

Follows (file:line relative to /root/reference):
    network/blocks.py:90-107    MultiplyBy            -> Gain
    network/blocks.py:110-132   NNConvUpsampling      -> UpConv (attribute .up = Sequential(UpsamplingNearest2d, Conv2d))
    network/blocks.py:135-181   SEWResBlock ('ADD')   -> SEWBlock (attributes conv1, sn1, conv2, sn2)
    network/blocks.py:40-83     ResBlock              -> AnnResBlock
    network/SNN_models.py:63-248    StereoSpike                          -> build('StereoSpike', ...)
    network/SNN_models.py:251-435   fromZero_..._Matt_SpikeFlowNetLike    -> build('PLIFNet', ...)
    network/SNN_models.py:438-622   ..._monocular_SpikeFlowNetLike        -> build('PLIFNetMono', ...)
    network/ANN_models.py:28-152    StereoSpike_equivalentANN             -> build('ANN', ...)
    network/loss.py:7-135       Total_Loss            -> total_loss
    network/metrics.py:83-95    MeanDepthError        -> mean_depth_error
state_dict key names equal the reference's, so weights move freely between the reference
modules, this restatement and the product modules.  tests/golden/make_golden.py checks this file against
the reference's own files (imported by path, in the build container only) before fixtures are written.

T > 1: `run_sequence` = reset_net, then one stateful single-step call per time step.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import sj_clock_driven as sj

ENC = [(32, 64), (64, 128), (128, 256), (256, 512)]                       # conv1..conv4 (k5 s2 p2)
DEC = [(512, 256, (33, 44)), (256, 128, (65, 87)), (128, 64, (130, 173)), (64, 32, (260, 346))]  # deconv4..1


class Gain(nn.Module):
    def __init__(self, scale_value=5., learnable=False):
        super().__init__()
        self.scale_value = nn.Parameter(torch.Tensor([scale_value])) if learnable else scale_value

    def forward(self, x):
        return torch.mul(x, self.scale_value)


class UpConv(nn.Module):
    def __init__(self, cin, cout, k, up_size, bias=False):
        super().__init__()
        self.up = nn.Sequential(nn.UpsamplingNearest2d(size=(up_size[0] + k - 1, up_size[1] + k - 1)),
                                nn.Conv2d(cin, cout, k, 1, 0, bias=bias))

    def forward(self, x):
        return self.up(x)


class SEWBlock(nn.Module):
    def __init__(self, c, make_node, gain):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(c, c, 3, 1, 1, bias=False), Gain(gain))
        self.sn1 = make_node()
        self.conv2 = nn.Sequential(nn.Conv2d(c, c, 3, 1, 1, bias=False), Gain(gain))
        self.sn2 = make_node()

    def forward(self, x):
        out = self.sn2(self.conv2(self.sn1(self.conv1(x))))
        out += x                     # blocks.py:171 (in place on the spike tensor)
        return out


class AnnResBlock(nn.Module):
    def __init__(self, c, act):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(c, c, 3, 1, 1, bias=True), act, nn.BatchNorm2d(c))
        self.conv2 = nn.Sequential(nn.Conv2d(c, c, 3, 1, 1, bias=True), act, nn.BatchNorm2d(c))

    def forward(self, x):
        out = self.conv2(self.conv1(x))
        out += x
        return out


class RefNet(nn.Module):
    """One class for the four topologically identical reference models."""
    RATE_KEYS = ['out_bottom', 'out_conv1', 'out_conv2', 'out_conv3', 'out_conv4', 'out_rconv', 'out_combined',
                 'out_deconv4', 'out_add4', 'out_deconv3', 'out_add3', 'out_deconv2', 'out_add2',
                 'out_deconv1', 'out_add1']

    def __init__(self, in_ch, tail, bottleneck_block, head_gain, ineuron, returns_spikes, conv_bias,
                 input_size=(260, 346)):
        super().__init__()
        # The reference hard-wires 260x346 (SNN_models.py:111-146).  `input_size` generalises the same arithmetic
        # (k5 s2 p2 pyramid) so tests can run the identical graph on small frames; the default is the reference.
        sizes = [tuple(input_size)]
        for _ in range(4):
            sizes.append(((sizes[-1][0] - 1) // 2 + 1, (sizes[-1][1] - 1) // 2 + 1))
        dec = [(ci, co, sizes[3 - j]) for j, (ci, co, _) in enumerate(DEC)]
        assert tuple(input_size) != (260, 346) or [d[2] for d in dec] == [d[2] for d in DEC]
        self.max_test_accuracy = float('inf')
        self.epoch = 0
        self.returns_spikes = returns_spikes
        self.bottom = nn.Sequential(nn.Conv2d(in_ch, 32, 5, 1, 2, bias=conv_bias), *tail(32))
        for i, (ci, co) in enumerate(ENC, 1):
            setattr(self, f'conv{i}', nn.Sequential(nn.Conv2d(ci, co, 5, 2, 2, bias=conv_bias), *tail(co)))
        self.bottleneck = nn.Sequential(bottleneck_block(), bottleneck_block())
        for lvl, (ci, co, size) in zip((4, 3, 2, 1), dec):
            setattr(self, f'deconv{lvl}', nn.Sequential(UpConv(ci, co, 5, size), *tail(co)))
        for lvl, c in zip((4, 3, 2, 1), (256, 128, 64, 32)):
            head = [UpConv(c, 1, 3, sizes[0], bias=True)] + ([Gain(head_gain)] if head_gain is not None else [])
            setattr(self, f'predict_depth{lvl}', nn.Sequential(*head))
        self.Ineurons = ineuron

    def _run(self, x, rates=None):
        def note(name, t):
            if rates is not None:
                rates[name] = t.count_nonzero() / t.numel()
            return t
        frame = x[:, 0, :, :, :]
        enc = [note('out_bottom', self.bottom(frame))]
        for i in range(1, 5):
            enc.append(note(f'out_conv{i}', getattr(self, f'conv{i}')(enc[-1])))
        cur = note('out_rconv', self.bottleneck(enc[4]))
        depths, spikes = [], [cur]
        for lvl in (4, 3, 2, 1):
            dec = note(f'out_deconv{lvl}', getattr(self, f'deconv{lvl}')(cur))
            cur = note(f'out_add{lvl}', dec + enc[lvl - 1])
            self.Ineurons(getattr(self, f'predict_depth{lvl}')(cur))
            depths.append(self.Ineurons.v)
            spikes.append(cur)
        return depths[::-1], spikes

    def forward(self, x):
        depths, spikes = self._run(x)
        return (depths, spikes) if self.returns_spikes else depths

    def calculate_firing_rates(self, x):
        rates = {k: 0. for k in self.RATE_KEYS}
        self._run(x, rates)
        return rates

    def set_init_depths_potentials(self, depth_prior):
        self.Ineurons.v = depth_prior

    def detach(self):
        for m in self.modules():
            if isinstance(m, sj.BaseNode) and isinstance(m.v, torch.Tensor):
                m.v.detach_()

    def count_trainable_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


def build(name, multiply_factor=1., surrogate_function=None, tau=10., v_threshold=1.0, v_reset=0.0, use_plif=False,
          activation_function=None, sigmoid_alpha=4.0, input_size=(260, 346)):
    """name in {'StereoSpike', 'PLIFNet', 'PLIFNetMono', 'ANN'}; kwargs as the reference constructors."""
    if name == 'StereoSpike':
        # SNN_models.py:71-72: v_threshold / v_reset arguments are swallowed (always 1.0 / 0.0); the bottleneck
        # does not receive surrogate_function and keeps SEWResBlock's default Sigmoid (blocks.py:142).
        sg = surrogate_function if surrogate_function is not None else sj.Sigmoid(sigmoid_alpha)
        node = lambda: sj.IFNode(1.0, 0.0, sg, True)
        bn_node = lambda: sj.IFNode(1.0, 0.0, sj.Sigmoid(sigmoid_alpha), True)
        tail = lambda c: [Gain(multiply_factor), node()]
        return RefNet(4, tail, lambda: SEWBlock(512, bn_node, multiply_factor), multiply_factor,
                      sj.IFNode(float('inf'), 0.0, sg), True, False, input_size)
    if name in ('PLIFNet', 'PLIFNetMono'):
        if use_plif:   # library-default surrogate (Sigmoid)
            node = lambda: sj.ParametricLIFNode(tau, v_threshold, v_reset, sj.Sigmoid(sigmoid_alpha), True)
        else:
            node = lambda: sj.LIFNode(tau, v_threshold, v_reset, sj.ATan(), True)
        bn_node = lambda: sj.ParametricLIFNode(tau, v_threshold, v_reset, sj.Sigmoid(sigmoid_alpha), True)  # :293-294
        tail = lambda c: [Gain(multiply_factor), node()]
        return RefNet(4 if name == 'PLIFNet' else 2, tail, lambda: SEWBlock(512, bn_node, multiply_factor),
                      multiply_factor, sj.IFNode(float('inf'), v_reset, sj.ATan()), name == 'PLIFNet', False, input_size)
    if name == 'ANN':
        act = activation_function if activation_function is not None else nn.Sigmoid()
        tail = lambda c: [act, nn.BatchNorm2d(c)]
        # ANN_models.py:41-66 encoder convs have bias, :75-94 the decoder up-convs keep bias=False, :98-109 no gain
        return RefNet(4, tail, lambda: AnnResBlock(512, act), None, sj.IFNode(float('inf'), 0., sj.ATan()), False, True,
                      input_size)
    raise ValueError(name)


def run_sequence(net, x_seq):
    """x_seq [B, T, C, H, W]: reset, then T stateful single-step calls; returns the last call's output."""
    sj.reset_net(net)
    out = None
    for t in range(x_seq.shape[1]):
        out = net(x_seq[:, t:t + 1])
    return out


# ---- float64-convolution mode ------------------------------------------------------------------------------
def ste_round(t, dtype):
    """Value rounded to `dtype` (nearest even) and returned as fp32; gradient = identity (straight-through): how a storage-format
    narrowing enters the float64-convolution oracle of the 16-bit activation modes."""
    return t + (t.detach().to(dtype).to(t.dtype) - t.detach())


class float64_convs:
    """Context manager: every nn.Conv2d of `net` evaluates its convolution in float64 and rounds ONCE to fp32 (forward and, through
    autograd, both gradients) — the reference graph with the synapse arithmetic taken to (almost) infinite precision.  This is the
    yard-stick for the product's synapse forms: MIOpen's fp32 convolutions, fp32 GEMM + gather, exact bf16x3 MFMA GEMMs all differ from
    each other by fp32 summation order; each is held against THIS value instead of against another fp32 summation order
    (tests/_pinned.py).  Neuron arithmetic, gains, adds, I-pool, loss stay the fp32 op-by-op restatement.

    narrow: optional callable  module name -> (weight dtype | None, output dtype | None): the 16-bit activation modes of the build
    (BASELINE configs 2 / 5; the reference itself is fp32-only) DEFINED end to end — at the named synapse the weight is rounded once to
    the weight dtype before the (float64) contraction and the result is rounded once to the output dtype, i.e. what is stored in HBM
    between layers is a 16-bit value, exactly at the points where the product narrows (oracle/np_x16.py is the per-kernel statement of
    the same semantics).  Gradients pass the roundings straight through (ste_round), so the oracle's backward is the exact gradient of
    the narrowed forward."""

    def __init__(self, net, narrow=None):
        self.convs = [(n, m) for n, m in net.named_modules() if isinstance(m, nn.Conv2d)]
        self.narrow = narrow

    def __enter__(self):
        for name, c in self.convs:
            wdt, odt = self.narrow(name) if self.narrow is not None else (None, None)

            def fwd(inp, weight, bias, c=c, wdt=wdt, odt=odt):
                w = weight if wdt is None else ste_round(weight, wdt)
                y = F.conv2d(inp.double(), w.double(), None if bias is None else bias.double(), c.stride, c.padding,
                             c.dilation, c.groups).float()
                return y if odt is None else ste_round(y, odt)
            c._conv_forward = fwd
        return self

    def __exit__(self, *a):
        for _, c in self.convs:
            c.__dict__.pop('_conv_forward', None)
        return False


# ---- loss.py / metrics.py ------------------------------------------------------------------------------
def _masked_residual(pred, gt):
    mask = ~torch.isnan(gt)
    n = torch.count_nonzero(mask)
    res = pred - gt
    res[mask == False] = 0  # noqa: E712  (as loss.py:19)
    return res, mask, n


def scale_invariant_loss(pred, gt):
    res, mask, n = _masked_residual(pred, gt)
    mse = 1 / n * torch.sum(torch.pow(res[mask], 2))
    quad = 1 / (n ** 2) * torch.pow(torch.sum(res[mask]), 2)
    return mse - quad


def gradient_matching_loss(pred, gt):
    res, mask, n = _masked_residual(pred, gt)
    sx = torch.Tensor([[1, 0, -1], [2, 0, -2], [1, 0, -1]]).view(1, 1, 3, 3).to(res.device)
    sy = torch.Tensor([[1, 2, 1], [0, 0, 0], [-1, -2, -1]]).view(1, 1, 3, 3).to(res.device)
    gx = F.conv2d(res, sx, stride=1, padding=1)
    gy = F.conv2d(res, sy, stride=1, padding=1)
    gx *= mask
    gy *= mask
    return 1 / n * torch.sum(torch.abs(gx[mask]) + torch.abs(gy[mask]))


def total_loss(preds, gt, spikes=None, alpha=0.5, scale_weights=(1., 1., 1., 1.), penalize_spikes=False, beta=1.):
    si, gm = 0.0, 0.0
    for w, p in zip(scale_weights, preds):
        g = F.interpolate(gt, size=p.shape[-2:], mode='bilinear', align_corners=False)   # loss.py:38,:90
        si = si + w * scale_invariant_loss(p, g)
    for w, p in zip(scale_weights, preds):
        g = F.interpolate(gt, size=p.shape[-2:], mode='bilinear', align_corners=False)
        gm = gm + w * gradient_matching_loss(p, g)
    loss = si + alpha * gm
    if penalize_spikes:
        pen = 0.0
        for s in spikes:
            pen = pen + 1 / (2 * s.numel()) * torch.sum(torch.pow(s, 2))
        loss = loss + beta * pen
    return loss


def mean_depth_error(pred, gt):
    res, mask, n = _masked_residual(pred, gt)
    return torch.sum(torch.abs(res[mask])) / n
