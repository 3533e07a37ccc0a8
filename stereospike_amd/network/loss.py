"""Training loss (/root/reference/network/loss.py): scale-invariant term (:7-24) + alpha * Sobel gradient-matching
term (:44-75), each summed over the 4 prediction scales (:27-41, :78-93), + optional spike penalisation (:96-107).

Same functions and signatures; evaluated without the reference's boolean-index gathers (`res[mask]` forces a
device->host sync for its data-dependent size): residuals are zeroed at invalid pixels, so plain sums over the
map are the same quantities.  Masks / sums are batch-wide exactly as in the reference (:16-24) — the samples
of a batch are coupled through n and the squared-mean term; see DESIGN.md "DP semantics".
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import config as _config
from ..config import current as _cfg


def _residual(predicted, groundtruth):
    mask = ~torch.isnan(groundtruth)
    n = torch.count_nonzero(mask)
    res = torch.where(mask, predicted - groundtruth, torch.zeros((), dtype=predicted.dtype, device=predicted.device))
    return res, mask, n


def ScaleInvariant_Loss(predicted, groundtruth):
    res, _, n = _residual(predicted, groundtruth)
    MSE = 1 / n * torch.sum(torch.pow(res, 2))
    quad = 1 / (n ** 2) * torch.pow(torch.sum(res), 2)
    return MSE - quad


_SOBEL = {}


def _sobel(ref):
    key = (ref.device, ref.dtype)
    if key not in _SOBEL:
        sx = torch.tensor([[1, 0, -1], [2, 0, -2], [1, 0, -1]], dtype=ref.dtype, device=ref.device)
        sy = torch.tensor([[1, 2, 1], [0, 0, 0], [-1, -2, -1]], dtype=ref.dtype, device=ref.device)
        _SOBEL[key] = torch.stack((sx, sy)).view(2, 1, 3, 3)
    return _SOBEL[key]


def GradientMatching_Loss(predicted, groundtruth):
    res, mask, n = _residual(predicted, groundtruth)
    grads = F.conv2d(res, _sobel(res), stride=1, padding=1)       # x- and y-Sobel of the [N,1,H,W] residual in one conv
    grads = grads * mask                                          # gradients count only at valid pixels
    return 1 / n * torch.sum(torch.abs(grads))


def _rescale(groundtruth, like):
    size = like.shape[-2:]
    if tuple(groundtruth.shape[-2:]) == tuple(size):
        # identity resize: the reference's F.interpolate returns the map unchanged on CPU; skipping it also keeps
        # a GPU kernel from spreading NaNs through 0 * NaN
        return groundtruth
    return F.interpolate(groundtruth, size=size, mode='bilinear', align_corners=False)


def Multiscale_ScaleInvariant_Loss(predicted, groundtruth, factors=(1., 1., 1., 1.)):
    total = 0.0
    for factor, pred in zip(factors, predicted):
        total = total + factor * ScaleInvariant_Loss(pred, _rescale(groundtruth, pred))
    return total


def MultiScale_GradientMatching_Loss(predicted, groundtruth, factors=(1., 1., 1., 1.)):
    total = 0.0
    for factor, pred in zip(factors, predicted):
        total = total + factor * GradientMatching_Loss(pred, _rescale(groundtruth, pred))
    return total


def SpikePenalization_Loss(intermediary_spike_tensors):
    total = 0.0
    for s in intermediary_spike_tensors:
        total = total + 1 / (2 * s.numel()) * torch.sum(torch.pow(s, 2))
    return total


# On the MI355X the two terms of every scale come from one statistics kernel (+ one stencil kernel backward) instead of ~25
# element-wise / reduction / convolution launches per scale; the functions above stay as the definition (and the CPU form).
# (EngineConfig.FUSED_LOSS; reads of `loss.FUSED_LOSS` answer with the configuration in effect.)


def _on_device(t):
    return t.is_cuda


def _fusable(pred, gt, cfg=None):
    return (cfg if cfg is not None else _cfg()).FUSED_LOSS and _on_device(pred) and pred.dim() == 4 and pred.shape[1] == 1 and pred.shape == gt.shape \
        and pred.dtype == torch.float32 and gt.dtype == torch.float32


def multiscale_terms(predicted, groundtruth, factors=(1., 1., 1., 1.)):
    """(sum_k f_k ScaleInvariant_k, sum_k f_k GradientMatching_k, MeanDepthError of predicted[0] — the final depth map, the one
    train.py:236 / test.py score) through the fused kernels."""
    from .. import fused
    si = gm = 0.0
    mde = None
    for factor, pred in zip(factors, predicted):
        terms = fused.scale_loss_terms(pred, _rescale(groundtruth, pred).expand_as(pred))
        si, gm = si + factor * terms[0], gm + factor * terms[1]
        if mde is None:
            mde = terms[2].detach()
    return si, gm, mde


class Total_Loss(nn.Module):
    """alpha = 0.5 for linear (metric) depth; scale_weights all 1; beta weighs the spike penalisation."""

    def __init__(self, alpha=0.5, scale_weights=(1., 1., 1., 1.), penalize_spikes=False, beta=1., config=None):
        """config (build-side addition): the EngineConfig whose FUSED_LOSS decides the kernel form — pass `net.config` to run the loss under a network's
        configuration.  The loss runs OUTSIDE the network's forward, so `net.configured(FUSED_LOSS=...)` alone does not reach it; None = the ambient
        configuration at call time (`config.engine_config(...)` / the shipped default), per thread."""
        super().__init__()
        self.config = config
        self.alpha = alpha
        self.scale_weights = scale_weights
        self.penalize_spikes = penalize_spikes
        self.beta = beta

    def forward(self, predicted, groundtruth, intermediary_spike_tensors=None):
        if all(_fusable(p, _rescale(groundtruth, p), self.config) for p in predicted):
            si, gm, _ = multiscale_terms(predicted, groundtruth, self.scale_weights)
            loss = si + self.alpha * gm
            if self.penalize_spikes:
                loss = loss + self.beta * SpikePenalization_Loss(intermediary_spike_tensors)
            return loss
        loss = Multiscale_ScaleInvariant_Loss(predicted, groundtruth, self.scale_weights) + \
            self.alpha * MultiScale_GradientMatching_Loss(predicted, groundtruth, self.scale_weights)
        if self.penalize_spikes:
            loss = loss + self.beta * SpikePenalization_Loss(intermediary_spike_tensors)
        return loss


_config.guard_module(__name__, 'loss')
