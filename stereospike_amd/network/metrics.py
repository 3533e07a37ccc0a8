"""Evaluation metric and depth-representation converters (/root/reference/network/metrics.py): MeanDepthError
(:83-95) is the "eval MDE" half of the headline metric; NaN in the ground truth marks an invalid pixel."""
import numpy as np
import torch

DISPARITY_MULTIPLIER = 7.0
FOCAL_LENGTH_X_BASELINE = {'indoor_flying': 19.941772}


def _valid(groundtruth):
    mask = ~torch.isnan(groundtruth)
    return mask, torch.count_nonzero(mask)


def mask_dead_pixels(predicted, groundtruth):
    """Copies of both maps with every invalid (NaN ground truth) pixel set to 0."""
    assert predicted.shape == groundtruth.shape, \
        "input and target tensors do not have the same shape, can't apply the same mask to them ! " \
        "Input is of shape {} and target of shape {}".format(predicted.shape, groundtruth.shape)
    mask, _ = _valid(groundtruth)
    zero = torch.zeros((), dtype=predicted.dtype, device=predicted.device)
    return torch.where(mask, predicted.detach(), zero), torch.where(mask, groundtruth.detach(), zero.to(groundtruth.dtype))


def depth_to_disparity(depth_maps):
    return DISPARITY_MULTIPLIER * FOCAL_LENGTH_X_BASELINE['indoor_flying'] / (depth_maps + 1e-15)


def disparity_to_depth(disparity_map):
    return DISPARITY_MULTIPLIER * FOCAL_LENGTH_X_BASELINE['indoor_flying'] / (disparity_map + 1e-7)


def lin_to_log_depths(depths_rect_lin, Dmax=10, alpha=6.):
    """Normalised log depth in [0, 1]:  Dlin = Dmax * exp(-alpha * (1 - Dlog))   (numpy in, numpy out)."""
    d = np.clip(depths_rect_lin, 0.0, Dmax) / Dmax
    return (1.0 + np.log(d) / alpha).clip(0, 1.0)


def log_to_lin_depths(depths_rect_log, Dmax=10, alpha=6.):
    return Dmax * torch.exp(alpha * (depths_rect_log - torch.ones_like(depths_rect_log)))


def MeanDepthError(predicted, groundtruth):
    """Mean absolute depth error over the valid pixels (metres).  Invalid residuals are zeroed, so the sum over
    the whole map equals the reference's sum over res[mask] without a data-dependent gather (no host sync).  On the MI355X
    the sum comes from the fused statistics kernel (ss_loss_stats_f32)."""
    from . import loss as _loss
    if _loss._fusable(predicted, groundtruth):
        from .. import fused
        return fused.scale_loss_terms(predicted.detach(), groundtruth)[2]
    mask, n = _valid(groundtruth)
    res = torch.where(mask, predicted - groundtruth, torch.zeros((), dtype=predicted.dtype, device=predicted.device))
    return torch.sum(torch.abs(res)) / n
