"""TEST INFRASTRUCTURE ONLY (see oracle/README.md) — numpy restatement of the per-scale loss statistics and their gradient,
the checker for `ss_loss_stats_f32` / `ss_loss_grad_f32` (include/ss_neuron.h).

Follows /root/reference/network/loss.py:7-24 (ScaleInvariant_Loss: res = (pred-gt)[mask]; 1/n*sum(res^2) - 1/n^2*sum(res)^2),
:44-75 (GradientMatching_Loss: res zeroed at invalid pixels, Sobel x / y cross-correlation with padding 1, masked, 1/n*sum(|gx|+|gy|))
and network/metrics.py:83-95 (MeanDepthError: sum|res|/n).  Pinned: tests/test_oracle.py checks these sums against the
reference's own outputs in tests/golden/loss_metric.npz, and the gradient against torch autograd of oracle/ref_network.total_loss.
"""
import numpy as np

KX = np.array([[1, 0, -1], [2, 0, -2], [1, 0, -1]], np.float64)      # loss.py:61-63
KY = np.array([[1, 2, 1], [0, 0, 0], [-1, -2, -1]], np.float64)      # loss.py:65-67


def _residual(pred, gt):
    pred = np.asarray(pred, np.float32).reshape(-1, pred.shape[-2], pred.shape[-1])
    gt = np.asarray(gt, np.float32).reshape(pred.shape)
    mask = ~np.isnan(gt)
    res = np.where(mask, pred - np.where(mask, gt, 0), np.float32(0)).astype(np.float32)
    return res, mask


def _xcorr(img, K):
    """3x3 cross-correlation with zero padding 1 (F.conv2d(..., padding=1)), batched over axis 0."""
    B, H, W = img.shape
    p = np.zeros((B, H + 2, W + 2), img.dtype)
    p[:, 1:-1, 1:-1] = img
    out = np.zeros((B, H, W), np.float64)
    for a in range(3):
        for b in range(3):
            out += K[a, b] * p[:, a:a + H, b:b + W]
    return out


def loss_stats(pred, gt):
    """float64[5] = n, sum r, sum r^2, sum_valid |gx|+|gy|, sum |r|."""
    res, mask = _residual(pred, gt)
    r = res.astype(np.float64)
    # Sobel responses from the fp32 residual; products by 1 / 2 are exact, so fp32-vs-fp64 differences are summation order only
    gx, gy = _xcorr(res.astype(np.float64), KX), _xcorr(res.astype(np.float64), KY)
    G = (np.abs(gx.astype(np.float32)) + np.abs(gy.astype(np.float32)))[mask].astype(np.float64).sum()
    return np.array([mask.sum(), r.sum(), (res * res).astype(np.float64).sum(), G, np.abs(r).sum()], np.float64)


def loss_grad(pred, gt, sums, coef):
    """d[c_si * ScaleInvariant + c_gm * GradientMatching] / d pred, float32, shape of pred."""
    res, mask = _residual(pred, gt)
    n, s1 = np.float32(sums[0]), np.float32(sums[1])
    gx = _xcorr(res.astype(np.float64), KX).astype(np.float32)
    gy = _xcorr(res.astype(np.float64), KY).astype(np.float32)
    sx = np.where(mask, np.sign(gx), 0).astype(np.float64)
    sy = np.where(mask, np.sign(gy), 0).astype(np.float64)
    # adjoint of a cross-correlation = cross-correlation with the kernel rotated by 180 degrees
    T = (_xcorr(sx, KX[::-1, ::-1]) + _xcorr(sy, KY[::-1, ::-1])).astype(np.float32)
    c_si, c_gm = np.float32(coef[0]), np.float32(coef[1])
    g = c_si * (np.float32(2) * res / n - np.float32(2) * s1 / (n * n)) + (c_gm / n) * T
    return np.where(mask, g, np.float32(0)).astype(np.float32).reshape(np.shape(pred))
