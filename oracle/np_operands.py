"""TEST INFRASTRUCTURE ONLY (see oracle/README.md) — numpy restatement of the operand preparation of the exact bf16x3 GEMM synapses
(include/ss_neuron.h: ss_im2col_cl_bf16, ss_split3_bf16; stereospike_amd/fused.py::_SpikeConvCL, _split3_bf16).

Nothing here comes from the reference (the reference calls nn.Conv2d, /root/reference/network/SNN_models.py:91-101, blocks.py:146-159);
the checker is the identity it must satisfy: with A = im2col(x) and W = Wh + Wm + Wl (bf16 terms),  A @ Wt  ==  conv2d(x, W)  up to
fp32 summation order, every product being exact when x holds small integers."""
import numpy as np


def bf16_round(a):
    """fp32 -> nearest-even bf16, returned widened back to fp32 (NaN kept quiet)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    out = r.astype(np.uint32).view(np.float32)
    return np.where(np.isnan(a), np.float32('nan'), out).astype(np.float32)


def split3(g):
    """g fp32 -> (hi, mid, lo) bf16 values (as fp32) with hi + mid + lo == g exactly."""
    g = np.asarray(g, np.float32)
    hi = bf16_round(g)
    r1 = (g - hi).astype(np.float32)
    mid = bf16_round(r1)
    lo = bf16_round((r1 - mid).astype(np.float32))
    return hi, mid, lo


def im2col_cl(x, k, stride, pad):
    """x [NB, h, w, C] -> A [NB*ho*wo, k*k*C] with column order (ky, kx, c), zero padding; values narrowed to bf16."""
    x = np.asarray(x, np.float32)
    NB, h, w, C = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    xp = np.zeros((NB, h + 2 * pad, w + 2 * pad, C), np.float32)
    xp[:, pad:pad + h, pad:pad + w] = x
    A = np.empty((NB, ho, wo, k, k, C), np.float32)
    for ky in range(k):
        for kx in range(k):
            A[:, :, :, ky, kx] = xp[:, ky:ky + stride * ho:stride, kx:kx + stride * wo:stride]
    return bf16_round(A.reshape(NB * ho * wo, k * k * C)), (ho, wo)


def wgrad_reduce3(parts, k, Cin, Cout):
    """Epilogue of the exact bf16x3 weight-gradient GEMM (ss_wgrad_reduce3_f32; autograd's Conv2d weight gradient of the reference's encoder / bottleneck
    convs, /root/reference/network/SNN_models.py:91-101, blocks.py:146-159): parts fp32 [S, k*k*Cin rows (ky, kx, ci), 3, Cout] -> g_w [Cout, Cin, k, k],
    slices added in ascending order, inside a slice ((hi + mid) + lo), every addition rounded to fp32."""
    parts = np.asarray(parts, np.float32).reshape(-1, k * k * Cin, 3, Cout)
    acc = np.zeros((k * k * Cin, Cout), np.float32)
    for s in range(parts.shape[0]):
        acc = acc + ((parts[s, :, 0] + parts[s, :, 1]) + parts[s, :, 2])
    return np.ascontiguousarray(acc.reshape(k, k, Cin, Cout).transpose(3, 2, 0, 1))
