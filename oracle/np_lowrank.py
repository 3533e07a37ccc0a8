"""TEST INFRASTRUCTURE — numpy checker of the low-rank second gradient of ss_neuron_bwd_fork_lr_f32.  Only tests/ may import it.

The map: each prediction head of the reference is NNConvUpsampling(C, 1, kernel_size=3) (/root/reference/network/SNN_models.py:150-163,
blocks.py:110-132), i.e. on the projected form P = x [rows, C] @ W2^T [C, 9] followed by a gather.  torch's autograd gives the head's input
the gradient g_x = g_P [rows, 9] @ W2 [9, C].  The kernel forms that product per element in fp32 with the taps in ascending order and every
multiply and add rounded separately (no fused multiply-add) — restated here so the kernel can be held bit-exact against it.
"""
import numpy as np

f32 = np.float32


def head_input_gradient(lr_p, lr_w):
    """lr_p [..., R] fp32, lr_w [R, C] fp32 -> [..., C] fp32: ((p0 w0 + p1 w1) + p2 w2) + ... in fp32, each op rounded once."""
    lr_p, lr_w = np.asarray(lr_p, f32), np.asarray(lr_w, f32)
    acc = lr_p[..., 0:1] * lr_w[0]
    for j in range(1, lr_w.shape[0]):
        acc = (acc + lr_p[..., j:j + 1] * lr_w[j]).astype(f32)
    return acc.astype(f32)


def head_input_gradient64(lr_p, lr_w):
    """The same product in float64 (the yard-stick for the dense fp32 GEMM it replaces)."""
    return np.asarray(lr_p, np.float64) @ np.asarray(lr_w, np.float64)
