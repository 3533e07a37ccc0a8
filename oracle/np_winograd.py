"""TEST INFRASTRUCTURE — numpy checker of the Winograd F(2x2, 3x3) data-gradient kernels (ss_wino_dgrad_*_f32).

The map itself is the autograd backward of a 3x3 / stride 1 / pad 1 convolution w.r.t. its input (the reference's SEWResBlock convs,
/root/reference/network/blocks.py:146-159, differentiated by torch).  This file restates (1) that gradient directly in float64
(`dgrad_direct64`: the yard-stick) and (2) the three transforms in fp32 with the kernels' operation order (`weights`, `input_tiles`,
`output_tiles`: the kernels are held bit-exact against these).  Only tests/ may import it.

Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks" (2016):
    B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]    G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]    A^T = [1 1 1 0; 0 1 -1 -1]
"""
import numpy as np

f32 = np.float32


def dgrad_direct64(g, w):
    """g [NB, H, W, Co], w [Co, Ci, 3, 3] -> g_in [NB, H, W, Ci] in float64: g_in[y][x][ci] = sum g[y + a - 1][x + b - 1][co] w[co][ci][2 - a][2 - b]."""
    g = g.astype(np.float64); w = w.astype(np.float64)
    NB, H, W, Co = g.shape
    gp = np.zeros((NB, H + 2, W + 2, Co)); gp[:, 1:-1, 1:-1] = g
    out = np.zeros((NB, H, W, w.shape[1]))
    for a in range(3):
        for b in range(3):
            out += gp[:, a:a + H, b:b + W] @ w[:, :, 2 - a, 2 - b]
    return out


def weights(w):
    """w [Co, Ci, 3, 3] fp32 -> U [16, Co, Ci] fp32 (kernel op order: 0.5 * ((f0 +- f1) + f2))."""
    w = w.astype(f32)
    f = w[:, :, ::-1, ::-1]                                            # f[a][b] = w[2 - a][2 - b]
    h = f32(0.5)
    t = np.stack((f[:, :, 0], h * ((f[:, :, 0] + f[:, :, 1]) + f[:, :, 2]), h * ((f[:, :, 0] - f[:, :, 1]) + f[:, :, 2]), f[:, :, 2]), 2)   # [Co, Ci, 4, 3]
    u = np.stack((t[..., 0], h * ((t[..., 0] + t[..., 1]) + t[..., 2]), h * ((t[..., 0] - t[..., 1]) + t[..., 2]), t[..., 2]), 3)           # [Co, Ci, 4, 4]
    return np.ascontiguousarray(u.reshape(w.shape[0], w.shape[1], 16).transpose(2, 0, 1)).astype(f32)


def input_tiles(g):
    """g [NB, H, W, C] fp32 -> V [16, T, C] fp32."""
    g = g.astype(f32)
    NB, H, W, C = g.shape
    th, tw = (H + 1) // 2, (W + 1) // 2
    gp = np.zeros((NB, 2 * th + 2, 2 * tw + 2, C), f32); gp[:, 1:H + 1, 1:W + 1] = g
    d = np.empty((4, 4, NB, th, tw, C), f32)
    for a in range(4):
        for b in range(4):
            d[a, b] = gp[:, a:a + 2 * th:2, b:b + 2 * tw:2]
    t = np.stack((d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]), 0)            # [4 (a), 4 (b), ...]
    v = np.stack((t[:, 0] - t[:, 2], t[:, 1] + t[:, 2], t[:, 2] - t[:, 1], t[:, 1] - t[:, 3]), 1)
    return np.ascontiguousarray(v.reshape(16, NB * th * tw, C))


def output_tiles(m, NB, H, W):
    """M [16, T, C] fp32 -> g_in [NB, H, W, C] fp32."""
    m = m.astype(f32)
    C = m.shape[2]
    th, tw = (H + 1) // 2, (W + 1) // 2
    m = m.reshape(4, 4, NB, th, tw, C)
    t = np.stack(((m[0] + m[1]) + m[2], (m[1] - m[2]) - m[3]), 0)                   # [2, 4 (b), ...]
    y = np.stack(((t[:, 0] + t[:, 1]) + t[:, 2], (t[:, 1] - t[:, 2]) - t[:, 3]), 1)  # [2, 2, NB, th, tw, C]
    out = np.zeros((NB, 2 * th, 2 * tw, C), f32)
    for a in range(2):
        for b in range(2):
            out[:, a::2, b::2] = y[a, b]
    return np.ascontiguousarray(out[:, :H, :W])


def dgrad(g, w):
    """The whole Winograd data gradient in fp32 numpy (transform-domain products in float64 accumulated, rounded once: the GEMM's accuracy class)."""
    NB, H, W, _ = g.shape
    v, u = input_tiles(g), weights(w)
    m = np.einsum('ktc,kcd->ktd', v.astype(np.float64), u.astype(np.float64)).astype(f32)
    return output_tiles(m, NB, H, W)
