"""TEST INFRASTRUCTURE (CPU checker, never imported by the product): numpy restatement of the decoder's backward in the BOX-SUM form the round-4
HIP kernels evaluate (stereospike_amd/csrc/ss_upconv_box.hip).

Reference: autograd through NNConvUpsampling (/root/reference/network/blocks.py:110-132: UpsamplingNearest2d(size = up + k - 1) -> Conv2d(k = 5, stride 1,
padding 0)), decoder call sites /root/reference/network/SNN_models.py:110-129.  With src_y / src_x the resize's source-index tables (torch's own
UpsamplingNearest2d: fused.nearest_tables) and [lo, hi) their inverse ranges,

    y[nb, Y, X, co]               = sum_{ky, kx, ci} W[co, ci, ky, kx] * x[nb, src_y[Y + ky], src_x[X + kx], ci]
    g_P[nb, iy, ix, (ky, kx), co] = sum over Y in [lo_y[iy] - ky, hi_y[iy] - ky) & [0, H), X in [lo_x[ix] - kx, hi_x[ix] - kx) & [0, W)  of g_y[nb, Y, X, co]
    g_x[nb, iy, ix, ci]           = sum_{tap, co} g_P[nb, iy, ix, tap, co] * W[co, ci, tap]
    g_W[co, ci, tap]              = sum_{nb, iy, ix} x[nb, iy, ix, ci] * g_P[nb, iy, ix, tap, co]

(oracle/ss_neuron_ref.c::ss_ref_upconv_cl_bwd_f32 is g_P with this summation order: rows top to bottom, inside a row left to right, every sum started from
+0).  g_P is 25 C_out floats per SOURCE pixel, but it only ever holds RECTANGLE sums of g_y, and the distinct rectangles are few: the distinct vertical
ranges {[lo_y[iy] - ky, hi_y[iy] - ky) & [0, H)} number about H + 4 (a nearest resize by ~2 makes (iy, ky) and (iy + 1, ky + 2) the same range), likewise
horizontally.  So with the range lists VR, HR (id 0 = the empty range) and the maps vmap[iy][ky], hmap[ix][kx] into them

    B[nb, j, i, co]               = sum_{Y in VR[j]} ( sum_{X in HR[i]} g_y[nb, Y, X, co] )                      ("box-sum image", ~ the size of g_y)
    g_P[nb, iy, ix, (ky, kx), co] = B[nb, vmap[iy][ky], hmap[ix][kx], co]                                       (bit for bit)

and both contractions read B through the two small maps — a strided 5 x 5 convolution over B and its weight gradient — instead of a 6.25 x larger g_P.
The kernels read B as three bf16 planes (round-to-nearest-even split: h + m + l == B exactly)."""
import numpy as np


def range_tables(lo, hi, n_out, k=5):
    """lo, hi [n_in]: inverse ranges of the resize's source-index table (positions of the up-sampled axis that read source index i);
    n_out: output extent (up-sampled extent - k + 1).  Returns (ranges [NR, 2] int32 rows (start, length), id 0 = the empty range, the others sorted by
    (start, length); rmap [n_in, k] int32: id of [lo[i] - t, hi[i] - t) & [0, n_out))."""
    lo = np.asarray(lo, np.int64)
    hi = np.asarray(hi, np.int64)
    n_in = lo.shape[0]
    pairs = {}
    for i in range(n_in):
        for t in range(k):
            a, b = max(int(lo[i]) - t, 0), min(int(hi[i]) - t, n_out)
            pairs[(i, t)] = (a, b - a) if b > a else (0, 0)
    uniq = sorted({p for p in pairs.values() if p[1] > 0})
    ids = {p: n + 1 for n, p in enumerate(uniq)}
    ids[(0, 0)] = 0
    ranges = np.array([(0, 0)] + uniq, np.int32)
    rmap = np.array([[ids[pairs[(i, t)]] for t in range(k)] for i in range(n_in)], np.int32)
    return ranges, rmap


def boxsum(g_y, vr, hr):
    """g_y [NB, H, W, C] fp32 -> B [NB, NVR, NHR, C] fp32 in ss_ref_upconv_cl_bwd_f32's order (rows top to bottom, columns left to right, sums from +0)."""
    g_y = np.asarray(g_y, np.float32)
    NB, H, W, C = g_y.shape
    hs = np.zeros((NB, H, len(hr), C), np.float32)                    # horizontal sums per output row and horizontal range
    for i, (x0, n) in enumerate(hr):
        cs = np.zeros((NB, H, C), np.float32)
        for x in range(x0, x0 + n):
            cs = cs + g_y[:, :, x]
        hs[:, :, i] = cs
    B = np.zeros((NB, len(vr), len(hr), C), np.float32)
    for j, (y0, n) in enumerate(vr):
        acc = np.zeros((NB, len(hr), C), np.float32)
        for y in range(y0, y0 + n):
            acc = acc + hs[:, y]
        B[:, j] = acc
    return B


def g_P_from_box(B, vmap, hmap):
    """B [NB, NVR, NHR, C], vmap [h, k], hmap [w, k] -> g_P [NB, h, w, k * k, C] (the gather B[vmap[iy][ky], hmap[ix][kx]])."""
    k = vmap.shape[1]
    return np.stack([B[:, vmap[:, ky]][:, :, hmap[:, kx]] for ky in range(k) for kx in range(k)], 3)


def bf16_rn(a):
    """fp32 array -> bf16 bit patterns (uint16), round to nearest even (NaN quieted) — v_cvt_pk_bf16_f32 / stereospike_amd's narrow<SS_DT_BF16>."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def bf16_to_f32(b):
    return (np.asarray(b, np.uint16).astype(np.uint32) << 16).view(np.float32)


def split3_rn(a):
    """fp32 -> (h, m, l) bf16 bit patterns with h + m + l == a exactly for finite a of moderate exponent (8 + 8 + 8 significant bits; the residuals are
    exact fp32 subtractions) — the split of the box-sum kernel and of ss_gemm6_f32's operands."""
    a = np.ascontiguousarray(a, np.float32)
    h = bf16_rn(a)
    r1 = a - bf16_to_f32(h)
    m = bf16_rn(r1)
    r2 = r1 - bf16_to_f32(m)
    return h, m, bf16_rn(r2)


def box_planes(B, co_chunk=8):
    """B [NB, NVR, NHR, C] fp32 -> the HBM layout of the kernels: uint16 [NB, C / co_chunk, 3, NVR, NHR, co_chunk] (bf16 bits; plane 0 = high term)."""
    NB, NVR, NHR, C = B.shape
    assert C % co_chunk == 0
    planes = np.stack(split3_rn(B), 0)                                 # [3, NB, NVR, NHR, C]
    return np.ascontiguousarray(planes.reshape(3, NB, NVR, NHR, C // co_chunk, co_chunk).transpose(1, 4, 0, 2, 3, 5))


def dgrad_from_box(B, weight, vmap, hmap, dtype=np.float64):
    """g_x [NB, h, w, C_in] = sum_{ky, kx, co} B[nb, vmap[iy][ky], hmap[ix][kx], co] * W[co, ci, ky, kx]   (weight [C_out, C_in, k, k])."""
    B = np.asarray(B, dtype)
    w = np.asarray(weight, dtype)
    k = w.shape[2]
    gx = 0
    for ky in range(k):
        for kx in range(k):
            gx = gx + B[:, vmap[:, ky]][:, :, hmap[:, kx]] @ w[:, :, ky, kx]
    return gx


def dgrad_magnitude(B, weight, vmap, hmap):
    """sum |B| |W| per output element (float64): the yard-stick of the six-term kernel's element-wise bound."""
    return dgrad_from_box(np.abs(np.asarray(B, np.float64)), np.abs(np.asarray(weight, np.float64)), vmap, hmap)


def wgrad_from_box(B, x, vmap, hmap, dtype=np.float64):
    """g_W [C_out, C_in, k, k] = sum_{nb, iy, ix} x[nb, iy, ix, ci] * B[nb, vmap[iy][ky], hmap[ix][kx], co]   (x [NB, h, w, C_in])."""
    B = np.asarray(B, dtype)
    x = np.asarray(x, dtype)
    k = vmap.shape[1]
    NB, h, w, Cin = x.shape
    Cout = B.shape[-1]
    gw = np.zeros((Cout, Cin, k, k), dtype)
    xf = x.reshape(-1, Cin)
    for ky in range(k):
        for kx in range(k):
            gp = B[:, vmap[:, ky]][:, :, hmap[:, kx]].reshape(-1, Cout)
            gw[:, :, ky, kx] = gp.T @ xf
    return gw
