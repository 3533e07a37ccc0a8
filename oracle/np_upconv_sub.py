"""TEST INFRASTRUCTURE (CPU checker, never imported by the product): numpy restatement of the decoder stage's FORWARD in the sub-pixel ("merged tap") form
the round-4 HIP kernel evaluates (stereospike_amd/csrc/ss_upconv_sub.hip).

Reference: NNConvUpsampling (/root/reference/network/blocks.py:110-132: UpsamplingNearest2d(size = up + k - 1) -> Conv2d(k = 5, stride 1, padding 0, no
bias)), decoder call sites /root/reference/network/SNN_models.py:110-129.  With src_y / src_x the resize's source-index tables (torch's own
UpsamplingNearest2d: fused.nearest_tables)

    y[nb, Y, X, co] = sum_{ky, kx, ci} W[co, ci, ky, kx] * x[nb, src_y[Y + ky], src_x[X + kx], ci]

A nearest resize by ~2 makes the 5 rows src_y[Y .. Y + 4] take only 2 or 3 DISTINCT values, in runs: the run lengths (2, 2, 1) / (1, 2, 2) on the regular
lattice, (3, 2) / (2, 3) / (1, 3, 1) at the few places where the resize's rounding repeats a row three times.  Taps that read the same source pixel can be
added BEFORE the multiplication: with the runs [k0_r, k0_r + n_r) of an output row (its vertical CLASS) and of an output column

    Wm[cv, ch][co, ci, r, c] = sum_{ky in run r of cv} sum_{kx in run c of ch} W[co, ci, ky, kx]          (<= 3 x 3 merged taps per class pair)
    y[nb, Y, X, co]          = sum_{r, c, ci} Wm[class(Y), class(X)][co, ci, r, c] * x[nb, src_y[Y + k0_r], src_x[X + k0_c], ci]

— 9 multiply-adds per input channel and output pixel instead of 25, i.e. 1.44 x the minimum of the projection form (25 per SOURCE pixel = 6.25 per output
pixel) without the projection's per-tap tensor P, its gather, or the fused kernel's window halo (1.9 x).  Output pixels of one class pair share their
weights, so 32 of them are the M dimension of one MFMA tile.

The kernel adds the merged taps in fp32 (ky outer, kx inner, from +0) and splits the sum into three bf16 terms (exact: 24 significant bits); spikes are
exact in bf16, so every product is exact and the result differs from the float64 value of the reference formula by (a) the fp32 rounding of the <= 9-term
weight sums and (b) the fp32 accumulation over 9 C_in products."""
import numpy as np


def axis_classes(src, n_out, k=5):
    """src [n_out + k - 1]: source index of every up-sampled position.  Returns (keys, cls, k0, kn):
    keys: sorted list of the distinct run-length tuples; cls [n_out]: class id of each output position; k0 / kn [n_classes, 3]: first tap / tap count of each
    run (kn = 0: no such run).  Raises if a position has more than 3 runs (a resize factor below ~1.7)."""
    src = np.asarray(src, np.int64)
    per = []
    for Y in range(n_out):
        runs, start = [], 0
        for t in range(1, k + 1):
            if t == k or src[Y + t] != src[Y + start]:
                runs.append(t - start)
                start = t
        if len(runs) > 3:
            raise ValueError('more than 3 distinct source positions under one 5-tap window')
        per.append(tuple(runs))
    keys = sorted(set(per))
    ids = {key: n for n, key in enumerate(keys)}
    cls = np.array([ids[p] for p in per], np.int32)
    k0 = np.zeros((len(keys), 3), np.int32)
    kn = np.zeros((len(keys), 3), np.int32)
    for n, key in enumerate(keys):
        a = 0
        for r, ln in enumerate(key):
            k0[n, r], kn[n, r] = a, ln
            a += ln
    return keys, cls, k0, kn


def axis_blocks(src, n_out, cls, k0, kn, max_out, max_src):
    """Cut the output positions of every class (ascending) into blocks of <= max_out positions that read <= max_src DISTINCT source positions.
    Returns a list of dicts: cls, out [n] (output positions), src [m] (distinct source positions, ascending), slot [n, 3] (index into src of run r's source
    position; 0 where the class has no run r)."""
    src = np.asarray(src, np.int64)
    blocks = []
    for c in range(k0.shape[0]):
        pos = [int(Y) for Y in np.nonzero(cls == c)[0]]
        ng = int((kn[c] > 0).sum())
        a = 0
        while a < len(pos):
            n, need = 0, set()
            while a + n < len(pos) and n < max_out:
                more = need | {int(src[pos[a + n] + k0[c, r]]) for r in range(ng)}
                if len(more) > max_src:
                    break
                need = more
                n += 1
            assert n >= 1
            srcs = sorted(need)
            where = {s: i for i, s in enumerate(srcs)}
            slot = np.zeros((n, 3), np.int32)
            for i in range(n):
                for r in range(ng):
                    slot[i, r] = where[int(src[pos[a + i] + k0[c, r]])]
            blocks.append(dict(cls=c, out=np.array(pos[a:a + n], np.int32), src=np.array(srcs, np.int32), slot=slot))
            a += n
    return blocks


def merged_weights(W, vk0, vkn, hk0, hkn, dtype=np.float32):
    """W [C_out, C_in, 5, 5] -> Wm [n_vclasses, n_hclasses, C_out, C_in, 3, 3]: the taps of run (r, c) added ky outer, kx inner, from +0, every addition
    rounded to `dtype` (fp32 = what ss_upconv_sub_prep does; float64 for the exact identity)."""
    W = np.asarray(W, dtype)
    Wm = np.zeros((vk0.shape[0], hk0.shape[0]) + W.shape[:2] + (3, 3), dtype)
    for cv in range(vk0.shape[0]):
        for ch in range(hk0.shape[0]):
            for r in range(3):
                for c in range(3):
                    acc = np.zeros(W.shape[:2], dtype)
                    for ky in range(vk0[cv, r], vk0[cv, r] + vkn[cv, r]):
                        for kx in range(hk0[ch, c], hk0[ch, c] + hkn[ch, c]):
                            acc = (acc + W[:, :, ky, kx]).astype(dtype)
                    Wm[cv, ch, :, :, r, c] = acc
    return Wm


def forward_direct(x, W, src_y, src_x, H, Wd, dtype=np.float64):
    """The reference formula: x [NB, h, w, C_in], W [C_out, C_in, 5, 5] -> y [NB, H, Wd, C_out]."""
    x = np.asarray(x, dtype)
    W = np.asarray(W, dtype)
    src_y, src_x = np.asarray(src_y, np.int64), np.asarray(src_x, np.int64)
    y = np.zeros((x.shape[0], H, Wd, W.shape[0]), dtype)
    for ky in range(5):
        for kx in range(5):
            y += x[:, src_y[ky:ky + H]][:, :, src_x[kx:kx + Wd]] @ W[:, :, ky, kx].T
    return y


def forward_merged(x, W, src_y, src_x, H, Wd, merge_dtype=np.float32, dtype=np.float64):
    """The sub-pixel form: weights merged per class pair in `merge_dtype`, then <= 9 taps per output pixel accumulated in `dtype`."""
    x = np.asarray(x, dtype)
    src_y, src_x = np.asarray(src_y, np.int64), np.asarray(src_x, np.int64)
    _, vcls, vk0, vkn = axis_classes(src_y, H)
    _, hcls, hk0, hkn = axis_classes(src_x, Wd)
    Wm = merged_weights(W, vk0, vkn, hk0, hkn, merge_dtype).astype(dtype)
    y = np.zeros((x.shape[0], H, Wd, Wm.shape[2]), dtype)
    for cv in range(vk0.shape[0]):
        Ys = np.nonzero(vcls == cv)[0]
        for ch in range(hk0.shape[0]):
            Xs = np.nonzero(hcls == ch)[0]
            if not len(Ys) or not len(Xs):
                continue
            acc = np.zeros((x.shape[0], len(Ys), len(Xs), Wm.shape[2]), dtype)
            for r in range(3):
                if not vkn[cv, r]:
                    continue
                for c in range(3):
                    if not hkn[ch, c]:
                        continue
                    acc += x[:, src_y[Ys + vk0[cv, r]]][:, :, src_x[Xs + hk0[ch, c]]] @ Wm[cv, ch, :, :, r, c].T
            y[:, Ys[:, None], Xs[None, :]] = acc
    return y


def magnitude(x, W, src_y, src_x, H, Wd):
    """sum |x| |W| per output element (float64): the yard-stick of the kernel's element-wise bound."""
    return forward_direct(np.abs(np.asarray(x, np.float64)), np.abs(np.asarray(W, np.float64)), src_y, src_x, H, Wd)
