"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of the 16-bit-activation variants behind
include/ss_neuron.h ss_neuron_{fwd,bwd}_x16 (BASELINE.json configs 2 and 5: bf16 / fp16 activations, fp32 membrane).

The reference itself is fp32-only (train.py:194-197 casts the inputs to float, no autocast anywhere): the low-precision mode
is a build-side addition whose semantics are DEFINED here — widen every 16-bit input to fp32, run exactly the fp32
recurrence of oracle/ss_neuron_ref.c (same op order, one rounding per op: numpy float32 arithmetic), round the stored
results to nearest-even.  Inputs / outputs are raw uint16 bit patterns so fp16 and bf16 share the code."""
import math

import numpy as np

F32 = np.float32


def widen(bits, dtype):
    bits = np.asarray(bits, np.uint16)
    if dtype == 'f16':
        return bits.view(np.float16).astype(F32)
    return (bits.astype(np.uint32) << 16).view(F32)


def narrow(f, dtype):
    f = np.asarray(f, F32)
    if dtype == 'f16':
        return f.astype(np.float16).view(np.uint16)
    u = f.view(np.uint32)
    r = ((u + np.uint32(0x7fff) + ((u >> 16) & np.uint32(1))) >> 16).astype(np.uint16)
    nan = (u & np.uint32(0x7fffffff)) > np.uint32(0x7f800000)
    return np.where(nan, ((u >> 16) | np.uint32(0x40)).astype(np.uint16), r)


def _charge(kind, v, xs, tau, k, v_reset):
    if kind == 'IF':
        return v + xs
    d = xs - (v - F32(v_reset))
    return v + d / F32(tau) if kind == 'LIF' else v + d * F32(k)


def neuron_fwd(x_bits, dtype, *, kind='IF', scale=1.0, tau=2.0, k=None, v_th=1.0, v_reset=0.0, v_init=None, skip_bits=None):
    T, N = x_bits.shape
    v = np.full(N, v_reset, F32) if v_init is None else np.asarray(v_init, F32).copy()
    out = np.empty((T, N), np.uint16)
    h = np.empty((T, N), F32)
    with np.errstate(invalid='ignore', over='ignore'):
        for t in range(T):
            xs = widen(x_bits[t], dtype) * F32(scale)
            ht = _charge(kind, v, xs, tau, k, v_reset)
            z = ((ht - F32(v_th)) >= 0).astype(F32)
            v = (F32(1) - z) * ht + z * F32(v_reset)
            o = z + widen(skip_bits[t], dtype) if skip_bits is not None else z
            out[t], h[t] = narrow(o, dtype), ht
    return dict(out=out, h=h, v_last=v)


def neuron_bwd(g_bits, h, dtype, *, kind='IF', scale=1.0, tau=2.0, k=None, v_th=1.0, v_reset=0.0, v_init=None,
               g_v_last=None, surrogate='ATan', alpha=2.0, g2_bits=None):
    """g2_bits: gradient of a second consumer of the output (ss_neuron_bwd_fork_x16): the two widened values are added in fp32 and the
    sum is used unrounded; `g_sum` = that sum narrowed once (dL/dskip of a stage with a fused skip add)."""
    T, N = h.shape
    gv = np.zeros(N, F32) if g_v_last is None else np.asarray(g_v_last, F32).copy()
    gx = np.empty((T, N), np.uint16)
    gsum = np.empty((T, N), np.uint16) if g2_bits is not None else None
    acc_k = 0.0
    c = F32(math.pi / 2.0 * alpha)
    ha = F32(alpha / 2.0)
    for t in range(T - 1, -1, -1):
        xh = h[t] - F32(v_th)
        z = (xh >= 0).astype(F32)
        g = widen(g_bits[t], dtype)
        if g2_bits is not None:
            g = g + widen(g2_bits[t], dtype)
            gsum[t] = narrow(g, dtype)
        if surrogate == 'ATan':
            u = xh * c
            sg = (F32(1) / (u * u + F32(1)) * ha) * g
        else:
            s = F32(1) / (F32(1) + np.exp(-(xh * F32(alpha)), dtype=F32))
            sg = ((g * (F32(1) - s)) * s) * F32(alpha)
        g_h = sg + gv * (F32(1) - z)
        if kind == 'IF':
            g_x, gv = g_h, g_h
        elif kind == 'LIF':
            g_x = g_h / F32(tau)
            gv = g_h - g_x
        else:
            g_x = g_h * F32(k)
            gv = g_h - g_x
            if t > 0:
                zp = ((h[t - 1] - F32(v_th)) >= 0).astype(F32)
                v_prev = (F32(1) - zp) * h[t - 1] + zp * F32(v_reset)
            else:
                v_prev = np.full(N, v_reset, F32) if v_init is None else np.asarray(v_init, F32)
            acc_k += float(np.sum((g_h * ((h[t] - v_prev) / F32(k))).astype(np.float64)))
        gx[t] = narrow(g_x * F32(scale), dtype)
    return dict(g_x=gx, g_v_init=gv, g_k=acc_k if kind == 'PLIF' else None, g_sum=gsum)
