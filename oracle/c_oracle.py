"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/ss_neuron_ref.c (numpy in / numpy out).

The functions mirror include/ss_neuron.h one for one (host pointers, no stream) so that GPU parity tests
can feed identical arguments to the HIP library and to this checker."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libss_oracle.so')

KIND = {'IF': 0, 'LIF': 1, 'PLIF': 2}
SURROGATE = {'ATan': 0, 'Sigmoid': 1}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'ss_neuron_ref.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'all'])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        fp, ull, i32, i64, f32 = C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_float
        _lib.ss_ref_neuron_fwd_f32.argtypes = [fp, fp, fp, fp, fp, fp, ull, i32, i64, f32, i32, f32, fp, f32, f32]
        _lib.ss_ref_neuron_bwd_f32.argtypes = [fp, fp, fp, fp, fp, fp, fp, i32, i64, f32, i32, f32, fp, f32, f32,
                                               i32, f32, i32]
        _lib.ss_ref_ipool_fwd_f32.argtypes = [fp, i64, i64, fp, fp, i32, i32, i64, f32, f32]
        _lib.ss_ref_ipool_bwd_f32.argtypes = [fp, fp, fp, i64, i64, fp, i32, i32, i64, f32]
        _lib.ss_ref_upconv1_fwd_f32.argtypes = [fp, fp, fp, fp, fp, i64, i32, i32, i32, i32, i32]
        _lib.ss_ref_upconv1_bwd_f32.argtypes = [fp, fp, fp, fp, fp, fp, i64, i32, i32, i32, i32, i32]
        _lib.ss_ref_upconv_cl_fwd_f32.argtypes = [fp, fp, fp, fp, fp, i64, i32, i32, i32, i32, i32, i32]
        _lib.ss_ref_upconv_cl_bwd_f32.argtypes = [fp, fp, fp, fp, fp, fp, i64, i32, i32, i32, i32, i32, i32]
        for f in (_lib.ss_ref_neuron_fwd_f32, _lib.ss_ref_neuron_bwd_f32, _lib.ss_ref_ipool_fwd_f32,
                  _lib.ss_ref_ipool_bwd_f32, _lib.ss_ref_upconv1_fwd_f32, _lib.ss_ref_upconv1_bwd_f32,
                  _lib.ss_ref_upconv_cl_fwd_f32, _lib.ss_ref_upconv_cl_bwd_f32):
            f.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def neuron_fwd(x_seq, *, kind='IF', scale=1.0, tau=2.0, k=None, v_th=1.0, v_reset=0.0, v_init=None, skip_seq=None,
               save_h=True, count=False):
    """x_seq: [T, N] float32.  Returns dict(out, h, v_last, nnz)."""
    x_seq = _f32(x_seq)
    T, N = x_seq.shape
    v_init, skip_seq = _f32(v_init), _f32(skip_seq)
    out = np.empty_like(x_seq)
    h = np.empty_like(x_seq) if save_h else None
    v_last = np.empty(N, np.float32)
    nnz = np.zeros(2, np.uint64) if count else None
    kk = None if k is None else np.asarray([k], np.float32)
    rc = lib().ss_ref_neuron_fwd_f32(_p(x_seq), _p(v_init), _p(skip_seq), _p(out), _p(h), _p(v_last), _p(nnz),
                                     T, N, scale, KIND[kind], tau, _p(kk), v_th, v_reset)
    if rc:
        raise ValueError(f'ss_ref_neuron_fwd_f32 -> {rc}')
    return dict(out=out, h=h, v_last=v_last, nnz=nnz)


def neuron_bwd(g_out_seq, h_seq, *, kind='IF', scale=1.0, tau=2.0, k=None, v_th=1.0, v_reset=0.0, v_init=None,
               g_v_last=None, surrogate='ATan', alpha=2.0, detach_reset=True):
    g_out_seq, h_seq = _f32(g_out_seq), _f32(h_seq)
    T, N = h_seq.shape
    v_init, g_v_last = _f32(v_init), _f32(g_v_last)
    g_x = np.empty_like(h_seq)
    g_v_init = np.empty(N, np.float32)
    g_k = np.zeros(1, np.float32) if kind == 'PLIF' else None
    kk = None if k is None else np.asarray([k], np.float32)
    rc = lib().ss_ref_neuron_bwd_f32(_p(g_out_seq), _p(g_v_last), _p(h_seq), _p(v_init), _p(g_x), _p(g_v_init),
                                     _p(g_k), T, N, scale, KIND[kind], tau, _p(kk), v_th, v_reset,
                                     SURROGATE[surrogate], alpha, int(detach_reset))
    if rc:
        raise ValueError(f'ss_ref_neuron_bwd_f32 -> {rc}')
    return dict(g_x=g_x, g_v_init=g_v_init, g_k=None if g_k is None else g_k[0])


def ipool_fwd(pd_seq, *, scale=1.0, v_reset=0.0, v_init=None):
    """pd_seq: [T, K, M] float32 (k=0 charged first).  Returns depth_seq [T, K, M]."""
    pd_seq = _f32(pd_seq)
    T, K, M = pd_seq.shape
    v_init = _f32(v_init)
    depth = np.empty_like(pd_seq)
    rc = lib().ss_ref_ipool_fwd_f32(_p(pd_seq), K * M, M, _p(v_init), _p(depth), T, K, M, scale, v_reset)
    if rc:
        raise ValueError(f'ss_ref_ipool_fwd_f32 -> {rc}')
    return depth


def ipool_bwd(g_depth_seq, *, scale=1.0, g_v_last=None):
    g_depth_seq = _f32(g_depth_seq)
    T, K, M = g_depth_seq.shape
    g_v_last = _f32(g_v_last)
    g_pd = np.empty_like(g_depth_seq)
    g_v_init = np.empty(M, np.float32)
    rc = lib().ss_ref_ipool_bwd_f32(_p(g_depth_seq), _p(g_v_last), _p(g_pd), K * M, M, _p(g_v_init), T, K, M, scale)
    if rc:
        raise ValueError(f'ss_ref_ipool_bwd_f32 -> {rc}')
    return dict(g_pd=g_pd, g_v_init=g_v_init)


def upconv1_fwd(P, src_y, src_x, bias, H, W):
    """P [NB, k*k, h, w] float32; src_y/src_x int32 tables; returns out [NB, H, W]."""
    P = _f32(P)
    NB, kk, h, w = P.shape
    k = int(round(kk ** 0.5))
    sy, sx = np.ascontiguousarray(src_y, np.int32), np.ascontiguousarray(src_x, np.int32)
    b = None if bias is None else np.asarray([bias], np.float32)
    out = np.empty((NB, H, W), np.float32)
    rc = lib().ss_ref_upconv1_fwd_f32(_p(P), _p(sy), _p(sx), _p(b), _p(out), NB, k, h, w, H, W)
    if rc:
        raise ValueError(f'ss_ref_upconv1_fwd_f32 -> {rc}')
    return out


def upconv1_bwd(g_out, y_lo, y_hi, x_lo, x_hi, k):
    g_out = _f32(g_out)
    NB, H, W = g_out.shape
    t = [np.ascontiguousarray(a, np.int32) for a in (y_lo, y_hi, x_lo, x_hi)]
    h, w = len(t[0]), len(t[2])
    g_P = np.empty((NB, k * k, h, w), np.float32)
    rc = lib().ss_ref_upconv1_bwd_f32(_p(g_out), _p(t[0]), _p(t[1]), _p(t[2]), _p(t[3]), _p(g_P), NB, k, h, w, H, W)
    if rc:
        raise ValueError(f'ss_ref_upconv1_bwd_f32 -> {rc}')
    return g_P


def upconv_cl_fwd(P, src_y, src_x, bias, k, C, H, W):
    """P [NB, h, w, k*k*C] float32 (channel = tap*C + c); returns out [NB, H, W, C]."""
    P = _f32(P)
    NB, h, w, _ = P.shape
    sy, sx = np.ascontiguousarray(src_y, np.int32), np.ascontiguousarray(src_x, np.int32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    out = np.empty((NB, H, W, C), np.float32)
    rc = lib().ss_ref_upconv_cl_fwd_f32(_p(P), _p(sy), _p(sx), _p(b), _p(out), NB, k, C, h, w, H, W)
    if rc:
        raise ValueError(f'ss_ref_upconv_cl_fwd_f32 -> {rc}')
    return out


def upconv_cl_bwd(g_out, y_lo, y_hi, x_lo, x_hi, k):
    g_out = _f32(g_out)
    NB, H, W, C = g_out.shape
    t = [np.ascontiguousarray(a, np.int32) for a in (y_lo, y_hi, x_lo, x_hi)]
    h, w = len(t[0]), len(t[2])
    g_P = np.empty((NB, h, w, k * k * C), np.float32)
    rc = lib().ss_ref_upconv_cl_bwd_f32(_p(g_out), _p(t[0]), _p(t[1]), _p(t[2]), _p(t[3]), _p(g_P), NB, k, C, h, w, H, W)
    if rc:
        raise ValueError(f'ss_ref_upconv_cl_bwd_f32 -> {rc}')
    return g_P
