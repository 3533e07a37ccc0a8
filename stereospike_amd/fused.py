"""torch.autograd front-end of the C-ABI (include/ss_neuron.h).

`fused_neuron(x_seq, ...)` replaces, for one layer and all T steps at once, what the reference spreads over
MultiplyBy.forward (/root/reference/network/blocks.py:106-107), the spikingjelly single-step node
(call sites SNN_models.py:78...128, 266...316, blocks.py:150,157), the skip / SEW adds (SNN_models.py:171-186,
blocks.py:171) and, in backward, the autograd chain through surrogate.ATan / surrogate.Sigmoid.

`ipool(pd_seq, ...)` replaces the shared I-neuron read-out (SNN_models.py:150,172-188).

Everything here needs the HIP library and HIP tensors; there is no eager fallback.
"""
import os as _os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from .config import current as _cfg, guard_module as _guard_module, note as _note, site as _site


@dataclass(frozen=True)
class NeuronCfg:
    kind: int                    # _lib.KIND_*
    scale: float = 1.0           # MultiplyBy gain folded into the kernel
    tau: float = 2.0             # LIF only
    v_th: float = 1.0
    v_reset: float = 0.0
    surrogate: int = _lib.SG_SIGMOID
    alpha: float = 4.0
    detach_reset: bool = True


class KernelTimer:
    """Optional HIP-event timing of every fused launch (bench.py's roofline leg).  Events are recorded on the
    stream the kernel is launched on (torch's current stream)."""

    def __init__(self):
        self.enabled = False
        self.records = []        # (tag, algorithmic_bytes, updates, start_event, end_event)

    def start(self):
        if not self.enabled:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, e0, tag, nbytes, updates):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((tag, nbytes, updates, e0, e1))

    def summary(self, by_shape=False):
        """{tag: dict(launches, bytes, updates, ms)} — call after torch.cuda.synchronize().  by_shape=True keys the
        groups by (tag, updates per launch) so one launch shape can be priced against the roofline on its own."""
        out = {}
        for tag, nbytes, updates, e0, e1 in self.records:
            d = out.setdefault((tag, updates) if by_shape else tag, dict(launches=0, bytes=0, updates=0, ms=0.0))
            d['launches'] += 1
            d['bytes'] += nbytes
            d['updates'] += updates
            d['ms'] += e0.elapsed_time(e1)
        return out

    def clear(self):
        self.records = []


TIMER = KernelTimer()

# fixed tuning constants (EngineConfig fields until round 5; none of the shipped configurations changed them)
_P_CHUNK_BYTES = 96 << 20             # NCHW projected form: P produced and consumed in chunks of frames that stay in the 256 MiB Infinity Cache
_P_MAX_BYTES_CL = 16 << 30            # NHWC projected form: frames are chunked only when P would exceed this
_EXACT_SPLIT_MIN_K = 128              # exact bf16x3 projection GEMM from this C_in on (below it the fp32 GEMM is as fast)
_WGRAD_SPLIT_ROWS = 8192              # library weight-gradient GEMM: split-K slice length
_SPIKE_CONV_WGRAD_SPLIT = 8           # encoder / bottleneck weight-gradient GEMM: split-K slices
_SPIKE_CONV_MIN_CIN = 128             # im2col + GEMM form of a conv on spikes from this C_in on


# Training keeps the layer INPUT (the conv output, which autograd would otherwise free) instead of a separately written h_seq and
# recomputes h inside the backward kernel: the forward launch writes 8 instead of 12 B/update, the backward reads the same 12.
# With 16-bit activations: 4 instead of 8 B/update forward, 6 instead of 8 backward.  Applies to the compile-time time-step counts
# (ss_neuron_bwd_rc_supported); any other T saves h.


# 2-bit packed spike tensors on the edges whose consumers can read them: the im2col of the exact-split convs
# (conv3, conv4, the bottleneck) and the skip / SEW-identity operand of a neuron launch.  A layer whose consumers are all of that kind
# writes NO dense output (forward 4.25 instead of 8 B/update); its autograd output is then a zero-strided "anchor" of the logical shape
# (4 bytes of storage) that only carries the graph edge — the data travels in the packed tensor next to it.


def spike_anchor(shape, dtype, device):
    return torch.zeros(1, dtype=dtype, device=device).as_strided(tuple(shape), (0,) * len(shape))


def x16_mode(device=None):
    """The activation dtype (torch.float16 | torch.bfloat16) when the caller runs under 16-bit torch.autocast AND the 16-bit activation modes are routed
    to the engine's own single-term kernels (EngineConfig.X16_OWN_KERNELS, round 5: include/ss_neuron.h "ABI 9"); None otherwise — fp32 mode, or the
    modes' round-2 .. 4 path (synapses = MIOpen convolutions under autocast)."""
    if device is not None and torch.device(device).type != 'cuda':
        return None
    if not (torch.is_autocast_enabled('cuda') and _cfg().X16_OWN_KERNELS):
        return None
    adt = torch.get_autocast_dtype('cuda')
    return adt if adt in (torch.float16, torch.bfloat16) else None


def unpack_dense(packed, shape, dtype=torch.float32):
    """Packed int32 [T, N/16] -> dense tensor of `shape` (for a consumer that cannot read the packed form)."""
    out = torch.empty(tuple(shape), dtype=dtype, device=packed.device)
    _lib.unpack_spikes(packed.contiguous(), out, out.numel())
    return out


# Low-rank gradient of a prediction head (3 x 3 taps, ONE output channel): d loss / d input = g_P [rows, 9] @ W2 [9, C].  Instead of running
# that GEMM and writing the C-channel result (which the neuron backward of the stage — and, through the fused skip add, of the full-resolution
# encoder layer — would then read at 4 B/update), the head's backward hands the PAIR to the consumer, and ss_neuron_bwd_fork_lr_f32 forms the
# gradient in registers from 36 / C B/update.
#
# How the pair travels (no module-level state: VERDICT r02 weak #9): ONE flat fp32 buffer [4 NaNs | g_P (rows x 9) | pad | W2 (9 x C)] is
# allocated per head and backward; the adjoint kernel writes g_P straight into it.  The gradient autograd carries is a ZERO-STRIDE view of the
# buffer's first element with the logical shape of the dense gradient — the pair lives exactly as long as autograd keeps that gradient, any
# view of it (fork handles, flatten) still shares the storage, two networks or retain_graph need no bookkeeping.  The consumer recognises it by
# layout alone (all strides 0, offset 0, storage longer than the 4-element header — an ordinary expanded scalar gradient has a 1-element
# storage) and checks that the storage length is the one its own shape implies; a mismatch RAISES.  A consumer that knows nothing about pairs
# and materialises the view sees NaNs everywhere.
_LR_HDR = 4                                  # floats before g_P (keeps g_P and W2 16-byte aligned)
_LR_RANK = 9


def _lr_layout(rows, C, rank=_LR_RANK):
    """(offset of g_P, offset of W2, total floats) of the pair buffer."""
    o_w = _LR_HDR + (rows * rank + 3) // 4 * 4
    return _LR_HDR, o_w, o_w + rank * C


def lowrank_buffer(shape, device, rank=_LR_RANK, dtype=torch.float32):
    """shape [..., C] of the dense gradient the pair stands for -> (anchor, g_P [rows, rank], W2 [rank, C]): the zero-stride gradient to
    return to autograd and the two views the producer fills (g_P by the adjoint kernel, W2 by a copy of the head's weight).
    dtype: the dtype of the gradient the anchor stands in for (autograd casts a gradient whose dtype differs from its output's — which would materialise
    the anchor — so in the 16-bit activation modes the anchor is a fp16 / bf16 VIEW of element 0 of the same fp32 buffer; the pair itself stays fp32).
    The header words are 0x7FC07FC0: a NaN as fp32 and, read as two 16-bit halves, as fp16 / bf16."""
    C = int(shape[-1])
    rows = 1
    for d in shape[:-1]:
        rows *= int(d)
    o_p, o_w, n = _lr_layout(rows, C, rank)
    buf = torch.empty(n, dtype=torch.float32, device=device)
    buf[:_LR_HDR].view(torch.int32).fill_(0x7FC07FC0)
    if dtype == torch.float32:
        anchor = buf.as_strided(tuple(shape), (0,) * len(shape))
    else:
        anchor = torch.empty(0, dtype=dtype, device=device).set_(buf.untyped_storage(), 0, tuple(shape), (0,) * len(shape))
    return anchor, buf[o_p:o_p + rows * rank].view(rows, rank), buf[o_w:o_w + rank * C].view(rank, C)


def lowrank_anchor(shape, lr_p, lr_w, dtype=torch.float32):
    """The anchor of an existing pair (copies both into a fresh buffer; the product path fills a lowrank_buffer in place instead)."""
    a, p, w = lowrank_buffer(shape, lr_p.device, lr_w.shape[0], dtype)
    p.copy_(lr_p)
    w.copy_(lr_w)
    return a


def lowrank_of(g, rank=_LR_RANK):
    """(lr_p [rows, rank], lr_w [rank, C]) when g is (a view of) a low-rank anchor, else None.

    A zero-stride gradient whose storage does not have the length the pair of ITS shape would have is not taken for a pair (e.g. the
    gradient of `(torch.stack(terms) * w).sum()` is an expanded view of element 0 of a len(terms)-float storage) and is handled as the dense
    gradient it is — unless its first element carries the pair buffer's NaN header, i.e. a real pair reached a layer it was not made for: that
    RAISES (one host read, on this rare path only and never inside a stream capture; without it the NaN header still surfaces as NaN gradients)."""
    if g is None or g.dtype not in (torch.float32, torch.float16, torch.bfloat16) or g.dim() == 0 or any(g.stride()) or g.storage_offset() != 0:
        return None
    st = g.untyped_storage()
    n = st.nbytes() // 4
    if n <= _LR_HDR:
        return None                          # an ordinary expanded scalar (e.g. the gradient of a sum)
    C = int(g.shape[-1])
    rows = g.numel() // C
    o_p, o_w, need = _lr_layout(rows, C, rank)
    if n != need:
        capturing = g.is_cuda and torch.cuda.is_current_stream_capturing()
        if not capturing and bool(torch.isnan(g.reshape(-1)[:1]).item()):
            raise _lib.SSNeuronError(f'zero-stride gradient of shape {tuple(g.shape)} over a {n}-float storage with the NaN header of a low-rank pair, but the '
                                     f'pair of this shape has {need} floats — a prediction head\'s gradient pair reached a layer it does not fit')
        return None
    flat = torch.empty(0, dtype=torch.float32, device=g.device).set_(st, 0, (n,), (1,))
    return flat[o_p:o_p + rows * rank].view(rows, rank), flat[o_w:o_w + rank * C].view(rank, C)


def lowrank_dense(lr, shape, dtype=torch.float32):
    """The dense gradient a low-rank pair stands for (consumers without the fused form)."""
    g = torch.mm(lr[0], lr[1]).view(tuple(shape))
    return g if dtype == torch.float32 else g.to(dtype)


class _FusedNeuron(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_seq, v_init, skip_seq, k, cfg: NeuronCfg, nnz, fork=False, pack=0, skip_packed=None, want_v=True):
        ctx.ecfg, ctx.site = _cfg(), _site()          # the engine configuration and plan site of THIS forward: the backward dispatches from them
        """pack: 0 dense output; 1 dense + packed; 2 packed only (the returned out_seq is an anchor).  skip_packed: the skip operand as a
        packed tensor (skip_seq then only carries the autograd edge and may be an anchor)."""
        T = x_seq.shape[0]
        N = x_seq.numel() // T
        x_seq = x_seq.contiguous()
        if v_init is not None:
            v_init = v_init.contiguous()
        need_grad = any(ctx.needs_input_grad[:4])
        half = x_seq.dtype in (torch.float16, torch.bfloat16)       # 16-bit activations, fp32 membrane (configs 2 / 5)
        recompute = need_grad and _cfg().RECOMPUTE_H and _lib.neuron_bwd_rc_supported(T)
        # the packed kernel form: compile-time T, whole words, no saved h (fp32 activations; 16-bit ones since ABI 9)
        can_pk = N % 16 == 0 and _lib.neuron_bwd_rc_supported(T) and (recompute or not need_grad)
        if skip_packed is not None and not can_pk:
            skip_seq, skip_packed = unpack_dense(skip_packed, x_seq.shape, x_seq.dtype), None
        if not can_pk:
            pack = 0
        has_skip = skip_seq is not None or skip_packed is not None
        if skip_packed is not None:
            skip_dense = None
        else:
            skip_dense = None if skip_seq is None else skip_seq.contiguous()
            if skip_dense is not None and skip_dense.dtype != x_seq.dtype:
                skip_dense = skip_dense.to(x_seq.dtype)              # spikes are small integers: exact in every format
        skip_seq = skip_dense
        out_seq = spike_anchor(x_seq.shape, x_seq.dtype, x_seq.device) if pack == 2 else torch.empty_like(x_seq)
        packed = torch.empty((T, N // 16), dtype=torch.int32, device=x_seq.device) if pack else None
        h_seq = torch.empty(x_seq.shape, dtype=torch.float32, device=x_seq.device) if (need_grad and not recompute) else None
        # want_v False (a training pass whose membrane nobody reads before the next reset: EngineConfig.LAZY_MEMBRANE): the packed kernel forms then do not
        # write v_last (4 B per neuron and layer: 10 - 36 % of what the forward moves) and None is returned; the other forms always write it
        lazy_v = (not want_v) and bool(pack or skip_packed is not None)
        v_last = None if lazy_v else torch.empty(x_seq.shape[1:], dtype=torch.float32, device=x_seq.device)
        if v_init is not None and v_init.dtype != torch.float32:
            v_init = v_init.float()
        e0 = TIMER.start()
        if nnz is not None or pack or skip_packed is not None:
            # firing-rate counters: per-workgroup partials + one fixed second pass (full grid, no same-address atomics)
            cnt_ws = torch.empty(_lib.cnt_ws_words(N), dtype=torch.int32, device=x_seq.device) if nnz is not None else None
            _lib.neuron_fwd_ex(x_seq, v_init, skip_seq, skip_packed, None if pack == 2 else out_seq, packed, h_seq, v_last, nnz, cnt_ws, T, N,
                               cfg.scale, cfg.kind, cfg.tau, k, cfg.v_th, cfg.v_reset)
        else:
            (_lib.neuron_fwd_x16 if half else _lib.neuron_fwd)(x_seq, v_init, skip_seq, out_seq, h_seq, v_last, nnz, T, N,
                                                              cfg.scale, cfg.kind, cfg.tau, k, cfg.v_th, cfg.v_reset)
        es = 2 if half else 4
        per = es * (1 + (0 if pack == 2 else 1)) + (0.25 if pack else 0) + (0.25 if skip_packed is not None else (es if skip_seq is not None else 0)) \
            + (4 if h_seq is not None else 0)
        tag = (('neuron_fwd_train' if need_grad else 'neuron_fwd_infer') + ('+skip' if has_skip else '') + ('+packed' if pack == 2 else ('+pkcopy' if pack == 1 else ''))
               + ('+h' if h_seq is not None else ''))            # '+h': the saved-h form (run-time T / RECOMPUTE_H off), never the benchmarked kernels
        TIMER.stop(e0, tag, int(per * T * N), T * N)
        _note('neuron_fwd', tag + ('+x16' if half else ''))
        ctx.cfg = cfg
        ctx.T, ctx.N = T, N
        ctx.has_vinit = v_init is not None
        ctx.has_skip = has_skip
        ctx.io_dtype = x_seq.dtype
        ctx.set_materialize_grads(False)
        ctx.recompute = recompute
        if recompute:
            # the backward kernel rebuilds h_0..h_{T-1} in registers from x_seq and v_init (same arithmetic, bit-identical)
            ctx.save_for_backward(x_seq, v_init, k)
        elif need_grad:
            # h_seq is all the backward needs (z_t, v_{t-1} are recomputed from it); the OUTPUT is not saved because
            # the reference mutates it in place (blocks.py:171).  v_init only feeds the PLIF dL/dk term at t = 0.
            keep_v = v_init if (cfg.kind == _lib.KIND_PLIF and ctx.needs_input_grad[3]) else None
            ctx.save_for_backward(h_seq, keep_v, k)
        # fork: a second handle on the same memory for the second consumer: autograd then delivers the two gradients separately and the
        # backward kernel adds them on load (ss_neuron_bwd_fork_f32) instead of a separate 12 B/element accumulation pass
        return out_seq, v_last, (out_seq.view_as(out_seq) if fork else None), packed

    @staticmethod
    def backward(ctx, g_out_seq, g_v_last, g_out2_seq=None, g_packed=None):
        h_seq, v_init, k = ctx.saved_tensors          # h_seq is x_seq (activation dtype) when ctx.recompute
        cfg, T, N = ctx.cfg, ctx.T, ctx.N
        half = ctx.io_dtype in (torch.float16, torch.bfloat16)
        lr = lr_anchor = None
        l1, l2 = lowrank_of(g_out_seq), lowrank_of(g_out2_seq)    # a prediction head's gradient as a low-rank pair (see lowrank_buffer)
        if l1 is not None or l2 is not None:
            if l1 is not None and l2 is not None:      # two pairs: one stays low-rank
                g_out_seq, l1 = lowrank_dense(l1, h_seq.shape, ctx.io_dtype), None
            if l1 is not None:
                g_out_seq, g_out2_seq, l2 = g_out2_seq, g_out_seq, l1
            if l2 is not None:
                C = int(l2[1].shape[1])
                lr_ok = (_lib.neuron_bwd_fork_lr_x16_supported if half else _lib.neuron_bwd_fork_lr_supported)(T, N, C, l2[1].shape[0])
                if ctx.recompute and h_seq.shape[-1] == C and lr_ok:
                    lr, lr_anchor, g_out2_seq = l2, g_out2_seq, None
                else:
                    g_out2_seq = lowrank_dense(l2, h_seq.shape, ctx.io_dtype)
        if g_out_seq is None and lr is None:
            g_out_seq, g_out2_seq = g_out2_seq, None
        if g_out_seq is None and lr is None:
            g_out_seq = torch.zeros(h_seq.shape, dtype=ctx.io_dtype, device=h_seq.device)
        if g_out_seq is not None:
            g_out_seq = g_out_seq.to(ctx.io_dtype).contiguous()
        fuse2 = g_out2_seq is not None and ctx.recompute
        if g_out2_seq is not None:
            g_out2_seq = g_out2_seq.to(ctx.io_dtype).contiguous()
            if not fuse2:
                g_out_seq, g_out2_seq = g_out_seq + g_out2_seq, None
        if g_v_last is not None:
            g_v_last = g_v_last.float().contiguous()
        g_x_seq = torch.empty(h_seq.shape, dtype=ctx.io_dtype, device=h_seq.device)
        want_gv = ctx.has_vinit and ctx.needs_input_grad[1]
        g_v_init = torch.empty(h_seq.shape[1:], dtype=torch.float32, device=h_seq.device) if want_gv else None
        want_gk = cfg.kind == _lib.KIND_PLIF and ctx.needs_input_grad[3]
        g_k = g_k_ws = None
        if want_gk:
            g_k = torch.empty((), dtype=torch.float32, device=h_seq.device)
            g_k_ws = torch.empty(_lib.gk_ws_floats(), dtype=torch.float32, device=h_seq.device)
        e0 = TIMER.start()
        want_gskip = ctx.has_skip and ctx.needs_input_grad[2]
        g_sum = torch.empty_like(g_out_seq) if ((fuse2 or lr is not None) and want_gskip and g_out_seq is not None) else None
        if lr is not None:
            (_lib.neuron_bwd_fork_lr_x16 if half else _lib.neuron_bwd_fork_lr)(g_out_seq, lr[0], lr[1], g_sum, g_v_last, h_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws,
                                    T, N, cfg.scale, cfg.kind, cfg.tau, k, cfg.v_th, cfg.v_reset, cfg.surrogate, cfg.alpha, cfg.detach_reset)
            es = 2 if half else 4
            per = 2 * es + 4 * lr[1].shape[0] / lr[1].shape[1] + (es if g_out_seq is not None else 0) + (es if g_sum is not None else 0)
            tag = 'neuron_bwd+lr' + ('+sum' if g_sum is not None else '') if g_out_seq is not None else 'neuron_bwd+lronly'
            TIMER.stop(e0, tag, int(per * T * N), T * N)
            _note('neuron_bwd', tag + ('+x16' if half else ''), ctx.site)
            # dL/dskip: the dense sum when there was a dense first gradient, else the low-rank pair itself travels on (identity)
            g_skip = (g_sum if g_out_seq is not None else lr_anchor.view(h_seq.shape)) if want_gskip else None
            return g_x_seq, g_v_init, g_skip, g_k, None, None, None, None, None, None
        if fuse2 and half:
            _lib.neuron_bwd_fork_x16(g_out_seq, g_out2_seq, g_sum, g_v_last, h_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws,
                                     T, N, cfg.scale, cfg.kind, cfg.tau, k, cfg.v_th, cfg.v_reset, cfg.surrogate, cfg.alpha, cfg.detach_reset)
        elif fuse2:
            _lib.neuron_bwd_fork(g_out_seq, g_out2_seq, g_sum, g_v_last, None, h_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws,
                                 T, N, cfg.scale, cfg.kind, cfg.tau, k, cfg.v_th, cfg.v_reset, cfg.surrogate, cfg.alpha, cfg.detach_reset)
        else:
            if ctx.recompute:
                bwd = _lib.neuron_bwd_rc_x16 if half else _lib.neuron_bwd_rc
            else:
                bwd = _lib.neuron_bwd_x16 if half else _lib.neuron_bwd
            bwd(g_out_seq, g_v_last, h_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws,
                T, N, cfg.scale, cfg.kind, cfg.tau, k, cfg.v_th, cfg.v_reset, cfg.surrogate, cfg.alpha, cfg.detach_reset)
        tag = (('neuron_bwd+fork+sum' if g_sum is not None else 'neuron_bwd+fork') if fuse2 else 'neuron_bwd') + ('' if ctx.recompute else '+savedh')
        TIMER.stop(e0, tag, (((6 if ctx.recompute else 8) + ((2 + (2 if g_sum is not None else 0)) if fuse2 else 0)) if half else ((20 if g_sum is not None else 16) if fuse2 else 12)) * T * N, T * N)
        _note('neuron_bwd', tag + ('+x16' if half else ''), ctx.site)
        g_skip = (g_sum if fuse2 else g_out_seq) if want_gskip else None                 # identity
        return g_x_seq, g_v_init, g_skip, g_k, None, None, None, None, None, None


def fused_neuron(x_seq: torch.Tensor, cfg: NeuronCfg, v_init: Optional[torch.Tensor] = None,
                 skip_seq: Optional[torch.Tensor] = None, k: Optional[torch.Tensor] = None,
                 nnz: Optional[torch.Tensor] = None, fork: bool = False, pack: int = 0, skip_packed: Optional[torch.Tensor] = None,
                 want_v: bool = True):
    """x_seq: [T, ...] conv output (before the gain).  Returns (out_seq [T, ...], v_last [...]) — with fork=True a third value: a
    second handle on out_seq for its second consumer (the two gradients are then added inside the backward kernel); with pack != 0 a
    last value: the 2-bit packed output (int32 [T, N/16]; None when the packed kernel form does not apply — the output is then dense).
    want_v=False: v_last may come back as None (packed kernel forms: the membrane after step T is not written; membrane_after() recomputes it)."""
    if cfg.kind == _lib.KIND_PLIF and k is None:
        raise _lib.SSNeuronError('PLIF needs k = sigmoid(w) as a 0-dim HIP tensor')
    out, v_last, out2, packed = _FusedNeuron.apply(x_seq, v_init, skip_seq, k, cfg, nnz, fork, pack, skip_packed, want_v)
    res = (out, v_last) + ((out2,) if fork else ())
    return res + ((packed,) if pack else ())


def membrane_after(x_seq: torch.Tensor, cfg: NeuronCfg, v_init: Optional[torch.Tensor] = None, k: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The membrane after the T steps of x_seq, when the training pass did not write it (want_v=False): one more launch of the packed forward form
    on the layer input the backward keeps anyway, its spike output thrown away.  No autograd history (what `net.detach()` leaves behind)."""
    with torch.no_grad():
        T = x_seq.shape[0]
        N = x_seq.numel() // T
        x = x_seq.detach().contiguous()
        v_last = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device)
        scratch = torch.empty((T, N // 16), dtype=torch.int32, device=x.device)
        vi = None if v_init is None else v_init.detach().float().contiguous()
        _lib.neuron_fwd_ex(x, vi, None, None, None, scratch, None, v_last, None, None, T, N, cfg.scale, cfg.kind, cfg.tau,
                           None if k is None else k.detach(), cfg.v_th, cfg.v_reset)
        _note('neuron_fwd', 'membrane_after')
    return v_last


class _IPool(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, pd_seq, v_init, scale, v_reset):
        # pd_seq: [K, T, M...] (k = 0 is the head charged first) — i.e. torch.stack of the K head outputs
        pd_seq = pd_seq.contiguous()
        K, T = pd_seq.shape[0], pd_seq.shape[1]
        M = pd_seq.numel() // (K * T)
        if v_init is not None:
            v_init = v_init.contiguous()
        depth_seq = torch.empty((T, K) + tuple(pd_seq.shape[2:]), dtype=pd_seq.dtype, device=pd_seq.device)
        _lib.ipool_fwd(pd_seq, M, T * M, v_init, depth_seq, T, K, M, scale, v_reset)
        ctx.dims = (T, K, M)
        ctx.scale = scale
        ctx.shape = pd_seq.shape
        ctx.has_vinit = v_init is not None
        return depth_seq

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g_depth_seq):
        T, K, M = ctx.dims
        g_depth_seq = g_depth_seq.float().contiguous()
        g_pd = torch.empty(ctx.shape, dtype=g_depth_seq.dtype, device=g_depth_seq.device)
        want_gv = ctx.has_vinit and ctx.needs_input_grad[1]
        g_v_init = torch.empty(ctx.shape[2:], dtype=g_depth_seq.dtype, device=g_depth_seq.device) if want_gv else None
        _lib.ipool_bwd(g_depth_seq, None, g_pd, M, T * M, g_v_init, T, K, M, ctx.scale)
        return g_pd, g_v_init, None, None


def ipool(pd_seq: torch.Tensor, scale: float, v_reset: float, v_init: Optional[torch.Tensor] = None):
    """pd_seq [K, T, ...] -> depth_seq [T, K, ...]: membrane snapshots after every charge (t outer, k inner)."""
    return _IPool.apply(pd_seq, v_init, float(scale), float(v_reset))


# ----------------------------------------------------------------------------------------------------------
# per-scale loss terms (the step after the path: SURVEY.md §8(f) rank 4)
# ----------------------------------------------------------------------------------------------------------
_fwd32 = torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)   # under autocast: inputs -> fp32, autocast off inside
_bwd32 = torch.amp.custom_bwd(device_type='cuda')



class _ScaleLossTerms(torch.autograd.Function):
    """pred, gt [B,1,H,W] -> fp32 [3] = (ScaleInvariant, GradientMatching, MeanDepthError) of the batch, from one statistics launch
    (+ a fixed-order fp64 second pass); backward = one stencil launch.  /root/reference/network/loss.py:7-24, :44-75;
    network/metrics.py:83-95.  Nothing but pred / gt is kept for backward (the 5 sums are recomputed into the gradient)."""

    @staticmethod
    @_fwd32
    def forward(ctx, pred, gt):
        pred, gt = pred.contiguous(), gt.contiguous()
        H, W = pred.shape[-2:]
        B = pred.numel() // (H * W)
        ws = torch.empty(_lib.loss_ws_doubles(), dtype=torch.float64, device=pred.device)   # 2.6 MB from the stream-aware caching allocator
        sums = torch.empty(5, dtype=torch.float64, device=pred.device)
        _lib.loss_stats(pred, gt, sums, ws, B, H, W)
        n = sums[0]
        mean = sums[1:] / n                                   # E r, E r^2, E(|gx|+|gy|), E|r|   (fp64, 4 elements)
        terms = torch.stack((mean[1] - mean[0] * mean[0], mean[2], mean[3])).float()
        ctx.save_for_backward(pred, gt, sums)
        ctx.dims = (B, H, W)
        return terms

    @staticmethod
    @_bwd32
    def backward(ctx, g_terms):
        pred, gt, sums = ctx.saved_tensors
        B, H, W = ctx.dims
        coef = g_terms[:2].float().contiguous()               # d loss / d (SI, GM); the MDE term is a metric (no gradient)
        g_pred = torch.empty_like(pred)
        _lib.loss_grad(pred, gt, sums, coef, g_pred, B, H, W)
        return g_pred, None


def scale_loss_terms(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """fp32 [3]: ScaleInvariant_Loss(pred, gt), GradientMatching_Loss(pred, gt), MeanDepthError(pred, gt) — differentiable w.r.t.
    pred through the first two."""
    if pred.shape != gt.shape:
        raise ValueError(f'pred {tuple(pred.shape)} and gt {tuple(gt.shape)} must have the same shape')
    return _ScaleLossTerms.apply(pred, gt)


# ----------------------------------------------------------------------------------------------------------
# predict_depth head synapse without the materialised up-sampled tensor
# ----------------------------------------------------------------------------------------------------------
def nearest_tables(in_size: int, out_size: int):
    """Source index of every up-sampled position, computed by torch's own UpsamplingNearest2d on the CPU (so it is the
    reference's mapping bit for bit, including its float rounding), and the inverse ranges [lo, hi) per source index."""
    src = torch.nn.functional.interpolate(torch.arange(in_size, dtype=torch.float32).view(1, 1, in_size, 1),
                                          size=(out_size, 1), mode='nearest').view(-1).to(torch.int32)
    assert bool((src[1:] >= src[:-1]).all())
    lo = torch.searchsorted(src, torch.arange(in_size, dtype=torch.int32), right=False).to(torch.int32)
    hi = torch.searchsorted(src, torch.arange(in_size, dtype=torch.int32), right=True).to(torch.int32)
    return src, lo, hi


# how the low-res per-tap projection is evaluated
#   'chunked' : frame-chunked GEMM + gather so the projection tensor P stays resident in the 256 MiB Infinity Cache
#               between its producer and its consumer (default; see _UpConvProjected)
#   'conv'    : one MIOpen 1x1 convolution over the whole batch, P materialised in HBM
#   'matmul'  : one rocBLAS batched GEMM over the whole batch, P materialised in HBM
#   'auto'    : 'chunked' when a chunk holds >= 4 frames (small maps: deconv4, deconv3, the heads — few, large GEMMs),
#               otherwise 'conv' (deconv2, deconv1: per-frame GEMMs with K = C_in <= 128 are launch/latency-bound;
#               measured on MI355X, profiles/r01/upconv_variants.log)


class _UpConv1(torch.autograd.Function):
    """The gather alone (P given): used by the 'conv' / 'matmul' variants, autograd reaches W through P."""

    @staticmethod
    @_fwd32
    def forward(ctx, P, bias, tables, k, H, W):
        P = P.contiguous()
        NB, kk, h, w = P.shape
        assert kk == k * k
        src_y, y_lo, y_hi, src_x, x_lo, x_hi = tables
        out = torch.empty((NB, 1, H, W), dtype=P.dtype, device=P.device)
        e0 = TIMER.start()
        _lib.upconv1_fwd(P, src_y, src_x, bias, out, NB, k, h, w, H, W)
        TIMER.stop(e0, 'upconv1_fwd', 4 * (P.numel() + out.numel()), out.numel())
        ctx.tables = tables
        ctx.dims = (NB, k, h, w, H, W)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @_bwd32
    def backward(ctx, g_out):
        NB, k, h, w, H, W = ctx.dims
        _, y_lo, y_hi, _, x_lo, x_hi = ctx.tables
        g_out = g_out.float().contiguous()
        g_P = torch.empty((NB, k * k, h, w), dtype=g_out.dtype, device=g_out.device)
        e0 = TIMER.start()
        _lib.upconv1_bwd(g_out, y_lo, y_hi, x_lo, x_hi, g_P, NB, k, h, w, H, W)
        TIMER.stop(e0, 'upconv1_bwd', 4 * (g_P.numel() + g_out.numel()), g_out.numel())
        g_bias = g_out.sum().view(1) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        return g_P, g_bias, None, None, None, None


class _UpConvProjected(torch.autograd.Function):
    """Projection GEMM + gather, walked over chunks of frames.  P (C_out*k*k channels at low resolution — 6x the size of
    the stage's output for deconv1) is produced and consumed chunk by chunk, each chunk <= P_CHUNK_BYTES, so it lives in
    the Infinity Cache and never makes the round trip to HBM; the same holds for g_P in backward, which feeds the two
    GEMMs (dgrad, wgrad) right after the gather adjoint writes it."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, weight, bias, tables, k, H, W):
        x = x.contiguous()
        NB, Cin, h, w = x.shape
        Cout, kk, hw = weight.shape[0], k * k, h * w
        W2 = weight.permute(0, 2, 3, 1).reshape(Cout * kk, Cin).contiguous()        # rows ordered (co, ky, kx)
        src_y, _, _, src_x, _, _ = tables
        out = torch.empty((NB, Cout, H, W), dtype=x.dtype, device=x.device)
        n = max(1, min(NB, _P_CHUNK_BYTES // (Cout * kk * hw * 4)))
        one_bias = bias if (bias is not None and Cout == 1) else None
        e0 = TIMER.start()
        for c0 in range(0, NB, n):
            c1 = min(NB, c0 + n)
            P = torch.matmul(W2, x[c0:c1].view(c1 - c0, Cin, hw))                     # [n, Cout*kk, hw]
            _lib.upconv1_fwd(P, src_y, src_x, one_bias, out[c0:c1], (c1 - c0) * Cout, k, h, w, H, W)
        if bias is not None and Cout != 1:
            out += bias.view(1, Cout, 1, 1)
        TIMER.stop(e0, 'upconv_projected_fwd', 4 * (x.numel() + out.numel()), out.numel())
        ctx.save_for_backward(x, W2)
        ctx.tables, ctx.k, ctx.n = tables, k, n
        ctx.wshape = weight.shape
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @_bwd32
    def backward(ctx, g_out):
        x, W2 = ctx.saved_tensors
        k, n = ctx.k, ctx.n
        _, y_lo, y_hi, _, x_lo, x_hi = ctx.tables
        NB, Cin, h, w = x.shape
        Cout, kk, hw = ctx.wshape[0], k * k, h * w
        H, W = g_out.shape[-2:]
        g_out = g_out.float().contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_x = torch.empty_like(x) if need_x else None
        g_W2 = torch.zeros_like(W2) if need_w else None
        W2t = W2.t().contiguous()
        e0 = TIMER.start()
        for c0 in range(0, NB, n):
            c1 = min(NB, c0 + n)
            m = c1 - c0
            g_P = torch.empty((m, Cout * kk, hw), dtype=x.dtype, device=x.device)
            _lib.upconv1_bwd(g_out[c0:c1], y_lo, y_hi, x_lo, x_hi, g_P, m * Cout, k, h, w, H, W)
            if need_x:
                torch.matmul(W2t, g_P, out=g_x[c0:c1].view(m, Cin, hw))               # dgrad
            if need_w:
                g_W2 += torch.bmm(g_P, x[c0:c1].view(m, Cin, hw).transpose(1, 2)).sum(0)   # wgrad
        TIMER.stop(e0, 'upconv_projected_bwd', 4 * (x.numel() + g_out.numel()), g_out.numel())
        g_w = g_W2.view(Cout, k, k, Cin).permute(0, 3, 1, 2).contiguous() if need_w else None
        g_b = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g_b = g_out.sum((0, 2, 3))
        return g_x, g_w, g_b, None, None, None, None


def upconv_projected(x: torch.Tensor, weight: torch.Tensor, bias, tables, k: int, H: int, W: int):
    """NNConvUpsampling(C_in -> C_out, k, up_size=(H, W)) applied to x [NB, C_in, h, w] (blocks.py:124-132) WITHOUT the
    up-sampled tensor.  Nearest-neighbour resize only replicates pixels, so the channel contraction commutes with it:
      1. per-tap projections at LOW resolution  P[nb][co][tap] = W[co, :, tap] . x[nb]   — a plain GEMM
         (C_in -> C_out*k*k on PyTorch-ROCm; (h*w)/(H*W) ~ 1/4.3 of the direct conv's MACs for the decoder stages,
         1/63 for predict_depth4);
      2. out[nb][co][y][x] = sum_taps P[nb][co][tap][src_y[y+ky]][src_x[x+kx]]  — the fused gather kernel
         ss_upconv1_fwd_f32 with (nb, co) as its image index.
    Same value as the reference's two-op form up to fp32 summation order (channels first, taps second).  weight is the
    Conv2d weight [C_out, C_in, k, k]."""
    NB, Cin, h, w = x.shape
    Cout = weight.shape[0]
    impl = 'auto'
    if impl == 'auto':
        impl = 'chunked' if _P_CHUNK_BYTES // (Cout * k * k * h * w * 4) >= 4 else 'conv'
    if impl == 'chunked':
        return _UpConvProjected.apply(x, weight, bias, tables, k, H, W)
    w_taps = weight.permute(0, 2, 3, 1).reshape(Cout * k * k, Cin)              # rows ordered (co, ky, kx)
    if impl == 'matmul':                 # rocBLAS strided-batched GEMM: [Cout*k*k, Cin] x [NB][Cin, h*w]
        P = torch.matmul(w_taps, x.reshape(NB, Cin, h * w))
    else:                                # MIOpen 1x1 convolution
        P = torch.nn.functional.conv2d(x, w_taps.view(Cout * k * k, Cin, 1, 1))
    one_bias = bias if (bias is not None and Cout == 1) else None
    out = _UpConv1.apply(P.view(NB * Cout, k * k, h, w), one_bias, tables, k, H, W).view(NB, Cout, H, W)
    if bias is not None and Cout != 1:
        out = out + bias.view(1, Cout, 1, 1)
    return out


# ----------------------------------------------------------------------------------------------------------
# channels-last (NHWC) form of the same up-conv: the decoder's fast path
# ----------------------------------------------------------------------------------------------------------


# Forward projection of a stage whose input is a spike tensor (values 0 / 1 / 2: exact in bf16): the fp32 weight is split into three
# bf16 terms W = Wh + Wm + Wl (8 + 8 + 8 mantissa bits: exact), and P = [X X X](bf16) @ [Wh; Wm; Wl](bf16) runs on the bf16 MFMA
# path with fp32 accumulation.  Every product is exact, so the result has the error profile of the fp32 GEMM (same 1e-6 relative
# difference to a float64 reference; profiles/r01/gemm_nhwc.log) at 1.2 - 2x its speed for K >= 128.


def _split3_cols(Wt):
    """Wt [K, N] fp32 -> [K, 3N] bf16 = [Wh | Wm | Wl] with Wh + Wm + Wl == Wt exactly (8 + 8 + 8 mantissa bits), in ONE launch of the
    kernel that splits the gradients (ss_split3_bf16) — was 2 subtractions, 5 casts and a concatenation, ~9 small launches per layer and step."""
    K, N = Wt.shape
    if Wt.is_cuda and N % 4 == 0:
        W3 = torch.empty((K, 3 * N), dtype=torch.bfloat16, device=Wt.device)
        _lib.split3_bf16(Wt.contiguous(), W3, K, N)
        return W3
    Wh = Wt.to(torch.bfloat16)
    r = Wt - Wh.float()
    Wm = r.to(torch.bfloat16)
    Wl = (r - Wm.float()).to(torch.bfloat16)
    return torch.cat((Wh, Wm, Wl), 1)


def _split3_bf16(Wt):
    K, N = Wt.shape
    return _split3_cols(Wt).view(K, 3, N).permute(1, 0, 2).reshape(3 * K, N)      # [3K, N]: rows = (term, k)






# ----------------------------------------------------------------------------------------------------------
# box-sum form of the decoder's backward (round 4; ss_upconv_box.hip): host-side index tables
# ----------------------------------------------------------------------------------------------------------
def _range_tables(lo, hi, n_out, k=5):
    """lo, hi: inverse ranges of a resize's source-index table (python lists); n_out: output extent.  -> (ranges [(start, length)], id 0 = the empty
    range, the others sorted; rmap[i][t] = id of [lo[i] - t, hi[i] - t) & [0, n_out)) — the distinct output ranges a (source index, tap) pair collects."""
    pairs = []
    for i in range(len(lo)):
        row = []
        for t in range(k):
            a, b = max(lo[i] - t, 0), min(hi[i] - t, n_out)
            row.append((a, b - a) if b > a else (0, 0))
        pairs.append(row)
    uniq = sorted({p for row in pairs for p in row if p[1] > 0})
    ids = {p: n + 1 for n, p in enumerate(uniq)}
    ids[(0, 0)] = 0
    return [(0, 0)] + uniq, [[ids[p] for p in row] for row in pairs]


def _tile_spans(rmap, tile):
    """[(first id, id count)] of the non-empty ranges the maps of `tile` consecutive source indices reach, per tile."""
    out = []
    for a in range(0, len(rmap), tile):
        ids = [v for row in rmap[a:a + tile] for v in row if v > 0]
        out.append((min(ids), max(ids) - min(ids) + 1) if ids else (1, 1))
    return out


def _row_tiles(vmap, max_rows, max_span):
    """Cut the source rows into tiles of <= max_rows rows whose non-empty vertical range ids span <= max_span: [(first row, rows, first id, id count)].  A tile
    with a triple-replicated row reaches more distinct ranges than the on-chip window holds and is cut short (a single row always fits: <= 5 ranges... its
    ids are consecutive taps of one row)."""
    out, a = [], 0
    while a < len(vmap):
        n = 1
        while n < max_rows and a + n < len(vmap):
            ids = [v for row in vmap[a:a + n + 1] for v in row if v > 0]
            if ids and max(ids) - min(ids) + 1 > max_span:
                break
            n += 1
        ids = [v for row in vmap[a:a + n] for v in row if v > 0]
        out.append((a, n, min(ids), max(ids) - min(ids) + 1) if ids else (a, n, 1, 1))
        a += n
    return out


def register_box_tables(tables, host_tables, H, W):
    """Called where the resize tables are built (NNConvUpsampling._tables): the range lists / maps / tiles of the box-sum backward, from the HOST copies
    (no device read-back, nothing that could land in a stream capture)."""
    y_lo, y_hi, x_lo, x_hi = (host_tables[i].tolist() for i in (1, 2, 4, 5))
    vr, vmap = _range_tables(y_lo, y_hi, H)
    hr, hmap = _range_tables(x_lo, x_hi, W)
    dev = tables[1].device
    if dev.type == 'cuda':
        max_rows, max_span, _ = _lib.upconv_box_window()
    else:
        max_rows, max_span = 4, 15
    tr, tc = _row_tiles(vmap, max_rows, max_span), _tile_spans(hmap, 32)

    def t(a):
        return torch.tensor(a, dtype=torch.int32).reshape(-1).to(dev)
    bt = dict(vr=t(vr), hr=t(hr), vmap=t(vmap), hmap=t(hmap), tile_rows=t(tr), tile_cols=t(tc), NVR=len(vr), NHR=len(hr), n_row_tiles=len(tr),
              max_tile_rows=max(n for _, _, _, n in tr), max_cols32=max(n for _, n in tc), H=H, W=W)
    _BOX[id(tables)] = (tables, bt)
    return bt


def box_tables(tables, H, W):
    """The box-sum tables of a resize-table tuple (registered by NNConvUpsampling._tables; tables built elsewhere are read back once, never during a
    stream capture)."""
    hit = _BOX.get(id(tables))
    if hit is not None and hit[0] is tables and hit[1]['H'] == H and hit[1]['W'] == W:
        return hit[1]
    if tables[1].is_cuda and torch.cuda.is_current_stream_capturing():
        raise _lib.SSNeuronError('box_tables: resize tables without registered box tables inside a stream capture')
    return register_box_tables(tables, tuple(t.cpu() for t in tables), H, W)


_BOX = {}


# ---- sub-pixel ("merged tap") forward of a decoder stage (ss_upconv_sub.hip): classes of output rows / columns by the run structure of the five source
#      indices under their 5-tap window, and per class blocks of output positions with the distinct source positions they read
def _axis_classes(src, n_out, k=5):
    per = []
    for Y in range(n_out):
        runs, start = [], 0
        for t in range(1, k + 1):
            if t == k or src[Y + t] != src[Y + start]:
                runs.append(t - start)
                start = t
        per.append(tuple(runs))
    keys = sorted(set(per))
    if any(len(key) not in (2, 3) for key in keys):
        return None                                   # (a resize factor far from 2: the kernel is compiled for 2 or 3 distinct source positions per window)
    ids = {key: n for n, key in enumerate(keys)}
    table = []
    for key in keys:
        k0, a = [], 0
        for ln in key:
            k0.append(a)
            a += ln
        table.append([len(key)] + (k0 + [0, 0, 0])[:3] + (list(key) + [0, 0, 0])[:3] + [0])
    return [ids[p] for p in per], table


def _axis_blocks(src, cls, table, max_out, max_src, rec_ints):
    recs = []
    for c, row in enumerate(table):
        ng, k0 = row[0], row[1:4]
        pos = [Y for Y, cc in enumerate(cls) if cc == c]
        a = 0
        while a < len(pos):
            n, need = 0, set()
            while a + n < len(pos) and n < max_out:
                more = need | {src[pos[a + n] + k0[r]] for r in range(ng)}
                if len(more) > max_src:
                    break
                need, n = more, n + 1
            srcs = sorted(need)
            where = {sv: i for i, sv in enumerate(srcs)}
            rec = [c, n, len(srcs)] + (pos[a:a + n] + [0] * max_out)[:max_out] + (srcs + [0] * max_src)[:max_src]
            for i in range(max_out):
                rec += [where[src[pos[a + i] + k0[r]]] if (i < n and r < ng) else 0 for r in range(3)]
            rec = (rec + [0] * rec_ints)[:rec_ints]
            rec[-1] = ng                                  # the kernel reads the class's run count from the record (no dependent class-table load)
            recs.append(rec)
            a += n
    return recs


def register_sub_tables(tables, host_tables, H, W):
    """Called where the resize tables are built (NNConvUpsampling._tables), from the HOST copies: class / block tables of the sub-pixel forward, or None
    when the geometry has more than 3 runs per window (the stage then keeps the projected form)."""
    src_y, src_x = host_tables[0].tolist(), host_tables[3].tolist()
    dev = tables[0].device
    geo = _lib.upconv_sub_geometry() if dev.type == 'cuda' else dict(block_rows=16, block_cols=32, window_rows=20, window_cols=36, vrec_ints=88, hrec_ints=168, runs=3,
                                                                      tall_rows=64, tall_window_rows=68, trec_ints=328, narrow_cols=8, window_pixels=720)
    cv, ch = _axis_classes(src_y, H), _axis_classes(src_x, W)
    st = None
    if cv is not None and ch is not None:
        vblk = _axis_blocks(src_y, cv[0], cv[1], geo['block_rows'], geo['window_rows'], geo['vrec_ints'])
        hblk = _axis_blocks(src_x, ch[0], ch[1], geo['block_cols'], geo['window_cols'], geo['hrec_ints'])
        # column blocks of <= narrow_cols columns get TALL row blocks (the four wavefronts stacked vertically) when their windows fit
        tblk = _axis_blocks(src_y, cv[0], cv[1], geo['tall_rows'], geo['tall_window_rows'], geo['trec_ints'])
        narrow = [i for i, hh in enumerate(hblk) if hh[1] <= geo['narrow_cols']]
        if not narrow or max(tb[2] for tb in tblk) * max(hblk[i][2] for i in narrow) > geo['window_pixels']:
            narrow, tblk = [], []
        NVB, NHB = len(vblk), len(hblk)
        pairs = [vb * NHB + hb for vb in range(NVB) for hb in range(NHB) if hb not in narrow] + [(NVB + tb) * NHB + hb for tb in range(len(tblk)) for hb in narrow]

        def t(a):
            return torch.tensor(a, dtype=torch.int32).reshape(-1).to(dev)

        # tile order: by cost — runs x runs x (4 M-blocks | 1) MFMA rounds per wavefront — most expensive first
        def cost(pair):
            v = vblk[pair // NHB] if pair < NVB * NHB else tblk[pair // NHB - NVB]
            hh = hblk[pair % NHB]
            return v[-1] * hh[-1] * (4 if v[1] > 4 else 1)
        order = sorted(pairs, key=lambda pr: (-cost(pr), pr))
        st = dict(vcls=t(cv[1]), hcls=t(ch[1]), vblk=t(vblk), hblk=t(hblk), tblk=t(tblk) if tblk else None, order=t(order), NVC=len(cv[1]), NHC=len(ch[1]),
                  NVB=NVB, NHB=NHB, NTB=len(tblk), NORD=len(order), vrec_ints=geo['vrec_ints'], hrec_ints=geo['hrec_ints'], trec_ints=geo['trec_ints'], H=H, W=W)
    _SUB[id(tables)] = (tables, st, H, W)
    return st


def sub_tables(tables, H, W):
    hit = _SUB.get(id(tables))
    if hit is not None and hit[0] is tables and hit[2] == H and hit[3] == W:
        return hit[1]
    if tables[0].is_cuda and torch.cuda.is_current_stream_capturing():
        raise _lib.SSNeuronError('sub_tables: resize tables without registered sub-pixel tables inside a stream capture')
    return register_sub_tables(tables, tuple(t.cpu() for t in tables), H, W)


_SUB = {}


# ---- which kernel forms a decoder stage takes: decided ONCE per (resize tables = module and input geometry), from the geometry (no
#      channel-count tuples standing in for "deconv1 .. 3").  Measured table behind the two thresholds (MI355X, BASELINE config 3, 80 frames):
#        sub-pixel forward vs exact GEMM + gather   deconv1 130x173 src  0.63 vs 1.34 ms | deconv2 65x87  0.69 vs 1.25 | deconv3 33x44  1.01 vs 1.26 | deconv4 17x22  slower
#                                                   (profiles/r04/bench_sub_fwd_v10.log)
#        box-sum backward vs adjoint -> g_P -> GEMMs deconv1 2.78 vs 3.72 ms | deconv2 2.15 vs 2.87 | deconv3 33x44 and deconv4 17x22: slower (tile fill 57 / 47 %;
#                                                   profiles/r04/bench_box_bwd_v5.log, bench_box_bwd_v14.log)
#      i.e. the per-frame tile kernels lose on source maps of few pixels with many channels (few, mostly empty tiles per frame, long per-tile weight streams).
_SUB_MIN_SRC_PIXELS = 1024          # wide stages (C_in > 128): source pixels per frame from which the sub-pixel forward wins
_BOX_MIN_SRC_PIXELS = 2048          # ... and the box-sum backward
_STAGE = {}


def stage_plan(tables, Cin: int, Cout: int, k: int, h: int, w: int, H: int, W: int) -> dict:
    """{'sub_fwd': bool, 'box_bwd': bool} of the decoder stage whose resize tables these are (spike input, k = 5): what the GEOMETRY admits and where it was
    measured to win.  The run-time switches (EngineConfig.SUB_FWD / BOX_BWD), dtypes and operand forms are checked at the dispatch site."""
    key = (id(tables), Cin, Cout, k, h, w, H, W)
    hit = _STAGE.get(key)
    if hit is not None and hit[0] is tables:
        return hit[1]
    plan = dict(sub_fwd=False, box_bwd=False)
    if k == 5 and tables[0].is_cuda:
        narrow_or_large = lambda floor: Cin <= 128 or h * w >= floor      # noqa: E731
        plan['sub_fwd'] = bool(_lib.upconv_sub_supported(Cin, Cout, k) and narrow_or_large(_SUB_MIN_SRC_PIXELS) and sub_tables(tables, H, W) is not None)
        if narrow_or_large(_BOX_MIN_SRC_PIXELS):
            bt = box_tables(tables, H, W)
            plan['box_bwd'] = bool(_lib.upconv_box_dgrad_supported(Cin, Cout, k, bt) and _lib.upconv_box_wgrad_supported(Cin, Cout, k, bt))
    _STAGE[key] = (tables, plan)
    return plan


@dataclass(frozen=True)
class StageOpts:
    """What upconv_projected_cl derives from the call and the autocast state for one stage / head pass (one object instead of nine positional arguments)."""
    k: int
    H: int
    W: int
    lowp: bool = False                          # bf16 autocast: GEMM operands in bf16 (fp32 accumulation / output)
    spikes_in: bool = False                     # the input holds spike counts (exact in bf16): exact-split forms apply
    lowp_bwd: Optional[bool] = None             # backward operand precision under ANY 16-bit autocast (None: as lowp)
    act_dtype: Optional[torch.dtype] = None     # decoder stages under 16-bit autocast write 16-bit activations
    lowrank_grad: bool = False                  # a one-channel 3 x 3 head may hand its input gradient on as the rank-9 pair
    own16: Optional[torch.dtype] = None         # the 16-bit mode's dtype when it runs on the engine's own single-term kernels
    fwd16: bool = False                         # fp16 mode on own kernels: the projection GEMM takes ONE fp16 weight term (as every other synapse of the mode) instead of bf16x3


# ---- the forms of a decoder stage / prediction head (dispatch: _UpConvProjectedCL below); each reads what the dispatcher derived and sets the ctx fields the
#      backward forms need
def _stage_fwd_packed_head(ctx, *, H, W, act_dtype, bias, k, lowp, lowrank_grad, tables, weight, x_cl, x_packed):
    """One-channel 3 x 3 prediction head on a 2-bit packed input: MFMA projection to the 9 taps (ss_head_proj_packed_f32) + gather."""
    NB, h, w, Cin = x_cl.shape
    Cout, kk = weight.shape[0], k * k
    if not (k == 3 and Cout == 1 and not lowp and act_dtype is None and _lib.head_packed_supported(Cin, Cout, k)):
        raise _lib.SSNeuronError('packed-only spike tensor handed to an up-conv that reads dense activations')
    ctx.lowrank_grad = bool(lowrank_grad)
    ctx.x_dtype = x_cl.dtype
    weight = weight.float()
    Wt = weight.permute(1, 2, 3, 0).reshape(Cin, kk * Cout).contiguous()
    src_y, _, _, src_x, _, _ = tables
    out = torch.empty((NB, H, W, Cout), dtype=torch.float32, device=x_cl.device)
    rows = NB * h * w
    P = torch.empty((rows, kk * Cout), dtype=torch.float32, device=x_cl.device)
    e0 = TIMER.start()
    _lib.head_proj_packed(x_packed, Wt, P, rows, Cin)
    _lib.upconv_cl_fwd(P, src_y, src_x, None if bias is None else bias.float(), out, NB, k, Cout, h, w, H, W)
    TIMER.stop(e0, 'upconv_cl_fwd', x_cl.numel() // 4 + 4 * out.numel(), out.numel())
    _note('synapse_fwd', 'head_proj_packed_mfma+gather')
    ctx.save_for_backward(x_cl, Wt, x_packed, weight)
    ctx.lowp, ctx.exact = False, False
    ctx.tables, ctx.k, ctx.n = tables, k, NB
    ctx.wshape = weight.shape
    ctx.has_bias = bias is not None
    return out

def _stage_fwd_sub_x16(ctx, *, Cin, Cout, H, NB, W, Wt, h, k, lowp, lowp_bwd, n, out, own16, tables, w, weight, x_cl, x_packed):
    """Decoder stage forward, 16-bit mode on own kernels: the sub-pixel (merged tap) implicit GEMM with two terms of the mode\'s format."""
    st16 = sub_tables(tables, H, W)
    e0 = TIMER.start()
    wm = _lib.upconv_sub_prep_x16(weight.contiguous(), st16, Cin, Cout, own16)
    _lib.upconv_sub_fwd_x16(None if x_packed is not None else x_cl, x_packed, wm, st16, out, NB, Cin, Cout, h, w)
    TIMER.stop(e0, 'upconv_cl_fwd', (x_cl.numel() // 4 if x_packed is not None else 2 * x_cl.numel()) + 2 * out.numel(), out.numel())
    _note('synapse_fwd', 'upconv_sub_mfma_x16' + ('(packed in)' if x_packed is not None else ''))
    ctx.save_for_backward(x_cl, Wt, x_packed, weight)
    ctx.lowp = lowp if lowp_bwd is None else lowp_bwd
    ctx.exact = False
    ctx.tables, ctx.k, ctx.n = tables, k, n
    ctx.wshape = weight.shape
    ctx.has_bias = False
    return out

def _stage_fwd_sub_f32(ctx, *, Cin, Cout, NB, Wt, h, k, lowp, lowp_bwd, n, out, st, tables, w, weight, x_cl, x_packed):
    """Decoder stage forward, fp32 mode: the sub-pixel (merged tap) implicit GEMM on exact bf16x3 products."""
    if _cfg().ASSERT_EXACT_SPLIT and x_packed is None:
        assert bool((x_cl.to(torch.bfloat16).float() == x_cl).all()), 'spikes_in=True but the input is not exact in bf16'
    e0 = TIMER.start()
    wm = _lib.upconv_sub_prep(weight.contiguous(), st, Cin, Cout)
    _lib.upconv_sub_fwd(None if x_packed is not None else x_cl, x_packed, wm, st, out, NB, Cin, Cout, h, w)
    TIMER.stop(e0, 'upconv_cl_fwd', (x_cl.numel() // 4 if x_packed is not None else 4 * x_cl.numel()) + 4 * out.numel(), out.numel())
    _note('synapse_fwd', 'upconv_sub_mfma' + ('(packed in)' if x_packed is not None else ''))
    ctx.save_for_backward(x_cl, Wt, x_packed, weight)
    ctx.lowp = lowp if lowp_bwd is None else lowp_bwd
    ctx.exact = True
    ctx.tables, ctx.k, ctx.n = tables, k, n
    ctx.wshape = weight.shape
    ctx.has_bias = False
    return out

def _stage_fwd_projected(ctx, *, fwd16, Cin, Cout, H, NB, W, Wt, act_dtype, bias, exact, h, half_in, k, lowp, lowp_bwd, n, out, src_x, src_y, tables, w, weight, x_cl):
    """Projection GEMM (exact bf16x3 / bf16 / fp32 on the library) + the channels-last gather kernel: geometries the sub-pixel kernel does not take, heads on dense inputs."""
    gdt = torch.bfloat16 if lowp else (torch.float16 if fwd16 else None)      # single-term GEMM: operands in the mode's format, fp32 accumulation / output
    if gdt is not None:
        xg = x_cl if x_cl.dtype == gdt else x_cl.to(gdt)
    elif exact or not half_in:
        xg = x_cl                                   # exact: any dtype feeds the bf16 triple copy; plain fp32 path: fp32
    else:
        xg = x_cl.float()                           # 16-bit input on the plain fp32 GEMM path (narrow stages, heads)
    Wg = Wt.to(gdt) if gdt is not None else Wt
    if exact:
        if _cfg().ASSERT_EXACT_SPLIT:
            assert bool((x_cl.to(torch.bfloat16).float() == x_cl.float()).all()), 'spikes_in=True but the input is not exact in bf16'
        W3 = _split3_bf16(Wt)
    e0 = TIMER.start()
    for c0 in range(0, NB, n):
        c1 = min(NB, c0 + n)
        xs = xg[c0:c1].view((c1 - c0) * h * w, Cin)
        if exact:
            x3 = torch.empty((xs.shape[0], 3, Cin), dtype=torch.bfloat16, device=xs.device)
            x3.copy_(xs.unsqueeze(1))                                                  # one cast kernel writing the 3 copies
            P = torch.mm(x3.view(-1, 3 * Cin), W3, out_dtype=torch.float32)
            del x3
        else:
            P = torch.mm(xs, Wg, out_dtype=torch.float32) if gdt is not None else torch.mm(xs, Wg)   # [(n*h*w), kk*Cout]
        (_lib.upconv_cl_fwd_x16 if act_dtype else _lib.upconv_cl_fwd)(P, src_y, src_x, bias, out[c0:c1], c1 - c0, k, Cout, h, w, H, W)
    TIMER.stop(e0, 'upconv_cl_fwd', 4 * (x_cl.numel() + out.numel()), out.numel())
    _note('synapse_fwd', ('exact_bf16x3_gemm' if exact else ('bf16_gemm' if lowp else ('fp16_gemm' if fwd16 else 'fp32_gemm'))) + '+gather' + ('_x16' if act_dtype else ''))
    ctx.save_for_backward(xg, Wt, None, weight)
    # backward operand precision: bf16 operands (fp32 accumulate / output) under ANY 16-bit autocast — bf16 has the fp32 exponent
    # range, so the fp16 mode needs no loss scaling for it; the forward of the fp16 mode: ONE fp16 term on the engine's own kernel path (round 6: as every other
    # synapse of the mode), the exact bf16x3 form on the round-2 .. 4 path (X16_OWN_KERNELS off)
    ctx.lowp = lowp if lowp_bwd is None else lowp_bwd
    ctx.exact = exact
    ctx.tables, ctx.k, ctx.n = tables, k, n
    ctx.wshape = weight.shape
    ctx.has_bias = bias is not None
    return out

def _stage_bwd_box_x16(ctx, *, Cin, Cout, H, NB, W, bt, g_out, h, need_w, need_x, own16, w, weight, x_cl, x_packed):
    """Stage backward on the box-sum image, 16-bit I/O (one plane of the mode\'s format, one weight term)."""
    e0 = TIMER.start()
    e1 = TIMER.start()
    box = _lib.upconv_boxsum_x16(g_out, bt, NB, Cout, H, W)
    TIMER.stop(e1, 'box_boxsum', 2 * g_out.numel() + 2 * box.numel(), g_out.numel())
    g_x = g_w = None
    wc = weight.detach().contiguous()
    if need_x:
        g_x = torch.empty(x_cl.shape, dtype=own16, device=g_out.device)
        e1 = TIMER.start()
        _lib.upconv_box_dgrad_x16(box, wc, bt, g_x, NB, Cin, Cout, h, w)
        TIMER.stop(e1, 'box_dgrad', 2 * box.numel() + 2 * g_x.numel(), g_x.numel())
    if need_w:
        g_w = torch.empty(wc.shape, dtype=torch.float32, device=g_out.device)
        e1 = TIMER.start()
        _lib.upconv_box_wgrad_x16(box, None if x_packed is not None else x_cl, x_packed, bt, g_w, NB, Cin, Cout, h, w)
        TIMER.stop(e1, 'box_wgrad', 2 * box.numel() + (x_cl.numel() // 4 if x_packed is not None else 2 * x_cl.numel()), g_out.numel())
    TIMER.stop(e0, 'upconv_cl_bwd', 2 * (x_cl.numel() + g_out.numel()), g_out.numel())
    _note('synapse_bwd', 'box_x16: boxsum' + ('+dgrad1_mfma' if need_x else '') + ('+wgrad1_mfma' if need_w else '') + ('(packed x)' if (need_w and x_packed is not None) else ''), ctx.site)
    if g_x is not None and g_x.dtype != ctx.x_dtype:
        g_x = g_x.to(ctx.x_dtype)
    g_b = g_out.float().sum((0, 1, 2)) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
    return g_x, g_w, g_b, None, None, None

def _stage_bwd_box_f32(ctx, *, Cin, Cout, H, NB, W, bt, g_out, h, need_w, need_x, w, weight, x_cl, x_packed):
    """Stage backward on the box-sum image, fp32 mode: box-sum launch, then both contractions over its three bf16 planes; no g_P."""
    e0 = TIMER.start()                                                   # the stage's whole backward ('upconv_cl_bwd', as every other form) ...
    e1 = TIMER.start()                                                   # ... and its three launches one by one ('box_*': inside the former, not additional)
    box = _lib.upconv_boxsum(g_out, bt, NB, Cout, H, W)
    TIMER.stop(e1, 'box_boxsum', 4 * g_out.numel() + 2 * box.numel(), g_out.numel())
    g_x = g_w = None
    wc = weight.detach().contiguous()
    if need_x:
        g_x = torch.empty(x_cl.shape, dtype=torch.float32, device=g_out.device)
        e1 = TIMER.start()
        _lib.upconv_box_dgrad(box, wc, bt, g_x, NB, Cin, Cout, h, w)
        TIMER.stop(e1, 'box_dgrad', 2 * box.numel() + 4 * g_x.numel(), g_x.numel())
    if need_w:
        g_w = torch.empty(wc.shape, dtype=torch.float32, device=g_out.device)
        e1 = TIMER.start()
        _lib.upconv_box_wgrad(box, None if x_packed is not None else x_cl, x_packed, bt, g_w, NB, Cin, Cout, h, w)
        TIMER.stop(e1, 'box_wgrad', 2 * box.numel() + (x_cl.numel() // 4 if x_packed is not None else 4 * x_cl.numel()), g_out.numel())
    TIMER.stop(e0, 'upconv_cl_bwd', 4 * (x_cl.numel() + g_out.numel()), g_out.numel())
    _note('synapse_bwd', 'box: boxsum' + ('+dgrad6_mfma' if need_x else '') + ('+wgrad3_mfma' if need_w else '') + ('(packed x)' if (need_w and x_packed is not None) else ''), ctx.site)
    if g_x is not None and g_x.dtype != ctx.x_dtype:
        g_x = g_x.to(ctx.x_dtype)
    # a stage built with bias=True (NNConvUpsampling accepts it; the shipped decoder stages have none): its gradient as in every other form
    g_b = g_out.sum((0, 1, 2)) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
    return g_x, g_w, g_b, None, None, None

def _stage_bwd_gp(ctx, *, Cin, Cout, H, NB, W, Wt, g16, g_out, h, k, kk, lowp, lowrank, n, need_w, need_x, w, x_cl, x_hi, x_lo, x_packed, y_hi, y_lo):
    """Stage backward through the per-tap tensor g_P: gather adjoint, then data / weight gradient GEMMs (ss_gemm6_f32, ss_spike_wgrad_f32, ss_head_wgrad_packed_f32 or the library); a head may hand its input gradient on as the rank-9 pair."""
    g_x = torch.empty(x_cl.shape, dtype=torch.float32, device=x_cl.device) if (need_x and not lowrank and not (lowp and n >= NB)) else None
    g_Wt = torch.zeros_like(Wt) if need_w else None
    W2 = Wt.t().contiguous()                                                   # [kk*Cout, Cin]
    # operand format of the two backward GEMMs under a 16-bit mode: bf16 — except the fp16 mode on own kernels, whose fp16 gradients make fp16 g_P, fp16 weights
    # (the mode's own rounding, what its forward multiplies with) and an fp16 data gradient straight from the GEMM's epilogue (round 6; it was bf16 operands,
    # an fp32 result and a conversion pass over it: ~1 ms at config 5's share)
    gdt = torch.float16 if (lowp and k == 5 and ctx.own16 == torch.float16 and g_out.dtype == torch.float16) else torch.bfloat16
    if lowp:
        W2 = W2.to(gdt)
    f32 = dict(out_dtype=torch.float32) if lowp else {}
    e0 = TIMER.start()
    for c0 in range(0, NB, n):
        c1 = min(NB, c0 + n)
        rows = (c1 - c0) * h * w
        if x_packed is not None and k == 5 and need_w:
            # a stage whose input arrived as packed spikes and whose geometry the box-sum kernels do not take: the g_P forms read the dense tensor
            # (a dense copy came along with the packed one — pack = 1 producers: use it instead of unpacking a second one).  16-bit modes whose
            # dense copy is not in the GEMMs' bf16 operand format (fp16 mode): unpack straight to bf16 — 0.25 B read per element instead of an fp16 -> bf16
            # conversion pass over the dense copy (round 6: 0.3 - 0.4 ms per stage at config 5's share)
            if lowp and x_cl.dtype != gdt:
                x_cl, x_packed = unpack_dense(x_packed, x_cl.shape, gdt), None
            else:
                x_cl, x_packed = (unpack_dense(x_packed, x_cl.shape, x_cl.dtype) if x_cl.stride(-1) == 0 else x_cl), None
        if lowp and k == 5:
            # 16-bit modes: the adjoint writes g_P in the operand format of both backward GEMMs (no fp32 round trip, no cast)
            g_P = torch.empty((rows, kk * Cout), dtype=gdt, device=x_cl.device)
            _lib.upconv_cl_bwd_lowp(g_out[c0:c1], y_lo, y_hi, x_lo, x_hi, g_P, c1 - c0, k, Cout, h, w, H, W)
        else:
            if lowrank:      # the adjoint writes g_P straight into the pair buffer the consumer's neuron backward will read
                lr_anchor, g_P, lr_w = lowrank_buffer(x_cl.shape, x_cl.device, kk * Cout, ctx.x_dtype)
                lr_w.copy_(W2)
            else:
                g_P = torch.empty((rows, kk * Cout), dtype=torch.float32, device=x_cl.device)
            (_lib.upconv_cl_bwd_x16 if g16 else _lib.upconv_cl_bwd)(g_out[c0:c1], y_lo, y_hi, x_lo, x_hi, g_P, c1 - c0, k, Cout, h, w, H, W)
            if lowp:
                g_P = g_P.to(gdt)
        # dense x dense on the bf16 matrix cores with six cross terms (fp32-product accuracy) where the fp32 library GEMM is compute-bound: wide stages
        # (measured: C_in 64 — deconv1 — is faster on the library's fp32 GEMM, 1.63 vs 2.48 ms, profiles/r04/bench_box_bwd_v5.log)
        gemm6 = bool(need_x and not lowrank and not lowp and ctx.ecfg.GEMM6_DGRAD and Cin >= 128 and g_P.dtype == torch.float32 and _lib.gemm6_supported(kk * Cout, Cin))
        if lowrank:
            g_x = lr_anchor                                                   # the pair was written in place (lowrank_buffer)
        elif gemm6:
            _lib.gemm6(g_P, W2, g_x[c0:c1].view(rows, Cin), rows, kk * Cout, Cin)
        elif need_x and lowp and c0 == 0 and c1 == NB:
            # one chunk (the usual case): the GEMM's own output IS the gradient — when the operand format IS the activation format (bf16 mode; fp16 mode on own kernels) written in it by the GEMM's epilogue (fp32
            # accumulation, one rounding: what fp32-then-narrow gives), in the fp16 mode fp32 and narrowed once below; no copy into a preallocated buffer
            g_x = (torch.mm(g_P, W2) if ctx.x_dtype == gdt else torch.mm(g_P, W2, **f32)).view(x_cl.shape)
        elif need_x:
            g_x[c0:c1].view(rows, Cin).copy_(torch.mm(g_P, W2, **f32)) if lowp else \
                torch.mm(g_P, W2, out=g_x[c0:c1].view(rows, Cin))             # dgrad
        spike_wgrad = bool(need_w and ctx.exact and not lowp and ctx.ecfg.EXACT_WGRAD_MFMA and x_cl.dtype == torch.float32 and _lib.spike_wgrad_supported(Cin, kk * Cout))
        if need_w and x_packed is not None and k == 3:
            _lib.head_wgrad_packed(x_packed, g_P, g_Wt, rows, Cin, accumulate=True)     # reads the 2-bit packed spikes (1/16 of the dense tensor)
        elif spike_wgrad:
            # x is a spike tensor: hand-written exact bf16x3 MFMA contraction over the rows, g_P split in registers (read once from HBM)
            _lib.spike_wgrad(g_P, x_cl[c0:c1].view(rows, Cin), g_Wt, rows, Cin, kk * Cout, accumulate=True)
        elif need_w:
            xs = x_cl[c0:c1].view(rows, Cin)
            if lowp and xs.dtype != gdt:
                xs = xs.to(gdt)                                               # spikes: exact
            elif not lowp and xs.dtype != torch.float32:
                xs = xs.float()
            S = max(1, rows // _WGRAD_SPLIT_ROWS)
            L = rows // S
            if S > 1:
                g_Wt += torch.bmm(xs[:S * L].view(S, L, Cin).transpose(1, 2), g_P[:S * L].view(S, L, kk * Cout), **f32).sum(0)
                if S * L < rows:
                    g_Wt += torch.mm(xs[S * L:].t(), g_P[S * L:], **f32)
            else:
                g_Wt += torch.mm(xs.t(), g_P, **f32)
    TIMER.stop(e0, 'upconv_cl_bwd', 4 * (x_cl.numel() + g_out.numel()), g_out.numel())
    how_x = 'none' if not need_x else ('lowrank_pair' if lowrank else ('adjoint+gemm6' if gemm6 else 'adjoint+library_gemm'))
    how_w = 'none' if not need_w else ('head_wgrad_packed_mfma' if (x_packed is not None and k == 3) else ('spike_wgrad_mfma' if spike_wgrad else 'library_gemm'))
    _note('synapse_bwd', f'g_x: {how_x}; g_w: {how_w}' + ('; 16-bit' if (lowp or g16) else ''), ctx.site)
    g_w = g_Wt.view(Cin, k, k, Cout).permute(3, 0, 1, 2).contiguous() if need_w else None
    g_b = g_out.float().sum((0, 1, 2)) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
    if g_x is not None and g_x.dtype != ctx.x_dtype:
        g_x = g_x.to(ctx.x_dtype)                    # the gradient of a 16-bit activation input is a 16-bit activation gradient
    return g_x, g_w, g_b, None, None, None


class _UpConvProjectedCL(torch.autograd.Function):
    """x_cl [NB, h, w, C_in] -> out_cl [NB, H, W, C_out], everything in NHWC memory: the autograd node of a decoder stage / prediction head.
    Since round 6 a DISPATCHER: it derives what the forms share (stage plan, dtypes, re-laid-out weight, output buffer) and hands the pass to
    ONE form, each a module-level function above —
      forward : _stage_fwd_packed_head | _stage_fwd_sub_x16 | _stage_fwd_sub_f32 | _stage_fwd_projected   (P = x @ W as ONE row-major GEMM + the gather kernel)
      backward: _stage_bwd_box_x16 | _stage_bwd_box_f32 | _stage_bwd_gp   (gather adjoint -> g_P -> data / weight gradient contractions)
    chosen from fused.stage_plan (geometry), StageOpts (call + autocast state) and the operand dtypes; net.plan() records the choice."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda')          # called with autocast disabled (upconv_projected_cl): dtypes are explicit
    def forward(ctx, x_cl, weight, bias, tables, opts: 'StageOpts', x_packed=None):
        k, H, W, lowp, spikes_in, lowp_bwd, act_dtype, lowrank_grad, own16 = (opts.k, opts.H, opts.W, opts.lowp, opts.spikes_in, opts.lowp_bwd, opts.act_dtype,
                                                                               opts.lowrank_grad, opts.own16)
        fwd16 = bool(opts.fwd16)
        ctx.ecfg, ctx.site = _cfg(), _site()          # the engine configuration and plan site of THIS forward: the backward dispatches from them
        # own16 (round 5): the 16-bit activation mode's dtype when the mode runs on the engine's own single-term kernels (x16_mode()) — decoder stages then take
        # the sub-pixel forward / box-sum backward on 16-bit I/O, and a packed head keeps its exact fp32 weights
        ctx.own16 = own16
        plan = stage_plan(tables, x_cl.shape[-1], weight.shape[0], k, x_cl.shape[1], x_cl.shape[2], H, W) if (k == 5 and spikes_in and x_cl.is_cuda) else dict(sub_fwd=False, box_bwd=False)
        ctx.stage_plan = plan
        sub_geo = bool(_cfg().SUB_FWD and plan['sub_fwd'] and bias is None and x_cl.numel() < 2 ** 32)       # the sub-pixel forward applies to this geometry
        sub16_ok = bool(own16 is not None and sub_geo and act_dtype == own16 and (x_packed is not None or x_cl.dtype == own16))
        sub32_ok = bool(sub_geo and not lowp and act_dtype is None and _cfg().EXACT_SPLIT_GEMM and x_cl.dtype == torch.float32)
        # x_packed (one-channel 3 x 3 head on a packed-only neuron output, fp32 mode): the input as a 2-bit packed spike tensor; x_cl is then a
        # data-less anchor that carries shape and autograd edge.  Projection and weight gradient read the packed form (ss_head_*_packed_f32)
        if x_packed is not None and k != 3:
            # a decoder stage on a packed(-only) input: the sub-pixel forward reads the packed form; any other form of this stage gets the dense tensor back first
            if not (sub16_ok or sub32_ok):
                x_cl, x_packed = (unpack_dense(x_packed, x_cl.shape, x_cl.dtype) if x_cl.stride(-1) == 0 else x_cl), None     # (a dense copy came along: use it)
        if x_packed is not None and k == 3:
            return _stage_fwd_packed_head(ctx, H=H, W=W, act_dtype=act_dtype, bias=bias, k=k, lowp=lowp, lowrank_grad=lowrank_grad, tables=tables, weight=weight, x_cl=x_cl, x_packed=x_packed)
        # lowrank_grad: the caller guarantees x_cl is consumed by nothing else and produced by a fused neuron layer (a forked handle), so
        # the input gradient of a one-channel 3 x 3 head may be handed over as the pair (g_P, W2) instead of their product (lowrank_anchor)
        ctx.lowrank_grad = bool(lowrank_grad)
        # lowp (only under bf16 autocast): the three GEMMs take bf16 operands with fp32 accumulation / output, exactly what
        # autocast does to the MIOpen convs of the encoder (spike inputs are exact in bf16; W and g_P are rounded).
        # P, the gather, its adjoint and every output stay fp32.
        # act_dtype (fp16 / bf16, decoder stages under 16-bit autocast): the stage output is written by the gather as 16-bit activations,
        # its gradient is read as such by the adjoint; the input may itself be a 16-bit spike tensor (exact)
        if x_packed is None:
            x_cl = x_cl.contiguous()                                 # (a packed-only input is a data-less anchor: nothing to lay out)
        ctx.x_dtype = x_cl.dtype
        half_in = x_cl.dtype in (torch.float16, torch.bfloat16)
        weight = weight.float()
        if bias is not None:
            bias = bias.float()
        NB, h, w, Cin = x_cl.shape
        Cout, kk = weight.shape[0], k * k
        Wt = weight.permute(1, 2, 3, 0).reshape(Cin, kk * Cout).contiguous()      # column index = tap*C_out + co
        src_y, _, _, src_x, _, _ = tables
        out = torch.empty((NB, H, W, Cout), dtype=act_dtype or torch.float32, device=x_cl.device)
        # one pass: measured on the MI355X (profiles/r01/chunk_sweep*.log) cache-sized chunks lose more in GEMM efficiency (M = n*h*w
        # rows) than they save in HBM traffic of P — 71.7 -> 69.9 ms/step for config 3; chunks only bound the memory of very large batches
        n = max(1, min(NB, _P_MAX_BYTES_CL // (Cout * kk * h * w * 4)))
        exact = spikes_in and not lowp and not fwd16 and _cfg().EXACT_SPLIT_GEMM and Cin >= _EXACT_SPLIT_MIN_K
        if sub16_ok:
            return _stage_fwd_sub_x16(ctx, Cin=Cin, Cout=Cout, H=H, NB=NB, W=W, Wt=Wt, h=h, k=k, lowp=lowp, lowp_bwd=lowp_bwd, n=n, out=out, own16=own16, tables=tables, w=w, weight=weight, x_cl=x_cl, x_packed=x_packed)
        # round 4: the sub-pixel (merged tap) implicit GEMM — 9 instead of 25 multiply-adds per output element and channel, no P, no gather, no halo
        st = sub_tables(tables, H, W) if sub32_ok else None
        if st is not None:
            return _stage_fwd_sub_f32(ctx, Cin=Cin, Cout=Cout, NB=NB, Wt=Wt, h=h, k=k, lowp=lowp, lowp_bwd=lowp_bwd, n=n, out=out, st=st, tables=tables, w=w, weight=weight, x_cl=x_cl, x_packed=x_packed)
        return _stage_fwd_projected(ctx, fwd16=fwd16, Cin=Cin, Cout=Cout, H=H, NB=NB, W=W, Wt=Wt, act_dtype=act_dtype, bias=bias, exact=exact, h=h, half_in=half_in, k=k, lowp=lowp, lowp_bwd=lowp_bwd, n=n, out=out, src_x=src_x, src_y=src_y, tables=tables, w=w, weight=weight, x_cl=x_cl)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g_out):
        x_cl, Wt, x_packed, weight = ctx.saved_tensors           # x_packed: the packed-only head input (x_cl is then a data-less anchor)
        k, n = ctx.k, ctx.n
        _, y_lo, y_hi, _, x_lo, x_hi = ctx.tables
        NB, h, w, Cin = x_cl.shape
        Cout, kk = ctx.wshape[0], k * k
        H, W = g_out.shape[1:3]
        g16 = g_out.dtype in (torch.float16, torch.bfloat16) and k == 5       # 16-bit gradient read directly by the adjoint kernel
        g_out = g_out.contiguous() if g16 else g_out.float().contiguous()
        lowp = ctx.lowp
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        # one-channel 3 x 3 head on a forked neuron output: g_x = g_P [rows, 9] @ W2 [9, C_in] is left to the consumer's backward kernel
        # (anomaly detection scans every backward output for NaNs and would trip over the anchor: it gets the dense form)
        # (round 6: also the dense-input heads of the 16-bit modes on own kernels — `lowp`: their weight-gradient GEMM keeps its bf16 operands, the pair is fp32 —
        #  instead of a [rows, 9] x [9, C] GEMM, an fp32 -> 16-bit conversion of its C-channel result and the dense second gradient in the stage's neuron backward)
        lowrank = (need_x and ctx.lowrank_grad and ctx.ecfg.LOWRANK_HEAD_GRAD and (not lowp or ctx.own16 is not None) and not g16 and kk * Cout == 9 and n >= NB
                   and (ctx.x_dtype == torch.float32 or ctx.own16 is not None) and Cin % 4 == 0 and 1024 % Cin == 0 and not torch.is_anomaly_enabled())
        # ---- round 5: the 16-bit activation modes on own kernels — the same box-sum backward on 16-bit I/O (box planes in the mode's format, ONE weight term)
        own16 = ctx.own16
        box_geo = ctx.ecfg.BOX_BWD and k == 5 and ctx.stage_plan['box_bwd'] and (need_x or need_w)
        if box_geo:
            bt = box_tables(ctx.tables, H, W)
            box_geo = _lib.upconv_box_dgrad_supported(Cin, Cout, k, bt, NB, h, w) and _lib.upconv_box_wgrad_supported(Cin, Cout, k, bt, NB, h, w)     # (+ the launch limits at this NB)
        box16_ok = bool(box_geo and own16 is not None and g16 and g_out.dtype == own16 and (x_packed is not None or x_cl.dtype == own16))
        if box16_ok:
            return _stage_bwd_box_x16(ctx, Cin=Cin, Cout=Cout, H=H, NB=NB, W=W, bt=bt, g_out=g_out, h=h, need_w=need_w, need_x=need_x, own16=own16, w=w, weight=weight, x_cl=x_cl, x_packed=x_packed)
        # ---- round 4: the whole stage backward on the box-sum image (ss_upconv_box.hip): one HBM-bound box-sum launch, then both contractions as implicit
        #      GEMMs over its three bf16 planes — no per-tap tensor g_P in HBM or on chip, no per-fragment operand arithmetic
        box_ok = bool(box_geo and not lowp and not g16 and ctx.exact and g_out.dtype == torch.float32 and not lowrank and (x_packed is not None or x_cl.dtype == torch.float32))
        if box_ok:
            return _stage_bwd_box_f32(ctx, Cin=Cin, Cout=Cout, H=H, NB=NB, W=W, bt=bt, g_out=g_out, h=h, need_w=need_w, need_x=need_x, w=w, weight=weight, x_cl=x_cl, x_packed=x_packed)
        return _stage_bwd_gp(ctx, Cin=Cin, Cout=Cout, H=H, NB=NB, W=W, Wt=Wt, g16=g16, g_out=g_out, h=h, k=k, kk=kk, lowp=lowp, lowrank=lowrank, n=n, need_w=need_w, need_x=need_x, w=w, x_cl=x_cl, x_hi=x_hi, x_lo=x_lo, x_packed=x_packed, y_hi=y_hi, y_lo=y_lo)


class _SpikeConvCL(torch.autograd.Function):
    """Conv2d on a spike tensor (NHWC array [NB, h, w, C_in], values exact in bf16) as exact bf16x3 GEMMs on the bf16 MFMA path:
    forward : A = im2col(x) in bf16 (ss_im2col_cl_bf16, kept for backward);  y = A @ [Wh | Wm | Wl], the three column blocks summed;
    wgrad   : g_W = A^T @ [gh | gm | gl] (ss_split3_bf16 of the fp32 output gradient), blocks summed — both with fp32 accumulation and
              exact products, i.e. the accuracy of the fp32 convolution at 1.3 - 2.4x MIOpen's fp32 speed (profiles/r01/conv_as_gemm.log);
    dgrad   : dense fp32 x dense fp32 — stays MIOpen's fp32 data-gradient convolution (aten.convolution_backward, input mask only).
    The reference call sites: conv3 / conv4 (SNN_models.py:91-101) and the SEW bottleneck convs (blocks.py:146-159)."""

    @staticmethod
    @_fwd32
    def forward(ctx, x_cl, weight, stride, pad, x_packed=None):
        ctx.ecfg, ctx.site = _cfg(), _site()          # the engine configuration and plan site of THIS forward: the backward dispatches from them
        # x_packed: the same input as a 2-bit packed spike tensor (x_cl then only carries shape and autograd edge; it may be an anchor)
        if x_packed is None:
            x_cl = x_cl.contiguous()
        NB, h, w, Cin = x_cl.shape
        Cout, _, k, _ = weight.shape
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        M, K = NB * ho * wo, k * k * Cin
        if _cfg().ASSERT_EXACT_SPLIT and x_packed is None:
            assert bool((x_cl.to(torch.bfloat16).float() == x_cl).all()), 'spikes_in=True but the input is not exact in bf16'
        A = torch.empty((M, K), dtype=torch.bfloat16, device=x_cl.device)
        if x_packed is not None:
            _lib.im2col_cl_bf16_packed(x_packed.contiguous(), A, NB, h, w, Cin, k, stride, pad, ho, wo)
        else:
            _lib.im2col_cl_bf16(x_cl, A, NB, h, w, Cin, k, stride, pad, ho, wo)
        Wt = weight.permute(2, 3, 1, 0).reshape(K, Cout)                          # row index = (ky, kx, c): the im2col column order
        y3 = torch.mm(A, _split3_cols(Wt.float()), out_dtype=torch.float32)       # [M, 3*Cout] = A @ [Wh | Wm | Wl]
        y = y3.view(M, 3, Cout).sum(1).view(NB, ho, wo, Cout)
        _note('synapse_fwd', 'im2col' + ('(packed in)' if x_packed is not None else '') + '+exact_bf16x3_gemm')
        ctx.save_for_backward(A, weight)
        ctx.geom = (NB, h, w, Cin, Cout, k, stride, pad, ho, wo)
        return y

    @staticmethod
    @_bwd32
    def backward(ctx, g):
        A, weight = ctx.saved_tensors
        NB, h, w, Cin, Cout, k, stride, pad, ho, wo = ctx.geom
        M, K = NB * ho * wo, k * k * Cin
        g = g.float().contiguous()
        g_x = g_w = None
        if ctx.needs_input_grad[1]:
            g3 = torch.empty((M, 3 * Cout), dtype=torch.bfloat16, device=g.device)
            _lib.split3_bf16(g, g3, M, Cout)
            S = next(d for d in (_SPIKE_CONV_WGRAD_SPLIT, 4, 2, 1) if M % d == 0)   # split-K: the output is only K x 3*Cout
            if S > 1:
                gw3 = torch.bmm(A.view(S, M // S, K).transpose(1, 2), g3.view(S, M // S, 3 * Cout), out_dtype=torch.float32)      # [S, K, 3*Cout]
            else:
                gw3 = torch.mm(A.t(), g3, out_dtype=torch.float32)                # [K, 3*Cout]
            if Cin % 8 == 0 and Cout % 32 == 0 and k <= 7:
                # slices and terms summed and the Conv2d layout written by one kernel (instead of two torch reductions and a permuting copy)
                g_w = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=g.device)
                _lib.wgrad_reduce3(gw3, g_w, S, k, Cin, Cout)
            else:
                if S > 1:
                    gw3 = gw3.sum(0)
                g_w = gw3.view(K, 3, Cout).sum(1).view(k, k, Cin, Cout).permute(3, 2, 0, 1).contiguous()
        how_x = 'none'
        if ctx.needs_input_grad[0] and ctx.ecfg.WINOGRAD_DGRAD and k == 3 and stride == 1 and pad == 1 and Cin % 4 == 0 and Cout % 4 == 0:
            g_x = winograd_dgrad_cl(g, weight)
            how_x = 'winograd_f2x2_3x3+batched_gemm'
        elif ctx.needs_input_grad[0] and ctx.ecfg.CONV_DGRAD_MFMA and _lib.conv_s2_dgrad_supported(Cin, Cout, k, stride, pad):
            g_x = conv_s2_dgrad_cl(g, weight, h, w)
            how_x = 'conv_s2_dgrad6_mfma'
        elif ctx.needs_input_grad[0]:
            how_x = 'miopen'
            x_meta = torch.empty((NB, Cin, h, w), dtype=torch.float32, device=g.device, memory_format=torch.channels_last)
            g_x = torch.ops.aten.convolution_backward(
                g.permute(0, 3, 1, 2), x_meta, weight.contiguous(memory_format=torch.channels_last), None,
                [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])[0]
            g_x = g_x.permute(0, 2, 3, 1)
            if not g_x.is_contiguous():
                g_x = g_x.contiguous()
        _note('synapse_bwd', f'g_x: {how_x}; g_w: ' + ('split3+exact_bf16x3_gemm' if ctx.needs_input_grad[1] else 'none'), ctx.site)
        return g_x, g_w, None, None, None






def conv_s2_dgrad_cl(g: torch.Tensor, weight: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """Data gradient of conv2d(x [NB, h, w, C_in], weight [C_out, C_in, 5, 5], stride 2, padding 2) for g [NB, ho, wo, C_out] (contiguous NHWC array,
    fp32) -> g_x [NB, h, w, C_in].  Autograd's backward of the reference's encoder convs w.r.t. their input (/root/reference/network/SNN_models.py:80-101)."""
    NB, ho, wo, Cout = g.shape
    Cin = weight.shape[1]
    g_x = torch.empty((NB, h, w, Cin), dtype=torch.float32, device=g.device)
    e0 = TIMER.start()
    _lib.conv_s2_dgrad(g, weight.detach().float().contiguous(), g_x, NB, Cin, Cout, h, w)
    TIMER.stop(e0, 'conv_s2_dgrad', 4 * (g.numel() + g_x.numel()), g_x.numel())
    return g_x




def winograd_dgrad_cl(g: torch.Tensor, weight: torch.Tensor, gemm6: Optional[bool] = None) -> torch.Tensor:
    """Data gradient of conv2d(x, weight, stride 1, padding 1), weight [C_out, C_in, 3, 3]: g [NB, H, W, C_out] (contiguous NHWC array, fp32)
    -> g_x [NB, H, W, C_in].  The autograd backward of the reference's SEWResBlock convs (/root/reference/network/blocks.py:146-159) w.r.t.
    their input, as ss_wino_dgrad_{weights,input,output}_f32 around torch.bmm (fp32)."""
    NB, H, W, Cout = g.shape
    Cin = weight.shape[1]
    T = _lib.wino_tiles(NB, H, W)
    U = torch.empty((16, Cout, Cin), dtype=torch.float32, device=g.device)
    V = torch.empty((16, T, Cout), dtype=torch.float32, device=g.device)
    _lib.wino_dgrad_weights(weight.detach().float().contiguous(), U, Cout, Cin)
    _lib.wino_dgrad_input(g, V, NB, H, W, Cout)
    if gemm6 and _lib.gemm6_supported(Cout, Cin) and (T * Cout) % 4 == 0:      # (A/B only: the library's batched fp32 GEMM is faster, profiles/r04/bench_wino_gemm6_*.json)
        M = torch.empty((16, T, Cin), dtype=torch.float32, device=g.device)          # the 16 products on the bf16 matrix cores, six cross terms
        _lib.gemm6_batched(V, U, M, 16, T, Cout, Cin)
    else:
        M = torch.bmm(V, U)
    del V
    g_x = torch.empty((NB, H, W, Cin), dtype=torch.float32, device=g.device)
    _lib.wino_dgrad_output(M, g_x, NB, H, W, Cin)
    return g_x




class _SpikeConvWgradCL(torch.autograd.Function):
    """Conv2d(k 5, stride 2, pad 2) of conv1 / conv2 (/root/reference/network/SNN_models.py:80-90) on a spike NHWC array: forward and data gradient
    are MIOpen's fp32 convolution (as before), the WEIGHT gradient — one operand is a spike tensor — is the exact bf16x3 MFMA contraction over the
    output pixels (ss_spike_conv_wgrad_f32)."""

    @staticmethod
    @_fwd32
    def forward(ctx, x_cl, weight, x_packed=None):
        ctx.ecfg, ctx.site = _cfg(), _site()          # the engine configuration and plan site of THIS forward: the backward dispatches from them
        NB, h, w, Cin = x_cl.shape
        Cout = weight.shape[0]
        if _cfg().SPIKE_CONV_FWD_MFMA and _lib.spike_conv_fwd_supported(Cin, Cout, 5, 2, 2):
            # forward as the exact bf16x3 implicit GEMM on the matrix cores, reading the packed spikes when the producer wrote them
            # (x_cl may then be a data-less anchor)
            if x_packed is None:
                x_cl = x_cl.contiguous()
                if _cfg().ASSERT_EXACT_SPLIT:
                    assert bool((x_cl.to(torch.bfloat16).float() == x_cl).all()), 'spike_conv: the input is not exact in bf16'
            y = torch.empty((NB, (h - 1) // 2 + 1, (w - 1) // 2 + 1, Cout), dtype=torch.float32, device=x_cl.device)
            e0 = TIMER.start()
            _lib.spike_conv_fwd(None if x_packed is not None else x_cl, None if x_packed is None else x_packed.contiguous(),
                                weight.detach().float().contiguous(), y, NB, Cin, Cout, h, w)
            TIMER.stop(e0, 'spike_conv_fwd', 4 * y.numel() + (x_cl.numel() // 4 if x_packed is not None else 4 * x_cl.numel()), y.numel())
            _note('synapse_fwd', 'spike_conv_fwd3_mfma' + ('(packed in)' if x_packed is not None else ''))
        else:
            _note('synapse_fwd', 'miopen')
            if x_cl.stride(-1) == 0:
                raise RuntimeError('packed-only spike tensor handed to a convolution that reads dense activations')
            x_cl = x_cl.contiguous()
            w_cl = weight.contiguous(memory_format=torch.channels_last)
            y = torch.nn.functional.conv2d(x_cl.permute(0, 3, 1, 2), w_cl, None, 2, 2).permute(0, 2, 3, 1)
            y = y if y.is_contiguous() else y.contiguous()
        # the backward's operand copy reads the 2-bit packed form when the producer wrote one (16x less to read than the fp32 tensor)
        ctx.save_for_backward(x_cl, weight, x_packed)
        return y

    @staticmethod
    @_bwd32
    def backward(ctx, g):
        x_cl, weight, x_packed = ctx.saved_tensors
        NB, h, w, Cin = x_cl.shape
        Cout = weight.shape[0]
        g = g.float().contiguous()
        g_x = g_w = None
        if ctx.needs_input_grad[1]:
            g_w = torch.empty(weight.shape, dtype=torch.float32, device=g.device)
            _lib.spike_conv_wgrad(g, x_cl, g_w, NB, Cin, Cout, h, w, x_packed=None if x_packed is None else x_packed.contiguous())
        how_x = 'none'
        if ctx.needs_input_grad[0] and ctx.ecfg.CONV_DGRAD_MFMA and _lib.conv_s2_dgrad_supported(Cin, Cout, 5, 2, 2):
            g_x = conv_s2_dgrad_cl(g, weight, h, w)
            how_x = 'conv_s2_dgrad6_mfma'
        elif ctx.needs_input_grad[0]:
            how_x = 'miopen'
            x_meta = torch.empty((NB, Cin, h, w), dtype=torch.float32, device=g.device, memory_format=torch.channels_last)
            g_x = torch.ops.aten.convolution_backward(
                g.permute(0, 3, 1, 2), x_meta, weight.contiguous(memory_format=torch.channels_last), None,
                [2, 2], [2, 2], [1, 1], False, [0, 0], 1, [True, False, False])[0].permute(0, 2, 3, 1)
            if not g_x.is_contiguous():
                g_x = g_x.contiguous()
        _note('synapse_bwd', f'g_x: {how_x}; g_w: ' + ('spike_conv_wgrad3_mfma' if ctx.needs_input_grad[1] else 'none'), ctx.site)
        return g_x, g_w, None


def spike_conv_fwd_applies(conv, device, dtype=torch.float32) -> bool:
    """True when `conv` on a spike NHWC array runs its FORWARD through ss_spike_conv_fwd_f32 (and can therefore take a packed-only input)."""
    import torch.nn as nn
    if x16_mode(device) is not None:
        return conv16_kind(conv, True) == 's2'
    return bool(_cfg().SPIKE_CONV_FWD_MFMA and _cfg().SPIKE_CONV_WGRAD_MFMA and isinstance(conv, nn.Conv2d) and device.type == 'cuda' and dtype == torch.float32
                and not torch.is_autocast_enabled('cuda') and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1)
                and conv.kernel_size == (5, 5) and conv.stride == (2, 2) and conv.padding == (2, 2)
                and _lib.spike_conv_fwd_supported(conv.in_channels, conv.out_channels, 5, 2, 2)
                and _lib.spike_conv_wgrad_supported(conv.in_channels, conv.out_channels, 5, 2, 2))


def spike_conv_wgrad_cl(x_cl: torch.Tensor, conv, x_packed: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """conv on a spike NHWC array through _SpikeConvWgradCL, or None when it does not apply (caller then uses the plain MIOpen convolution)."""
    import torch.nn as nn
    if not (_cfg().SPIKE_CONV_WGRAD_MFMA and isinstance(conv, nn.Conv2d) and x_cl.is_cuda and x_cl.dtype == torch.float32
            and not torch.is_autocast_enabled('cuda') and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1)
            and conv.kernel_size == (5, 5) and conv.stride == (2, 2) and conv.padding == (2, 2)
            and (x_cl.stride(-1) == 1 or (x_packed is not None and spike_conv_fwd_applies(conv, x_cl.device, x_cl.dtype)))   # anchor: packed-only input
            and _lib.spike_conv_wgrad_supported(conv.in_channels, conv.out_channels, 5, 2, 2)):
        return None
    return _SpikeConvWgradCL.apply(x_cl, conv.weight, x_packed)




class _DenseConvS1CL(torch.autograd.Function):
    """The first encoder layer (/root/reference/network/SNN_models.py:75-79: Conv2d(4 | 2, 32, 5, stride 1, padding 2) on the event-voxel input)
    on an NHWC array: forward and weight gradient = six-term bf16 MFMA contractions (no precondition on the input values); an input gradient, if
    ever asked for, = MIOpen's convolution backward."""

    @staticmethod
    @_fwd32
    def forward(ctx, x_cl, weight):
        ctx.ecfg, ctx.site = _cfg(), _site()          # the engine configuration and plan site of THIS forward: the backward dispatches from them
        x_cl = x_cl.contiguous()
        NB, h, w, Cin = x_cl.shape
        Cout = weight.shape[0]
        y = torch.empty((NB, h, w, Cout), dtype=torch.float32, device=x_cl.device)
        e0 = TIMER.start()
        _lib.dense_conv_s1_fwd(x_cl, weight.detach().float().contiguous(), y, NB, Cin, Cout, h, w)
        TIMER.stop(e0, 'dense_conv_s1_fwd', 4 * (y.numel() + x_cl.numel()), y.numel())
        _note('synapse_fwd', 'dense_conv_s1_fwd6_mfma')
        ctx.save_for_backward(x_cl, weight)
        return y

    @staticmethod
    @_bwd32
    def backward(ctx, g):
        x_cl, weight = ctx.saved_tensors
        g = g.float().contiguous()
        NB, h, w, Cin = x_cl.shape
        own_w = bool(ctx.needs_input_grad[1]) and ctx.ecfg.DENSE_CONV_S1_WGRAD_MFMA and _lib.dense_conv_s1_wgrad_supported(Cin, weight.shape[0], 5, 1, 2)
        gw_own = None
        if own_w:                                            # six-term MFMA contraction over the pixels (was MIOpen's igemm_wrw: the step's last MIOpen call)
            gw_own = torch.empty(weight.shape, dtype=torch.float32, device=g.device)
            e0 = TIMER.start()
            _lib.dense_conv_s1_wgrad(g, x_cl, gw_own, NB, Cin, weight.shape[0], h, w)
            TIMER.stop(e0, 'dense_conv_s1_wgrad', 4 * (g.numel() + x_cl.numel()), g.numel())
            if not ctx.needs_input_grad[0]:
                _note('synapse_bwd', 'g_x: none; g_w: dense_conv_s1_wgrad6_mfma', ctx.site)
                return None, gw_own
        gx, gw, _ = torch.ops.aten.convolution_backward(
            g.permute(0, 3, 1, 2), x_cl.permute(0, 3, 1, 2), weight.contiguous(memory_format=torch.channels_last), None,
            [1, 1], [2, 2], [1, 1], False, [0, 0], 1, [bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1]) and not own_w, False])
        if own_w:
            gw = gw_own
        _note('synapse_bwd', 'g_x: ' + ('miopen' if ctx.needs_input_grad[0] else 'none') + '; g_w: ' + ('dense_conv_s1_wgrad6_mfma' if own_w else 'miopen'), ctx.site)
        if gx is not None:
            gx = gx.permute(0, 2, 3, 1)
            gx = gx if gx.is_contiguous() else gx.contiguous()
        return gx, gw


def dense_conv_s1_cl(x_cl: torch.Tensor, conv) -> Optional[torch.Tensor]:
    """conv (the first encoder layer's geometry) on a dense NHWC array through _DenseConvS1CL, or None when it does not apply."""
    import torch.nn as nn
    if not (_cfg().DENSE_CONV_S1_MFMA and isinstance(conv, nn.Conv2d) and x_cl.is_cuda and x_cl.dtype == torch.float32
            and not torch.is_autocast_enabled('cuda') and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1)
            and conv.kernel_size == (5, 5) and conv.stride == (1, 1) and conv.padding == (2, 2)
            and _lib.dense_conv_s1_fwd_supported(conv.in_channels, conv.out_channels, 5, 1, 2)):
        return None
    return _DenseConvS1CL.apply(x_cl, conv.weight)




# ----------------------------------------------------------------------------------------------------------
# 16-bit activation modes on the engine's own kernels (round 5; EngineConfig.X16_OWN_KERNELS, include/ss_neuron.h ABI 9): the encoder / bottleneck synapses.
# Called with autocast DISABLED (conv_cl16): every dtype below is explicit.  adt = torch.float16 | torch.bfloat16.
# ----------------------------------------------------------------------------------------------------------
class _DenseConvS1CL16(torch.autograd.Function):
    """The first encoder layer (/root/reference/network/SNN_models.py:75-79) in a 16-bit activation mode: x_cl fp32 NHWC (the event-voxel input) ->
    y in adt; input and weight rounded once to adt inside the kernels (autocast's semantics for this convolution), fp32 accumulation, fp32 weight gradient."""

    @staticmethod
    def forward(ctx, x_cl, weight, adt):
        ctx.ecfg, ctx.site = _cfg(), _site()
        x_cl = x_cl.float().contiguous()
        NB, h, w, Cin = x_cl.shape
        Cout = weight.shape[0]
        y = torch.empty((NB, h, w, Cout), dtype=adt, device=x_cl.device)
        e0 = TIMER.start()
        _lib.dense_conv_s1_fwd_x16(x_cl, weight.detach().float().contiguous(), y, NB, Cin, Cout, h, w)
        TIMER.stop(e0, 'dense_conv_s1_fwd', 2 * y.numel() + 4 * x_cl.numel(), y.numel())
        _note('synapse_fwd', 'dense_conv_s1_fwd1_mfma_x16')
        ctx.save_for_backward(x_cl, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        x_cl, weight = ctx.saved_tensors
        NB, h, w, Cin = x_cl.shape
        g = g.contiguous()
        gw = None
        if ctx.needs_input_grad[1]:
            gw = torch.empty(weight.shape, dtype=torch.float32, device=g.device)
            e0 = TIMER.start()
            _lib.dense_conv_s1_wgrad_x16(g, x_cl, gw, NB, Cin, weight.shape[0], h, w)
            TIMER.stop(e0, 'dense_conv_s1_wgrad', 2 * g.numel() + 4 * x_cl.numel(), g.numel())
        gx = None
        if ctx.needs_input_grad[0]:          # (never in the shipped models: the network input needs no gradient)
            gx = torch.ops.aten.convolution_backward(g.float().permute(0, 3, 1, 2), x_cl.permute(0, 3, 1, 2), weight.float().contiguous(memory_format=torch.channels_last),
                                                     None, [1, 1], [2, 2], [1, 1], False, [0, 0], 1, [True, False, False])[0].permute(0, 2, 3, 1).contiguous()
        _note('synapse_bwd', 'g_x: ' + ('miopen' if gx is not None else 'none') + '; g_w: ' + ('dense_conv_s1_wgrad1_mfma_x16' if gw is not None else 'none'), ctx.site)
        return gx, gw, None


class _SpikeConvS2CL16(torch.autograd.Function):
    """conv1 / conv2 (/root/reference/network/SNN_models.py:80-90) in a 16-bit activation mode: forward, weight gradient and data gradient as single-term MFMA
    implicit GEMMs on 16-bit I/O; the spike input is read 2-bit packed when the producer wrote it (x_cl may then be a data-less anchor)."""

    @staticmethod
    def forward(ctx, x_cl, weight, x_packed, adt):
        ctx.ecfg, ctx.site = _cfg(), _site()
        NB, h, w, Cin = x_cl.shape
        Cout = weight.shape[0]
        if x_packed is None:
            x_cl = x_cl.to(adt).contiguous()                      # spike counts: exact in either format
        y = torch.empty((NB, (h - 1) // 2 + 1, (w - 1) // 2 + 1, Cout), dtype=adt, device=x_cl.device)
        e0 = TIMER.start()
        _lib.spike_conv_fwd_x16(None if x_packed is not None else x_cl, None if x_packed is None else x_packed.contiguous(),
                                weight.detach().float().contiguous(), y, NB, Cin, Cout, h, w)
        TIMER.stop(e0, 'spike_conv_fwd', 2 * y.numel() + (x_cl.numel() // 4 if x_packed is not None else 2 * x_cl.numel()), y.numel())
        _note('synapse_fwd', 'spike_conv_fwd1_mfma_x16' + ('(packed in)' if x_packed is not None else ''))
        ctx.save_for_backward(x_cl, weight, x_packed)
        ctx.adt = adt
        return y

    @staticmethod
    def backward(ctx, g):
        x_cl, weight, x_packed = ctx.saved_tensors
        NB, h, w, Cin = x_cl.shape
        Cout = weight.shape[0]
        g = g.to(ctx.adt).contiguous()
        g_x = g_w = None
        if ctx.needs_input_grad[1]:
            g_w = torch.empty(weight.shape, dtype=torch.float32, device=g.device)
            _lib.spike_conv_wgrad_x16(g, None if x_packed is not None else x_cl, g_w, NB, Cin, Cout, h, w, x_packed=None if x_packed is None else x_packed.contiguous())
        if ctx.needs_input_grad[0]:
            g_x = torch.empty((NB, h, w, Cin), dtype=ctx.adt, device=g.device)
            e0 = TIMER.start()
            _lib.conv_s2_dgrad_x16(g, weight.detach().float().contiguous(), g_x, NB, Cin, Cout, h, w)
            TIMER.stop(e0, 'conv_s2_dgrad', 2 * (g.numel() + g_x.numel()), g_x.numel())
        _note('synapse_bwd', 'g_x: ' + ('conv_s2_dgrad1_mfma_x16' if g_x is not None else 'none') + '; g_w: ' + ('spike_conv_wgrad1_mfma_x16' if g_w is not None else 'none'), ctx.site)
        return g_x, g_w, None, None


class _SpikeConvGemmCL16(torch.autograd.Function):
    """conv3 / conv4 and the SEW bottleneck convs (/root/reference/network/SNN_models.py:91-107, blocks.py:145-154) in a 16-bit activation mode:
    forward : A = im2col(x) in adt (from the packed spikes when the producer wrote them; kept for backward), y = A @ W_adt — ONE single-term library GEMM with the
              adt output written by its epilogue (the fp32 mode: three terms + a summing pass);
    wgrad   : g_W = A^T @ g, both adt, fp32 accumulation AND fp32 output (split-K slices summed in fp32);
    dgrad   : stride-2 5 x 5: ss_conv_s2_dgrad_x16; 3 x 3 stride 1: the same convolution of g with the flipped, transposed kernel — im2col of the dense adt
              gradient (ss_im2col_cl_x16) + one single-term GEMM (the fp32 mode's Winograd transform would round its 16-bit intermediates)."""

    @staticmethod
    def forward(ctx, x_cl, weight, stride, pad, x_packed, adt):
        ctx.ecfg, ctx.site = _cfg(), _site()
        NB, h, w, Cin = x_cl.shape
        Cout, _, k, _ = weight.shape
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        M, K = NB * ho * wo, k * k * Cin
        A = torch.empty((M, K), dtype=adt, device=x_cl.device)
        if x_packed is not None:
            _lib.im2col_cl_packed_x16(x_packed.contiguous(), A, NB, h, w, Cin, k, stride, pad, ho, wo)
        else:
            _lib.im2col_cl_x16(x_cl.to(adt).contiguous(), A, NB, h, w, Cin, k, stride, pad, ho, wo)
        Wt = weight.detach().to(adt).permute(2, 3, 1, 0).reshape(K, Cout)            # rows (ky, kx, c): the im2col column order; rounded once
        y = torch.mm(A, Wt).view(NB, ho, wo, Cout)
        _note('synapse_fwd', 'im2col' + ('(packed in)' if x_packed is not None else '') + '+gemm1_x16')
        ctx.save_for_backward(A, weight)
        ctx.geom = (NB, h, w, Cin, Cout, k, stride, pad, ho, wo)
        ctx.adt = adt
        return y

    @staticmethod
    def backward(ctx, g):
        A, weight = ctx.saved_tensors
        NB, h, w, Cin, Cout, k, stride, pad, ho, wo = ctx.geom
        adt = ctx.adt
        M, K = NB * ho * wo, k * k * Cin
        g = g.to(adt).contiguous()
        g_x = g_w = None
        if ctx.needs_input_grad[1]:
            S = next(d for d in (_SPIKE_CONV_WGRAD_SPLIT, 4, 2, 1) if M % d == 0)
            g2 = g.view(M, Cout)
            if S > 1:
                gw = torch.bmm(A.view(S, M // S, K).transpose(1, 2), g2.view(S, M // S, Cout), out_dtype=torch.float32).sum(0)
            else:
                gw = torch.mm(A.t(), g2, out_dtype=torch.float32)
            g_w = gw.view(k, k, Cin, Cout).permute(3, 2, 0, 1).contiguous()
        how_x = 'none'
        if ctx.needs_input_grad[0] and k == 5 and _lib.conv_s2_dgrad_supported(Cin, Cout, k, stride, pad):
            g_x = torch.empty((NB, h, w, Cin), dtype=adt, device=g.device)
            e0 = TIMER.start()
            _lib.conv_s2_dgrad_x16(g, weight.detach().float().contiguous(), g_x, NB, Cin, Cout, h, w)
            TIMER.stop(e0, 'conv_s2_dgrad', 2 * (g.numel() + g_x.numel()), g_x.numel())
            how_x = 'conv_s2_dgrad1_mfma_x16'
        elif ctx.needs_input_grad[0] and stride == 1 and 2 * pad == k - 1 and Cout % 8 == 0:
            Ag = torch.empty((NB * h * w, k * k * Cout), dtype=adt, device=g.device)
            _lib.im2col_cl_x16(g, Ag, NB, ho, wo, Cout, k, 1, pad, h, w)
            Wd = weight.detach().to(adt).flip(2, 3).permute(2, 3, 0, 1).reshape(k * k * Cout, Cin)     # [(jy, jx, co)][ci] = W[co][ci][k-1-jy][k-1-jx]
            g_x = torch.mm(Ag, Wd).view(NB, h, w, Cin)
            how_x = 'im2col+gemm1_x16'
        elif ctx.needs_input_grad[0]:
            how_x = 'miopen'
            x_meta = torch.empty((NB, Cin, h, w), dtype=torch.float32, device=g.device, memory_format=torch.channels_last)
            g_x = torch.ops.aten.convolution_backward(g.float().permute(0, 3, 1, 2), x_meta, weight.float().contiguous(memory_format=torch.channels_last), None,
                                                      [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])[0].permute(0, 2, 3, 1).contiguous().to(adt)
        _note('synapse_bwd', f'g_x: {how_x}; g_w: ' + ('gemm1_x16(fp32 out)' if g_w is not None else 'none'), ctx.site)
        return g_x, g_w, None, None, None, None


def conv16_kind(conv, spikes_in: bool):
    """Which own 16-bit kernel family runs `conv` (an nn.Conv2d of the encoder / bottleneck) in the 16-bit activation modes: 's1' (first layer), 's2'
    (conv1 / conv2 implicit GEMMs), 'gemm' (im2col + single-term GEMM: conv3, conv4, the bottleneck), or None (the caller keeps MIOpen under autocast)."""
    import torch.nn as nn
    if not (isinstance(conv, nn.Conv2d) and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1) and not isinstance(conv.padding, str)):
        return None
    ks, st, pd = conv.kernel_size, conv.stride, conv.padding
    if ks[0] != ks[1] or st[0] != st[1] or pd[0] != pd[1]:
        return None
    k, s_, p_ = ks[0], st[0], pd[0]
    if not spikes_in:
        return 's1' if _lib.dense_conv_s1_fwd_supported(conv.in_channels, conv.out_channels, k, s_, p_) and _lib.dense_conv_s1_wgrad_supported(conv.in_channels, conv.out_channels, k, s_, p_) else None
    if (k, s_, p_) == (5, 2, 2) and _lib.spike_conv_fwd_supported(conv.in_channels, conv.out_channels, 5, 2, 2) and _lib.spike_conv_wgrad_supported(conv.in_channels, conv.out_channels, 5, 2, 2) \
            and _lib.conv_s2_dgrad_supported(conv.in_channels, conv.out_channels, 5, 2, 2):
        return 's2'
    if conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0 and conv.in_channels >= _SPIKE_CONV_MIN_CIN:
        return 'gemm'
    return None


def conv_cl16(conv, x_cl: torch.Tensor, spikes_in: bool, x_packed: Optional[torch.Tensor], adt) -> Optional[torch.Tensor]:
    """conv on an NHWC array in the 16-bit activation mode `adt` through the engine's own kernels, or None when no family applies."""
    kind = conv16_kind(conv, spikes_in)
    if kind is None:
        return None
    with torch.autocast('cuda', enabled=False):
        if kind == 's1':
            return _DenseConvS1CL16.apply(x_cl, conv.weight, adt)
        if kind == 's2':
            return _SpikeConvS2CL16.apply(x_cl, conv.weight, x_packed, adt)
        return _SpikeConvGemmCL16.apply(x_cl, conv.weight, conv.stride[0], conv.padding[0], x_packed, adt)


def spike_conv_applies(conv, device, dtype=torch.float32) -> bool:
    """True when `conv` (nn.Conv2d) on a spike NHWC array of `dtype` on `device` runs through _SpikeConvCL (exact bf16x3 GEMM form)."""
    import torch.nn as nn
    if not isinstance(conv, nn.Conv2d):
        return False
    if x16_mode(device) is not None:                       # 16-bit activation modes on own kernels: the im2col + single-term GEMM family reads packed spikes
        return conv16_kind(conv, True) == 'gemm'
    k = conv.kernel_size[0]
    return bool(_cfg().EXACT_SPLIT_GEMM and device.type == 'cuda' and dtype == torch.float32 and not torch.is_autocast_enabled('cuda')
                and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1) and conv.kernel_size == (k, k)
                and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1] and not isinstance(conv.padding, str)
                and conv.in_channels % 8 == 0 and conv.out_channels % 4 == 0 and conv.in_channels >= _SPIKE_CONV_MIN_CIN)


def spike_conv_cl(x_cl: torch.Tensor, conv, x_packed: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """conv (nn.Conv2d, square kernel, no bias / groups / dilation) on a spike NHWC array through _SpikeConvCL, or None when the
    exact-split form does not apply (caller then uses the MIOpen convolution).  x_packed: the input as a packed spike tensor."""
    if not spike_conv_applies(conv, x_cl.device, x_cl.dtype):
        return None
    return _SpikeConvCL.apply(x_cl, conv.weight, conv.stride[0], conv.padding[0], x_packed)




def stage_takes_packed_copy(stage, h: int, w: int, device) -> bool:
    """True when the decoder stage `stage` (an NNConvUpsampling with k = 5) reads its input as 2-bit packed spikes on an h x w source map: its sub-pixel forward
    (ss_upconv_sub_fwd_*) and box-sum weight gradient read them when they are given them — the producer then writes the packed form (beside its dense output, or
    instead of it when its other consumer reads packed spikes too) and the stage never touches a dense tensor (32 x fewer input bytes, no conversion while
    staging).  Should the backward's geometry check still send the stage down the g_P forms, _UpConvProjectedCL unpacks: correct, just not free."""
    conv = stage.up[1]
    k = conv.kernel_size[0]
    if not (_cfg().PACK_SPIKES and _cfg().SUB_FWD and _cfg().EXACT_SPLIT_GEMM and k == 5 and torch.device(device).type == 'cuda'):
        return False
    Hu, Wu = stage.up[0].size
    return bool(stage_plan(stage._tables(h, w, device), conv.in_channels, conv.out_channels, k, h, w, Hu - k + 1, Wu - k + 1)['sub_fwd'])


class _UnpackLastStep(torch.autograd.Function):
    """The last time step of a packed-only spike sequence as a dense tensor (what the models return as `out_add*`; the reference returns the dense
    tensors of its single step).  Backward (only when a loss term reads the spikes, e.g. Total_Loss(penalize_spikes=True)): a dense gradient for
    the whole sequence, zero except at the last step."""

    @staticmethod
    def forward(ctx, anchor, packed):
        T = anchor.shape[0]
        ctx.shape = tuple(anchor.shape)
        return unpack_dense(packed[T - 1:T], anchor.shape[1:], anchor.dtype)

    @staticmethod
    def backward(ctx, g):
        g_seq = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        g_seq[-1] = g
        return g_seq, None


def unpack_last_step(anchor: torch.Tensor, packed: torch.Tensor) -> torch.Tensor:
    return _UnpackLastStep.apply(anchor, packed)


def upconv_projected_cl(x_cl: torch.Tensor, weight: torch.Tensor, bias, tables, k: int, H: int, W: int, spikes_in: bool = False,
                        lowrank_grad: bool = False, x_packed: Optional[torch.Tensor] = None):
    """Channels-last NNConvUpsampling: x_cl [NB, h, w, C_in] (plain contiguous NHWC array) -> [NB, H, W, C_out].
    spikes_in: the caller guarantees x_cl holds spike counts (small integers, exact in bf16) — enables the exact bf16x3 projection."""
    amp = x_cl.is_cuda and torch.is_autocast_enabled('cuda')
    adt = torch.get_autocast_dtype('cuda') if amp else None
    lowp = amp and adt == torch.bfloat16
    lowp_bwd = amp and adt in (torch.bfloat16, torch.float16)
    # decoder stages (k = 5) hand 16-bit activations to their neuron layer under 16-bit autocast; the heads (k = 3) feed the fp32 I-pool
    act_dtype = adt if (lowp_bwd and k == 5) else None
    own16 = x16_mode(x_cl.device)
    if own16 is not None and k == 3 and x_packed is not None:
        # a prediction head on packed spikes in a 16-bit mode: the packed kernels with the exact fp32 weight, fp32 P and output (the head feeds the fp32 I-pool)
        lowp = lowp_bwd = False
    with torch.autocast('cuda', enabled=False):      # dtypes are handled explicitly inside (no blanket casts in either direction)
        # fp16 mode on the engine's own kernels: the library projection (deconv4, heads 3 / 4) on ONE fp16 weight term — the mode's definition everywhere else —
        # instead of the exact bf16x3 form (a [X X X] copy + three times the products); the packed heads keep the exact fp32 weight
        fwd16 = bool(own16 == torch.float16 and adt == torch.float16 and not (k == 3 and x_packed is not None))
        return _UpConvProjectedCL.apply(x_cl, weight, bias, tables, StageOpts(k=k, H=H, W=W, lowp=lowp, spikes_in=spikes_in, lowp_bwd=lowp_bwd, act_dtype=act_dtype,
                                                                             lowrank_grad=lowrank_grad, own16=own16, fwd16=fwd16), x_packed)


_guard_module(__name__, 'fused')
