"""GPU tests of the round-6 forms of the 16-bit fused neuron kernels (stereospike_amd/csrc/ss_neuron16_v2.hpp, include/ss_neuron.h ABI 10) and of the
unwritten ("lazy") membrane.  The kernels keep the arithmetic of round 5 — every comparison below is bit for bit:

  * ss_neuron_fwd_ex on 16-bit activations with packed I/O (neuron_fwd16_pk8_kernel: 8 neurons per lane, bit-parallel codes / counters) against the
    numpy definition oracle/np_x16.py, incl. shapes with an odd number of lane pairs and inf / NaN / signed-zero inputs (the reset select is claimed
    equal to (1 - z) * h + z * v_reset on ALL of them);
  * v_last == NULL in the packed forms: same spikes / counters, and fused.membrane_after == the membrane the writing call produces;
  * ss_neuron_bwd_fork_lr_x16 on the segmented kernel (wavefront-staged pair, Newton reciprocal): partial last wavefronts, every (T, C) it takes, the
    exact redo when a surrogate denominator reaches 2^126 (v_th = inf, overflowing inputs);
  * EngineConfig.LAZY_MEMBRANE: a training step with the membrane left unwritten == the same step with it written (loss, every gradient, every node.v).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DTS = [torch.bfloat16, torch.float16]
IDS = ['bf16', 'f16']
NAME = {torch.bfloat16: 'bf16', torch.float16: 'f16'}
KIND = {'IF': 0, 'LIF': 1, 'PLIF': 2}


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _from_bits(b, dt):
    return torch.from_numpy(b.view(np.int16)).to(DEV).view(dt)


@pytest.mark.parametrize('dt', DTS, ids=IDS)
@pytest.mark.parametrize('kind', ['IF', 'LIF', 'PLIF'])
@pytest.mark.parametrize('T,N', [(5, 16), (10, 48), (4, 16 * 129), (8, 16 * 1000 + 32), (5, 32 * 33 * 44), (2, 80), (1, 16 * 7)])
def test_fwd16_packed_forms_vs_numpy_definition(dt, kind, T, N):
    """spikes (packed), counters, v_last of the packed 16-bit forward == oracle/np_x16.neuron_fwd on the same bit patterns; special values included."""
    from stereospike_amd import _lib
    from oracle import np_pack, np_x16
    rng = np.random.default_rng(T * 1000 + N)
    x = (rng.standard_normal((T, N)) * 0.12).astype(np.float32)
    xb = _bits(torch.tensor(x).to(dt))
    # special patterns in the first neurons: +-0, +-inf, NaN, the largest finite value, a denormal
    sp = {'f16': [0x0000, 0x8000, 0x7c00, 0xfc00, 0x7e00, 0x7bff, 0x0001, 0xfbff], 'bf16': [0x0000, 0x8000, 0x7f80, 0xff80, 0x7fc0, 0x7f7f, 0x0001, 0xff7f]}[NAME[dt]]
    for t in range(T):
        for j, b in enumerate(sp):
            xb[t, (j + t) % 16] = b
    skip = rng.integers(0, 3, (T, N)).astype(np.float32)
    v0 = (rng.standard_normal(N) * 0.4).astype(np.float32)
    v0[:4] = [0.0, -0.0, np.inf, -1.0]
    kw = dict(kind=kind, scale=7.5, tau=2.0, k=np.float32(0.3) if kind == 'PLIF' else None, v_th=1.0, v_reset=0.0)
    xd = _from_bits(xb, dt)
    kd = torch.tensor([0.3], device=DEV) if kind == 'PLIF' else None
    for use_skip, use_v0 in ((False, False), (True, True)):
        ref = np_x16.neuron_fwd(xb, NAME[dt], v_init=v0 if use_v0 else None, skip_bits=_bits(torch.tensor(skip).to(dt)) if use_skip else None, **kw)
        want_codes = np.minimum(np_x16.widen(ref['out'], NAME[dt]), 3).astype(np.float32)
        sk_pk = torch.from_numpy(np_pack.pack(skip.reshape(T, N)).view(np.int32)).to(DEV) if use_skip else None
        v0d = torch.tensor(v0, device=DEV) if use_v0 else None
        outs = {}
        for with_v in (True, False):
            out_pk = torch.full((T, N // 16), -1, dtype=torch.int32, device=DEV)
            v_last = torch.full((N,), float('nan'), device=DEV) if with_v else None
            nnz = torch.zeros(2, dtype=torch.int64, device=DEV)
            ws = torch.empty(_lib.cnt_ws_words(N), dtype=torch.int32, device=DEV)
            _lib.neuron_fwd_ex(xd, v0d, None, sk_pk, None, out_pk, None, v_last, nnz, ws, T, N, 7.5, KIND[kind], 2.0, kd, 1.0, 0.0)
            outs[with_v] = (out_pk, v_last, nnz)
            got = np_pack.unpack(out_pk.cpu().numpy().view(np.uint32).reshape(T, N // 16)).reshape(T, N)
            nanmask = np.isnan(want_codes)
            assert np.array_equal(got[~nanmask], want_codes[~nanmask]), (use_skip, with_v)
            z = (ref['h'] - np.float32(1.0)) >= 0
            assert int(nnz[0]) == int(z.sum()) and int(nnz[1]) == int((got != 0).sum())
        assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][2], outs[False][2])
        vl = outs[True][1].cpu().numpy()
        vnan = np.isnan(ref['v_last'])               # (a NaN is a NaN: its sign / payload differ between the host's and the GPU's invalid-operation result)
        assert np.array_equal(np.isnan(vl), vnan) and np.array_equal(vl.view(np.uint32)[~vnan], ref['v_last'].view(np.uint32)[~vnan]), 'v_last bit pattern'
        # the dense copy beside the packed output (layers with a consumer that cannot read packed spikes)
        out_d = torch.empty_like(xd)
        out_pk2 = torch.empty((T, N // 16), dtype=torch.int32, device=DEV)
        _lib.neuron_fwd_ex(xd, v0d, None, sk_pk, out_d, out_pk2, None, None, None, None, T, N, 7.5, KIND[kind], 2.0, kd, 1.0, 0.0)
        assert torch.equal(out_pk2, outs[True][0])
        assert np.array_equal(out_d.float().cpu().numpy()[~nanmask], want_codes[~nanmask])


@pytest.mark.parametrize('dt', [torch.float32] + DTS, ids=['f32'] + IDS)
def test_membrane_after_equals_written_membrane(dt):
    """fused.membrane_after (the on-demand recomputation of an unwritten membrane) == v_last of the call that writes it."""
    from stereospike_amd import _lib
    from stereospike_amd.fused import NeuronCfg, membrane_after
    T, N = 5, 16 * 321
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn(T, N, device=DEV, generator=g) * 0.2).to(dt)
    v0 = torch.randn(N, device=DEV, generator=g) * 0.3
    k = torch.tensor(0.4, device=DEV)
    for kind in (0, 2):
        cfg = NeuronCfg(kind=kind, scale=9.0, v_th=1.0, v_reset=0.0, surrogate=_lib.SG_ATAN, alpha=2.0)
        out_pk = torch.empty((T, N // 16), dtype=torch.int32, device=DEV)
        v_last = torch.empty(N, device=DEV)
        _lib.neuron_fwd_ex(x, v0, None, None, None, out_pk, None, v_last, None, None, T, N, cfg.scale, kind, cfg.tau, k if kind == 2 else None, 1.0, 0.0)
        got = membrane_after(x, cfg, v0, k if kind == 2 else None)
        assert torch.equal(got.view(torch.int32), v_last.view(torch.int32))
    with pytest.raises(_lib.SSNeuronError):        # v_last may be NULL in the packed forms only
        _lib.neuron_fwd_ex(x, None, None, None, torch.empty_like(x), None, None, None, None, None, T, N, 9.0, 0, 2.0, None, 1.0, 0.0)


@pytest.mark.parametrize('dt', DTS, ids=IDS)
@pytest.mark.parametrize('sg', [0, 1], ids=['atan', 'sigmoid'])
@pytest.mark.parametrize('T,rows,C', [(10, 2307, 32), (5, 89960 // 8 + 3, 32), (8, 19, 64), (4, 7, 128), (10, 5, 256), (5, 66, 8), (10, 1, 32)])
def test_lr_x16_segmented_kernel_partial_wavefronts_and_special_values(dt, sg, T, rows, C):
    """The segmented low-rank backward on shapes whose last wavefront is partial (rows not a multiple of the pixels per wavefront, N / 4 not a multiple of
    64) and with x values that drive the ATan denominator to inf (the exact redo pass): == the fp32 recompute backward on the widened operands."""
    from stereospike_amd import _lib
    from oracle import np_lowrank
    N = rows * C
    assert _lib.neuron_bwd_fork_lr_x16_supported(T, N, C, 9)
    rng = np.random.default_rng(N + 7 * T + sg)
    x = (rng.standard_normal((T, N)) * 0.25).astype(np.float32)
    big = 6.0e4 if dt == torch.float16 else 3.0e38
    x[0, :: max(1, N // 50)] = big                      # x * scale overflows: h = inf, the surrogate denominator inf -> the exact redo
    x[T - 1, 1:: max(1, N // 37)] = -big
    x = torch.tensor(x, device=DEV).to(dt)
    g1 = torch.tensor(rng.standard_normal((T, N)).astype(np.float32), device=DEV).to(dt)
    lr_p = (rng.standard_normal((T, rows, 9)) * 2).astype(np.float32)
    lr_w = rng.standard_normal((9, C)).astype(np.float32)
    g2 = torch.tensor(np_lowrank.head_input_gradient(lr_p, lr_w).reshape(T, N), device=DEV)
    P, Wl = torch.tensor(lr_p, device=DEV), torch.tensor(lr_w, device=DEV)
    for v_th, scale in ((1.0, 1e30 if dt == torch.bfloat16 else 1e35), (float('inf'), 7.5), (1.0, 7.5)):
        args = (T, N, scale, 0, 2.0, None, v_th, 0.0, sg, 2.0, True)
        gsum32 = g1.float() + g2
        gx_a = torch.empty(T, N, device=DEV)
        _lib.neuron_bwd_rc(gsum32, None, x.float(), None, gx_a, None, None, None, *args)
        gx_b, gsum = torch.full_like(x, float('nan')), torch.full_like(x, float('nan'))
        _lib.neuron_bwd_fork_lr_x16(g1, P, Wl, gsum, None, x, None, gx_b, None, None, None, *args)
        a, b = gx_a.to(dt).view(torch.int16), gx_b.view(torch.int16)
        nan_a = torch.isnan(gx_a)
        assert torch.equal(a[~nan_a], b[~nan_a]) and bool(torch.isnan(gx_b.float()[nan_a]).all()), (v_th, scale)
        assert torch.equal(gsum, gsum32.to(dt))
        gx_c, gx_d = torch.empty(T, N, device=DEV), torch.full_like(x, float('nan'))
        _lib.neuron_bwd_rc(g2, None, x.float(), None, gx_c, None, None, None, *args)
        _lib.neuron_bwd_fork_lr_x16(None, P, Wl, None, None, x, None, gx_d, None, None, None, *args)
        nan_c = torch.isnan(gx_c)
        assert torch.equal(gx_c.to(dt).view(torch.int16)[~nan_c], gx_d.view(torch.int16)[~nan_c])


@pytest.mark.parametrize('amp', [None, torch.bfloat16], ids=['f32', 'bf16'])
def test_lazy_membrane_training_step_equals_written_membrane(amp):
    """EngineConfig.LAZY_MEMBRANE: the training pass leaves the membranes unwritten where the kernel form allows; loss, every gradient and — recomputed on
    first access — every node.v equal the pass that writes them, bit for bit; a second pass WITHOUT reset continues from the recomputed membranes."""
    from stereospike_amd.clock_driven import functional, neuron, surrogate
    from stereospike_amd.config import EngineConfig
    from stereospike_amd.engine import synthetic_batch
    from stereospike_amd.network import SNN_models as S
    from stereospike_amd.network.loss import Total_Loss
    res = {}
    x, gt = synthetic_batch(2, 4, H=64, W=80, seed=11, device=DEV, lam=0.3)
    for lazy in (True, False):
        torch.manual_seed(3)
        cfg = EngineConfig.default().replace(LAZY_MEMBRANE=lazy)
        net = S.StereoSpike(surrogate_function=surrogate.ATan(), detach_reset=True, multiply_factor=10., input_size=(64, 80), config=cfg).to(DEV)
        functional.reset_net(net)
        with torch.autocast('cuda', dtype=amp, enabled=amp is not None):
            pred, spk = net.forward_sequence(x)
            loss = Total_Loss(alpha=0.5, scale_weights=(1., 1., 1., 1.), penalize_spikes=False)(pred, gt, spk)
        nodes = [m for m in net.modules() if isinstance(m, neuron.BaseNode)]
        n_lazy = sum(isinstance(m._v, neuron._LazyMembrane) for m in nodes)
        assert (n_lazy > 0) == lazy, n_lazy
        loss.backward()
        grads = [p.grad.clone() for p in net.parameters()]
        vs = [m.v.detach().clone() for m in nodes]       # (first access recomputes the unwritten ones; layers on a non-packed kernel form wrote theirs)
        net.detach()
        with torch.no_grad(), torch.autocast('cuda', dtype=amp, enabled=amp is not None):
            pred2, _ = net.forward_sequence(x)              # no reset: continues from the membranes
        res[lazy] = (loss.detach().clone(), grads, vs, [p.clone() for p in pred2], n_lazy)
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0])
    assert all(torch.equal(p, q) for p, q in zip(a[1], b[1]))
    assert all(torch.equal(p.view(torch.int32), q.view(torch.int32)) for p, q in zip(a[2], b[2]))
    assert all(torch.equal(p, q) for p, q in zip(a[3], b[3]))
