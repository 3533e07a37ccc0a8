"""GPU parity tests of the 16-bit activation modes' OWN synapse kernels (round 5, include/ss_neuron.h "ABI 9"): single-term operands on the native
matrix-core type, 16-bit I/O.  The reference is fp32-only (/root/reference/train.py:194-197); the modes are a build-side addition whose semantics are
"operands as stored (activations / gradients in the 16-bit format, weights rounded ONCE to it), exact products, fp32 accumulation, one narrowing on
store".  Every kernel is held against exactly that, evaluated in float64 from the SAME rounded operands:

    |out - float64| <= u |float64| (the one narrowing, u = 2^-8 bf16 / 2^-11 fp16)  +  2^-20 sum |a||b| (fp32 accumulation)      element-wise,

and at most a small fraction of the elements may differ from the once-rounded float64 value at all (an fp32 sum that lands on the other side of a
rounding boundary).  fp32 outputs (weight gradients): the second term alone.  Neuron kernels: bit-exact against the existing x16 / fp32 kernels.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
DTS = [torch.bfloat16, torch.float16]
IDS = ['bf16', 'f16']
U = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}
KIND = {'IF': 0, 'LIF': 1, 'PLIF': 2}


def rt(t, dt):
    """round once to the 16-bit format, back in float64 (the operand the kernel multiplies)"""
    return t.to(dt).double()


def spikes(shape, gen, dt=None):
    x = ((torch.rand(shape, device=DEV, generator=gen) < 0.3).float() + (torch.rand(shape, device=DEV, generator=gen) < 0.1).float()
         + (torch.rand(shape, device=DEV, generator=gen) < 0.03).float())                      # values 0 .. 3
    return x if dt is None else x.to(dt)


def pack(x):
    from oracle import np_pack
    return torch.from_numpy(np_pack.pack(x.float().cpu().numpy().reshape(-1)).view(np.int32)).to(DEV)


def assert_narrowed(out, ref64, mag64, dt, what, frac=5e-3):
    assert out.dtype == dt and bool(torch.isfinite(out.float()).all()), f'{what}: an element was not written'
    err = (out.double() - ref64).abs()
    tiny = 2.0 ** -24 if dt == torch.float16 else 1e-37
    bound = ref64.abs() * U[dt] * (1 + 1e-6) + mag64 * 2.0 ** -20 + tiny
    assert bool((err <= bound).all()), (what, float((err / bound).max()))
    differ = float((out != ref64.to(dt)).float().mean())
    assert differ <= frac, (what, 'fraction differing from the once-rounded float64 value', differ)


def assert_f32(out, ref64, mag64, what, scale=1.0):
    assert out.dtype == torch.float32 and bool(torch.isfinite(out).all()), what
    err = (out.double() - ref64).abs()
    bound = mag64 * 2.0 ** -20 * scale + 1e-30
    assert bool((err <= bound).all()), (what, float((err / bound).max()))


# ---------------------------------------------------------------------------------------------------------------------------------------
# neuron kernels
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', DTS, ids=IDS)
@pytest.mark.parametrize('kind', ['IF', 'LIF', 'PLIF'])
@pytest.mark.parametrize('T,N', [(1, 16), (5, 4096 + 16), (10, 16 * 1001), (5, 32 * 65 * 87 * 4), (4, 48)])
def test_packed_spike_io_on_16bit_activations(dt, kind, T, N):
    """ss_neuron_fwd_ex with act_dtype != 0 and packed output / packed skip == the dense x16 kernel (itself bit-exact against oracle/np_x16.py): dense
    output equal, packed output == oracle/np_pack.pack of it, v_last equal, counters equal; packed-only output; packed skip == dense skip."""
    from stereospike_amd import _lib
    from oracle import np_pack
    rng = np.random.default_rng(T * 131 + N)
    x = torch.tensor((rng.standard_normal((T, N)) * 0.3).astype(np.float32), device=DEV).to(dt)
    skip = torch.tensor(rng.integers(0, 3, (T, N)).astype(np.float32), device=DEV).to(dt)
    v0 = torch.tensor((rng.standard_normal(N) * 0.3).astype(np.float32), device=DEV)
    k = torch.tensor([0.4], device=DEV) if kind == 'PLIF' else None
    args = (T, N, 6.0, KIND[kind], 3.0, k, 1.0, 0.0)
    for sk in (None, skip):
        out_d, v_d = torch.empty_like(x), torch.empty(N, device=DEV)
        nnz_d = torch.zeros(2, dtype=torch.int64, device=DEV)
        _lib.neuron_fwd_x16(x, v0, sk, out_d, None, v_d, nnz_d, *args)
        out_p, v_p = torch.full_like(x, float('nan')), torch.empty(N, device=DEV)
        pk = torch.empty((T, N // 16), dtype=torch.int32, device=DEV)
        nnz_p = torch.zeros(2, dtype=torch.int64, device=DEV)
        ws = torch.empty(_lib.cnt_ws_words(N), dtype=torch.int32, device=DEV)
        _lib.neuron_fwd_ex(x, v0, sk, None, out_p, pk, None, v_p, nnz_p, ws, *args)
        assert torch.equal(out_d, out_p) and torch.equal(v_d, v_p) and torch.equal(nnz_d, nnz_p)
        want = np_pack.pack(out_d.float().cpu().numpy())
        assert np.array_equal(pk.cpu().numpy().view(np.uint32), want.view(np.uint32).reshape(T, N // 16))
        pk2 = torch.empty_like(pk)
        _lib.neuron_fwd_ex(x, v0, sk, None, None, pk2, None, v_p, None, None, *args)           # packed only
        assert torch.equal(pk, pk2)
        if sk is not None:                                                                       # the skip operand read from its packed form
            skp = torch.from_numpy(np_pack.pack(skip.float().cpu().numpy()).view(np.int32)).to(DEV)
            out_q, pk3 = torch.empty_like(x), torch.empty_like(pk)
            _lib.neuron_fwd_ex(x, v0, None, skp, out_q, pk3, None, v_p, None, None, *args)
            assert torch.equal(out_q, out_d) and torch.equal(pk3, pk)
    with pytest.raises(_lib.SSNeuronError):                                                       # saved h and packed I/O exclude each other
        _lib.neuron_fwd_ex(x, v0, None, None, out_p, pk, torch.empty((T, N), device=DEV), v_p, None, None, *args)


@pytest.mark.parametrize('dt', DTS, ids=IDS)
@pytest.mark.parametrize('kind', ['IF', 'PLIF'])
@pytest.mark.parametrize('T,rows,C', [(5, 96, 32), (5, 17, 64), (10, 8, 128), (1, 3, 256), (4, 5, 512), (2, 7, 4), (5, 20000, 32), (8, 33, 16)])
def test_x16_backward_with_low_rank_second_gradient(dt, kind, T, rows, C):
    """ss_neuron_bwd_fork_lr_x16 == the fp32 recompute backward fed widen(g1) + oracle.np_lowrank.head_input_gradient(lr_p, lr_w) on the widened input, its
    g_x narrowed once (the x16 kernels' definition: fp32 arithmetic on widened values, one narrowing on store): g_x, g_sum bit for bit, g_v_init and dL/dk
    exactly; the pair alone (no dense first gradient)."""
    from stereospike_amd import _lib
    from oracle import np_lowrank
    N = rows * C
    rng = np.random.default_rng(N + T)
    x = torch.tensor((rng.standard_normal((T, N)) * 0.25).astype(np.float32), device=DEV).to(dt)
    g1 = torch.tensor(rng.standard_normal((T, N)).astype(np.float32), device=DEV).to(dt)
    lr_p = (rng.standard_normal((T, rows, 9)) * 2).astype(np.float32)
    lr_w = rng.standard_normal((9, C)).astype(np.float32)
    g2 = torch.tensor(np_lowrank.head_input_gradient(lr_p, lr_w).reshape(T, N), device=DEV)
    P, Wl = torch.tensor(lr_p, device=DEV), torch.tensor(lr_w, device=DEV)
    v0 = torch.tensor((rng.standard_normal(N) * 0.5).astype(np.float32), device=DEV)
    k = torch.tensor([0.3], device=DEV) if kind == 'PLIF' else None
    args = (T, N, 7.5, KIND[kind], 2.0, k, 1.0, 0.0, 0, 2.0, True)
    ws = torch.empty(_lib.gk_ws_floats(), device=DEV) if kind == 'PLIF' else None

    def gk():
        return torch.zeros(1, device=DEV) if kind == 'PLIF' else None
    if not _lib.neuron_bwd_fork_lr_x16_supported(T, N, C, 9):
        assert C < 4 or (256 * (4 if T <= 5 else 2)) % C != 0 or C % 4 != 0, (T, N, C)
        return
    gsum32 = g1.float() + g2
    gx_a, gv_a, gk_a = torch.empty(T, N, device=DEV), torch.empty(N, device=DEV), gk()
    _lib.neuron_bwd_rc(gsum32, None, x.float(), v0, gx_a, gv_a, gk_a, ws, *args)
    gx_b, gv_b, gk_b = torch.empty_like(x), torch.empty(N, device=DEV), gk()
    gsum = torch.full_like(x, float('nan'))
    _lib.neuron_bwd_fork_lr_x16(g1, P, Wl, gsum, None, x, v0, gx_b, gv_b, gk_b, ws, *args)
    assert torch.equal(gx_a.to(dt), gx_b) and torch.equal(gv_a, gv_b) and torch.equal(gsum, gsum32.to(dt))
    if kind == 'PLIF':
        assert abs(float(gk_a) - float(gk_b)) <= 1e-6 * abs(float(gk_a)) + 1e-12      # (fp64 lane sums in another lane order: VEC 4 vs 2)
    gx_c, gx_d = torch.empty(T, N, device=DEV), torch.empty_like(x)
    _lib.neuron_bwd_rc(g2, None, x.float(), v0, gx_c, None, gk(), ws, *args)
    _lib.neuron_bwd_fork_lr_x16(None, P, Wl, None, None, x, v0, gx_d, None, gk(), ws, *args)
    assert torch.equal(gx_c.to(dt), gx_d)
    with pytest.raises(_lib.SSNeuronError):
        _lib.neuron_bwd_fork_lr_x16(None, P, Wl, gsum, None, x, v0, gx_d, None, gk(), ws, *args)     # a "sum" needs a dense first gradient
    assert not _lib.neuron_bwd_fork_lr_x16_supported(T, N, C, 25) and not _lib.neuron_bwd_fork_lr_x16_supported(3, N, C, 9)


# ---------------------------------------------------------------------------------------------------------------------------------------
# encoder
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dt', DTS, ids=IDS)
@pytest.mark.parametrize('NB,Cin,hw', [(2, 4, (64, 80)), (3, 2, (33, 45)), (1, 4, (7, 9)), (2, 4, (260, 346)), (2, 4, (50, 70))])
def test_dense_conv_s1_x16(dt, NB, Cin, hw):
    """First encoder layer (/root/reference/network/SNN_models.py:75-79) in the 16-bit modes: forward and weight gradient."""
    from stereospike_amd import _lib
    h, w = hw
    gen = torch.Generator(device=DEV).manual_seed(NB + h)
    x = torch.poisson(torch.full((NB, h, w, Cin), 0.3, device=DEV), generator=gen) + 0.37 * (torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.05).float()
    wt = torch.randn(32, Cin, 5, 5, device=DEV, generator=gen) * 0.1

    def conv(xx, ww):
        return F.conv2d(xx.permute(0, 3, 1, 2), ww, None, 1, 2).permute(0, 2, 3, 1)
    y = torch.full((NB, h, w, 32), float('nan'), device=DEV).to(dt)
    _lib.dense_conv_s1_fwd_x16(x, wt, y, NB, Cin, 32, h, w)
    assert_narrowed(y, conv(rt(x, dt), rt(wt, dt)), conv(rt(x, dt).abs(), rt(wt, dt).abs()), dt, 'dense_conv_s1_fwd_x16')
    g = (torch.randn(NB, h, w, 32, device=DEV, generator=gen) * torch.exp(torch.randn(NB, h, w, 1, device=DEV, generator=gen)) * 1e-2).to(dt)
    gw = torch.full((32, Cin, 5, 5), float('nan'), device=DEV)
    _lib.dense_conv_s1_wgrad_x16(g, x, gw, NB, Cin, 32, h, w)
    xr = rt(x, dt).permute(0, 3, 1, 2)

    def wgrad(gg, xx):
        return torch.ops.aten.convolution_backward(gg.permute(0, 3, 1, 2), xx, torch.zeros(32, Cin, 5, 5, dtype=torch.float64, device=DEV), None,
                                                   [1, 1], [2, 2], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    assert_f32(gw, wgrad(g.double(), xr), wgrad(g.double().abs(), xr.abs()), 'dense_conv_s1_wgrad_x16', scale=4.0)
    gw2 = gw.clone()
    _lib.dense_conv_s1_wgrad_x16(g, x, gw2, NB, Cin, 32, h, w, accumulate=True)
    assert float((gw2 - 2 * gw).abs().max()) <= 1e-6 * float(gw.abs().max())


@pytest.mark.parametrize('dt', DTS, ids=IDS)
@pytest.mark.parametrize('NB,Cin,Cout,hw', [(2, 32, 64, (64, 80)), (3, 64, 128, (33, 45)), (1, 32, 64, (7, 9)), (3, 64, 128, (130, 173)), (2, 32, 64, (260, 346)),
                                            (2, 32, 64, (50, 70))])
def test_spike_conv_x16(dt, NB, Cin, Cout, hw):
    """conv1 / conv2 (/root/reference/network/SNN_models.py:80-90) in the 16-bit modes: forward (dense 16-bit and packed spike input), weight gradient
    (dense and packed), data gradient."""
    from stereospike_amd import _lib
    h, w = hw
    gen = torch.Generator(device=DEV).manual_seed(NB + h)
    x = spikes((NB, h, w, Cin), gen, dt)
    wt = torch.randn(Cout, Cin, 5, 5, device=DEV, generator=gen) * 0.05
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1

    def conv(xx, ww):
        return F.conv2d(xx.permute(0, 3, 1, 2), ww, None, 2, 2).permute(0, 2, 3, 1)
    y = torch.full((NB, ho, wo, Cout), float('nan'), device=DEV).to(dt)
    _lib.spike_conv_fwd_x16(x, None, wt, y, NB, Cin, Cout, h, w)
    assert_narrowed(y, conv(x.double(), rt(wt, dt)), conv(x.double(), rt(wt, dt).abs()), dt, 'spike_conv_fwd_x16')
    packable = (NB * h * w * Cin) % 16 == 0
    if packable:
        xp = pack(x)
        y2 = torch.full_like(y, float('nan'))
        _lib.spike_conv_fwd_x16(None, xp, wt, y2, NB, Cin, Cout, h, w)
        assert torch.equal(y, y2)
    # ---- weight gradient
    g = (torch.randn(NB, ho, wo, Cout, device=DEV, generator=gen) * torch.exp(torch.randn(NB, ho, wo, 1, device=DEV, generator=gen)) * 1e-2).to(dt)
    gw = torch.full((Cout, Cin, 5, 5), float('nan'), device=DEV)
    _lib.spike_conv_wgrad_x16(g, x, gw, NB, Cin, Cout, h, w)
    xr = x.double().permute(0, 3, 1, 2)

    def wgrad(gg):
        return torch.ops.aten.convolution_backward(gg.permute(0, 3, 1, 2), xr, torch.zeros(Cout, Cin, 5, 5, dtype=torch.float64, device=DEV), None,
                                                   [2, 2], [2, 2], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    assert_f32(gw, wgrad(g.double()), wgrad(g.double().abs()), 'spike_conv_wgrad_x16', scale=4.0)
    if packable:
        gw2 = torch.full_like(gw, float('nan'))
        _lib.spike_conv_wgrad_x16(g, None, gw2, NB, Cin, Cout, h, w, x_packed=xp)        # packed input: the window / transposed-read form (another summation order)
        assert_f32(gw2, wgrad(g.double()), wgrad(g.double().abs()), 'spike_conv_wgrad_x16(packed)', scale=4.0)
        gw3 = torch.full_like(gw, float('nan'))
        _lib.spike_conv_wgrad_x16(g, None, gw3, NB, Cin, Cout, h, w, x_packed=xp)
        assert torch.equal(gw2, gw3)
    # ---- data gradient
    gx = torch.full((NB, h, w, Cin), float('nan'), device=DEV).to(dt)
    _lib.conv_s2_dgrad_x16(g, wt, gx, NB, Cin, Cout, h, w)

    def dgrad(gg, ww):
        return F.conv_transpose2d(gg.permute(0, 3, 1, 2), ww, None, 2, 2, output_padding=((h - 1) % 2, (w - 1) % 2)).permute(0, 2, 3, 1)
    assert_narrowed(gx, dgrad(g.double(), rt(wt, dt)), dgrad(g.double().abs(), rt(wt, dt).abs()), dt, 'conv_s2_dgrad_x16')


@pytest.mark.parametrize('dt', DTS, ids=IDS)
@pytest.mark.parametrize('NB,Cin,hw', [(5, 128, (17, 21)), (3, 256, (9, 10)), (2, 256, (33, 44)), (1, 128, (65, 87)), (7, 64, (1, 70)), (1, 32, (5, 3))])
def test_conv_s2_dgrad_x16_wide(dt, NB, Cin, hw):
    """conv3 / conv4 data gradients (and the small / odd maps of the fp32 test) in the 16-bit modes."""
    from stereospike_amd import _lib
    h, w = hw
    Cout = 2 * Cin
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    gen = torch.Generator(device=DEV).manual_seed(NB + h + Cin)
    g = (torch.randn(NB, ho, wo, Cout, device=DEV, generator=gen) * torch.exp(torch.randn(NB, ho, wo, 1, device=DEV, generator=gen)) * 1e-2).to(dt)
    wt = torch.randn(Cout, Cin, 5, 5, device=DEV, generator=gen) * 0.05
    gx = torch.full((NB, h, w, Cin), float('nan'), device=DEV).to(dt)
    _lib.conv_s2_dgrad_x16(g, wt, gx, NB, Cin, Cout, h, w)

    def dgrad(gg, ww):
        return F.conv_transpose2d(gg.permute(0, 3, 1, 2), ww, None, 2, 2, output_padding=((h - 1) % 2, (w - 1) % 2)).permute(0, 2, 3, 1)
    assert_narrowed(gx, dgrad(g.double(), rt(wt, dt)), dgrad(g.double().abs(), rt(wt, dt).abs()), dt, 'conv_s2_dgrad_x16')
    gx2 = torch.full_like(gx, float('nan'))
    _lib.conv_s2_dgrad_x16(g, wt, gx2, NB, Cin, Cout, h, w)
    assert torch.equal(gx, gx2)


@pytest.mark.parametrize('dt', DTS, ids=IDS)
def test_im2col_x16(dt):
    """Patch matrices of the 16-bit modes' library GEMMs: from a packed spike tensor (values as 16-bit patterns of `dt`) and from a dense 16-bit array."""
    from stereospike_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(3)
    for (NB, h, w, C, k, s, p) in ((3, 9, 11, 64, 3, 1, 1), (2, 17, 22, 128, 5, 2, 2), (2, 8, 8, 16, 3, 1, 1)):
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        x = spikes((NB, h, w, C), gen, dt)
        want = F.unfold(x.float().permute(0, 3, 1, 2), k, padding=p, stride=s).view(NB, C, k * k, ho * wo).permute(0, 3, 2, 1).reshape(NB * ho * wo, k * k * C).to(dt)
        A = torch.full((NB * ho * wo, k * k * C), float('nan'), device=DEV).to(dt)
        _lib.im2col_cl_packed_x16(pack(x), A, NB, h, w, C, k, s, p, ho, wo)
        assert torch.equal(A, want)
        xd = (torch.randn(NB, h, w, C, device=DEV, generator=gen)).to(dt)
        wantd = F.unfold(xd.float().permute(0, 3, 1, 2), k, padding=p, stride=s).view(NB, C, k * k, ho * wo).permute(0, 3, 2, 1).reshape(NB * ho * wo, k * k * C).to(dt)
        A2 = torch.full_like(A, float('nan'))
        _lib.im2col_cl_x16(xd, A2, NB, h, w, C, k, s, p, ho, wo)
        assert torch.equal(A2, wantd)


# ---------------------------------------------------------------------------------------------------------------------------------------
# decoder
# ---------------------------------------------------------------------------------------------------------------------------------------
GEOS = [(64, 32, (130, 173), (260, 346), 1), (128, 64, (65, 87), (130, 173), 2), (256, 128, (33, 44), (65, 87), 2), (64, 32, (32, 40), (64, 80), 3),
        (128, 64, (16, 20), (32, 40), 2), (64, 64, (13, 18), (25, 35), 3), (64, 32, (4, 5), (8, 10), 2)]


def _two_op(x64, w64, H, W):
    up = F.interpolate(x64.permute(0, 3, 1, 2), size=(H + 4, W + 4), mode='nearest')
    return F.conv2d(up, w64).permute(0, 2, 3, 1)


@pytest.mark.parametrize('dt', DTS, ids=IDS)
@pytest.mark.parametrize('Cin,Cout,hw,HW,NB', GEOS)
def test_upconv_sub_forward_x16(dt, Cin, Cout, hw, HW, NB):
    """Decoder stage forward (NNConvUpsampling, /root/reference/network/blocks.py:110-132) in the 16-bit modes: the two-op form evaluated in float64 with
    the taps rounded once to the format; the kernel's merged-tap sums travel as two terms of the format (relative 2^-16 bf16 / 2^-22 fp16 of a merged weight:
    inside the narrowing bound).  Dense 16-bit input == packed input."""
    from stereospike_amd import _lib, fused
    from stereospike_amd.network.blocks import NNConvUpsampling
    (h, w), (H, W) = hw, HW
    up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(DEV)
    tables = up._tables(h, w, torch.device(DEV))
    st = fused.sub_tables(tables, H, W)
    assert st is not None and _lib.upconv_sub_supported(Cin, Cout, 5)
    gen = torch.Generator(device=DEV).manual_seed(Cin + h)
    x = spikes((NB, h, w, Cin), gen, dt)
    wt = up.up[1].weight.detach().contiguous()
    wm = _lib.upconv_sub_prep_x16(wt, st, Cin, Cout, dt)
    y = torch.full((NB, H, W, Cout), float('nan'), device=DEV).to(dt)
    _lib.upconv_sub_fwd_x16(x, None, wm, st, y, NB, Cin, Cout, h, w)
    ref, mag = _two_op(x.double(), rt(wt, dt), H, W), _two_op(x.double(), rt(wt, dt).abs(), H, W)
    # + the two-term representation of the merged weights: 2^-16 (bf16) / 2^-22 (fp16) of the magnitude sum
    assert_narrowed(y, ref, mag * (1 + (2.0 ** -16 if dt == torch.bfloat16 else 2.0 ** -21) * 2.0 ** 20), dt, 'upconv_sub_fwd_x16', frac=2e-2 if dt == torch.bfloat16 else 5e-3)
    if (NB * h * w * Cin) % 16 == 0:
        y2 = torch.full_like(y, float('nan'))
        _lib.upconv_sub_fwd_x16(None, pack(x), wm, st, y2, NB, Cin, Cout, h, w)
        assert torch.equal(y, y2)


# (64, 24, ..) / (32, 8, ..): C_out % 16 != 0 — the one-chunk-per-window forms (ss_upconv_box.hip: NG = 1); every other geometry runs two chunks per window (round 6)
@pytest.mark.parametrize('dt', DTS, ids=IDS)
@pytest.mark.parametrize('Cin,Cout,hw,HW,NB', GEOS + [(64, 24, (13, 18), (25, 35), 2), (128, 8, (16, 20), (32, 40), 2)])
def test_upconv_box_kernels_x16(dt, Cin, Cout, hw, HW, NB):
    """Decoder stage backward on the box-sum image in the 16-bit modes (autograd of /root/reference/network/blocks.py:110-132): the box plane is the fp32 box sum of the
    widened gradients (the fp32 kernel's summation order) rounded once to the format; data gradient and weight gradient are
    held against float64 evaluated from the planes AS STORED and the weight rounded once."""
    from oracle import np_upconv_box as nbx
    from stereospike_amd import _lib, fused
    from stereospike_amd.network.blocks import NNConvUpsampling
    (h, w), (H, W) = hw, HW
    up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(DEV)
    tables = up._tables(h, w, torch.device(DEV))
    bt = fused.box_tables(tables, H, W)
    if not _lib.upconv_box_dgrad_supported(Cin, Cout, 5, bt, NB, h, w):
        assert bt['max_cols32'] > 76, bt          # deconv3's geometry: 32 source columns reach 78 horizontal ranges, the window holds 76 (the stage runs the g_P forms)
        pytest.skip('geometry exceeds the box window')
    vr, vmap = nbx.range_tables(tables[1].cpu().numpy(), tables[2].cpu().numpy(), H)
    hr, hmap = nbx.range_tables(tables[4].cpu().numpy(), tables[5].cpu().numpy(), W)
    gen = torch.Generator(device=DEV).manual_seed(23 + Cin + h)
    g = (torch.randn(NB, H, W, Cout, device=DEV, generator=gen) * torch.exp(torch.randn(NB, H, W, 1, device=DEV, generator=gen)) * 1e-2).to(dt)
    g[:, : H // 5] = 0
    wt = up.up[1].weight.detach().contiguous()
    NP = _lib.upconv_box_planes_x16(dt)
    assert NP == 1
    box = _lib.upconv_boxsum_x16(g, bt, NB, Cout, H, W)
    assert box.shape == (NB, Cout // 8, NP, len(vr), len(hr), 8)
    B32 = torch.tensor(nbx.boxsum(g.float().cpu().numpy(), vr, hr), device=DEV)                 # [NB, NVR, NHR, Cout] fp32, the fp32 kernel's summation order
    Beff = box.double().sum(2).permute(0, 2, 3, 1, 4).reshape(NB, len(vr), len(hr), Cout)        # the planes as stored
    assert torch.equal(box[:, :, 0], B32.view(NB, len(vr), len(hr), Cout // 8, 8).permute(0, 3, 1, 2, 4).to(dt))      # the first plane is the once-rounded fp32 box sum
    vm, hm = torch.tensor(vmap, device=DEV, dtype=torch.long), torch.tensor(hmap, device=DEV, dtype=torch.long)
    w64 = rt(wt, dt)
    ref = torch.zeros(NB, h, w, Cin, dtype=torch.float64, device=DEV)
    mag = torch.zeros_like(ref)
    for ky in range(5):
        for kx in range(5):
            gp = Beff[:, vm[:, ky]][:, :, hm[:, kx]]
            ref += gp @ w64[:, :, ky, kx]
            mag += gp.abs() @ w64[:, :, ky, kx].abs()
    g_x = torch.full((NB, h, w, Cin), float('nan'), device=DEV).to(dt)
    _lib.upconv_box_dgrad_x16(box, wt, bt, g_x, NB, Cin, Cout, h, w)
    assert_narrowed(g_x, ref, mag, dt, 'upconv_box_dgrad_x16')
    x = spikes((NB, h, w, Cin), gen, dt)
    x64 = x.double().reshape(-1, Cin)
    refw = torch.zeros(Cout, Cin, 5, 5, dtype=torch.float64, device=DEV)
    magw = torch.zeros_like(refw)
    for ky in range(5):
        for kx in range(5):
            gp = Beff[:, vm[:, ky]][:, :, hm[:, kx]].reshape(-1, Cout)
            refw[:, :, ky, kx] = gp.t() @ x64
            magw[:, :, ky, kx] = gp.abs().t() @ x64
    g_w = torch.full((Cout, Cin, 5, 5), float('nan'), device=DEV)
    _lib.upconv_box_wgrad_x16(box, x, None, bt, g_w, NB, Cin, Cout, h, w)
    assert_f32(g_w, refw, magw, 'upconv_box_wgrad_x16', scale=1.0)
    if x.numel() % 16 == 0:
        g_w3 = torch.empty_like(g_w)
        _lib.upconv_box_wgrad_x16(box, None, pack(x), bt, g_w3, NB, Cin, Cout, h, w)
        assert torch.equal(g_w3, g_w)
    g_w2 = g_w.clone()
    _lib.upconv_box_wgrad_x16(box, x, None, bt, g_w2, NB, Cin, Cout, h, w, accumulate=True)
    assert torch.equal(g_w2, g_w + g_w)
