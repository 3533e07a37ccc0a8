"""GPU parity tests proper: the HIP kernels, called through the C-ABI (stereospike_amd._lib -> libss_neuron.so),
against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): spikes / h / v BIT-EXACT; gradients bit-exact with the ATan surrogate and
within 1e-6 relative (stated here) with the Sigmoid surrogate (device expf vs libm/torch); dL/dk within 1e-5
relative (fp32 tree reduction vs the oracle's double accumulation), run-to-run bit-reproducible.
"""
import numpy as np
import pytest
import torch

from _util import c_oracle, kat_cases, bit_equal, rel_err

pytestmark = pytest.mark.gpu

KIND = {'IF': 0, 'LIF': 1, 'PLIF': 2}
SG = {'ATan': 0, 'Sigmoid': 1}
DEV = 'cuda:0'


def _dev(a):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def hip_fwd(x, *, kind, scale, tau, k, v_th, v_reset, v_init=None, skip=None, save_h=True, count=False):
    from stereospike_amd import _lib
    T, N = x.shape
    xd, vd, sd = _dev(x), _dev(v_init), _dev(skip)
    out = torch.empty_like(xd)
    h = torch.empty_like(xd) if save_h else None
    v_last = torch.empty(N, device=DEV)
    nnz = torch.zeros(2, dtype=torch.int64, device=DEV) if count else None
    kd = None if k is None else torch.tensor([k], dtype=torch.float32, device=DEV)
    _lib.neuron_fwd(xd, vd, sd, out, h, v_last, nnz, T, N, scale, KIND[kind], tau, kd, v_th, v_reset)
    torch.cuda.synchronize()
    return dict(out=out.cpu().numpy(), h=None if h is None else h.cpu().numpy(), v_last=v_last.cpu().numpy(),
                nnz=None if nnz is None else nnz.cpu().numpy())


def hip_bwd(g_out, h, *, kind, scale, tau, k, v_th, v_reset, surrogate, alpha, v_init=None, g_v_last=None,
            detach_reset=True):
    from stereospike_amd import _lib
    T, N = h.shape
    gd, hd, vd, gvd = _dev(g_out), _dev(h), _dev(v_init), _dev(g_v_last)
    g_x = torch.empty_like(hd)
    g_v_init = torch.empty(N, device=DEV)
    plif = kind == 'PLIF'
    g_k = torch.zeros(1, device=DEV) if plif else None
    ws = torch.empty(_lib.gk_ws_floats(), device=DEV) if plif else None
    kd = None if k is None else torch.tensor([k], dtype=torch.float32, device=DEV)
    _lib.neuron_bwd(gd, gvd, hd, vd, g_x, g_v_init, g_k, ws, T, N, scale, KIND[kind], tau, kd, v_th, v_reset,
                    SG[surrogate], alpha, detach_reset)
    torch.cuda.synchronize()
    return dict(g_x=g_x.cpu().numpy(), g_v_init=g_v_init.cpu().numpy(), g_k=None if g_k is None else float(g_k.item()))


def test_library_is_the_hip_one():
    from stereospike_amd import _lib
    assert _lib.lib().ss_abi_version() == _lib.ABI_VERSION
    assert _lib.LIB_PATH.endswith('stereospike_amd/lib/libss_neuron.so')


def test_kat_fixtures_bit_exact():
    """Every committed known-answer vector: spikes, h, v_last bit-exact; g_x per the stated tolerance."""
    n = 0
    for i, cfg, c in kat_cases():
        kw = dict(kind=cfg['kind'], scale=cfg['scale'], tau=cfg['tau'], k=cfg['k'], v_th=cfg['v_th'],
                  v_reset=cfg['v_reset'])
        skip = c['skip'].astype(np.float32) if 'skip' in c else None
        f = hip_fwd(c['x'], v_init=c.get('v_init'), skip=skip, count=True, **kw)
        assert np.array_equal(f['out'], c['out'].astype(np.float32)), f'case {i} {cfg}: spike mask differs'
        assert bit_equal(f['h'], c['h']), f'case {i} {cfg}: h differs'
        assert bit_equal(f['v_last'], c['v_last']), f'case {i} {cfg}: v_last differs'
        z = (c['h'] - np.float32(cfg['v_th'])) >= 0
        assert int(f['nnz'][0]) == int(z.sum()) and int(f['nnz'][1]) == int((c['out'] != 0).sum())
        b = hip_bwd(c['g_out'], c['h'], v_init=c.get('v_init'), g_v_last=c['g_v_last'], surrogate=cfg['surrogate'],
                    alpha=cfg['alpha'], **kw)
        if cfg['surrogate'] == 'ATan':
            assert bit_equal(b['g_x'], c['g_x']), f'case {i} {cfg}: g_x not bit-exact (rel {rel_err(b["g_x"], c["g_x"])})'
        else:
            assert rel_err(b['g_x'], c['g_x']) < 1e-6, f'case {i} {cfg}'
        if 'g_v_init' in c:
            assert rel_err(b['g_v_init'], c['g_v_init']) < 1e-6
        if cfg['kind'] == 'PLIF':
            k = cfg['k']
            g_w = b['g_k'] * k * (1 - k)
            assert abs(g_w - float(c['g_w'])) <= 2e-5 * max(1.0, abs(float(c['g_w']))) + 1e-4, (i, g_w, float(c['g_w']))
        n += 1
    assert n >= 60


@pytest.mark.parametrize('kind', ['IF', 'LIF', 'PLIF'])
@pytest.mark.parametrize('T', [1, 3, 5, 7, 10])
@pytest.mark.parametrize('N', [4, 1000, 4099, 262144 + 4])
def test_random_vs_c_oracle(kind, T, N):
    """Seeded random inputs, ragged sizes (N % 4 != 0 -> scalar path; templated and runtime-T paths)."""
    rng = np.random.default_rng(T * 1000003 + N)
    x = (rng.standard_normal((T, N)) * 0.2).astype(np.float32)
    v_init = (rng.standard_normal(N) * 0.5).astype(np.float32) if N % 2 == 0 else None
    skip = rng.integers(0, 3, (T, N)).astype(np.float32) if T % 2 == 1 else None
    kw = dict(kind=kind, scale=10.0, tau=3.0, k=0.3333333 if kind == 'PLIF' else None, v_th=1.0, v_reset=0.0)
    ref = c_oracle.neuron_fwd(x, v_init=v_init, skip_seq=skip, count=True, **kw)
    got = hip_fwd(x, v_init=v_init, skip=skip, count=True, **kw)
    assert np.array_equal(ref['out'], got['out'])
    assert bit_equal(ref['h'], got['h']) and bit_equal(ref['v_last'], got['v_last'])
    assert np.array_equal(ref['nnz'].astype(np.int64), got['nnz'])
    inf = hip_fwd(x, v_init=v_init, skip=skip, save_h=False, **kw)       # inference variant (no h)
    assert np.array_equal(inf['out'], got['out']) and bit_equal(inf['v_last'], got['v_last'])
    g = rng.standard_normal((T, N)).astype(np.float32)
    gv = rng.standard_normal(N).astype(np.float32)
    for sg, alpha in (('ATan', 2.0), ('Sigmoid', 4.0)):
        rb = c_oracle.neuron_bwd(g, ref['h'], v_init=v_init, g_v_last=gv, surrogate=sg, alpha=alpha, **kw)
        gb = hip_bwd(g, ref['h'], v_init=v_init, g_v_last=gv, surrogate=sg, alpha=alpha, **kw)
        if sg == 'ATan':
            assert bit_equal(rb['g_x'], gb['g_x']) and bit_equal(rb['g_v_init'], gb['g_v_init'])
        else:
            assert rel_err(gb['g_x'], rb['g_x']) < 1e-6 and rel_err(gb['g_v_init'], rb['g_v_init']) < 1e-6
        if kind == 'PLIF':
            assert abs(gb['g_k'] - rb['g_k']) <= 1e-5 * abs(rb['g_k']) + 2e-7 * T * N
            again = hip_bwd(g, ref['h'], v_init=v_init, g_v_last=gv, surrogate=sg, alpha=alpha, **kw)
            assert again['g_k'] == gb['g_k'], 'dL/dk must be bit-reproducible run to run'


def test_edge_cases():
    """Exact threshold, +-1 ulp, infinite threshold (I-neuron use), non-detached reset, v_reset != 0."""
    one = np.float32(1.0)
    x = np.array([[one, np.nextafter(one, np.float32(2)), np.nextafter(one, np.float32(0)), 0.0, -0.0, 5.0, -5.0, 1e-38]],
                 np.float32)
    kw = dict(kind='IF', scale=1.0, tau=2.0, k=None, v_th=1.0, v_reset=0.0)
    ref, got = c_oracle.neuron_fwd(x, **kw), hip_fwd(x, **kw)
    assert np.array_equal(got['out'][0], [1, 1, 0, 0, 0, 1, 0, 0])
    assert bit_equal(ref['v_last'], got['v_last']) and bit_equal(ref['h'], got['h'])
    kw_inf = dict(kw, v_th=float('inf'))
    x5 = np.tile(x, (5, 1))
    ref, got = c_oracle.neuron_fwd(x5, **kw_inf), hip_fwd(x5, **kw_inf)
    assert not got['out'].any() and bit_equal(ref['v_last'], got['v_last'])
    g = np.ones_like(x5)
    for sg, alpha in (('ATan', 2.0), ('Sigmoid', 4.0)):
        rb = c_oracle.neuron_bwd(g, ref['h'], surrogate=sg, alpha=alpha, g_v_last=np.ones(8, np.float32), **kw_inf)
        gb = hip_bwd(g, ref['h'], surrogate=sg, alpha=alpha, g_v_last=np.ones(8, np.float32), **kw_inf)
        assert np.array_equal(rb['g_x'], gb['g_x'])          # surrogate derivative is exactly 0 at -inf
    rng = np.random.default_rng(5)
    xr = rng.standard_normal((5, 4096)).astype(np.float32)
    kw2 = dict(kind='LIF', scale=2.0, tau=10.0, k=None, v_th=1.0, v_reset=0.1)
    ref, got = c_oracle.neuron_fwd(xr, **kw2), hip_fwd(xr, **kw2)
    assert bit_equal(ref['h'], got['h'])
    rb = c_oracle.neuron_bwd(xr, ref['h'], surrogate='ATan', alpha=2.0, detach_reset=False, **kw2)
    gb = hip_bwd(xr, ref['h'], surrogate='ATan', alpha=2.0, detach_reset=False, **kw2)
    assert bit_equal(rb['g_x'], gb['g_x'])


def test_empty_and_invalid():
    from stereospike_amd import _lib
    x = torch.zeros(0, device=DEV)
    out, v = torch.zeros(0, device=DEV), torch.zeros(0, device=DEV)
    L = _lib.lib()
    import ctypes as C
    p = lambda t: C.c_void_p(t.data_ptr())
    one = torch.zeros(4, device=DEV)
    assert L.ss_neuron_fwd_f32(p(one), None, None, p(one.clone()), None, p(one.clone()), None, 1, 0, 1.0, 0, 2.0, None, 1.0, 0.0, None) == 0
    assert L.ss_neuron_fwd_f32(None, None, None, p(one), None, p(one), None, 1, 4, 1.0, 0, 2.0, None, 1.0, 0.0, None) == -22
    assert L.ss_neuron_fwd_f32(p(one), None, None, p(one.clone()), None, p(one.clone()), None, 0, 4, 1.0, 0, 2.0, None, 1.0, 0.0, None) == -22
    assert L.ss_neuron_fwd_f32(p(one), None, None, p(one.clone()), None, p(one.clone()), None, 1, 4, 1.0, 7, 2.0, None, 1.0, 0.0, None) == -22
    assert L.ss_neuron_fwd_f32(p(one), None, None, p(one.clone()), None, p(one.clone()), None, 1, 4, 1.0, 2, 2.0, None, 1.0, 0.0, None) == -22  # PLIF without k
    with pytest.raises(_lib.SSNeuronError):
        _lib.neuron_fwd(torch.zeros(4), None, None, torch.zeros(4), None, torch.zeros(4), None, 1, 4, 1.0, 0, 2.0, None, 1.0, 0.0)


def test_ipool_vs_oracle():
    from stereospike_amd import _lib
    rng = np.random.default_rng(9)
    for T, K, M in ((1, 4, 89960), (5, 4, 4100), (3, 2, 7)):
        pd = rng.standard_normal((T, K, M)).astype(np.float32)
        v0 = rng.standard_normal(M).astype(np.float32)
        ref = c_oracle.ipool_fwd(pd, scale=10.0, v_reset=0.0, v_init=v0)
        pdd = _dev(np.ascontiguousarray(pd.transpose(1, 0, 2)))       # [K, T, M] as torch.stack gives
        depth = torch.empty(T, K, M, device=DEV)
        _lib.ipool_fwd(pdd, M, T * M, _dev(v0), depth, T, K, M, 10.0, 0.0)
        assert bit_equal(depth, ref)
        g = rng.standard_normal((T, K, M)).astype(np.float32)
        rb = c_oracle.ipool_bwd(g, scale=10.0)
        g_pd = torch.empty(K, T, M, device=DEV)
        g_v0 = torch.empty(M, device=DEV)
        _lib.ipool_bwd(_dev(g), None, g_pd, M, T * M, g_v0, T, K, M, 10.0)
        assert bit_equal(g_pd.permute(1, 0, 2).contiguous(), rb['g_pd']) and bit_equal(g_v0, rb['g_v_init'])


def test_full_size_properties():
    """Config-3 sized layer (B16 x T5 x 32x260x346 = 2.3e8 updates): size-independent checks only.
    (a) linearity of the backward in g_out, (b) counters == count_nonzero of the outputs, (c) sub-range equals
    the small-N result bit for bit (no cross-lane leakage), (d) T-split equivalence: running T=5 equals T=2 then T=3
    from the carried membrane."""
    from stereospike_amd import _lib
    T, N = 5, 16 * 32 * 260 * 346
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(T, N, device=DEV, generator=g) * 0.15
    out, h, v = torch.empty_like(x), torch.empty_like(x), torch.empty(N, device=DEV)
    nnz = torch.zeros(2, dtype=torch.int64, device=DEV)
    _lib.neuron_fwd(x, None, None, out, h, v, nnz, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)
    assert int(nnz[0]) == int(out.count_nonzero()) == int(nnz[1])
    assert set(out.unique().tolist()) <= {0.0, 1.0}
    # (c)
    n = 4096
    sub = c_oracle.neuron_fwd(x[:, 12345 * 4:12345 * 4 + n].cpu().numpy(), kind='IF', scale=10.0, v_th=1.0, v_reset=0.0)
    assert bit_equal(h[:, 12345 * 4:12345 * 4 + n], sub['h'])
    # (d)
    o1, h1, v1 = torch.empty(2, N, device=DEV), torch.empty(2, N, device=DEV), torch.empty(N, device=DEV)
    _lib.neuron_fwd(x[:2].contiguous(), None, None, o1, h1, v1, None, 2, N, 10.0, 0, 2.0, None, 1.0, 0.0)
    o2, h2, v2 = torch.empty(3, N, device=DEV), torch.empty(3, N, device=DEV), torch.empty(N, device=DEV)
    _lib.neuron_fwd(x[2:].contiguous(), v1, None, o2, h2, v2, None, 3, N, 10.0, 0, 2.0, None, 1.0, 0.0)
    assert torch.equal(torch.cat([o1, o2]), out) and torch.equal(v2, v) and torch.equal(torch.cat([h1, h2]), h)
    del o1, o2, h1, h2
    # (a)
    ga = torch.randn(T, N, device=DEV, generator=g)
    gx1, gx2 = torch.empty_like(x), torch.empty_like(x)
    _lib.neuron_bwd(ga, None, h, None, gx1, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)
    _lib.neuron_bwd(ga * 2, None, h, None, gx2, None, None, None, T, N, 10.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)
    assert torch.equal(gx1 * 2, gx2)        # scaling by 2 is exact in fp32


def test_upconv1_heads_vs_oracle_and_torch():
    """Fused predict_depth head (nearest-upsample + valid 3x3 conv to one channel) == the oracle's gather (same
    summation order => bit-exact) and == torch's UpsamplingNearest2d + Conv2d on the CPU within 2e-6."""
    import torch.nn as nn
    from stereospike_amd import _lib
    from stereospike_amd.fused import nearest_tables, upconv_projected
    torch.manual_seed(0)
    for (C, h, w, H, W, k) in [(256, 33, 44, 260, 346, 3), (32, 260, 346, 260, 346, 3), (5, 7, 9, 20, 30, 3), (6, 17, 22, 33, 44, 5)]:
        up = nn.Sequential(nn.UpsamplingNearest2d(size=(H + k - 1, W + k - 1)), nn.Conv2d(C, 1, k, bias=True))
        x = (torch.rand(3, C, h, w) < 0.4).float().requires_grad_()
        y = up(x)
        g = torch.randn_like(y)
        y.backward(g)
        sy, ylo, yhi = nearest_tables(h, H + k - 1)
        sx, xlo, xhi = nearest_tables(w, W + k - 1)
        tables = tuple(t.to(DEV) for t in (sy, ylo, yhi, sx, xlo, xhi))
        xd = x.detach().to(DEV).requires_grad_()
        wd = up[1].weight.detach().to(DEV).requires_grad_()
        bd = up[1].bias.detach().to(DEV).requires_grad_()
        yd = upconv_projected(xd, wd, bd, tables, k, H, W)
        yd.backward(g.to(DEV))
        assert rel_err(yd, y) < 2e-6
        assert rel_err(xd.grad, x.grad) < 1e-5 and rel_err(wd.grad, up[1].weight.grad) < 1e-5
        assert rel_err(bd.grad, up[1].bias.grad) < 1e-5
        # kernel vs C oracle on the same projections: identical order => bit-exact
        P = torch.nn.functional.conv2d(x.detach(), up[1].weight.detach().view(C, k * k).t().reshape(k * k, C, 1, 1))
        ref = c_oracle.upconv1_fwd(P.numpy(), sy.numpy(), sx.numpy(), float(up[1].bias), H, W)
        out = torch.empty(3, 1, H, W, device=DEV)
        _lib.upconv1_fwd(P.to(DEV), tables[0], tables[3], bd.detach(), out, 3, k, h, w, H, W)
        assert bit_equal(out[:, 0], ref)
        gref = c_oracle.upconv1_bwd(g.numpy()[:, 0], ylo.numpy(), yhi.numpy(), xlo.numpy(), xhi.numpy(), k)
        gP = torch.empty(3, k * k, h, w, device=DEV)
        _lib.upconv1_bwd(g.to(DEV).contiguous(), tables[1], tables[2], tables[4], tables[5], gP, 3, k, h, w, H, W)
        assert bit_equal(gP, gref)


def test_upconv_projected_multichannel_vs_torch():
    """Decoder synapse NNConvUpsampling(C_in -> C_out, k=5) through projection + gather == torch's two-op form (CPU)."""
    import torch.nn as nn
    from stereospike_amd.fused import nearest_tables, upconv_projected
    torch.manual_seed(1)
    for (Cin, Cout, h, w, H, W, k) in [(512, 256, 17, 22, 33, 44, 5), (64, 32, 130, 173, 260, 346, 5), (8, 4, 9, 11, 20, 25, 5)]:
        up = nn.Sequential(nn.UpsamplingNearest2d(size=(H + k - 1, W + k - 1)), nn.Conv2d(Cin, Cout, k, bias=False))
        x = (torch.rand(2, Cin, h, w) < 0.4).float().requires_grad_()
        y = up(x)
        g = torch.randn_like(y)
        y.backward(g)
        tables = tuple(t.to(DEV) for t in (nearest_tables(h, H + k - 1) + nearest_tables(w, W + k - 1)))
        xd = x.detach().to(DEV).requires_grad_()
        wd = up[1].weight.detach().to(DEV).requires_grad_()
        yd = upconv_projected(xd, wd, None, tables, k, H, W)
        yd.backward(g.to(DEV))
        assert rel_err(yd, y) < 3e-6, (Cin, Cout)
        assert rel_err(xd.grad, x.grad) < 2e-5 and rel_err(wd.grad, up[1].weight.grad) < 2e-5, (Cin, Cout)


def test_upconv_channels_last_vs_oracle_and_torch():
    """NHWC decoder synapse: one GEMM + channels-last gather == C oracle (bit-exact kernels) == torch two-op (2e-6)."""
    import torch.nn as nn
    from stereospike_amd import _lib
    from stereospike_amd.fused import nearest_tables, upconv_projected_cl
    torch.manual_seed(2)
    for (Cin, Cout, h, w, H, W, k, B) in [(512, 256, 17, 22, 33, 44, 5, 3), (64, 32, 130, 173, 260, 346, 5, 2),
                                           (8, 4, 9, 11, 20, 25, 5, 2), (32, 1, 65, 87, 260, 346, 3, 2), (6, 3, 5, 7, 40, 50, 3, 2),
                                           # prediction heads (3 x 3, ONE channel; round 6): the LDS-staged gather — more frames than workgroup rows (23 > 8192 / 390 tiles),
                                           # tiles ragged in both directions — and the union-scan adjoint at x15 / x4 up-sampling; then the x2 row-scan path
                                           (8, 1, 17, 22, 260, 346, 3, 23), (8, 1, 9, 11, 37, 90, 3, 3), (8, 1, 33, 44, 66, 88, 3, 2)]:
        up = nn.Sequential(nn.UpsamplingNearest2d(size=(H + k - 1, W + k - 1)), nn.Conv2d(Cin, Cout, k, bias=(Cout == 1)))
        x = (torch.rand(B, Cin, h, w) < 0.4).float().requires_grad_()
        y = up(x)
        g = torch.randn_like(y)
        y.backward(g)
        tabs = nearest_tables(h, H + k - 1) + nearest_tables(w, W + k - 1)
        tables = tuple(t.to(DEV) for t in tabs)
        x_cl = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_()
        wd = up[1].weight.detach().to(DEV).requires_grad_()
        bd = up[1].bias.detach().to(DEV).requires_grad_() if Cout == 1 else None
        y_cl = upconv_projected_cl(x_cl, wd, bd, tables, k, H, W)
        y_cl.backward(g.permute(0, 2, 3, 1).contiguous().to(DEV))
        assert rel_err(y_cl.permute(0, 3, 1, 2), y) < 3e-6, (Cin, Cout)
        assert rel_err(x_cl.grad.permute(0, 3, 1, 2), x.grad) < 2e-5, (Cin, Cout)
        assert rel_err(wd.grad, up[1].weight.grad) < 2e-5, (Cin, Cout)
        if bd is not None:       # a plain sum over every output element: cancelling — held against its magnitude sum (both sides are fp32 sums in different orders)
            assert float((bd.grad.cpu() - up[1].bias.grad).abs().max()) <= 1e-6 * float(g.abs().sum()), (Cin, Cout)
        # kernels vs the C oracle on identical P / g_out: same summation order => bit-exact
        Wt = up[1].weight.detach().permute(1, 2, 3, 0).reshape(Cin, k * k * Cout)
        P = (x.detach().permute(0, 2, 3, 1).reshape(-1, Cin) @ Wt).view(B, h, w, k * k * Cout).contiguous()
        bias_np = None if Cout != 1 else up[1].bias.detach().numpy()
        ref = c_oracle.upconv_cl_fwd(P.numpy(), tabs[0].numpy(), tabs[3].numpy(), bias_np, k, Cout, H, W)
        out = torch.empty(B, H, W, Cout, device=DEV)
        _lib.upconv_cl_fwd(P.to(DEV), tables[0], tables[3], None if bd is None else bd.detach(), out, B, k, Cout, h, w, H, W)
        assert bit_equal(out, ref), (Cin, Cout)
        g_cl = g.permute(0, 2, 3, 1).contiguous()
        gref = c_oracle.upconv_cl_bwd(g_cl.numpy(), tabs[1].numpy(), tabs[2].numpy(), tabs[4].numpy(), tabs[5].numpy(), k)
        gP = torch.empty(B, h, w, k * k * Cout, device=DEV)
        _lib.upconv_cl_bwd(g_cl.to(DEV), tables[1], tables[2], tables[4], tables[5], gP, B, k, Cout, h, w, H, W)
        assert bit_equal(gP, gref), (Cin, Cout)


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('kind', ['IF', 'LIF', 'PLIF'])
@pytest.mark.parametrize('T,N', [(1, 8), (5, 4104), (10, 100003), (5, 16 * 32 * 65 * 87)])
def test_x16_kernels_vs_numpy_oracle(dt, kind, T, N):
    """16-bit activation I/O kernels (configs 2 / 5) == oracle/np_x16.py: spikes / h / v bit-exact, g_x bit-exact (ATan)."""
    from oracle import np_x16
    from stereospike_amd import _lib
    name = 'f16' if dt == torch.float16 else 'bf16'
    g = torch.Generator().manual_seed(T * 7 + N)
    x = (torch.randn(T, N, generator=g) * 0.2).to(dt)
    skip = torch.randint(0, 3, (T, N), generator=g).to(dt) if T % 2 == 1 else None
    v0 = torch.randn(N, generator=g) * 0.5 if N % 2 == 0 else None
    kw = dict(kind=kind, scale=10.0, tau=3.0, k=np.float32(1 / 3.) if kind == 'PLIF' else None, v_th=1.0, v_reset=0.0)
    bits = lambda t: None if t is None else t.view(torch.int16).numpy().view(np.uint16)
    ref = np_x16.neuron_fwd(bits(x), name, v_init=None if v0 is None else v0.numpy(), skip_bits=bits(skip), **kw)
    xd, sd, vd = x.to(DEV), None if skip is None else skip.to(DEV), None if v0 is None else v0.to(DEV)
    out, h, v = torch.empty_like(xd), torch.empty(T, N, device=DEV), torch.empty(N, device=DEV)
    nnz = torch.zeros(2, dtype=torch.int64, device=DEV)
    kd = None if kind != 'PLIF' else torch.tensor([1 / 3.], device=DEV)
    _lib.neuron_fwd_x16(xd, vd, sd, out, h, v, nnz, T, N, 10.0, KIND[kind], 3.0, kd, 1.0, 0.0)
    assert np.array_equal(bits(out.cpu()), ref['out'])
    assert bit_equal(h, ref['h']) and bit_equal(v, ref['v_last'])
    assert int(nnz[1]) == int((np_x16.widen(ref['out'], name) != 0).sum())
    go = torch.randn(T, N, generator=g).to(dt)
    rb = np_x16.neuron_bwd(bits(go), ref['h'], name, v_init=None if v0 is None else v0.numpy(), surrogate='ATan', alpha=2.0, **kw)
    gx, gvi = torch.empty_like(xd), torch.empty(N, device=DEV)
    gk = torch.zeros(1, device=DEV) if kind == 'PLIF' else None
    ws = torch.empty(_lib.gk_ws_floats(), device=DEV) if kind == 'PLIF' else None
    _lib.neuron_bwd_x16(go.to(DEV), None, h, vd, gx, gvi, gk, ws, T, N, 10.0, KIND[kind], 3.0, kd, 1.0, 0.0, 0, 2.0, True)
    assert np.array_equal(bits(gx.cpu()), rb['g_x'])
    assert bit_equal(gvi, rb['g_v_init'])
    if kind == 'PLIF':
        assert abs(float(gk) - rb['g_k']) <= 1e-5 * abs(rb['g_k']) + 2e-7 * T * N
    # recompute form (h rebuilt from the 16-bit layer input; the forward then runs without h_seq): bit-identical
    out2, v2 = torch.empty_like(xd), torch.empty(N, device=DEV)
    _lib.neuron_fwd_x16(xd, vd, sd, out2, None, v2, None, T, N, 10.0, KIND[kind], 3.0, kd, 1.0, 0.0)
    assert torch.equal(out2.view(torch.int16), out.view(torch.int16)) and torch.equal(v2, v)
    for detach in (True, False):
        gx1, gv1 = torch.empty_like(xd), torch.empty(N, device=DEV)
        gx2, gv2 = torch.empty_like(xd), torch.empty(N, device=DEV)
        gk1 = torch.zeros(1, device=DEV) if kind == 'PLIF' else None
        gk2 = torch.zeros(1, device=DEV) if kind == 'PLIF' else None
        gvl = torch.randn(N, generator=g).to(DEV)
        _lib.neuron_bwd_x16(go.to(DEV), gvl, h, vd, gx1, gv1, gk1, ws, T, N, 10.0, KIND[kind], 3.0, kd, 1.0, 0.0, 0, 2.0, detach)
        _lib.neuron_bwd_rc_x16(go.to(DEV), gvl, xd, vd, gx2, gv2, gk2, ws, T, N, 10.0, KIND[kind], 3.0, kd, 1.0, 0.0, 0, 2.0, detach)
        assert torch.equal(gx1.view(torch.int16), gx2.view(torch.int16)) and torch.equal(gv1.view(torch.int32), gv2.view(torch.int32))
        if kind == 'PLIF':     # lane width differs (8 vs 4 / 2 neurons): same terms, another fp32 summation order
            assert abs(float(gk1) - float(gk2)) <= 1e-5 * abs(float(gk1)) + 2e-7 * T * N


def test_voxelizer_vs_reference_fixture_and_oracle():
    """HIP voxeliser == the reference's own output (fixture) bit for bit (integer counts), and == the numpy oracle on a
    large random stream with colliding pixels and out-of-range coordinates."""
    from oracle import np_voxelize as nv
    from _util import load_npz
    from stereospike_amd.data import mvsecCumulateSpikesIntoFrames
    z = load_npz('voxelizer.npz')
    for ci in range(int(z['n_cases'])):
        n_chunks, nfpdm = (int(v) for v in z[f'v{ci}_cfg'])
        ev = torch.tensor(z[f'v{ci}_events'], dtype=torch.float64, device=DEV)
        fr = mvsecCumulateSpikesIntoFrames(ev, n_chunks, nfpdm)
        assert fr.dtype == torch.float32 and tuple(fr.shape) == (n_chunks, nfpdm, 2, 260, 346)
        assert np.array_equal(fr.cpu().numpy(), z[f'v{ci}_frames'].astype(np.float32))
    rng = np.random.default_rng(5)
    E = 2_000_000
    t = np.sort(rng.uniform(5.0, 5.0 + 0.26, E))
    ev = np.stack([rng.uniform(-2, 348, E), rng.uniform(-2, 262, E), t, rng.choice([1.0, 0.0, -1.0], E)], 1)
    ev[:, :2] = np.where(rng.random((E, 2)) < 0.3, np.floor(ev[:, :2] / 8) * 8, ev[:, :2])     # many collisions
    ref = nv.cumulate_spikes_into_frames(ev, 5, 5)
    got = mvsecCumulateSpikesIntoFrames(torch.tensor(ev, device=DEV), 5, 5)
    assert np.array_equal(got.cpu().numpy(), ref.astype(np.float32))
    assert ref.max() > 3
    empty = mvsecCumulateSpikesIntoFrames(torch.zeros(1, 4, dtype=torch.float64, device=DEV), 2, 1)
    assert float(empty.sum()) == 0


def _loss_case(rng, B, H, W, nan_frac, scale=3.0):
    pred = (rng.standard_normal((B, 1, H, W)) * scale).astype(np.float32)
    gt = (rng.standard_normal((B, 1, H, W)) * scale + 5).astype(np.float32)
    gt[rng.random(gt.shape) < nan_frac] = np.nan
    if nan_frac >= 1.0:
        gt[0, 0, H // 2, W // 2] = 1.0                    # one valid pixel: n = 1
    return pred, gt


@pytest.mark.parametrize('B,H,W,nan_frac', [(1, 1, 1, 0.0), (2, 15, 17, 0.3), (3, 33, 47, 0.5), (1, 16, 16, 0.0), (2, 48, 64, 1.0),
                                            (1, 260, 346, 0.6), (16, 260, 346, 0.45)])
def test_loss_statistics_and_gradient_kernels_vs_oracle(B, H, W, nan_frac):
    """ss_loss_stats_f32: the 5 batch-wide sums within 1e-6 relative of the oracle's fp64 sums (fp32 wavefront partials, fp64 above);
    ss_loss_grad_f32: BIT-EXACT against the oracle's fp32 formula given the same sums (the Sobel adjoint is small-integer exact)."""
    from oracle import np_loss
    from stereospike_amd import _lib
    rng = np.random.default_rng(B * 1000 + H)
    pred, gt = _loss_case(rng, B, H, W, nan_frac)
    pd, gd = torch.tensor(pred, device=DEV), torch.tensor(gt, device=DEV)
    sums = torch.empty(5, dtype=torch.float64, device=DEV)
    ws = torch.empty(_lib.loss_ws_doubles(), dtype=torch.float64, device=DEV)
    _lib.loss_stats(pd, gd, sums, ws, B, H, W)
    got = sums.cpu().numpy()
    ref = np_loss.loss_stats(pred, gt)
    assert got[0] == ref[0]
    assert abs(got[1] - ref[1]) <= 1e-6 * ref[4] and abs(got[4] - ref[4]) <= 1e-6 * ref[4]
    assert abs(got[2] - ref[2]) <= 1e-6 * ref[2] and abs(got[3] - ref[3]) <= 1e-6 * ref[3] + 1e-12
    sums2 = torch.empty_like(sums)
    _lib.loss_stats(pd, gd, sums2, ws, B, H, W)
    assert torch.equal(sums, sums2)                       # deterministic (fixed-order reduction)
    coef = torch.tensor([0.7, 0.35], dtype=torch.float32, device=DEV)
    g = torch.full_like(pd, float('nan'))
    _lib.loss_grad(pd, gd, sums, coef, g, B, H, W)
    want = np_loss.loss_grad(pred, gt, got, coef.cpu().numpy())
    assert bit_equal(g.cpu().numpy(), want)


def test_fused_total_loss_vs_reference_fixture_and_unfused_form():
    """Total_Loss on the device (fused kernels) against the reference's own loss value / MDE / d loss / d pred (fixture), and against the
    torch-composed form of the same module (FUSED_LOSS = False) on a full-size batch."""
    from _util import load_npz
    from stereospike_amd.network import loss as L
    from stereospike_amd import fused
    z = load_npz('loss_metric.npz')
    for ci in range(int(z['n_cases'])):
        preds = [torch.tensor(z[f'l{ci}_pred{i}'], device=DEV, requires_grad=True) for i in range(4)]
        gt = torch.tensor(z[f'l{ci}_gt'], device=DEV)
        spikes = [torch.tensor(z[f'l{ci}_spk{i}'].astype(np.float32), device=DEV) for i in range(5)]
        for pen in (False, True):
            tag = 'pen' if pen else 'nopen'
            loss = L.Total_Loss(penalize_spikes=pen, beta=0.5)(preds, gt, spikes)
            ref = float(z[f'l{ci}_{tag}_loss'])
            assert abs(float(loss) - ref) <= 2e-6 * abs(ref), (ci, pen)
            grads = torch.autograd.grad(loss, preds)
            for i, g in enumerate(grads):
                want = z[f'l{ci}_{tag}_gpred{i}']
                if want.ndim:
                    assert float(np.abs(g.cpu().numpy() - want).max()) <= 1e-6 * float(np.abs(want).max()) + 1e-9
        mde = fused.scale_loss_terms(preds[0].detach(), gt)[2]
        assert abs(float(mde) - float(z[f'l{ci}_mde'])) <= 2e-6 * float(z[f'l{ci}_mde'])
    rng = np.random.default_rng(3)
    B, H, W = 16, 260, 346
    gt = torch.tensor(_loss_case(rng, B, H, W, 0.45)[1], device=DEV)
    preds = [torch.tensor(_loss_case(rng, B, H, W, 0.0)[0] + 5, device=DEV, requires_grad=True) for _ in range(4)]
    out = {}
    from stereospike_amd import config
    for fusedflag in (True, False):
        with config.engine_config(FUSED_LOSS=fusedflag):
            loss = L.Total_Loss(scale_weights=(1., 0.5, 0.25, 2.))(preds, gt)
            out[fusedflag] = (float(loss), [g.clone() for g in torch.autograd.grad(loss, preds)])
    assert abs(out[True][0] - out[False][0]) <= 2e-6 * abs(out[False][0])
    for a, b in zip(out[True][1], out[False][1]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
        assert torch.equal(a == 0, b == 0) or float(((a == 0) != (b == 0)).float().mean()) < 1e-6


@pytest.mark.parametrize('surrogate', ['ATan', 'Sigmoid'])
@pytest.mark.parametrize('kind', ['IF', 'LIF', 'PLIF'])
@pytest.mark.parametrize('T,N', [(1, 1000), (2, 4096), (4, 777), (5, 260 * 346 * 4), (8, 4100), (10, 64 * 1024 + 3)])
def test_backward_with_recomputed_h_is_bit_identical(kind, T, N, surrogate):
    """ss_neuron_bwd_rc_f32 (h rebuilt in registers from x_seq / v_init) == ss_neuron_bwd_f32 fed the forward's saved h_seq, bit for
    bit on g_x, g_v_init and dL/dk — and == the C oracle (ATan) — for every compile-time T, the vector and the scalar-tail path,
    detach_reset on and off; unsupported T is refused."""
    from stereospike_amd import _lib
    rng = np.random.default_rng(T * 7 + N)
    x = (rng.standard_normal((T, N)) * 0.25).astype(np.float32)
    v0 = (rng.standard_normal(N) * 0.5).astype(np.float32)
    g = rng.standard_normal((T, N)).astype(np.float32)
    gvl = rng.standard_normal(N).astype(np.float32)
    k = float(np.float32(1 / 3.)) if kind == 'PLIF' else None
    kw = dict(kind=kind, scale=7.5, tau=2.5, k=k, v_th=1.0, v_reset=0.25)
    alpha = 2.0 if surrogate == 'ATan' else 4.0
    assert _lib.neuron_bwd_rc_supported(T) and not _lib.neuron_bwd_rc_supported(3)
    for v_init in (v0, None):
        f = hip_fwd(x, v_init=v_init, **kw)
        for detach in (True, False):
            a = hip_bwd(g, f['h'], v_init=v_init, g_v_last=gvl, surrogate=surrogate, alpha=alpha, detach_reset=detach, **kw)
            gd, xd, vd, gvd = _dev(g), _dev(x), _dev(v_init), _dev(gvl)
            g_x = torch.full_like(xd, float('nan'))
            g_vi = torch.empty(N, device=DEV)
            g_k = torch.zeros(1, device=DEV) if kind == 'PLIF' else None
            ws = torch.empty(_lib.gk_ws_floats(), device=DEV) if kind == 'PLIF' else None
            kd = None if k is None else torch.tensor([k], dtype=torch.float32, device=DEV)
            _lib.neuron_bwd_rc(gd, gvd, xd, vd, g_x, g_vi, g_k, ws, T, N, kw['scale'], KIND[kind], kw['tau'], kd, kw['v_th'],
                               kw['v_reset'], SG[surrogate], alpha, detach)
            assert bit_equal(g_x.cpu().numpy(), a['g_x']) and bit_equal(g_vi.cpu().numpy(), a['g_v_init'])
            if kind == 'PLIF':
                assert float(g_k.item()) == a['g_k']
            assert torch.equal(xd.cpu(), torch.from_numpy(x))          # the layer input is left intact
    if surrogate == 'ATan' and N <= 70000:
        ref = c_oracle.neuron_bwd(g, f['h'], v_init=None, g_v_last=gvl, surrogate='ATan', alpha=alpha, detach_reset=False, **kw)
        assert bit_equal(g_x.cpu().numpy(), ref['g_x'])
    xd3 = torch.zeros(3, 8, device=DEV)
    with pytest.raises(_lib.SSNeuronError):
        _lib.neuron_bwd_rc(xd3, None, xd3, None, torch.empty_like(xd3), None, None, None, 3, 8, 1.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, True)


@pytest.mark.parametrize('Cin,Cout,hw,HW', [(512, 256, (17, 22), (33, 44)), (128, 64, (9, 11), (18, 23))])
def test_exact_bf16x3_projection_has_fp32_gemm_accuracy(Cin, Cout, hw, HW):
    """The decoder's forward projection on spike inputs (fused.EXACT_SPLIT_GEMM: fp32 weight split into 3 bf16 terms, bf16 MFMA,
    fp32 accumulate) against a float64 reference: its error is at the level of the plain fp32 GEMM's (stated: <= 1.5x that error
    + 1e-7 of max|out|), backward untouched (same tensors, bit for bit), non-spike input refused in checking mode."""
    from stereospike_amd import fused
    from stereospike_amd.network.blocks import NNConvUpsampling
    torch.manual_seed(7)
    k, NB = 5, 6
    m = NNConvUpsampling(Cin, Cout, k, HW, bias=False).to(DEV)
    x = torch.randint(0, 3, (NB, hw[0], hw[1], Cin), device=DEV).float()
    g = torch.randn(NB, HW[0], HW[1], Cout, device=DEV)
    outs = {}
    from stereospike_amd import config
    for exact in (True, False):
        # (BOX_BWD off: this test is about the g_P forms of the backward, which share their data gradient between the two projection forms; SUB_FWD off: it is
        #  about the PROJECTION forms of the forward — the sub-pixel kernel has its own element-wise bound in test_upconv_sub_forward)
        with config.engine_config(EXACT_SPLIT_GEMM=exact, ASSERT_EXACT_SPLIT=True, BOX_BWD=False, SUB_FWD=False):
            xin = x.clone().requires_grad_()
            y = m.forward_projected_cl(xin, spikes_in=True)
            gx, gw = torch.autograd.grad(y, (xin, m.up[1].weight), g)
            outs[exact] = (y.detach(), gx, gw)
    wd = m.up[1].weight.detach().double().requires_grad_()
    ref = torch.nn.functional.conv2d(m.up[0](x.permute(0, 3, 1, 2).double()), wd).permute(0, 2, 3, 1)
    gw_ref, = torch.autograd.grad(ref, wd, g.double())
    e_exact = float((outs[True][0].double() - ref).abs().max())
    e_fp32 = float((outs[False][0].double() - ref).abs().max())
    assert e_exact <= 1.5 * e_fp32 + 1e-7 * float(ref.abs().max()), (e_exact, e_fp32)
    assert e_exact <= 1e-5 * float(ref.abs().max())
    assert torch.equal(outs[True][1], outs[False][1])              # data gradient: the same fp32 GEMM either way
    # weight gradient: the exact MFMA contraction (ss_spike_wgrad_f32) where it is compiled, else the same fp32 GEMM
    ew_exact = float((outs[True][2].double() - gw_ref).abs().max())
    ew_fp32 = float((outs[False][2].double() - gw_ref).abs().max())
    assert ew_exact <= 1.5 * ew_fp32 + 1e-7 * float(gw_ref.abs().max()), (ew_exact, ew_fp32)
    from stereospike_amd import _lib as _l
    if not _l.spike_wgrad_supported(Cin, 25 * Cout):
        assert torch.equal(outs[True][2], outs[False][2])
    for sub in (False, True):                                       # checking mode refuses a non-spike input on either forward form
        with config.engine_config(ASSERT_EXACT_SPLIT=True, SUB_FWD=sub), pytest.raises(AssertionError):
            m.forward_projected_cl(x + 0.3, spikes_in=True)


@pytest.mark.parametrize('Cin,Cout,k,s,pad,hw', [(512, 512, 3, 1, 1, (17, 22)), (256, 512, 5, 2, 2, (33, 44)), (128, 256, 5, 2, 2, (21, 30)),
                                                   (8, 4, 3, 2, 0, (7, 9)), (256, 512, 5, 2, 2, (4, 5)), (512, 512, 3, 1, 1, (2, 3)),
                                                   (128, 256, 5, 2, 2, (8, 10))])
def test_spike_conv_as_exact_bf16x3_gemm(Cin, Cout, k, s, pad, hw):
    """ss_im2col_cl_bf16 == the patch matrix of torch's unfold, bit for bit; ss_split3_bf16 terms sum back to the fp32 input exactly;
    fused._SpikeConvCL (forward + weight gradient as bf16x3 GEMMs, data gradient on MIOpen) against a float64 convolution: errors at the
    level of the fp32 convolution's own (stated: <= 2x + 1e-7 of max)."""
    import torch.nn.functional as F
    from stereospike_amd import _lib, fused
    torch.manual_seed(3)
    NB = 5
    h, w = hw
    x = torch.randint(0, 3, (NB, h, w, Cin), device=DEV).float()
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    M, K = NB * ho * wo, k * k * Cin
    A = torch.empty(M, K, dtype=torch.bfloat16, device=DEV)
    _lib.im2col_cl_bf16(x, A, NB, h, w, Cin, k, s, pad, ho, wo)
    cols = F.unfold(x.permute(0, 3, 1, 2), k, padding=pad, stride=s)              # [NB, Cin*k*k, L], row index = (c, ky, kx)
    ref = cols.view(NB, Cin, k * k, ho * wo).permute(0, 3, 2, 1).reshape(M, K)     # -> (ky, kx, c) column order
    assert torch.equal(A.float(), ref)
    g = torch.randn(M, Cout, device=DEV) * torch.logspace(-6, 3, M, device=DEV).unsqueeze(1)
    g3 = torch.empty(M, 3 * Cout, dtype=torch.bfloat16, device=DEV)
    _lib.split3_bf16(g, g3, M, Cout)
    parts = g3.view(M, 3, Cout).float()
    assert torch.equal((parts[:, 0].double() + parts[:, 1].double() + parts[:, 2].double()).float(), g)
    if M * K <= 2_000_000:                                       # against the numpy oracle, bit for bit
        from oracle import np_operands as no
        A_ref, _ = no.im2col_cl(x.cpu().numpy(), k, s, pad)
        assert np.array_equal(A.float().cpu().numpy(), A_ref)
        hi, mid, lo = no.split3(g.cpu().numpy())
        assert np.array_equal(parts.cpu().numpy(), np.stack((hi, mid, lo), 1))
    if Cin < fused._SPIKE_CONV_MIN_CIN:
        return
    conv = torch.nn.Conv2d(Cin, Cout, k, s, pad, bias=False).to(DEV)
    gy = torch.randn(NB, ho, wo, Cout, device=DEV)
    from stereospike_amd import config
    with config.engine_config(ASSERT_EXACT_SPLIT=True):
        xin = x.clone().requires_grad_()
        y = fused.spike_conv_cl(xin, conv)
        assert y is not None and tuple(y.shape) == (NB, ho, wo, Cout)
        gx, gw = torch.autograd.grad(y, (xin, conv.weight), gy)
    xr = x.clone().requires_grad_()
    y32 = F.conv2d(xr.permute(0, 3, 1, 2), conv.weight, None, s, pad)
    gx32, gw32 = torch.autograd.grad(y32, (xr, conv.weight), gy.permute(0, 3, 1, 2))
    xd = x.double().requires_grad_()
    wd = conv.weight.detach().double().requires_grad_()
    y64 = F.conv2d(xd.permute(0, 3, 1, 2), wd, None, s, pad)
    gx64, gw64 = torch.autograd.grad(y64, (xd, wd), gy.double().permute(0, 3, 1, 2))

    def err(a, b):
        return float((a.double() - b).abs().max())
    assert err(y.permute(0, 3, 1, 2), y64) <= 2 * err(y32, y64) + 1e-7 * float(y64.abs().max())
    assert err(gw, gw64) <= 2 * err(gw32, gw64) + 1e-7 * float(gw64.abs().max())
    if k == 5 and s == 2 and pad == 2 and fused.CONV_DGRAD_MFMA:
        # the data gradient of the stride-2 5x5 geometries is the six-term MFMA kernel (ss_conv_s2_dgrad_f32): held to ITS OWN element-wise bound
        # 2^-21 sum |g| |w| (test_conv_s2_dgrad_mfma), not to a bar relative to whatever error the library's fp32 kernel happens to have on this box
        # (a bar of that kind failed once at 2.9e-6 vs 1.8e-6, profiles/r03/pytest_gpu_r03_clean_run3_failed_at_158.log)
        mag = torch.autograd.grad(F.conv2d(xd.permute(0, 3, 1, 2), wd.abs(), None, s, pad), xd, gy.double().abs().permute(0, 3, 1, 2))[0]
        assert bool(((gx.double() - gx64).abs() <= 2.0 ** -21 * mag + 1e-30).all()), float(((gx.double() - gx64).abs() / (mag + 1e-30)).max()) * 2 ** 21
    else:
        assert err(gx, gx64) <= 2 * err(gx32, gx64) + 3e-7 * float(gx64.abs().max())


@pytest.mark.parametrize('kind', ['IF', 'PLIF'])
@pytest.mark.parametrize('T,N', [(5, 4096 * 3), (5, 1003), (10, 8192), (1, 64)])
def test_backward_with_forked_output_gradients(kind, T, N):
    """ss_neuron_bwd_fork_f32: two output gradients added on load == the recompute backward fed their fp32 sum, bit for bit (g_x, g_v_init,
    dL/dk); g_sum_seq == g1 + g2; the saved-h form refuses a second gradient."""
    from stereospike_amd import _lib
    rng = np.random.default_rng(N + T)
    x = _dev((rng.standard_normal((T, N)) * 0.25).astype(np.float32))
    g1 = _dev(rng.standard_normal((T, N)).astype(np.float32))
    g2 = _dev((rng.standard_normal((T, N)) * 3).astype(np.float32))
    v0 = _dev((rng.standard_normal(N) * 0.5).astype(np.float32))
    k = torch.tensor([0.3], device=DEV) if kind == 'PLIF' else None
    args = (T, N, 7.5, KIND[kind], 2.0, k, 1.0, 0.0, SG['ATan'], 2.0, True)
    ws = torch.empty(_lib.gk_ws_floats(), device=DEV) if kind == 'PLIF' else None
    gx_a, gv_a, gk_a = torch.empty_like(x), torch.empty(N, device=DEV), (torch.zeros(1, device=DEV) if kind == 'PLIF' else None)
    _lib.neuron_bwd_rc(g1 + g2, None, x, v0, gx_a, gv_a, gk_a, ws, *args)
    gx_b, gv_b, gk_b = torch.empty_like(x), torch.empty(N, device=DEV), (torch.zeros(1, device=DEV) if kind == 'PLIF' else None)
    gsum = torch.full_like(x, float('nan'))
    _lib.neuron_bwd_fork(g1, g2, gsum, None, None, x, v0, gx_b, gv_b, gk_b, ws, *args)
    assert torch.equal(gx_a, gx_b) and torch.equal(gv_a, gv_b) and torch.equal(gsum, g1 + g2)
    if kind == 'PLIF':
        assert float(gk_a) == float(gk_b)
    gx_c = torch.empty_like(x)
    _lib.neuron_bwd_fork(g1, g2, None, None, None, x, v0, gx_c, None, None if kind != 'PLIF' else torch.zeros(1, device=DEV), ws, *args)
    assert torch.equal(gx_c, gx_a)
    with pytest.raises(_lib.SSNeuronError):
        _lib.neuron_bwd_fork(g1, g2, None, None, x, None, v0, gx_c, None, None, None, *args)      # saved-h form + second gradient


def test_weight_split_in_one_launch_equals_the_torch_definition():
    """fused._split3_cols / _split3_bf16 (the exact three-term bf16 split of a synapse's weights through ss_split3_bf16, one launch) ==
    the cast / subtract / cast chain they replace, bit for bit; the three terms sum back to the fp32 weight exactly."""
    from stereospike_amd import fused
    torch.manual_seed(3)
    for K, N in ((3200, 256), (800, 64), (27, 12)):
        Wt = (torch.randn(K, N, device=DEV) * torch.logspace(-6, 1, N, device=DEV)).contiguous()
        Wh = Wt.to(torch.bfloat16)
        r = Wt - Wh.float()
        Wm = r.to(torch.bfloat16)
        Wl = (r - Wm.float()).to(torch.bfloat16)
        cols = fused._split3_cols(Wt)
        assert cols.dtype == torch.bfloat16 and torch.equal(cols, torch.cat((Wh, Wm, Wl), 1))
        assert torch.equal(fused._split3_bf16(Wt), torch.cat((Wh, Wm, Wl), 0))
        assert torch.equal(cols[:, :N].double() + cols[:, N:2 * N].double() + cols[:, 2 * N:].double(), Wt.double())


@pytest.mark.parametrize('kind', ['IF', 'LIF', 'PLIF'])
@pytest.mark.parametrize('T,rows,C', [(5, 96, 32), (5, 17, 64), (10, 8, 128), (1, 3, 256), (4, 5, 512), (2, 7, 4), (5, 40000, 32), (8, 33, 16)])
def test_backward_with_low_rank_second_gradient(kind, T, rows, C):
    """ss_neuron_bwd_fork_lr_f32: the second gradient as the rank-9 pair of a prediction head (lr_p [T, rows, 9], lr_w [9, C]), formed in
    registers == the recompute backward fed g1 + oracle.np_lowrank.head_input_gradient(lr_p, lr_w), bit for bit (g_x, g_v_init, g_sum, dL/dk);
    without a dense first gradient == the backward fed the expansion alone; unsupported shapes are refused."""
    from stereospike_amd import _lib
    from oracle import np_lowrank
    N = rows * C
    rng = np.random.default_rng(N + T)
    x = _dev((rng.standard_normal((T, N)) * 0.25).astype(np.float32))
    g1 = _dev(rng.standard_normal((T, N)).astype(np.float32))
    lr_p = (rng.standard_normal((T, rows, 9)) * 2).astype(np.float32)
    lr_w = rng.standard_normal((9, C)).astype(np.float32)
    g2 = _dev(np_lowrank.head_input_gradient(lr_p, lr_w).reshape(T, N))
    assert float(np.abs(g2.cpu().numpy().reshape(T, rows, C) - np_lowrank.head_input_gradient64(lr_p, lr_w)).max()) < 1e-4
    P, Wl = _dev(lr_p), _dev(lr_w)
    v0 = _dev((rng.standard_normal(N) * 0.5).astype(np.float32))
    k = torch.tensor([0.3], device=DEV) if kind == 'PLIF' else None
    args = (T, N, 7.5, KIND[kind], 2.0, k, 1.0, 0.0, SG['ATan'], 2.0, True)
    ws = torch.empty(_lib.gk_ws_floats(), device=DEV) if kind == 'PLIF' else None

    def gk():
        return torch.zeros(1, device=DEV) if kind == 'PLIF' else None
    assert _lib.neuron_bwd_fork_lr_supported(T, N, C, 9)
    gx_a, gv_a, gk_a = torch.empty_like(x), torch.empty(N, device=DEV), gk()
    _lib.neuron_bwd_rc(g1 + g2, None, x, v0, gx_a, gv_a, gk_a, ws, *args)
    gx_b, gv_b, gk_b = torch.empty_like(x), torch.empty(N, device=DEV), gk()
    gsum = torch.full_like(x, float('nan'))
    _lib.neuron_bwd_fork_lr(g1, P, Wl, gsum, None, x, v0, gx_b, gv_b, gk_b, ws, *args)
    assert torch.equal(gx_a, gx_b) and torch.equal(gv_a, gv_b) and torch.equal(gsum, g1 + g2)
    if kind == 'PLIF':
        assert float(gk_a) == float(gk_b)
    # the pair alone (the full-resolution stage: its only consumer is the head)
    gx_c, gx_d = torch.empty_like(x), torch.empty_like(x)
    _lib.neuron_bwd_rc(g2, None, x, v0, gx_c, None, gk(), ws, *args)
    _lib.neuron_bwd_fork_lr(None, P, Wl, None, None, x, v0, gx_d, None, gk(), ws, *args)
    assert torch.equal(gx_c, gx_d)
    with pytest.raises(_lib.SSNeuronError):
        _lib.neuron_bwd_fork_lr(None, P, Wl, gsum, None, x, v0, gx_d, None, gk(), ws, *args)     # a "sum" needs a dense first gradient
    assert not _lib.neuron_bwd_fork_lr_supported(T, N, C, 25) and not _lib.neuron_bwd_fork_lr_supported(3, N, C, 9)
    assert not _lib.neuron_bwd_fork_lr_supported(T, 96 * 48, 48, 9)                                 # 1024 % C != 0
    assert not _lib.neuron_bwd_fork_lr_supported(T, 4 * 1024, 1024, 9)                              # the weight matrix must fit the LDS budget (C <= 512)


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
def test_upconv_cl_x16_equals_fp32_gather_with_narrowed_io(dt):
    """ss_upconv_cl_fwd_x16 == nearest-even narrowing of ss_upconv_cl_fwd_f32's output; ss_upconv_cl_bwd_x16 on a 16-bit gradient ==
    ss_upconv_cl_bwd_f32 on the same values widened — both bit for bit (the 16-bit modes change the I/O format only)."""
    from stereospike_amd import _lib
    from stereospike_amd.fused import nearest_tables
    torch.manual_seed(1)
    NB, C, k, (h, w), (H, W) = 3, 64, 5, (9, 11), (18, 23)
    sy, ylo, yhi = (t.to(DEV) for t in nearest_tables(h, H + k - 1))
    sx, xlo, xhi = (t.to(DEV) for t in nearest_tables(w, W + k - 1))
    P = torch.randn(NB, h, w, k * k * C, device=DEV)
    o32 = torch.empty(NB, H, W, C, device=DEV)
    o16 = torch.empty(NB, H, W, C, device=DEV, dtype=dt)
    _lib.upconv_cl_fwd(P, sy, sx, None, o32, NB, k, C, h, w, H, W)
    _lib.upconv_cl_fwd_x16(P, sy, sx, None, o16, NB, k, C, h, w, H, W)
    assert torch.equal(o16, o32.to(dt))
    g16 = torch.randn(NB, H, W, C, device=DEV).to(dt)
    gp_a, gp_b = torch.empty_like(P), torch.empty_like(P)
    _lib.upconv_cl_bwd(g16.float(), ylo, yhi, xlo, xhi, gp_a, NB, k, C, h, w, H, W)
    _lib.upconv_cl_bwd_x16(g16, ylo, yhi, xlo, xhi, gp_b, NB, k, C, h, w, H, W)
    assert torch.equal(gp_a, gp_b)
    # ss_upconv_cl_bwd_lowp: g_P written as bf16 == nearest-even narrowing of the fp32 adjoint, for 16-bit and fp32 gradients
    gp_c = torch.empty(P.shape, dtype=torch.bfloat16, device=DEV)
    _lib.upconv_cl_bwd_lowp(g16, ylo, yhi, xlo, xhi, gp_c, NB, k, C, h, w, H, W)
    assert torch.equal(gp_c, gp_a.to(torch.bfloat16))
    if dt == torch.float16:     # ABI 10: g_P in fp16 for an fp16 gradient (the fp16 mode's g_P forms); any other pairing is refused
        gp_h = torch.empty(P.shape, dtype=torch.float16, device=DEV)
        _lib.upconv_cl_bwd_lowp(g16, ylo, yhi, xlo, xhi, gp_h, NB, k, C, h, w, H, W)
        assert torch.equal(gp_h, gp_a.to(torch.float16))
    else:
        with pytest.raises(_lib.SSNeuronError):
            _lib.upconv_cl_bwd_lowp(g16, ylo, yhi, xlo, xhi, torch.empty(P.shape, dtype=torch.float16, device=DEV), NB, k, C, h, w, H, W)
    g32 = torch.randn(NB, H, W, C, device=DEV)
    _lib.upconv_cl_bwd(g32, ylo, yhi, xlo, xhi, gp_a, NB, k, C, h, w, H, W)
    _lib.upconv_cl_bwd_lowp(g32, ylo, yhi, xlo, xhi, gp_c, NB, k, C, h, w, H, W)
    assert torch.equal(gp_c, gp_a.to(torch.bfloat16))


def test_c_caller_runs_against_the_library():
    """examples/c_caller.c built with gcc and run on the device: the C-ABI used from plain C (no Python, no torch in that process)."""
    import os
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, 'stereospike_amd', 'lib')
    if shutil.which('gcc') is None:
        pytest.skip('no gcc on this box')
    torch_lib = os.path.join(os.path.dirname(torch.__file__), 'lib')           # the HIP runtime the library was loaded with in-process
    hip_dir = '/opt/rocm/lib' if os.path.exists('/opt/rocm/lib/libamdhip64.so') else torch_lib
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, 'c_caller')
        r = subprocess.run(['gcc', '-std=c99', '-I' + os.path.join(root, 'include'), os.path.join(root, 'examples', 'c_caller.c'),
                            '-L' + lib_dir, '-lss_neuron', '-L' + hip_dir, '-lamdhip64', '-Wl,-rpath-link,' + hip_dir, '-o', exe],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        env = dict(os.environ, LD_LIBRARY_PATH=lib_dir + ':' + hip_dir + ':' + os.environ.get('LD_LIBRARY_PATH', ''))
        r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert 'spikes out of 20480 updates' in r.stdout and ' 0 spikes' not in r.stdout, r.stdout


# ======================================================================================================
# ABI 2: descriptor forward (2-bit packed spikes, counter partials), forked x16 backward, packed readers
# ======================================================================================================
@pytest.mark.parametrize('kind', ['IF', 'LIF', 'PLIF'])
@pytest.mark.parametrize('T,N', [(1, 16), (5, 4096 + 16), (10, 16 * 1001), (5, 32 * 65 * 87 * 4), (4, 48)])
def test_packed_spike_io_vs_oracle(kind, T, N):
    """ss_neuron_fwd_ex: packed output == pack(oracle output) bit for bit (with and without the dense output, with a dense and a packed
    skip operand), v_last and the counters unchanged; unpack(packed) == the dense tensor (oracle/np_pack.py)."""
    from oracle import np_pack
    from stereospike_amd import _lib
    rng = np.random.default_rng(T * 31 + N)
    x = (rng.standard_normal((T, N)) * 0.2).astype(np.float32)
    skip = rng.integers(0, 3, (T, N)).astype(np.float32)
    v0 = (rng.standard_normal(N) * 0.5).astype(np.float32) if N % 32 == 0 else None
    kw = dict(kind=kind, scale=10.0, tau=3.0, k=np.float32(1 / 3.) if kind == 'PLIF' else None, v_th=1.0, v_reset=0.0)
    kd = None if kind != 'PLIF' else torch.tensor([1 / 3.], device=DEV)
    xd, vd = _dev(x), _dev(v0)
    for use_skip in (False, True):
        ref = c_oracle.neuron_fwd(x, v_init=v0, skip_seq=skip if use_skip else None, **kw)
        want = np_pack.pack(ref['out']).view(np.int32)
        sk_dense = _dev(skip) if use_skip else None
        sk_packed = torch.from_numpy(np_pack.pack(skip).view(np.int32)).to(DEV) if use_skip else None
        for skip_form in (('dense', 'packed') if use_skip else ('none',)):
            for with_dense in (True, False):
                out = torch.empty(T, N, device=DEV) if with_dense else None
                pk = torch.zeros(T, N // 16, dtype=torch.int32, device=DEV)
                v = torch.empty(N, device=DEV)
                nnz = torch.zeros(2, dtype=torch.int64, device=DEV)
                ws = torch.empty(_lib.cnt_ws_words(N), dtype=torch.int32, device=DEV)
                _lib.neuron_fwd_ex(xd, vd, sk_dense if skip_form == 'dense' else None, sk_packed if skip_form == 'packed' else None,
                                   out, pk, None, v, nnz, ws, T, N, 10.0, KIND[kind], 3.0, kd, 1.0, 0.0)
                assert np.array_equal(pk.cpu().numpy(), want), (use_skip, skip_form, with_dense)
                assert bit_equal(v, ref['v_last'])
                assert int(nnz[1]) == int((ref['out'] != 0).sum()) and int(nnz[0]) == int(((ref['h'] - 1.0) >= 0).sum())
                if with_dense:
                    assert np.array_equal(out.cpu().numpy(), ref['out'])
                for dt in (torch.float32, torch.bfloat16, torch.float16):
                    dense = torch.empty(T, N, dtype=dt, device=DEV)
                    _lib.unpack_spikes(pk, dense, T * N)
                    assert np.array_equal(dense.float().cpu().numpy(), ref['out'])
    # the training pair: forward without h (packed only) + recompute backward == the saved-h pair
    g = rng.standard_normal((T, N)).astype(np.float32)
    rb = c_oracle.neuron_bwd(g, ref['h'], v_init=v0, surrogate='ATan', alpha=2.0, **kw)
    gx, gvi = torch.empty(T, N, device=DEV), torch.empty(N, device=DEV)
    gk = torch.zeros(1, device=DEV) if kind == 'PLIF' else None
    gws = torch.empty(_lib.gk_ws_floats(), device=DEV) if kind == 'PLIF' else None
    _lib.neuron_bwd_rc(_dev(g), None, xd, vd, gx, gvi, gk, gws, T, N, 10.0, KIND[kind], 3.0, kd, 1.0, 0.0, 0, 2.0, True)
    assert bit_equal(gx, rb['g_x'])


def test_packed_io_argument_validation():
    from stereospike_amd import _lib
    x = torch.zeros(3, 32, device=DEV)
    v = torch.empty(32, device=DEV)
    pk = torch.zeros(3, 2, dtype=torch.int32, device=DEV)
    with pytest.raises(_lib.SSNeuronError):                                  # T = 3 is not a compile-time T
        _lib.neuron_fwd_ex(x, None, None, None, None, pk, None, v, None, None, 3, 32, 1.0, 0, 2.0, None, 1.0, 0.0)
    x5 = torch.zeros(5, 32, device=DEV)
    pk5 = torch.zeros(5, 2, dtype=torch.int32, device=DEV)
    with pytest.raises(_lib.SSNeuronError):                                  # packed output together with a saved h
        _lib.neuron_fwd_ex(x5, None, None, None, None, pk5, torch.empty(5, 32, device=DEV), v, None, None, 5, 32, 1.0, 0, 2.0, None, 1.0, 0.0)
    with pytest.raises(_lib.SSNeuronError):                                  # neither output
        _lib.neuron_fwd_ex(x5, None, None, None, None, None, None, v, None, None, 5, 32, 1.0, 0, 2.0, None, 1.0, 0.0)
    with pytest.raises(_lib.SSNeuronError):                                  # N % 16 != 0
        _lib.neuron_fwd_ex(torch.zeros(5, 24, device=DEV), None, None, None, None, torch.zeros(5, 1, dtype=torch.int32, device=DEV), None,
                           torch.empty(24, device=DEV), None, None, 5, 24, 1.0, 0, 2.0, None, 1.0, 0.0)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_counter_partials_match_count_nonzero_at_full_size(dt):
    """Config-3 bottom layer (2.3e8 updates): the per-workgroup-partials counters (full grid) == count_nonzero, twice (accumulating)."""
    from stereospike_amd import _lib
    T, N = 5, 16 * 32 * 260 * 346
    g = torch.Generator(device=DEV).manual_seed(3)
    x = (torch.randn(T, N, device=DEV, generator=g) * 0.15).to(dt)
    out, v = torch.empty_like(x), torch.empty(N, device=DEV)
    nnz = torch.zeros(2, dtype=torch.int64, device=DEV)
    ws = torch.empty(_lib.cnt_ws_words(N), dtype=torch.int32, device=DEV)
    for rep in (1, 2):
        _lib.neuron_fwd_ex(x, None, None, None, out, None, None, v, nnz, ws, T, N, 10.0, 0, 2.0, None, 1.0, 0.0)
        assert int(nnz[0]) == rep * int(out.count_nonzero()) == int(nnz[1])


def test_unpack_copies_and_packed_im2col_vs_oracle():
    from oracle import np_pack, np_operands
    from stereospike_amd import _lib
    rng = np.random.default_rng(5)
    rows, Cc = 37 * 16, 64
    xs = rng.integers(0, 4, (rows, Cc)).astype(np.float32)
    pk = torch.from_numpy(np_pack.pack(xs.reshape(-1)).view(np.int32)).to(DEV)
    x3 = torch.empty(rows, 3 * Cc, dtype=torch.bfloat16, device=DEV)
    _lib.unpack_spikes(pk, x3, rows * Cc, row_len=Cc, copies=3)
    assert np.array_equal(x3.float().cpu().numpy(), np.concatenate([xs, xs, xs], 1))
    for (NB, h, w, Cin, k, s, pad) in ((2, 9, 12, 128, 5, 2, 2), (1, 7, 8, 512, 3, 1, 1), (3, 6, 6, 16, 5, 1, 2)):
        x = rng.integers(0, 4, (NB, h, w, Cin)).astype(np.float32)
        ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        xp = torch.from_numpy(np_pack.pack(x.reshape(-1)).view(np.int32)).to(DEV)
        A1 = torch.empty(NB * ho * wo, k * k * Cin, dtype=torch.bfloat16, device=DEV)
        A2 = torch.empty_like(A1)
        _lib.im2col_cl_bf16_packed(xp, A1, NB, h, w, Cin, k, s, pad, ho, wo)
        _lib.im2col_cl_bf16(torch.from_numpy(x).to(DEV), A2, NB, h, w, Cin, k, s, pad, ho, wo)
        assert torch.equal(A1.view(torch.int16), A2.view(torch.int16)), (NB, h, w, Cin, k, s, pad)
        ref = np_operands.im2col_cl(x, k, s, pad)[0]
        assert np.array_equal(A1.float().cpu().numpy(), ref.reshape(A1.shape))


def test_wgrad_reduce3_vs_oracle():
    """ss_wgrad_reduce3_f32 (slices + bf16 terms of the encoder weight-gradient GEMM summed, Conv2d layout written) == oracle/np_operands.wgrad_reduce3
    bit for bit; the network's geometries (conv3, conv4, a bottleneck conv) and ragged ones."""
    from oracle import np_operands
    from stereospike_amd import _lib
    rng = np.random.default_rng(11)
    for (S, k, Cin, Cout) in ((4, 5, 128, 256), (4, 5, 256, 512), (2, 3, 512, 512), (1, 3, 8, 32), (3, 7, 16, 64), (5, 1, 24, 96)):
        parts = (rng.standard_normal((S, k * k * Cin, 3, Cout)) * np.array([1.0, 2.0 ** -8, 2.0 ** -16]).reshape(1, 1, 3, 1)).astype(np.float32)
        g_w = torch.full((Cout, Cin, k, k), float('nan'), device=DEV)
        _lib.wgrad_reduce3(torch.from_numpy(parts).to(DEV), g_w, S, k, Cin, Cout)
        assert np.array_equal(g_w.cpu().numpy(), np_operands.wgrad_reduce3(parts, k, Cin, Cout)), (S, k, Cin, Cout)
    with pytest.raises(_lib.SSNeuronError):
        _lib.wgrad_reduce3(torch.zeros(1, 25 * 12, 3, 32, device=DEV), torch.zeros(32, 12, 5, 5, device=DEV), 1, 5, 12, 32)      # C_in % 8 != 0


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('kind', ['IF', 'PLIF'])
@pytest.mark.parametrize('T,N', [(5, 4096 * 3), (5, 1003), (10, 8192), (1, 64)])
def test_x16_backward_with_forked_output_gradients(dt, kind, T, N):
    """ss_neuron_bwd_fork_x16 == oracle/np_x16.py with g2_bits: g_x, g_v_init, g_sum bit-exact (ATan)."""
    from oracle import np_x16
    from stereospike_amd import _lib
    name = 'f16' if dt == torch.float16 else 'bf16'
    g = torch.Generator().manual_seed(T * 13 + N)
    x = (torch.randn(T, N, generator=g) * 0.2).to(dt)
    g1, g2 = torch.randn(T, N, generator=g).to(dt), torch.randn(T, N, generator=g).to(dt)
    bits = lambda t: t.view(torch.int16).numpy().view(np.uint16)
    kw = dict(kind=kind, scale=10.0, tau=3.0, k=np.float32(1 / 3.) if kind == 'PLIF' else None, v_th=1.0, v_reset=0.0)
    ref = np_x16.neuron_fwd(bits(x), name, **kw)
    rb = np_x16.neuron_bwd(bits(g1), ref['h'], name, surrogate='ATan', alpha=2.0, g2_bits=bits(g2), **kw)
    kd = None if kind != 'PLIF' else torch.tensor([1 / 3.], device=DEV)
    gx, gs, gvi = torch.empty(T, N, dtype=dt, device=DEV), torch.empty(T, N, dtype=dt, device=DEV), torch.empty(N, device=DEV)
    gk = torch.zeros(1, device=DEV) if kind == 'PLIF' else None
    ws = torch.empty(_lib.gk_ws_floats(), device=DEV) if kind == 'PLIF' else None
    _lib.neuron_bwd_fork_x16(g1.to(DEV), g2.to(DEV), gs, None, x.to(DEV), None, gx, gvi, gk, ws, T, N, 10.0, KIND[kind], 3.0, kd, 1.0, 0.0, 0, 2.0, True)
    assert np.array_equal(bits(gx.cpu()), rb['g_x']) and np.array_equal(bits(gs.cpu()), rb['g_sum'])
    assert bit_equal(gvi, rb['g_v_init'])
    if kind == 'PLIF':
        assert abs(float(gk) - rb['g_k']) <= 1e-5 * abs(rb['g_k']) + 2e-7 * T * N


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('T,N', [(2, 4096), (3, 8 * 777), (4, 1024), (7, 2048 + 8), (8, 4104), (10, 8 * 12345)])
def test_x16_forward_every_time_step_count(dt, T, N):
    """Compile-time-T (1, 2, 4, 5, 8, 10) and run-time-T instantiations of the 16-bit forward, 8- and 4-neuron lanes == np_x16."""
    from oracle import np_x16
    from stereospike_amd import _lib
    name = 'f16' if dt == torch.float16 else 'bf16'
    g = torch.Generator().manual_seed(T * 3 + N)
    x = (torch.randn(T, N, generator=g) * 0.2).to(dt)
    skip = torch.randint(0, 3, (T, N), generator=g).to(dt)
    bits = lambda t: t.view(torch.int16).numpy().view(np.uint16)
    for sk in (None, skip):
        ref = np_x16.neuron_fwd(bits(x), name, kind='LIF', scale=10.0, tau=3.0, v_th=1.0, v_reset=0.0, skip_bits=None if sk is None else bits(sk))
        for save_h in (True, False):
            out, v = torch.empty(T, N, dtype=dt, device=DEV), torch.empty(N, device=DEV)
            h = torch.empty(T, N, device=DEV) if save_h else None
            _lib.neuron_fwd_x16(x.to(DEV), None, None if sk is None else sk.to(DEV), out, h, v, None, T, N, 10.0, 1, 3.0, None, 1.0, 0.0)
            assert np.array_equal(bits(out.cpu()), ref['out']) and bit_equal(v, ref['v_last'])
            if save_h:
                assert bit_equal(h, ref['h'])






# ======================================================================================================
# Winograd F(2x2, 3x3) data gradient of the bottleneck convs (ss_wino_dgrad_*_f32)
# ======================================================================================================
@pytest.mark.parametrize('NB,H,W,Co,Ci', [(3, 17, 22, 64, 32), (2, 5, 7, 8, 12), (1, 4, 4, 4, 4), (2, 1, 3, 8, 4), (80, 17, 22, 512, 512)])
def test_winograd_dgrad_kernels(NB, H, W, Co, Ci):
    """The three transform kernels are BIT-EXACT against oracle/np_winograd.py; the composed data gradient (transforms + torch.bmm) matches the
    float64 gradient of the 3x3 / pad 1 convolution (reference blocks.py:146-159 under autograd) at fp32-GEMM accuracy, and MIOpen's own
    data gradient to the same tolerance; deterministic."""
    from oracle import np_winograd as nw
    from stereospike_amd import _lib, fused
    rng = np.random.default_rng(NB * 100 + H)
    big = NB * H * W * Co > 4e6
    g = rng.standard_normal((NB, H, W, Co)).astype(np.float32)
    w = (rng.standard_normal((Co, Ci, 3, 3)) * 0.05).astype(np.float32)
    gd, wd = torch.tensor(g, device=DEV), torch.tensor(w, device=DEV)
    T = _lib.wino_tiles(NB, H, W)
    U = torch.full((16, Co, Ci), float('nan'), device=DEV)
    V = torch.full((16, T, Co), float('nan'), device=DEV)
    _lib.wino_dgrad_weights(wd, U, Co, Ci)
    _lib.wino_dgrad_input(gd, V, NB, H, W, Co)
    assert np.array_equal(U.cpu().numpy().view(np.int32), nw.weights(w).view(np.int32))
    if not big:
        assert np.array_equal(V.cpu().numpy().view(np.int32), nw.input_tiles(g).view(np.int32))
        m = rng.standard_normal((16, T, Ci)).astype(np.float32)
        out = torch.full((NB, H, W, Ci), float('nan'), device=DEV)
        _lib.wino_dgrad_output(torch.tensor(m, device=DEV), out, NB, H, W, Ci)
        assert np.array_equal(out.cpu().numpy().view(np.int32), nw.output_tiles(m, NB, H, W).view(np.int32))
    else:                                                  # config-3 size: a sub-range against the oracle
        sub = nw.input_tiles(g[:2])
        Vv = V.view(16, NB, -1, Co)[:, :2].reshape(16, -1, Co)
        assert np.array_equal(Vv.cpu().numpy().view(np.int32), sub.view(np.int32))
    got = fused.winograd_dgrad_cl(gd, wd)
    assert torch.equal(got, fused.winograd_dgrad_cl(gd, wd))
    if big:
        gx64 = torch.ops.aten.convolution_backward(gd.double().permute(0, 3, 1, 2), torch.empty(NB, Ci, H, W, dtype=torch.float64, device=DEV),
                                                   wd.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        ref = gx64.permute(0, 2, 3, 1)
        scale = float(ref.abs().max())
        assert float((got.double() - ref).abs().max()) / scale <= 3e-6
    else:
        ref = nw.dgrad_direct64(g, w)
        assert np.abs(got.cpu().numpy() - ref).max() <= 3e-6 * max(1.0, np.abs(ref).max())
    mi = torch.ops.aten.convolution_backward(gd.permute(0, 3, 1, 2), torch.empty(NB, Ci, H, W, device=DEV), wd, None,
                                             [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0].permute(0, 2, 3, 1)
    assert float((got - mi).abs().max()) <= 2e-5 * max(1.0, float(mi.abs().max()))


def test_winograd_dgrad_inside_the_spike_conv():
    """_SpikeConvCL with WINOGRAD_DGRAD on / off: same input gradient to fp32-GEMM accuracy, identical forward and weight gradient."""
    from stereospike_amd import fused
    torch.manual_seed(1)
    conv = torch.nn.Conv2d(128, 128, 3, 1, 1, bias=False).to(DEV)
    x = (torch.rand(4, 9, 11, 128, device=DEV) < 0.3).float().requires_grad_()
    gy = torch.randn(4, 9, 11, 128, device=DEV)
    res = {}
    from stereospike_amd import config
    for flag in (True, False):
        with config.engine_config(WINOGRAD_DGRAD=flag):
            x.grad = None; conv.weight.grad = None
            y = fused.spike_conv_cl(x, conv)
            assert y is not None
            y.backward(gy)
            res[flag] = (y.detach().clone(), x.grad.clone(), conv.weight.grad.clone())
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][2], res[False][2])
    assert float((res[True][1] - res[False][1]).abs().max()) <= 2e-5 * float(res[False][1].abs().max())


# ======================================================================================================
# exact bf16x3 MFMA weight gradient of a synapse on spike inputs (ss_spike_wgrad_f32)
# ======================================================================================================
@pytest.mark.parametrize('R,Cin,N', [(1000, 64, 800), (4099, 128, 1600), (37, 64, 96), (16 * 130 * 173, 64, 800), (5 * 65 * 87, 128, 1600),
                                     (3001, 256, 3200), (777, 512, 6400), (80 * 33 * 44, 256, 3200)])
def test_spike_wgrad_mfma(R, Cin, N):
    """g_w[ci][n] = sum_r x[r][ci] g[r][n] for spike x: products exact, fp32 accumulation -> within fp32 summation error of the float64
    contraction (far inside it: the error of an fp32 GEMM); ragged row counts; accumulate flag; run-to-run bit-identical."""
    from stereospike_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(R + N)
    x = ((torch.rand(R, Cin, device=DEV, generator=gen) < 0.3).float() + (torch.rand(R, Cin, device=DEV, generator=gen) < 0.1).float())
    g = torch.randn(R, N, device=DEV, generator=gen) * torch.exp(torch.randn(R, 1, device=DEV, generator=gen))       # wide dynamic range
    assert _lib.spike_wgrad_supported(Cin, N)
    out = torch.full((Cin, N), float('nan'), device=DEV)
    _lib.spike_wgrad(g, x, out, R, Cin, N)
    ref = x.double().t() @ g.double()
    bound = (x.double().t().abs() @ g.double().abs()) * 2.0 ** -22 + 1e-30          # a few ulp of the magnitude sum (fp32 accumulation)
    assert bool(((out.double() - ref).abs() <= bound).all()), float(((out.double() - ref).abs() / bound).max())
    lib32 = x.t() @ g
    assert float((out - ref.float()).abs().max()) <= 2.0 * float((lib32 - ref.float()).abs().max()) + 1e-6 * float(ref.abs().max())
    out2 = torch.empty_like(out)
    _lib.spike_wgrad(g, x, out2, R, Cin, N)
    assert torch.equal(out, out2)
    _lib.spike_wgrad(g, x, out2, R, Cin, N, accumulate=True)
    assert torch.equal(out2, out + out)
    assert not _lib.spike_wgrad_supported(32, 800) and not _lib.spike_wgrad_supported(64, 100)










# ======================================================================================================
# dense x dense GEMM with six bf16 cross terms (ss_gemm6_f32): the decoder's data gradient
# ======================================================================================================
@pytest.mark.parametrize('R,K,N', [(1000, 800, 64), (5000, 1600, 128), (777, 3200, 256), (300, 6400, 512), (33, 16, 64),
                                   (16 * 130 * 173, 800, 64), (8 * 65 * 87, 1600, 128)])
def test_gemm6(R, K, N):
    """|C - float64| <= 2^-21 sum_k |a||b| element-wise (six exact cross terms + fp32 accumulation: the dropped terms are one fp32 product
    rounding); at least as accurate as the library's fp32 GEMM in max error (x 2 slack); ragged row counts; deterministic."""
    from stereospike_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(R + K)
    A = torch.randn(R, K, device=DEV, generator=gen) * torch.exp(torch.randn(R, 1, device=DEV, generator=gen))
    B = torch.randn(K, N, device=DEV, generator=gen) * 0.05
    assert _lib.gemm6_supported(K, N) and not _lib.gemm6_supported(K + 1, N) and not _lib.gemm6_supported(K, 96)
    C = torch.full((R, N), float('nan'), device=DEV)
    _lib.gemm6(A, B, C, R, K, N)
    ref = A.double() @ B.double()
    bound = (A.double().abs() @ B.double().abs()) * 2.0 ** -21 + 1e-30
    assert bool(((C.double() - ref).abs() <= bound).all()), float(((C.double() - ref).abs() / bound).max())
    lib32 = A @ B
    assert float((C.double() - ref).abs().max()) <= 2.0 * float((lib32.double() - ref).abs().max()) + 1e-7 * float(ref.abs().max())
    if R * N >= 50000 and K >= 800:
        # no coherent drift: the bf16 MFMA's accumulation drifts down by ~5e-9 of the magnitude sum unless the running sum's sign alternates
        # (ss_gemm6 does); a cancelling reduction over the result (a PLIF node's dL/dw) sees the mean, not the rms
        assert abs(float(((C.double() - ref) / bound).mean())) * 2.0 ** -21 <= 1e-9
    C2 = torch.empty_like(C)
    _lib.gemm6(A, B, C2, R, K, N)
    assert torch.equal(C, C2)


def test_gemm6_batched():
    from stereospike_amd import _lib
    gen = torch.Generator(device=DEV).manual_seed(3)
    b, R, K, N = 16, 1000, 512, 512
    A = torch.randn(b, R, K, device=DEV, generator=gen); B = torch.randn(b, K, N, device=DEV, generator=gen) * 0.05
    C = torch.full((b, R, N), float('nan'), device=DEV)
    _lib.gemm6_batched(A, B, C, b, R, K, N)
    ref = torch.bmm(A.double(), B.double())
    bound = torch.bmm(A.double().abs(), B.double().abs()) * 2.0 ** -21
    assert bool(((C.double() - ref).abs() <= bound).all())
    one = torch.empty(R, N, device=DEV)
    _lib.gemm6(A[5], B[5], one, R, K, N)
    assert torch.equal(one, C[5])


# ======================================================================================================
# exact MFMA weight gradient of the stride-2 5x5 encoder convs on spike inputs (ss_spike_conv_wgrad_f32)
# ======================================================================================================
@pytest.mark.parametrize('NB,Cin,Cout,hw', [(2, 32, 64, (64, 80)), (3, 64, 128, (33, 45)), (1, 32, 64, (7, 9)), (5, 64, 128, (130, 173)), (4, 32, 64, (260, 346))])
def test_spike_conv_wgrad_mfma(NB, Cin, Cout, hw):
    """== the float64 weight gradient of conv2d(x, w, stride 2, padding 2) within fp32 accumulation error of exact products (bound 2^-22 of
    the magnitude sum per element); at least as close as MIOpen's fp32 weight gradient (x 2); odd sizes / ragged row tails; deterministic."""
    import torch.nn.functional as F
    from stereospike_amd import _lib
    h, w = hw
    gen = torch.Generator(device=DEV).manual_seed(NB + h)
    x = ((torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.3).float() + (torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.1).float())
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    g = torch.randn(NB, ho, wo, Cout, device=DEV, generator=gen) * torch.exp(torch.randn(NB, ho, wo, 1, device=DEV, generator=gen))
    assert _lib.spike_conv_wgrad_supported(Cin, Cout, 5, 2, 2) and not _lib.spike_conv_wgrad_supported(Cin, Cout, 3, 1, 1)
    gw = torch.full((Cout, Cin, 5, 5), float('nan'), device=DEV)
    _lib.spike_conv_wgrad(g, x, gw, NB, Cin, Cout, h, w)

    def wgrad(xx, gg):
        return torch.ops.aten.convolution_backward(gg.permute(0, 3, 1, 2), xx.permute(0, 3, 1, 2), torch.empty(Cout, Cin, 5, 5, dtype=xx.dtype, device=DEV),
                                                   None, [2, 2], [2, 2], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    ref = wgrad(x.double(), g.double())
    mag = wgrad(x.double(), g.double().abs())
    assert bool(((gw.double() - ref).abs() <= mag * 2.0 ** -22 + 1e-30).all()), float(((gw.double() - ref).abs() / (mag * 2.0 ** -22 + 1e-30)).max())
    mi = wgrad(x, g)
    assert float((gw.double() - ref).abs().max()) <= 2.0 * float((mi.double() - ref).abs().max()) + 1e-7 * float(ref.abs().max())
    gw2 = torch.empty_like(gw)
    _lib.spike_conv_wgrad(g, x, gw2, NB, Cin, Cout, h, w)
    assert torch.equal(gw, gw2)
    if (NB * h * w * Cin) % 16 == 0:                         # packed spike input: the window / transposed-read form (round 5) — same bound, every element written, deterministic
        from oracle import np_pack
        xp = torch.from_numpy(np_pack.pack(x.cpu().numpy().reshape(-1)).view(np.int32)).to(DEV)
        gw3 = torch.full_like(gw, float('nan'))
        _lib.spike_conv_wgrad(g, None, gw3, NB, Cin, Cout, h, w, x_packed=xp)
        assert bool(((gw3.double() - ref).abs() <= mag * 2.0 ** -22 + 1e-30).all()), float(((gw3.double() - ref).abs() / (mag * 2.0 ** -22 + 1e-30)).max())
        assert float((gw3.double() - ref).abs().max()) <= 2.0 * float((mi.double() - ref).abs().max()) + 1e-7 * float(ref.abs().max())
        gw4 = torch.empty_like(gw)
        _lib.spike_conv_wgrad(g, None, gw4, NB, Cin, Cout, h, w, x_packed=xp)
        assert torch.equal(gw3, gw4)
    _lib.spike_conv_wgrad(g, x, gw2, NB, Cin, Cout, h, w, accumulate=True)
    assert torch.equal(gw2, gw + gw)


# ======================================================================================================
# exact MFMA FORWARD of the stride-2 5x5 encoder convs on spike inputs (ss_spike_conv_fwd_f32)
# ======================================================================================================
@pytest.mark.parametrize('NB,Cin,Cout,hw', [(2, 32, 64, (64, 80)), (3, 64, 128, (33, 45)), (1, 32, 64, (7, 9)), (5, 64, 128, (130, 173)), (4, 32, 64, (260, 346)),
                                            (2, 32, 64, (50, 70)),
                                            (3, 128, 256, (65, 87)), (2, 256, 512, (33, 44)), (1, 128, 256, (9, 7))])     # the wide (sliced, A/B-only) shapes
def test_spike_conv_fwd_mfma(NB, Cin, Cout, hw):
    """== conv2d(x, w, stride 2, padding 2) evaluated in float64 within fp32 accumulation error of EXACT products (element-wise bound 2^-21 of
    the magnitude sum sum |x||w| over K = 800 / 1600 terms: the exact 3-way bf16 split of the weight times spike counts, fp32 accumulation in the MFMA); within 4x (worst element) / 2x (rms) of
    MIOpen's fp32 convolution's distance to float64; packed input == dense input bit for bit; odd sizes / ragged tiles / frame edges; deterministic."""
    import torch.nn.functional as F
    from stereospike_amd import _lib
    h, w = hw
    gen = torch.Generator(device=DEV).manual_seed(NB + h)
    x = ((torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.3).float() + (torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.1).float()
         + (torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.03).float())                      # values 0 .. 3
    wt = torch.randn(Cout, Cin, 5, 5, device=DEV, generator=gen) * 0.05
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    assert (_lib.spike_conv_fwd_supported(Cin, Cout, 5, 2, 2) != _lib.spike_conv_fwd_wide_supported(Cin, Cout, 5, 2, 2)) and not _lib.spike_conv_fwd_supported(Cin, Cout, 3, 1, 1)
    y = torch.full((NB, ho, wo, Cout), float('nan'), device=DEV)
    _lib.spike_conv_fwd(x, None, wt, y, NB, Cin, Cout, h, w)

    def conv(xx, ww):
        return F.conv2d(xx.permute(0, 3, 1, 2), ww, None, 2, 2).permute(0, 2, 3, 1)
    ref = conv(x.double(), wt.double())
    mag = conv(x.double(), wt.double().abs())
    err = (y.double() - ref).abs()
    assert bool(torch.isfinite(y).all()) and bool((err <= mag * 2.0 ** -21 + 1e-30).all()), float((err / (mag * 2.0 ** -21 + 1e-30)).max())
    mi = conv(x, wt)
    # one fp32 accumulator walks all 50 / 100 k-steps in order (MIOpen's implicit GEMM sums blocked partials): worst element 2.5x MIOpen's at K = 1600
    assert float(err.max()) <= 4.0 * float((mi.double() - ref).abs().max()) + 1e-7 * float(ref.abs().max())
    assert float(err.double().pow(2).mean().sqrt()) <= 2.0 * float((mi.double() - ref).pow(2).mean().sqrt()) + 1e-9      # rms: the same accuracy class
    y2 = torch.empty_like(y)
    _lib.spike_conv_fwd(x, None, wt, y2, NB, Cin, Cout, h, w)
    assert torch.equal(y, y2)
    if (NB * h * w * Cin) % 16 == 0:                         # the 2-bit packed spike input: same values, bit for bit
        from oracle import np_pack
        xp = torch.from_numpy(np_pack.pack(x.cpu().numpy().reshape(-1)).view(np.int32)).to(DEV)
        y3 = torch.full_like(y, float('nan'))
        _lib.spike_conv_fwd(None, xp, wt, y3, NB, Cin, Cout, h, w)
        assert torch.equal(y, y3)


@pytest.mark.parametrize('NB,Cin,hw', [(2, 32, (64, 80)), (3, 32, (33, 45)), (1, 64, (65, 87)), (2, 64, (20, 22)), (5, 128, (17, 21)), (3, 256, (9, 10)),
                                       (2, 256, (33, 44)), (1, 32, (5, 3)), (4, 32, (260, 346)), (1, 128, (65, 87)), (7, 64, (1, 70))])
def test_conv_s2_dgrad_mfma(NB, Cin, hw):
    """ss_conv_s2_dgrad_f32 == the data gradient of conv2d(x, w, stride 2, padding 2) (conv1 .. conv4 of the encoder) evaluated in float64, within fp32
    accumulation error of fp32-accurate products: element-wise |g_x - float64| <= 2^-21 sum |g||w| (six bf16 cross terms per product; a tap's MFMAs
    run on a scratch accumulator, the running sum of up to 9 C_out = 4608 terms takes one fp32 addition per tap: measured <= 0.3 x 2^-21), rms within
    2x of MIOpen's fp32 data gradient's own distance to float64; gradients spanning e^{+-4} between pixels; odd / even sizes, maps smaller than a
    tile, rows crossing frame boundaries (the padded row space); every element written; deterministic."""
    import torch.nn.functional as F
    from stereospike_amd import _lib
    h, w = hw
    Cout = 2 * Cin
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    gen = torch.Generator(device=DEV).manual_seed(NB + h + Cin)
    g = torch.randn(NB, ho, wo, Cout, device=DEV, generator=gen) * torch.exp(2 * torch.randn(NB, ho, wo, 1, device=DEV, generator=gen)) * 1e-4
    wt = torch.randn(Cout, Cin, 5, 5, device=DEV, generator=gen) * 0.05
    assert _lib.conv_s2_dgrad_supported(Cin, Cout, 5, 2, 2) and not _lib.conv_s2_dgrad_supported(Cin, Cout, 3, 1, 1) and not _lib.conv_s2_dgrad_supported(Cin, Cin, 5, 2, 2)
    gx = torch.full((NB, h, w, Cin), float('nan'), device=DEV)
    _lib.conv_s2_dgrad(g, wt, gx, NB, Cin, Cout, h, w)

    def dgrad(gg, ww):
        return F.conv_transpose2d(gg.permute(0, 3, 1, 2), ww, None, 2, 2, output_padding=((h - 1) % 2, (w - 1) % 2)).permute(0, 2, 3, 1)
    ref = dgrad(g.double(), wt.double())
    assert ref.shape == gx.shape
    mag = dgrad(g.double().abs(), wt.double().abs())
    err = (gx.double() - ref).abs()
    assert bool(torch.isfinite(gx).all()), 'an element of g_x was not written'
    assert bool((err <= mag * 2.0 ** -21 + 1e-300).all()), float((err / (mag * 2.0 ** -21 + 1e-300)).max())
    x_meta = torch.empty((NB, Cin, h, w), dtype=torch.float32, device=DEV, memory_format=torch.channels_last)
    mi = torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), x_meta, wt.contiguous(memory_format=torch.channels_last), None,
                                             [2, 2], [2, 2], [1, 1], False, [0, 0], 1, [True, False, False])[0].permute(0, 2, 3, 1)
    rms = lambda e: float(e.double().pow(2).mean().sqrt())
    # (+ a floor of 2^-23 of the rms magnitude sum: should MIOpen pick a solver that accumulates in higher precision on some box, this kernel is still
    # held to fp32-accumulation accuracy, not to that solver's)
    assert rms(err) <= 2.0 * rms(mi.double() - ref) + 2.0 ** -23 * rms(mag), (rms(err), rms(mi.double() - ref), rms(mag))
    gx2 = torch.full_like(gx, float('nan'))
    _lib.conv_s2_dgrad(g, wt, gx2, NB, Cin, Cout, h, w)
    assert torch.equal(gx, gx2)
    if NB * h * w * Cin <= 200_000:                          # the oracle's parity-class restatement (oracle/np_conv_dgrad.py, float64): the same bound
        from oracle import np_conv_dgrad
        orc = torch.from_numpy(np_conv_dgrad.conv_s2_dgrad(g.cpu().numpy(), wt.cpu().numpy(), h, w)).to(DEV)
        assert float((orc - ref).abs().max()) <= 1e-12 * float(mag.max()) + 1e-30
        assert bool(((gx.double() - orc).abs() <= mag * 2.0 ** -21 + 1e-300).all())
    # linearity in g (size-independent property): dgrad(2 g) == 2 dgrad(g) bit for bit (powers of two commute with the bf16 split)
    _lib.conv_s2_dgrad(g * 2, wt, gx2, NB, Cin, Cout, h, w)
    assert torch.equal(gx2, gx * 2)


@pytest.mark.parametrize('NB,Cin,hw', [(2, 4, (64, 80)), (3, 2, (33, 45)), (1, 4, (7, 9)), (4, 4, (260, 346)), (2, 2, (130, 173)), (5, 4, (50, 70)), (1, 2, (1, 1))])
def test_dense_conv_s1_wgrad_mfma(NB, Cin, hw):
    """The first encoder layer's WEIGHT gradient (Conv2d(4 | 2, 32, 5, stride 1, pad 2)) as the six-term bf16 MFMA contraction over the pixels: on
    integer event counts and on arbitrary fp32 inputs |g_w - float64| <= 2^-20 sum |g||x| element-wise (six cross terms per product, one fp32 addition
    of the running sum per tile, fp64 second pass), rms within 2x of MIOpen's fp32 weight gradient's own distance to float64; odd sizes / ragged tiles
    / a single pixel; accumulate flag; bit-reproducible."""
    import torch.nn.functional as F
    from stereospike_amd import _lib
    h, w = hw
    gen = torch.Generator(device=DEV).manual_seed(NB + h + Cin)
    assert _lib.dense_conv_s1_wgrad_supported(Cin, 32, 5, 1, 2) and not _lib.dense_conv_s1_wgrad_supported(8, 32, 5, 1, 2)
    g = torch.randn(NB, h, w, 32, device=DEV, generator=gen) * torch.exp(2 * torch.randn(NB, h, w, 1, device=DEV, generator=gen)) * 1e-4
    wt = torch.randn(32, Cin, 5, 5, device=DEV, generator=gen) * 0.1
    for kind in ('counts', 'real'):
        x = torch.poisson(torch.full((NB, h, w, Cin), 0.3, device=DEV), generator=gen) if kind == 'counts' else \
            torch.randn(NB, h, w, Cin, device=DEV, generator=gen) * torch.exp(3 * torch.randn(NB, h, w, 1, device=DEV, generator=gen))
        gw = torch.full((32, Cin, 5, 5), float('nan'), device=DEV)
        _lib.dense_conv_s1_wgrad(g, x, gw, NB, Cin, 32, h, w)

        def wgrad64(gg, xx):                                  # float64 on the GPU: g^T @ unfold(x), summed over the frames
            cols = F.unfold(xx.permute(0, 3, 1, 2), 5, padding=2)                          # [NB, Cin * 25, h * w], row index = (ci, ky, kx)
            return torch.bmm(gg.reshape(NB, h * w, 32).transpose(1, 2), cols.transpose(1, 2)).sum(0).view(32, Cin, 5, 5)

        def wgrad(gg, xx):
            return torch.ops.aten.convolution_backward(gg.permute(0, 3, 1, 2), xx.permute(0, 3, 1, 2), wt, None,
                                                       [1, 1], [2, 2], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        ref = wgrad64(g.double(), x.double())
        mag = wgrad64(g.double().abs(), x.double().abs())
        err = (gw.double() - ref).abs()
        assert bool(torch.isfinite(gw).all()) and bool((err <= mag * 2.0 ** -20 + 1e-300).all()), (kind, float((err / (mag * 2.0 ** -20 + 1e-300)).max()))
        mi = wgrad(g, x)
        rms = lambda e: float(e.double().pow(2).mean().sqrt())
        assert rms(err) <= 2.0 * rms(mi.double() - ref) + 2.0 ** -23 * rms(mag), (kind, rms(err), rms(mi.double() - ref), rms(mag))
        gw2 = gw.clone()
        _lib.dense_conv_s1_wgrad(g, x, gw2, NB, Cin, 32, h, w, accumulate=True)
        assert float((gw2.double() - 2 * ref).abs().max()) <= 2.0 ** -19 * float(mag.max()) + 1e-30
        gw3 = torch.empty_like(gw)
        _lib.dense_conv_s1_wgrad(g, x, gw3, NB, Cin, 32, h, w)
        assert torch.equal(gw, gw3)


@pytest.mark.parametrize('rows,C', [(64 * 80 * 3, 32), (1, 32), (33, 64), (260 * 346 * 2, 32), (12345, 64), (31, 32)])
def test_head_on_packed_spikes(rows, C):
    """ss_head_proj_packed_f32 / ss_head_wgrad_packed_f32 (the full-resolution prediction head reading 2-bit packed spikes) against float64 on the
    unpacked codes: projection within 2^-22 of the magnitude sum (exact products — the weight splits exactly into three bf16 terms — and one fp32
    accumulation over C terms), weight gradient within 2^-18 of it (fma per row, per-wavefront partials, fixed-order fp64 second pass) and within
    4x of the fp32 library GEMM's own distance to float64 (rms); ragged row counts; accumulate flag; bit-reproducible."""
    from stereospike_amd import _lib
    from oracle import np_pack
    assert _lib.head_packed_supported(C, 1, 3) and not _lib.head_packed_supported(C, 1, 5) and not _lib.head_packed_supported(128, 1, 3)
    rng = np.random.default_rng(rows + C)
    n = rows * C
    codes = rng.choice(4, size=n, p=[0.62, 0.3, 0.07, 0.01]).astype(np.float32)
    xp = torch.from_numpy(np_pack.pack(codes).view(np.int32)).to(DEV)          # C % 16 == 0: whole words
    x = torch.from_numpy(codes).to(DEV).view(rows, C)
    gen = torch.Generator(device=DEV).manual_seed(rows)
    Wt = torch.randn(C, 9, device=DEV, generator=gen) * 0.1
    P = torch.full((rows, 9), float('nan'), device=DEV)
    _lib.head_proj_packed(xp, Wt, P, rows, C)
    ref = x.double() @ Wt.double()
    mag = x.double() @ Wt.double().abs()
    assert bool(torch.isfinite(P).all())
    assert bool(((P.double() - ref).abs() <= mag * 2.0 ** -22 + 1e-300).all()), float(((P.double() - ref).abs() / (mag * 2.0 ** -22 + 1e-300)).max())
    P2 = torch.empty_like(P)
    _lib.head_proj_packed(xp, Wt, P2, rows, C)
    assert torch.equal(P, P2)
    g = torch.randn(rows, 9, device=DEV, generator=gen) * torch.exp(2 * torch.randn(rows, 1, device=DEV, generator=gen)) * 1e-5
    gW = torch.full((C, 9), float('nan'), device=DEV)
    _lib.head_wgrad_packed(xp, g, gW, rows, C)
    refw = x.double().t() @ g.double()
    magw = x.double().t() @ g.double().abs()
    errw = (gW.double() - refw).abs()
    assert bool(torch.isfinite(gW).all()) and bool((errw <= magw * 2.0 ** -18 + 1e-300).all()), float((errw / (magw * 2.0 ** -18 + 1e-300)).max())
    lib32 = x.t() @ g
    rms = lambda e: float(e.double().pow(2).mean().sqrt())
    assert rms(errw) <= 4.0 * rms(lib32.double() - refw) + 2.0 ** -23 * rms(magw), (rms(errw), rms(lib32.double() - refw))
    gW2 = gW.clone()
    _lib.head_wgrad_packed(xp, g, gW2, rows, C, accumulate=True)
    assert float((gW2.double() - 2 * refw).abs().max()) <= 2.0 ** -17 * float(magw.max()) + 1e-30
    gW3 = torch.empty_like(gW)
    _lib.head_wgrad_packed(xp, g, gW3, rows, C)
    assert torch.equal(gW, gW3)


def test_packed_head_matches_the_dense_head(monkeypatch):
    """StereoSpike with the full-resolution head on packed spikes (fused.PACKED_HEAD, default) against the same step with the dense head: identical
    spikes and depth maps up to the two GEMMs' fp32 summation order (depths <= 1e-6 relative), every gradient tensor <= 1e-5 relative L2 (the one-element biases 1e-4), the returned
    last-step `out_add1` tensor identical; and the stage's dense output is really gone (peak memory)."""
    from stereospike_amd import fused
    from stereospike_amd.clock_driven import functional, surrogate
    from stereospike_amd.network.SNN_models import StereoSpike
    from stereospike_amd.network.loss import Total_Loss
    H, W, T, B = 64, 80, 5, 2
    res = []
    from stereospike_amd.config import EngineConfig
    for on in (True, False):
        torch.manual_seed(5)
        net = StereoSpike(surrogate_function=surrogate.ATan(), multiply_factor=10., input_size=(H, W), config=EngineConfig.default().replace(PACKED_HEAD=on)).to(DEV)
        gen = torch.Generator().manual_seed(6)
        x = torch.poisson(torch.full((B, T, 4, H, W), 0.08), generator=gen).to(DEV)
        gt = (0.5 + 9.5 * torch.rand(B, 1, H, W, generator=gen)).to(DEV)
        functional.reset_net(net)
        fused.TIMER.enabled = True
        fused.TIMER.clear()
        pred, spks = net.forward_sequence(x)
        loss = Total_Loss()(pred, gt, spks)
        loss.backward()
        torch.cuda.synchronize()
        tags = {k: v['launches'] for k, v in fused.TIMER.summary().items()}
        fused.TIMER.enabled = False
        assert tags.get('neuron_fwd_train+skip+packed', 0) == (3 if on else 1), tags         # + deconv1 and deconv2 (fused.PACKED_DECONV2) when on
        res.append(([p.detach().clone() for p in pred], [s.detach().clone() for s in spks], {k: p.grad.clone() for k, p in net.named_parameters()}))
    (p1, s1, g1), (p0, s0, g0) = res
    for a, b in zip(s1, s0):
        assert a.shape == b.shape and torch.equal(a, b)
    for a, b in zip(p1, p0):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
    for k in g0:                                             # (one-element tensors — the heads' biases — are cancelling sums over every pixel: 1e-4)
        assert float((g1[k] - g0[k]).norm()) <= (1e-4 if g0[k].numel() == 1 else 1e-5) * float(g0[k].norm()) + 1e-12, k


def test_spike_conv_stage_packed_only_input_matches_dense(monkeypatch):
    """SpikingStage.forward_sequence_conv_cl on a packed-only input (a data-less anchor + the packed tensor: what bottom / conv1 hand on in the
    default configuration) == on the dense tensor: output spikes, weight gradient and input gradient."""
    from stereospike_amd import fused
    from stereospike_amd.clock_driven import functional, neuron, surrogate
    from stereospike_amd.network.blocks import MultiplyBy, SpikingStage
    from oracle import np_pack
    T, B, h, w, Cin, Cout = 5, 2, 64, 80, 32, 64
    torch.manual_seed(3)
    st = SpikingStage(torch.nn.Conv2d(Cin, Cout, 5, 2, 2, bias=False), MultiplyBy(10.), neuron.IFNode(surrogate_function=surrogate.ATan(), detach_reset=True)).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(4)
    x = (torch.rand(T, B, h, w, Cin, device=DEV, generator=gen) < 0.2).float()
    xp = torch.from_numpy(np_pack.pack(x.cpu().numpy().reshape(T, -1)).view(np.int32)).to(DEV)
    g = torch.randn(T, B, h // 2, w // 2, Cout, device=DEV, generator=gen)
    outs = []
    for packed in (False, True):
        st.zero_grad()
        functional.reset_net(st)
        xin = (fused.spike_anchor(x.shape, x.dtype, x.device) if packed else x.clone()).requires_grad_()
        y = st.forward_sequence_conv_cl(xin, spikes_in=True, x_packed=xp if packed else None)
        (y * g).sum().backward()
        outs.append((y.detach().clone(), st[0].weight.grad.clone(), xin.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and float(outs[0][0].mean()) > 0.01
    # weight gradient: the packed input runs the window / transposed-read kernel (round 5), the dense one the first form — exact products, another fp32 summation order
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 2e-6 * float(outs[0][1].abs().max())
    assert float((outs[0][2] - outs[1][2]).abs().max()) <= 1e-6 * float(outs[0][2].abs().max())       # MIOpen's data gradient (atomics: not bit-stable)


@pytest.mark.parametrize('NB,Cin,hw', [(2, 4, (64, 80)), (3, 2, (33, 45)), (1, 4, (7, 9)), (4, 4, (260, 346)), (2, 2, (260, 346)), (2, 4, (50, 70))])
def test_dense_conv_s1_fwd_mfma(NB, Cin, hw):
    """The first encoder layer's forward (Conv2d(4 | 2, 32, 5, stride 1, pad 2)) as the six-term bf16 MFMA implicit GEMM: on integer event counts
    (what the voxeliser produces) |y - float64| <= 2^-21 sum |x||w| element-wise, on ARBITRARY fp32 inputs <= 2^-20; within 2x of MIOpen's fp32
    convolution's distance to float64 (+ its own rms); odd sizes / ragged tiles; deterministic."""
    import torch.nn.functional as F
    from stereospike_amd import _lib
    h, w = hw
    gen = torch.Generator(device=DEV).manual_seed(NB + h + Cin)
    wt = torch.randn(32, Cin, 5, 5, device=DEV, generator=gen) * 0.1
    assert _lib.dense_conv_s1_fwd_supported(Cin, 32, 5, 1, 2) and not _lib.dense_conv_s1_fwd_supported(8, 32, 5, 1, 2)
    for kind in ('counts', 'real'):
        x = torch.poisson(torch.full((NB, h, w, Cin), 0.3, device=DEV), generator=gen) if kind == 'counts' else \
            torch.randn(NB, h, w, Cin, device=DEV, generator=gen) * torch.exp(torch.randn(NB, h, w, 1, device=DEV, generator=gen))
        y = torch.full((NB, h, w, 32), float('nan'), device=DEV)
        _lib.dense_conv_s1_fwd(x, wt, y, NB, Cin, 32, h, w)

        def conv(xx, ww):
            return F.conv2d(xx.permute(0, 3, 1, 2), ww, None, 1, 2).permute(0, 2, 3, 1)
        ref = conv(x.double(), wt.double())
        mag = conv(x.double().abs(), wt.double().abs())
        err = (y.double() - ref).abs()
        bound = mag * (2.0 ** -21 if kind == 'counts' else 2.0 ** -20) + 1e-30      # 'real': magnitudes spanning e^(+-3) inside one 5 x 5 window (measured 1.004 x 2^-21)
        assert bool(torch.isfinite(y).all()) and bool((err <= bound).all()), (kind, float((err / bound).max()))
        mi = conv(x, wt)
        assert float(err.max()) <= 2.0 * float((mi.double() - ref).abs().max()) + 1e-7 * float(ref.abs().max()), kind
        y2 = torch.empty_like(y)
        _lib.dense_conv_s1_fwd(x, wt, y2, NB, Cin, 32, h, w)
        assert torch.equal(y, y2)


# ======================================================================================================
# decoder stage forward in the sub-pixel (merged tap) form (ss_upconv_sub_prep_f32 / ss_upconv_sub_fwd_f32)
# ======================================================================================================
@pytest.mark.parametrize('Cin,Cout,hw,HW,NB', [(64, 32, (130, 173), (260, 346), 1), (128, 64, (65, 87), (130, 173), 2), (256, 128, (33, 44), (65, 87), 2),
                                                (64, 32, (17, 22), (33, 44), 3), (16, 32, (13, 9), (26, 18), 2), (32, 96, (5, 7), (10, 13), 2),
                                                (64, 32, (8, 8), (15, 17), 1), (64, 64, (32, 40), (64, 80), 2)])
def test_upconv_sub_forward(Cin, Cout, hw, HW, NB):
    """The merged weights are the oracle's fp32 tap sums, split exactly (hi + mid + lo == the sum, sign flipped on odd channel groups); the output is the
    reference formula evaluated in float64 to within 2^-21 sum |x||W| element-wise and within 2e-6 (max norm) of today's projected kernel; packed input ==
    dense input bit for bit; every element written; deterministic."""
    from oracle import np_pack, np_upconv_sub as ns
    from stereospike_amd import _lib, fused
    from stereospike_amd.network.blocks import NNConvUpsampling
    (h, w), (H, W) = hw, HW
    up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(DEV)
    tabs = up._tables(h, w, torch.device(DEV))
    st = fused.sub_tables(tabs, H, W)
    assert st is not None and _lib.upconv_sub_supported(Cin, Cout, 5) and not _lib.upconv_sub_supported(Cin, Cout, 3) and not _lib.upconv_sub_supported(24, Cout, 5)
    gen = torch.Generator(device=DEV).manual_seed(NB + h)
    x = ((torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.3).float() + (torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.1).float()
         + (torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.03).float())                      # values 0 .. 3
    wt = torch.randn(Cout, Cin, 5, 5, device=DEV, generator=gen) * 0.05
    wm = _lib.upconv_sub_prep(wt, st, Cin, Cout)
    # ---- the merged weights
    vc, hc = st['vcls'].view(-1, 8).cpu().numpy(), st['hcls'].view(-1, 8).cpu().numpy()
    Wm = ns.merged_weights(wt.cpu().numpy(), vc[:, 1:4], vc[:, 4:7], hc[:, 1:4], hc[:, 4:7], np.float32)          # [cv, ch, co, ci, r, c]
    terms = wm.view(st['NVC'], st['NHC'], Cout // 32, Cin // 16, 3, 3, 3, 64, 8).float().cpu().numpy().astype(np.float64)
    got = terms.sum(6)                                                                                          # [cv, ch, cot, g, r, c, lane, e]
    lane, e = np.arange(64)[:, None], np.arange(8)[None, :]
    for cot in range(Cout // 32):
        for g in range(Cin // 16):
            ref = Wm[:, :, 32 * cot + (lane & 31), 16 * g + 8 * (lane >> 5) + e]                                # [cv, ch, lane, e, r, c]
            assert np.array_equal(got[:, :, cot, g], (-1.0) ** g * ref.transpose(0, 1, 4, 5, 2, 3).astype(np.float64)), (cot, g)
    # ---- the forward
    y = torch.full((NB, H, W, Cout), float('nan'), device=DEV)
    _lib.upconv_sub_fwd(x, None, wm, st, y, NB, Cin, Cout, h, w)
    sy, sx = tabs[0].cpu().numpy(), tabs[3].cpu().numpy()
    xd, wd = x.cpu().numpy(), wt.cpu().numpy()
    ref = ns.forward_direct(xd, wd, sy, sx, H, W)
    mag = ns.magnitude(xd, wd, sy, sx, H, W)
    err = np.abs(y.double().cpu().numpy() - ref)
    assert bool(torch.isfinite(y).all()) and (err <= 2.0 ** -21 * mag + 1e-30).all(), float((err / (2.0 ** -21 * mag + 1e-30)).max())
    from stereospike_amd import config
    with torch.no_grad(), config.engine_config(SUB_FWD=False):                    # today's projected form (fused projection + gather, or GEMM + gather)
        up.up[1].weight.copy_(wt)
        assert up.up[1].bias is None
        y_old = up.forward_projected_cl(x, spikes_in=True)
    assert float((y - y_old).abs().max()) <= 2e-6 * float(y_old.abs().max())
    y2 = torch.empty_like(y)
    _lib.upconv_sub_fwd(x, None, wm, st, y2, NB, Cin, Cout, h, w)
    assert torch.equal(y, y2)
    if (NB * h * w * Cin) % 16 == 0:
        xp = torch.from_numpy(np_pack.pack(xd.reshape(-1)).view(np.int32)).to(DEV)
        y3 = torch.full_like(y, float('nan'))
        _lib.upconv_sub_fwd(None, xp, wm, st, y3, NB, Cin, Cout, h, w)
        assert torch.equal(y, y3)


@pytest.mark.parametrize('Cin,Cout,hw,HW,NB', [(64, 32, (130, 173), (260, 346), 1), (128, 64, (65, 87), (130, 173), 2), (256, 128, (33, 44), (65, 87), 2),
                                                (512, 256, (17, 22), (33, 44), 2), (64, 32, (32, 40), (64, 80), 3), (128, 64, (16, 20), (32, 40), 2),
                                                (64, 64, (13, 18), (25, 35), 3), (64, 32, (4, 5), (8, 10), 2), (64, 32, (9, 11), (17, 19), 1)])
def test_upconv_box_kernels(Cin, Cout, hw, HW, NB):
    """The decoder's backward on the box-sum image (ss_upconv_box.hip, round 4) against oracle/np_upconv_box.py (pinned on the CPU to ss_ref_upconv_cl_bwd_f32
    bit for bit and to torch's autograd through UpsamplingNearest2d -> Conv2d: tests/test_oracle.py):
      ss_upconv_boxsum_f32    the three bf16 planes BIT-EQUAL to the oracle's (same rectangles, same summation order, same round-to-nearest split);
      ss_upconv_box_dgrad_f32 |g_x - float64| <= 2^-20 sum |B| |W| element-wise and <= 1.5 x ss_gemm6_f32's own worst element on the same operands (six
                              cross terms), every element written, no coherent drift, bit-reproducible;
      ss_upconv_box_wgrad_f32 |g_w - float64| <= 2^-20 sum |x| |B| element-wise (exact products, fp32 accumulation) and <= 1.5 x ss_spike_wgrad_f32's, accumulate mode, 2-bit packed input == dense input bit
                              for bit, bit-reproducible —
    on the four decoder geometries of the 260x346 pyramid, the 64x80 pyramid of the parity tests, odd sizes with triple-replicated rows / columns, ragged
    tiles, a map smaller than a tile; a resize ratio the on-chip window does not hold is refused by *_supported (the caller then keeps the g_P forms)."""
    from oracle import np_upconv_box as nbx
    from stereospike_amd import _lib, fused
    from stereospike_amd.network.blocks import NNConvUpsampling
    (h, w), (H, W) = hw, HW
    up = NNConvUpsampling(Cin, Cout, 5, (H, W)).to(DEV)
    tables = up._tables(h, w, torch.device(DEV))
    bt = fused.box_tables(tables, H, W)
    vr, vmap = nbx.range_tables(tables[1].cpu().numpy(), tables[2].cpu().numpy(), H)
    hr, hmap = nbx.range_tables(tables[4].cpu().numpy(), tables[5].cpu().numpy(), W)
    assert np.array_equal(bt['vr'].cpu().numpy().reshape(-1, 2), vr) and np.array_equal(bt['hmap'].cpu().numpy().reshape(-1, 5), hmap)
    assert np.array_equal(bt['hr'].cpu().numpy().reshape(-1, 2), hr) and np.array_equal(bt['vmap'].cpu().numpy().reshape(-1, 5), vmap)
    # the host cuts the source rows into tiles of <= 4 rows whose vertical ranges fit the on-chip window (15 ids): every geometry's ROWS fit by construction;
    # a resize whose 32 source columns reach more than 76 horizontal ranges is refused (the caller then keeps the g_P forms)
    tr = bt['tile_rows'].cpu().numpy().reshape(-1, 4)
    assert tr[0, 0] == 0 and (tr[1:, 0] == tr[:-1, 0] + tr[:-1, 1]).all() and tr[-1, 0] + tr[-1, 1] == h and tr[:, 1].max() <= 4 and tr[:, 3].max() <= 15
    if not _lib.upconv_box_dgrad_supported(Cin, Cout, 5, bt):
        assert bt['max_cols32'] > 76, bt
        return
    assert _lib.upconv_box_wgrad_supported(Cin, Cout, 5, bt) and not _lib.upconv_box_dgrad_supported(Cin, Cout, 3, bt)
    gen = torch.Generator(device=DEV).manual_seed(23 + Cin + h)
    g = torch.randn(NB, H, W, Cout, device=DEV, generator=gen) * torch.exp(2.0 * torch.randn(NB, H, W, 1, device=DEV, generator=gen))
    g[:, : H // 5] = 0
    weight = up.up[1].weight.detach().contiguous()
    # ---- K1
    box = _lib.upconv_boxsum(g, bt, NB, Cout, H, W)
    B = nbx.boxsum(g.cpu().numpy(), vr, hr)
    want = nbx.box_planes(B, 8)
    got = box.cpu().numpy().view(np.uint16)
    assert got.shape == want.shape and np.array_equal(got, want), int((got != want).sum())
    # ---- K2
    B64 = torch.tensor(B, dtype=torch.float64, device=DEV)
    vm, hm = torch.tensor(vmap, device=DEV, dtype=torch.long), torch.tensor(hmap, device=DEV, dtype=torch.long)
    w64 = weight.double()
    ref = torch.zeros(NB, h, w, Cin, dtype=torch.float64, device=DEV)
    mag = torch.zeros_like(ref)
    for ky in range(5):
        for kx in range(5):
            gp = B64[:, vm[:, ky]][:, :, hm[:, kx]]
            ref += gp @ w64[:, :, ky, kx]
            mag += gp.abs() @ w64[:, :, ky, kx].abs()
    g_x = torch.full((NB, h, w, Cin), float('nan'), device=DEV)
    _lib.upconv_box_dgrad(box, weight, bt, g_x, NB, Cin, Cout, h, w)
    err = (g_x.double() - ref).abs()
    bound = mag * 2.0 ** -21 + 1e-30
    # the gradients here span e^(+-4) between neighbouring output pixels, i.e. WITHIN a box sum's row of taps: on such data the six-term kernels reach
    # ~1.0 x 2^-21 (ss_gemm6_f32 on the same operands); asserted at 2^-20 and against ss_gemm6_f32's own worst element
    assert bool(torch.isfinite(g_x).all()) and bool((err <= 2.0 * bound).all()), float((err / bound).max())
    if _lib.gemm6_supported(25 * Cout, Cin):
        gPm = B64[:, vm][:, :, :, hm].permute(0, 1, 3, 2, 4, 5).reshape(NB * h * w, 25 * Cout).float()      # g_P [rows][(ky, kx, co)]
        W2 = weight.permute(2, 3, 0, 1).reshape(25 * Cout, Cin).contiguous()
        c6 = torch.empty(NB * h * w, Cin, device=DEV)
        _lib.gemm6(gPm.contiguous(), W2, c6, NB * h * w, 25 * Cout, Cin)
        assert float((err / bound).max()) <= 1.5 * float(((c6.view_as(ref).double() - ref).abs() / bound).max()) + 0.1
    if g_x.numel() >= 50000:
        assert abs(float(((g_x.double() - ref) / bound).mean())) * 2.0 ** -21 <= 1e-9          # no coherent drift (alternating accumulator sign)
    g_x2 = torch.empty_like(g_x)
    _lib.upconv_box_dgrad(box, weight, bt, g_x2, NB, Cin, Cout, h, w)
    assert torch.equal(g_x, g_x2)
    # ---- K3
    x = ((torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.3).float() + (torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.1).float()
         + (torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.02).float())
    x64 = x.double().reshape(-1, Cin)
    refw = torch.zeros(Cout, Cin, 5, 5, dtype=torch.float64, device=DEV)
    magw = torch.zeros_like(refw)
    for ky in range(5):
        for kx in range(5):
            gp = B64[:, vm[:, ky]][:, :, hm[:, kx]].reshape(-1, Cout)
            refw[:, :, ky, kx] = gp.t() @ x64
            magw[:, :, ky, kx] = gp.abs().t() @ x64
    g_w = torch.full((Cout, Cin, 5, 5), float('nan'), device=DEV)
    _lib.upconv_box_wgrad(box, x, None, bt, g_w, NB, Cin, Cout, h, w)
    errw = (g_w.double() - refw).abs()
    boundw = magw * 2.0 ** -22 + 1e-30
    # every product is exact; what is left is the fp32 accumulation over the source pixels (22 000 - 60 000 per element here, 1.8 M at config 3, in <= 128
    # fixed-order slices): measured <= 1.8 x 2^-22 of the magnitude sum, asserted at 2^-20 and against ss_spike_wgrad_f32 (the same exact products in
    # another order) on the materialised g_P
    assert bool(torch.isfinite(g_w).all()) and bool((errw <= 4.0 * boundw).all()), float((errw / boundw).max())
    if _lib.spike_wgrad_supported(Cin, 25 * Cout):
        gPm = B64[:, vm][:, :, :, hm].permute(0, 1, 3, 2, 4, 5).reshape(NB * h * w, 25 * Cout).float().contiguous()
        gw_sp = torch.empty(Cin, 25 * Cout, device=DEV)
        _lib.spike_wgrad(gPm, x.view(NB * h * w, Cin), gw_sp, NB * h * w, Cin, 25 * Cout)
        e_sp = (gw_sp.view(Cin, 5, 5, Cout).permute(3, 0, 1, 2).double() - refw).abs()
        assert float((errw / boundw).max()) <= 1.5 * float((e_sp / boundw).max()) + 0.5, (float((errw / boundw).max()), float((e_sp / boundw).max()))
    g_w2 = g_w.clone()
    _lib.upconv_box_wgrad(box, x, None, bt, g_w2, NB, Cin, Cout, h, w, accumulate=True)
    assert torch.equal(g_w2, g_w + g_w)
    if x.numel() % 16 == 0:
        xp = torch.empty(x.numel() // 16, dtype=torch.int32, device=DEV)
        from oracle import np_pack
        xp.copy_(torch.from_numpy(np_pack.pack(x.cpu().numpy().reshape(1, -1)).view(np.int32).reshape(-1)))
        g_w3 = torch.empty_like(g_w)
        _lib.upconv_box_wgrad(box, None, xp, bt, g_w3, NB, Cin, Cout, h, w)
        assert torch.equal(g_w3, g_w)


def test_upconv_box_backward_with_bias_and_oversized_geometry():
    """(a) a k = 5 spike-input stage built with bias=True takes the exact-split forward and the box-sum backward — the bias gradient must come back
    (it used to be dropped: bias.grad stayed None); (b) the *_supported predicates know the launch entry points' own limits (<= 64 row tiles, <= 16 column tiles,
    32-bit pixel indices), so an oversized geometry is 'unsupported' (the caller falls back) instead of an SS_EINVAL raised inside backward."""
    from stereospike_amd import _lib, fused, config
    from stereospike_amd.network.blocks import NNConvUpsampling
    Cin, Cout, (h, w), (H, W), NB = 128, 64, (16, 20), (32, 40), 2
    torch.manual_seed(5)
    up = NNConvUpsampling(Cin, Cout, 5, (H, W), bias=True).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(7)
    x = (torch.rand(NB, h, w, Cin, device=DEV, generator=gen) < 0.3).float().requires_grad_()
    g = torch.randn(NB, H, W, Cout, device=DEV, generator=gen)
    plan = {}
    with config.recording(plan), config.layer('stage'):
        y = up.forward_projected_cl(x, spikes_in=True)
        y.backward(g)
    assert plan['stage']['synapse_bwd'].startswith('box:'), plan
    assert up.up[1].bias.grad is not None
    want = g.double().sum((0, 1, 2))
    assert float((up.up[1].bias.grad.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    # the reference's two-op form, fp32 autograd
    xr = x.detach().permute(0, 3, 1, 2).clone().requires_grad_()
    wr, br = up.up[1].weight.detach().clone().requires_grad_(), up.up[1].bias.detach().clone().requires_grad_()
    yr = torch.nn.functional.conv2d(torch.nn.functional.interpolate(xr, size=(H + 4, W + 4), mode='nearest'), wr, br)
    yr.backward(g.permute(0, 3, 1, 2))
    assert float((y.detach().permute(0, 3, 1, 2) - yr.detach()).abs().max()) <= 1e-4 * float(yr.detach().abs().max())
    assert float((up.up[1].weight.grad - wr.grad).norm() / wr.grad.norm()) <= 1e-5
    assert float((x.grad.permute(0, 3, 1, 2) - xr.grad).norm() / xr.grad.norm()) <= 1e-5
    # (b)
    bt = fused.box_tables(up._tables(h, w, torch.device(DEV)), H, W)
    assert _lib.upconv_box_dgrad_supported(Cin, Cout, 5, bt, NB, h, w) and _lib.upconv_box_wgrad_supported(Cin, Cout, 5, bt, NB, h, w)
    big = dict(bt, n_row_tiles=65)
    assert not _lib.upconv_box_dgrad_supported(Cin, Cout, 5, big, NB, h, w) and not _lib.upconv_box_wgrad_supported(Cin, Cout, 5, big, NB, h, w)
    assert not _lib.upconv_box_dgrad_supported(Cin, Cout, 5, bt, NB, h, 513) and not _lib.upconv_box_wgrad_supported(Cin, Cout, 5, bt, 2 ** 31, 1, 1)
