"""CPU suite, part 1: pin the oracle.

  * oracle/ss_neuron_ref.c (fused recurrence) == oracle/sj_clock_driven.py run through torch autograd, on every
    committed known-answer vector and on fresh seeded inputs: forward bit-exact; backward bit-exact (ATan) /
    <= 3e-7 relative (Sigmoid: libm expf vs torch's vectorised sigmoid);
  * oracle/ref_network.py == the committed fixtures that the reference's own network/*.py produced
    (tests/golden/make_golden.py) — bit-exact on the machine/torch build that generated them (recorded in `meta`),
    tolerance otherwise;
  * the I-neuron pool restatement, the loss / MDE restatement against the PURE-reference fixture.
"""
import json
import platform

import numpy as np
import pytest
import torch

from _util import bit_equal, c_oracle, kat_cases, load_npz, ref_network as rn, rel_err, sj


def test_c_oracle_matches_kat_fixtures():
    n = 0
    for i, cfg, c in kat_cases():
        kw = dict(kind=cfg['kind'], scale=cfg['scale'], tau=cfg['tau'], k=cfg['k'], v_th=cfg['v_th'], v_reset=cfg['v_reset'])
        skip = c['skip'].astype(np.float32) if 'skip' in c else None
        f = c_oracle.neuron_fwd(c['x'], v_init=c.get('v_init'), skip_seq=skip, count=True, **kw)
        assert np.array_equal(f['out'], c['out'].astype(np.float32)), (i, cfg)
        assert bit_equal(f['h'], c['h']) and bit_equal(f['v_last'], c['v_last']), (i, cfg)
        assert int(f['nnz'][1]) == int((c['out'] != 0).sum())
        b = c_oracle.neuron_bwd(c['g_out'], c['h'], v_init=c.get('v_init'), g_v_last=c['g_v_last'],
                                surrogate=cfg['surrogate'], alpha=cfg['alpha'], **kw)
        if cfg['surrogate'] == 'ATan':
            assert bit_equal(b['g_x'], c['g_x']), (i, cfg)
        else:
            assert rel_err(b['g_x'], c['g_x']) < 3e-7, (i, cfg)
        if 'g_v_init' in c:
            assert rel_err(b['g_v_init'], c['g_v_init']) < 3e-7
        if cfg['kind'] == 'PLIF':
            k = cfg['k']
            assert abs(b['g_k'] * k * (1 - k) - float(c['g_w'])) <= 1e-5 * abs(float(c['g_w'])) + 1e-4
        n += 1
    assert n == 60


def test_kat_fixtures_contain_threshold_edge_cases():
    hit = 0
    for i, cfg, c in kat_cases():
        if cfg['kind'] == 'IF' and 'v_init' not in c and cfg['v_reset'] == 0.0:
            h0 = c['h'][0, :96].reshape(32, 3)
            one = np.float32(1.0)
            # exactly at threshold fires, +1ulp fires, -1ulp does not
            ok = (np.abs(h0[:, 0] - one) <= np.spacing(one))
            z = c['out'][0, :96].reshape(32, 3) - (c['skip'][0, :96].reshape(32, 3) if 'skip' in c else 0)
            at = h0 == one
            assert (z[at] == 1).all() and (z[h0 > one] == 1).all() and (z[h0 < one] == 0).all()
            hit += int(at.sum() > 0)
    assert hit > 0


@pytest.mark.parametrize('kind', ['IF', 'LIF', 'PLIF'])
@pytest.mark.parametrize('sg,alpha', [('ATan', 2.0), ('Sigmoid', 4.0), ('Sigmoid', 1.0)])
def test_c_oracle_vs_eager_autograd(kind, sg, alpha):
    from golden.make_golden import eager_neuron
    import zlib
    rng = np.random.default_rng(zlib.crc32(f'{kind}-{sg}-{alpha}'.encode()))   # (hash() of a str is salted per process: a different data set every run)
    for T, N, scale, v_reset in ((1, 257, 1.0, 0.0), (4, 1000, 10.0, 0.1), (7, 333, 3.0, 0.0)):
        x = (rng.standard_normal((T, N)) * 1.5 / scale).astype(np.float32)
        g = rng.standard_normal((T, N)).astype(np.float32)
        gv = rng.standard_normal(N).astype(np.float32)
        vi = (rng.standard_normal(N) * 0.5).astype(np.float32)
        skip = rng.integers(0, 3, (T, N)).astype(np.float32)
        e = eager_neuron(kind, x, scale, 2.5, 1.0, v_reset, sg, alpha, g, gv, skip, vi)
        kw = dict(kind=kind, scale=scale, tau=2.5, k=e['k'], v_th=1.0, v_reset=v_reset, v_init=vi)
        f = c_oracle.neuron_fwd(x, skip_seq=skip, **kw)
        assert np.array_equal(f['out'], e['out']) and bit_equal(f['h'], e['h']) and bit_equal(f['v_last'], e['v_last'])
        b = c_oracle.neuron_bwd(g, f['h'], g_v_last=gv, surrogate=sg, alpha=alpha, **kw)
        if sg == 'ATan':
            assert bit_equal(b['g_x'], e['g_x']) and bit_equal(b['g_v_init'], e['g_v_init'])
        else:
            # C expf vs torch.sigmoid: 1e-6 (the bar stated for the Sigmoid surrogate everywhere else; 3e-7 held for ~99.8 % of random data sets only)
            assert rel_err(b['g_x'], e['g_x']) < 1e-6 and rel_err(b['g_v_init'], e['g_v_init']) < 1e-6
        if kind == 'PLIF':
            k = e['k']
            assert abs(b['g_k'] * k * (1 - k) - e['g_w']) <= 1e-5 * abs(e['g_w']) + 1e-4


def test_non_detached_reset_matches_autograd():
    """detach_reset=False (library default; the reference's I-neuron pool) — the adjoint keeps the reset path."""
    rng = np.random.default_rng(3)
    T, N = 4, 500
    x = rng.standard_normal((T, N)).astype(np.float32)
    g = rng.standard_normal((T, N)).astype(np.float32)
    xt = torch.tensor(x, requires_grad=True)
    node = sj.IFNode(1.0, 0.1, sj.ATan(), False)
    outs = [node(xt[t]) for t in range(T)]
    (torch.stack(outs) * torch.tensor(g)).sum().backward()
    f = c_oracle.neuron_fwd(x, kind='IF', v_th=1.0, v_reset=0.1)
    b = c_oracle.neuron_bwd(g, f['h'], kind='IF', v_th=1.0, v_reset=0.1, surrogate='ATan', alpha=2.0, detach_reset=False)
    assert rel_err(b['g_x'], xt.grad.numpy()) < 1e-6


def test_ipool_oracle_vs_eager_ifnode():
    """IFNode(v_threshold=inf) charged by K heads per step == ss_ref_ipool (SNN_models.py:150,172-188)."""
    rng = np.random.default_rng(4)
    T, K, M = 3, 4, 1000
    pd = rng.standard_normal((T, K, M)).astype(np.float32)
    g = rng.standard_normal((T, K, M)).astype(np.float32)
    pdt = torch.tensor(pd, requires_grad=True)
    pool = sj.IFNode(float('inf'), 0.0, sj.ATan())
    snaps = []
    for t in range(T):
        for k in range(K):
            pool(torch.mul(pdt[t, k], 10.0))
            snaps.append(pool.v)
    depth = torch.stack(snaps).view(T, K, M)
    (depth * torch.tensor(g)).sum().backward()
    ref = c_oracle.ipool_fwd(pd, scale=10.0, v_reset=0.0)
    assert bit_equal(ref, depth.detach().numpy())
    rb = c_oracle.ipool_bwd(g, scale=10.0)
    assert bit_equal(rb['g_pd'], pdt.grad.numpy())


def _same_machine(meta):
    return meta['torch'] == torch.__version__ and meta['threads'] == torch.get_num_threads() and \
        meta['cpu'] == (platform.processor() or platform.machine())


@pytest.mark.parametrize('tag,name,kw,T', [
    ('stereospike_T1', 'StereoSpike', dict(multiply_factor=10.), 1),
    ('plif_T5', 'PLIFNet', dict(tau=3., use_plif=True, multiply_factor=30.), 5),
    ('mono_plif_T1', 'PLIFNetMono', dict(tau=3., use_plif=True, multiply_factor=30.), 1),
    ('ann_T1', 'ANN', dict(), 1),
])
def test_oracle_network_vs_reference_fixture(tag, name, kw, T):
    """The fixture holds what /root/reference/network/*.py itself computed (spikingjelly stand-in = sj_clock_driven)."""
    z = load_npz(f'model_{tag}.npz')
    meta = json.loads(str(z['meta']))
    torch.manual_seed(int(z['seed']))
    if name == 'StereoSpike':
        kw = dict(kw, surrogate_function=sj.ATan())
    net = rn.build(name, sigmoid_alpha=meta['sigmoid_alpha'], **kw) if name != 'ANN' else rn.build(name)
    import hashlib
    h = hashlib.sha256()
    for k, v in net.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    assert h.hexdigest() == str(z['state_sha']), 'default init under the recorded seed must reproduce the weights'
    x = torch.tensor(z['x'].astype(np.float32))
    gt = torch.tensor(z['gt'])
    res = rn.run_sequence(net, x)
    depths, spikes = res if isinstance(res, tuple) else (res, [])
    loss = rn.total_loss(depths, gt, spikes)
    mde = rn.mean_depth_error(depths[0].detach(), gt)
    exact = _same_machine(meta)
    scale = float(np.abs(z['depth1']).max())
    for i, d in enumerate(depths):
        if exact:
            assert bit_equal(d, z[f'depth{i + 1}']), f'depth{i + 1}'
        else:
            assert float(np.abs(d.detach().numpy() - z[f'depth{i + 1}']).max()) <= 2e-2 * scale
    for nm, s in zip(('out_rconv', 'out_add4', 'out_add3', 'out_add2', 'out_add1'), spikes):
        mism = float((s.detach().numpy() != z[nm].astype(np.float32)).mean())
        assert mism == 0.0 if exact else mism <= 2e-3, (nm, mism)
    assert abs(float(loss) - float(z['loss'])) <= (0 if exact else 2e-3 * abs(float(z['loss'])))
    assert abs(float(mde) - float(z['mde'])) <= (0 if exact else 2e-3 * abs(float(z['mde'])))
    # membranes carried over the T steps (state really is stateful: SURVEY.md §3.4)
    import hashlib as _h
    vs = {k: m.v for k, m in net.named_modules() if isinstance(m, sj.BaseNode) and isinstance(m.v, torch.Tensor)}
    assert list(vs.keys()) == json.loads(str(z['v_names']))
    if exact:
        shas = [_h.sha256(v.detach().contiguous().numpy().tobytes()).hexdigest() for v in vs.values()]
        assert shas == json.loads(str(z['v_sha']))
    else:
        assert np.allclose([v.detach().double().sum().item() for v in vs.values()], z['v_sum'], rtol=5e-2)
    if tag == 'stereospike_T1':      # one backward is enough for the CPU budget
        loss.backward()
        l2 = np.array([p.grad.double().norm().item() for p in net.parameters()])
        # forward values are bit-stable on one machine; oneDNN's backward kernels reduce in an order that depends on the thread partition of
        # the moment (this test alone vs inside the suite differ by 4e-8 relative): fp32 reduction-order tolerance, not 1e-12
        assert np.allclose(l2, z['grad_l2'], rtol=1e-6 if exact else 2e-2, atol=0)


def test_loss_and_mde_restatement_vs_pure_reference_fixture():
    z = load_npz('loss_metric.npz')
    for ci in range(int(z['n_cases'])):
        preds = [torch.tensor(z[f'l{ci}_pred{i}'], requires_grad=True) for i in range(4)]
        gt = torch.tensor(z[f'l{ci}_gt'])
        spikes = [torch.tensor(z[f'l{ci}_spk{i}'].astype(np.float32)) for i in range(5)]
        for pen in (False, True):
            L = rn.total_loss(preds, gt, spikes, penalize_spikes=pen, beta=0.5)
            assert float(L) == float(z[f'l{ci}_{"pen" if pen else "nopen"}_loss'])
        assert float(rn.mean_depth_error(preds[0].detach(), gt)) == float(z[f'l{ci}_mde'])


def test_loss_statistics_restatement_vs_pure_reference_fixture():
    """oracle/np_loss.py (the checker of ss_loss_stats_f32 / ss_loss_grad_f32) against the reference's own loss values, MDE and
    d loss / d pred stored in tests/golden/loss_metric.npz (generated by the reference's network/loss.py, network/metrics.py)."""
    from oracle import np_loss
    z = load_npz('loss_metric.npz')
    for ci in range(int(z['n_cases'])):
        gt = z[f'l{ci}_gt']
        total, stats = 0.0, []
        for i in range(4):
            s = np_loss.loss_stats(z[f'l{ci}_pred{i}'], gt)
            stats.append(s)
            n = s[0]
            total += (s[2] / n - (s[1] / n) ** 2) + 0.5 * s[3] / n
        ref = float(z[f'l{ci}_nopen_loss'])
        assert abs(total - ref) <= 2e-6 * abs(ref), (ci, total, ref)
        assert abs(stats[0][4] / stats[0][0] - float(z[f'l{ci}_mde'])) <= 2e-6 * float(z[f'l{ci}_mde'])
        for i in range(4):
            want = z[f'l{ci}_nopen_gpred{i}']
            if want.ndim == 0:
                continue                                     # the full-size case stores no gradient maps
            got = np_loss.loss_grad(z[f'l{ci}_pred{i}'], gt, stats[i], (1.0, 0.5))
            assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max() + 1e-9, (ci, i)
            assert np.array_equal(got == 0, want == 0)       # zero exactly at the invalid pixels


@pytest.mark.parametrize('dtype', ['f16', 'bf16'])
@pytest.mark.parametrize('kind', ['IF', 'LIF', 'PLIF'])
def test_x16_numpy_oracle_equals_c_oracle_on_widened_inputs(dtype, kind):
    """oracle/np_x16.py (16-bit activation I/O, fp32 membrane) == oracle/ss_neuron_ref.c fed the widened inputs:
    h / v bit-exact, outputs = nearest-even narrowing, so the low-precision mode adds nothing but the I/O rounding."""
    from oracle import np_x16
    rng = np.random.default_rng(11)
    T, N = 5, 1000
    xb = np_x16.narrow((rng.standard_normal((T, N)) * 0.2).astype(np.float32), dtype)
    sb = np_x16.narrow(rng.integers(0, 3, (T, N)).astype(np.float32), dtype)
    v0 = (rng.standard_normal(N) * 0.5).astype(np.float32)
    kw = dict(kind=kind, scale=10.0, tau=3.0, k=np.float32(1 / 3.) if kind == 'PLIF' else None, v_th=1.0, v_reset=0.0)
    a = np_x16.neuron_fwd(xb, dtype, v_init=v0, skip_bits=sb, **kw)
    b = c_oracle.neuron_fwd(np_x16.widen(xb, dtype), v_init=v0, skip_seq=np_x16.widen(sb, dtype), **kw)
    assert bit_equal(a['h'], b['h']) and bit_equal(a['v_last'], b['v_last'])
    assert np.array_equal(np_x16.widen(a['out'], dtype), b['out'])
    gb = np_x16.narrow(rng.standard_normal((T, N)).astype(np.float32), dtype)
    ga = np_x16.neuron_bwd(gb, a['h'], dtype, v_init=v0, surrogate='ATan', alpha=2.0, **kw)
    gc = c_oracle.neuron_bwd(np_x16.widen(gb, dtype), b['h'], v_init=v0, surrogate='ATan', alpha=2.0, **kw)
    assert np.array_equal(ga['g_x'], np_x16.narrow(gc['g_x'], dtype))
    assert bit_equal(ga['g_v_init'], gc['g_v_init'])
    # round trip of the bit conversions
    f = rng.standard_normal(4096).astype(np.float32)
    assert np.array_equal(np_x16.narrow(np_x16.widen(np_x16.narrow(f, dtype), dtype), dtype), np_x16.narrow(f, dtype))
    t = torch.tensor(f)
    td = t.to(torch.float16 if dtype == 'f16' else torch.bfloat16)
    assert np.array_equal(np_x16.narrow(f, dtype), td.view(torch.int16).numpy().view(np.uint16))


def test_c_oracle_under_address_and_ub_sanitizers():
    """Every entry point of oracle/ss_neuron_ref.c on small ragged inputs under -fsanitize=address,undefined."""
    import os
    import subprocess
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
    subprocess.check_call(['make', '-s', '-C', root, '_build/ss_oracle_asan_check'])
    out = subprocess.run([os.path.join(root, '_build', 'ss_oracle_asan_check')], capture_output=True, text=True)
    assert out.returncode == 0 and 'rc=0' in out.stdout, out.stdout + out.stderr


def test_voxelizer_oracle_vs_reference_fixture():
    """oracle/np_voxelize.py == what the reference's own mvsecCumulateSpikesIntoFrames produced (tests/golden/voxelizer.npz),
    including events sitting exactly on frame boundaries and the -1 polarity encoding."""
    from oracle import np_voxelize as nv
    z = load_npz('voxelizer.npz')
    for ci in range(int(z['n_cases'])):
        n_chunks, nfpdm = (int(v) for v in z[f'v{ci}_cfg'])
        fr = nv.cumulate_spikes_into_frames(z[f'v{ci}_events'], n_chunks, nfpdm)
        assert np.array_equal(fr, z[f'v{ci}_frames'].astype(np.float64))
        assert fr.sum() > 0


def test_exact_split_operand_oracle_identities():
    """oracle/np_operands.py (checker of ss_im2col_cl_bf16 / ss_split3_bf16): the three bf16 terms sum back to every fp32 value exactly;
    im2col(x) @ (Wh + Wm + Wl) equals torch's conv2d on spike inputs to fp32 summation-order accuracy; each product is exact."""
    from oracle import np_operands as no
    rng = np.random.default_rng(0)
    g = (rng.standard_normal(20000) * np.logspace(-20, 20, 20000)).astype(np.float32)
    hi, mid, lo = no.split3(g)
    assert np.array_equal((hi.astype(np.float64) + mid + lo).astype(np.float32), g)
    for t in (hi, mid, lo):                                        # each term is a bf16 value: low 16 bits clear
        assert not (t.view(np.uint32) & 0xffff).any()
    for (Cin, Cout, k, s, p, hw) in ((16, 8, 3, 1, 1, (6, 7)), (8, 4, 5, 2, 2, (9, 11)), (8, 4, 3, 2, 0, (7, 9))):
        x = rng.integers(0, 3, (2, hw[0], hw[1], Cin)).astype(np.float32)
        W = (rng.standard_normal((Cout, Cin, k, k)) * 0.1).astype(np.float32)
        A, (ho, wo) = no.im2col_cl(x, k, s, p)
        assert np.array_equal(A, no.im2col_cl(x, k, s, p)[0]) and set(np.unique(A)) <= {0.0, 1.0, 2.0}
        Wt = W.transpose(2, 3, 1, 0).reshape(k * k * Cin, Cout)
        Wh, Wm, Wl = no.split3(Wt)
        y = (A.astype(np.float64) @ Wh + A.astype(np.float64) @ Wm + A.astype(np.float64) @ Wl).reshape(2, ho, wo, Cout)
        ref = torch.nn.functional.conv2d(torch.tensor(x).permute(0, 3, 1, 2).double(), torch.tensor(W).double(), None, s, p)
        assert np.abs(y - ref.permute(0, 2, 3, 1).numpy()).max() <= 1e-12      # exact products, float64 accumulation on both sides


def test_wgrad_reduce3_oracle_is_the_two_reductions_and_the_permute():
    """oracle/np_operands.wgrad_reduce3 (checker of ss_wgrad_reduce3_f32) == the torch expression it replaces in fused._SpikeConvCL.backward —
    parts.sum(0).view(K, 3, C_out).sum(1) permuted to the Conv2d layout — to fp32 summation-order accuracy, and exactly when every partial sum is
    representable."""
    from oracle import np_operands as no
    rng = np.random.default_rng(3)
    for (S, k, Cin, Cout) in ((4, 5, 8, 32), (1, 3, 16, 32), (3, 1, 8, 64)):
        parts = rng.standard_normal((S, k * k * Cin, 3, Cout)).astype(np.float32)
        ref = torch.tensor(parts).double().sum(0).sum(1).view(k, k, Cin, Cout).permute(3, 2, 0, 1).numpy()
        got = no.wgrad_reduce3(parts, k, Cin, Cout)
        assert got.shape == (Cout, Cin, k, k) and np.abs(got - ref).max() <= 2e-6 * np.abs(parts).sum((0, 2)).max()
        ints = rng.integers(-8, 9, parts.shape).astype(np.float32)                   # small integers: every order gives the same fp32 value
        assert np.array_equal(no.wgrad_reduce3(ints, k, Cin, Cout), torch.tensor(ints).sum(0).sum(1).view(k, k, Cin, Cout).permute(3, 2, 0, 1).numpy())


def test_winograd_dgrad_oracle_matches_autograd():
    """oracle/np_winograd.py: the float64 direct form == torch autograd's conv data gradient; the fp32 Winograd restatement is within fp32
    accumulation error of it (odd sizes, ragged last tile)."""
    import torch.nn.functional as F
    from oracle import np_winograd as nw
    rng = np.random.default_rng(5)
    for NB, H, W, Co, Ci in ((2, 17, 22, 8, 12), (1, 4, 4, 4, 4), (3, 5, 7, 8, 4), (1, 1, 1, 4, 4)):
        g = rng.standard_normal((NB, H, W, Co)).astype(np.float32)
        w = (rng.standard_normal((Co, Ci, 3, 3)) * 0.1).astype(np.float32)
        x = torch.zeros(NB, Ci, H, W, dtype=torch.float64, requires_grad=True)
        F.conv2d(x, torch.tensor(w, dtype=torch.float64), padding=1).backward(torch.tensor(g, dtype=torch.float64).permute(0, 3, 1, 2))
        ref = x.grad.permute(0, 2, 3, 1).numpy()
        assert np.abs(nw.dgrad_direct64(g, w) - ref).max() <= 1e-12
        assert np.abs(nw.dgrad(g, w) - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('NB,Cin,hw', [(2, 4, (9, 10)), (1, 3, (1, 7)), (2, 2, (6, 5)), (1, 8, (13, 12)), (3, 2, (4, 4))])
def test_conv_s2_dgrad_oracle_matches_autograd(NB, Cin, hw):
    """oracle/np_conv_dgrad.py — the parity-class form of the stride-2 5x5 data gradient that ss_conv_s2_dgrad_f32 evaluates — against torch's own
    input gradient of conv2d(x, w, stride 2, padding 2) (the reference's autograd, /root/reference/network/SNN_models.py:80-101) in float64, and
    against autograd through an actual forward; odd / even sizes, a one-row map."""
    from oracle import np_conv_dgrad
    h, w = hw
    Cout = 2 * Cin
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    rng = np.random.default_rng(NB + h + w)
    g = rng.standard_normal((NB, ho, wo, Cout))
    wt = rng.standard_normal((Cout, Cin, 5, 5))
    got = np_conv_dgrad.conv_s2_dgrad(g, wt, h, w)
    ref = torch.nn.grad.conv2d_input((NB, Cin, h, w), torch.from_numpy(wt), torch.from_numpy(g).permute(0, 3, 1, 2), stride=2, padding=2)
    assert np.abs(got - ref.permute(0, 2, 3, 1).numpy()).max() <= 1e-12
    x = torch.from_numpy(rng.standard_normal((NB, Cin, h, w))).requires_grad_()
    y = torch.nn.functional.conv2d(x, torch.from_numpy(wt), None, 2, 2)
    assert tuple(y.shape) == (NB, Cout, ho, wo)
    (gx,) = torch.autograd.grad(y, x, torch.from_numpy(g).permute(0, 3, 1, 2))
    assert np.abs(got - gx.permute(0, 2, 3, 1).numpy()).max() <= 1e-12
    assert sum(len(v) for v in np_conv_dgrad.CLASS_TAPS.values()) == 25 and sorted(len(v) for v in np_conv_dgrad.CLASS_TAPS.values()) == [4, 6, 6, 9]


@pytest.mark.parametrize('NB,Cin,Cout,hw,up', [(2, 8, 4, (17, 22), (33, 44)), (1, 8, 8, (13, 9), (26, 18)), (1, 4, 4, (33, 44), (65, 87)), (2, 4, 4, (5, 7), (10, 13)),
                                               (1, 4, 4, (8, 8), (15, 17))])
def test_upconv_sub_oracle_is_the_reference_formula_and_the_product_tables(NB, Cin, Cout, hw, up):
    """oracle/np_upconv_sub.py (checker of ss_upconv_sub_*): (1) forward_direct == torch's UpsamplingNearest2d -> Conv2d in float64 (the reference's two ops,
    /root/reference/network/blocks.py:110-132); (2) the merged-tap form with float64 weight sums is the SAME function (<= 1e-12: it only re-associates the
    weight additions); (3) with the weight sums rounded to fp32 (what the kernel's preparation does) it stays within 2^-24 sum |x||W| per element;
    (4) the product's host tables (fused.register_sub_tables) are the oracle's classes and blocks, every output row / column in exactly one block."""
    from oracle import np_upconv_sub as ns
    from stereospike_amd import fused
    (h, w), (H, W) = hw, up
    rng = np.random.default_rng(NB + h)
    tabs = fused.nearest_tables(h, H + 4) + fused.nearest_tables(w, W + 4)
    sy, sx = tabs[0].numpy(), tabs[3].numpy()
    x = rng.integers(0, 4, (NB, h, w, Cin)).astype(np.float32)
    Wt = (rng.standard_normal((Cout, Cin, 5, 5)) * 0.1).astype(np.float32)
    yd = ns.forward_direct(x, Wt, sy, sx, H, W)
    xt = torch.tensor(x).permute(0, 3, 1, 2).double()
    ref = torch.nn.functional.conv2d(torch.nn.UpsamplingNearest2d(size=(H + 4, W + 4))(xt), torch.tensor(Wt).double()).permute(0, 2, 3, 1).numpy()
    assert np.abs(yd - ref).max() <= 1e-12
    assert np.abs(ns.forward_merged(x, Wt, sy, sx, H, W, merge_dtype=np.float64) - yd).max() <= 1e-12
    mag = ns.magnitude(x, Wt, sy, sx, H, W)
    assert (np.abs(ns.forward_merged(x, Wt, sy, sx, H, W, merge_dtype=np.float32) - yd) <= 2.0 ** -24 * mag + 1e-30).all()
    st = fused.register_sub_tables(tabs, tabs, H, W)
    for src, n_out, cls_t, blk_t, rec, mo, ms in ((sy, H, st['vcls'], st['vblk'], 88, 16, 20), (sx, W, st['hcls'], st['hblk'], 168, 32, 36)):
        keys, cls, k0, kn = ns.axis_classes(src, n_out)
        ct = cls_t.view(-1, 8).numpy()
        assert np.array_equal(ct[:, 0], (kn > 0).sum(1)) and np.array_equal(ct[:, 1:4], k0) and np.array_equal(ct[:, 4:7], kn)
        blocks = ns.axis_blocks(src, n_out, cls, k0, kn, mo, ms)
        bt = blk_t.view(-1, rec).numpy()
        assert len(blocks) == bt.shape[0]
        seen = []
        for b, r in zip(blocks, bt):
            n, m = len(b['out']), len(b['src'])
            assert (r[0], r[1], r[2]) == (b['cls'], n, m) and np.array_equal(r[3:3 + n], b['out']) and np.array_equal(r[3 + mo:3 + mo + m], b['src'])
            assert np.array_equal(r[3 + mo + ms:3 + mo + ms + 3 * n].reshape(n, 3), b['slot'])
            assert r[-1] == (kn[b['cls']] > 0).sum()                         # the record's last word: runs of its class
            seen += list(b['out'])
        assert sorted(seen) == list(range(n_out))
    # the tall row blocks (<= 64 rows, paired with the column blocks of <= 8 columns) are the oracle's blocks at those limits
    NVB, NHB, NTB = st['NVB'], st['NHB'], st['NTB']
    vb, hb = st['vblk'].view(-1, 88).numpy(), st['hblk'].view(-1, 168).numpy()
    tb = st['tblk'].view(-1, 328).numpy() if NTB else np.zeros((0, 328), np.int32)
    keys, cls, k0, kn = ns.axis_classes(sy, H)
    tall = ns.axis_blocks(sy, H, cls, k0, kn, 64, 68)
    assert NTB in (0, len(tall))
    for b, r in zip(tall, tb):
        n, m = len(b['out']), len(b['src'])
        assert (r[0], r[1], r[2], r[-1]) == (b['cls'], n, m, (kn[b['cls']] > 0).sum()) and np.array_equal(r[3:3 + n], b['out']) and np.array_equal(r[67:67 + m], b['src'])
        assert np.array_equal(r[135:135 + 3 * n].reshape(n, 3), b['slot'])
    # the tiles of a frame: every output pixel in exactly one (row block, column block) pair; narrow column blocks meet tall row blocks only; cost order
    order = st['order'].numpy()
    assert len(order) == st['NORD'] == len(set(order.tolist()))
    cover = np.zeros((H, W), np.int32)
    costs = []
    for p in order:
        v = vb[p // NHB] if p < NVB * NHB else tb[p // NHB - NVB]
        hh = hb[p % NHB]
        assert (hh[1] <= 8) == (p >= NVB * NHB) or NTB == 0
        rows, cols = v[3:3 + v[1]], hh[3:3 + hh[1]]
        cover[np.ix_(rows, cols)] += 1
        assert v[2] * hh[2] <= 720                                                   # the tile's window fits the 720 pixels held on chip
        costs.append(v[-1] * hh[-1] * (4 if v[1] > 4 else 1))
    assert (cover == 1).all()
    assert costs == sorted(costs, reverse=True)


@pytest.mark.parametrize('NB,Cin,Cout,hw,up', [(2, 8, 8, (17, 22), (33, 44)), (1, 8, 16, (5, 7), (9, 13)), (2, 4, 8, (3, 3), (8, 5)), (1, 8, 8, (1, 6), (2, 11)),
                                               (1, 4, 8, (9, 11), (9, 11))])
def test_upconv_box_oracle_matches_the_adjoint_oracle_and_autograd(NB, Cin, Cout, hw, up):
    """oracle/np_upconv_box.py (the box-sum form of the decoder's backward the round-4 kernels evaluate) pinned on the CPU:
      (1) g_P gathered from the box-sum image B == ss_ref_upconv_cl_bwd_f32's g_P BIT FOR BIT (same rectangles, same summation order), on the pyramid's
          own geometries incl. triple-replicated rows / columns, resize factors != 2, a source map of one row, and the identity resize;
      (2) data and weight gradient through B == torch's autograd through the reference's two-op form nn.UpsamplingNearest2d -> nn.Conv2d
          (/root/reference/network/blocks.py:124-132) in float64;
      (3) the three bf16 planes sum back to B exactly, and the range tables are consistent (id 0 empty, maps inside the lists)."""
    from oracle import np_upconv_box as nb
    from stereospike_amd.fused import nearest_tables
    h, w = hw
    H, W = up
    k = 5
    rng = np.random.default_rng(h * 100 + W)
    ty, tx = nearest_tables(h, H + k - 1), nearest_tables(w, W + k - 1)
    y_lo, y_hi, x_lo, x_hi = (t.numpy() for t in (ty[1], ty[2], tx[1], tx[2]))
    g = (rng.standard_normal((NB, H, W, Cout)) * np.exp(rng.uniform(-3, 3, (NB, H, W, 1)))).astype(np.float32)
    vr, vmap = nb.range_tables(y_lo, y_hi, H, k)
    hr, hmap = nb.range_tables(x_lo, x_hi, W, k)
    assert tuple(vr[0]) == (0, 0) and tuple(hr[0]) == (0, 0) and vmap.max() < len(vr) and hmap.max() < len(hr)
    assert len(vr) <= H + 2 * k + 3 * h and (np.diff(vr[1:, 0]) >= 0).all()
    B = nb.boxsum(g, vr, hr)
    gP = nb.g_P_from_box(B, vmap, hmap)                                                       # [NB, h, w, 25, Cout]
    ref = c_oracle.upconv_cl_bwd(g, y_lo, y_hi, x_lo, x_hi, k).reshape(NB, h, w, k * k, Cout)
    assert bit_equal(gP, ref)
    hp, mp, lp = nb.split3_rn(B)
    assert np.array_equal((nb.bf16_to_f32(hp).astype(np.float64) + nb.bf16_to_f32(mp) + nb.bf16_to_f32(lp)).astype(np.float32), B)
    pl = nb.box_planes(B, 8)
    assert pl.shape == (NB, Cout // 8, 3, len(vr), len(hr), 8) and np.array_equal(pl[:, Cout // 8 - 1, 2, :, :, 3], lp[..., Cout - 8 + 3])
    # (2) autograd of the two-op form, float64
    x = rng.integers(0, 3, (NB, h, w, Cin)).astype(np.float64)
    wt = rng.standard_normal((Cout, Cin, k, k)) * 0.1
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_()
    wtt = torch.tensor(wt).requires_grad_()
    y = torch.nn.functional.conv2d(torch.nn.UpsamplingNearest2d(size=(H + k - 1, W + k - 1))(xt), wtt)
    assert tuple(y.shape) == (NB, Cout, H, W)
    gx_ref, gw_ref = torch.autograd.grad(y, (xt, wtt), torch.tensor(g.astype(np.float64)).permute(0, 3, 1, 2))
    B64 = nb.boxsum(g, vr, hr).astype(np.float64)
    gx = nb.dgrad_from_box(B64, wt, vmap, hmap)
    gw = nb.wgrad_from_box(B64, x, vmap, hmap)
    mag = nb.dgrad_magnitude(B64, wt, vmap, hmap)
    assert np.abs(gx - gx_ref.permute(0, 2, 3, 1).numpy()).max() <= 1e-6 * mag.max()         # B itself is an fp32 sum of <= 9 terms
    assert np.abs(gw - gw_ref.numpy()).max() <= 1e-6 * np.abs(gw_ref.numpy()).max()
