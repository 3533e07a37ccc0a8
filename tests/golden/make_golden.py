#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (it needs /root/reference, which never
travels to the GPU box); what it writes — inputs and expected outputs, *.npz next to this file — is data.

It imports the reference's OWN graph files by path
    /root/reference/network/{blocks,SNN_models,ANN_models,loss,metrics}.py
under a synthetic package name (so the broken network/__init__.py:2 is never executed) with
oracle/sj_clock_driven.py standing in for the absent third-party `spikingjelly.clock_driven`
(requirements.txt:3, un-pinned, not installed, no network).  Consequently:
  * loss_metric.npz is a PURE reference fixture (loss.py / metrics.py have no spikingjelly dependency);
  * blocks.npz / model_*.npz pin everything the reference itself defines — conv hyper-parameters and order,
    MultiplyBy placement, SEW add, skip adds, I-neuron accumulation order, firing-rate keys, state_dict
    names, loss, MDE — around the restated neuron arithmetic;
  * neuron_kat.npz are known-answer vectors of the restated neuron arithmetic alone (sj_clock_driven.py run
    through torch autograd): they pin the C oracle and the HIP kernel to that restatement, NOT to upstream
    spikingjelly ("parity unpinned" — see oracle/README.md).  Assumed upstream constants are recorded in
    every file (`sigmoid_alpha`, `atan_alpha`, `charge_statement`).

Before writing, the script asserts that oracle/ref_network.py reproduces the reference modules bit for bit
on the same weights and inputs (depths, spikes, loss, MDE, every parameter gradient).

usage:  python tests/golden/make_golden.py            (≈ 2 min on 8 cores)
"""
import hashlib
import importlib.util
import json
import os
import platform
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import sj_clock_driven as sj          # noqa: E402
from oracle import ref_network as rn              # noqa: E402

REF = '/root/reference/network'
SIGMOID_ALPHA = 4.0
META = dict(sigmoid_alpha=SIGMOID_ALPHA, atan_alpha=2.0, charge_statement='self.v = self.v + x (rebinding)',
            torch=torch.__version__, cpu=platform.processor() or platform.machine(),
            threads=torch.get_num_threads())


def load_reference():
    sys.modules.update(sj.as_modules('spikingjelly'))
    pkg = types.ModuleType('refnet')
    pkg.__path__ = [REF]
    sys.modules['refnet'] = pkg
    out = {}
    for name in ('blocks', 'SNN_models', 'ANN_models', 'loss', 'metrics'):
        spec = importlib.util.spec_from_file_location(f'refnet.{name}', f'{REF}/{name}.py')
        m = importlib.util.module_from_spec(spec)
        sys.modules[f'refnet.{name}'] = m
        spec.loader.exec_module(m)
        out[name] = m
    return out


def sha(t):
    a = t.detach().cpu().contiguous().numpy() if isinstance(t, torch.Tensor) else np.ascontiguousarray(t)
    return hashlib.sha256(a.tobytes()).hexdigest()


def state_sha(net):
    h = hashlib.sha256()
    for k, v in net.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def save(name, **arrays):
    arrays['meta'] = np.array(json.dumps(META))
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1e3:.1f} kB')


# --------------------------------------------------------------------------------------------------
def gen_neuron_kat():
    """Kernel-boundary KATs.  x spans ±3 v_th incl. exact-threshold and ±1 ulp cases."""
    rng = np.random.default_rng(20211121)
    cases = {}
    idx = 0
    N = 512
    for kind in ('IF', 'LIF', 'PLIF'):
        for sg_name, alpha in (('ATan', 2.0), ('Sigmoid', SIGMOID_ALPHA)):
            for scale in (1.0, 10.0):
                for v_reset in (0.0, 0.1):
                    for T in ((1, 5, 10) if scale == 10.0 else (1, 5)):
                        tau = 3.0 if (idx % 2 == 0) else 10.0
                        with_skip = (idx % 3 == 0)
                        with_vinit = (T > 1 and idx % 2 == 1)
                        x = (rng.standard_normal((T, N)) * 1.5 / scale).astype(np.float32)
                        # exact-threshold / ±1ulp inputs at t=0 for the first 96 lanes (v_init is v_reset there)
                        if not with_vinit:
                            one = np.float32(1.0)
                            tgt = np.array([one, np.nextafter(one, np.float32(2)), np.nextafter(one, np.float32(0))],
                                           np.float32)
                            if kind == 'IF':
                                x[0, :96] = np.tile((tgt - np.float32(v_reset)) / np.float32(scale), 32)
                        g = rng.standard_normal((T, N)).astype(np.float32)
                        gv = rng.standard_normal(N).astype(np.float32)
                        skip = rng.integers(0, 3, (T, N)).astype(np.float32) if with_skip else None
                        vi = (rng.standard_normal(N) * 0.5).astype(np.float32) if with_vinit else None
                        r = eager_neuron(kind, x, scale, tau, 1.0, v_reset, sg_name, alpha, g, gv, skip, vi)
                        p = f'c{idx:03d}_'
                        cases[p + 'cfg'] = np.array(json.dumps(dict(kind=kind, surrogate=sg_name, alpha=alpha,
                                                                    scale=scale, v_reset=v_reset, v_th=1.0, T=T,
                                                                    tau=tau, k=r['k'])))
                        cases[p + 'x'] = x
                        cases[p + 'g_out'] = g
                        cases[p + 'g_v_last'] = gv
                        if skip is not None:
                            cases[p + 'skip'] = skip.astype(np.uint8)
                        if vi is not None:
                            cases[p + 'v_init'] = vi
                            cases[p + 'g_v_init'] = r['g_v_init']
                        cases[p + 'out'] = r['out'].astype(np.uint8)
                        cases[p + 'h'] = r['h']
                        cases[p + 'v_last'] = r['v_last']
                        cases[p + 'g_x'] = r['g_x']
                        if kind == 'PLIF':
                            cases[p + 'g_w'] = np.float32(r['g_w'])
                        idx += 1
    cases['n_cases'] = np.array(idx)
    save('neuron_kat.npz', **cases)


def eager_neuron(kind, x_seq, scale, tau, v_th, v_reset, sg, alpha, g_out, g_v_last, skip=None, v_init=None):
    T, N = x_seq.shape
    x = torch.tensor(x_seq, requires_grad=True)
    sgf = sj.ATan(alpha) if sg == 'ATan' else sj.Sigmoid(alpha)
    if kind == 'IF':
        node = sj.IFNode(v_th, v_reset, sgf, True)
    elif kind == 'LIF':
        node = sj.LIFNode(tau, v_th, v_reset, sgf, True)
    else:
        node = sj.ParametricLIFNode(tau, v_th, v_reset, sgf, True)
    vi = None
    if v_init is not None:
        vi = torch.tensor(v_init, requires_grad=True)
        node.v = vi
    outs, hs = [], []
    for t in range(T):
        xs = torch.mul(x[t], scale)                       # MultiplyBy, blocks.py:107
        node.neuronal_charge(xs)
        hs.append(node.v.detach().clone())
        node.neuronal_fire()
        node.neuronal_reset()
        o = node.spike
        if skip is not None:
            o = o + torch.tensor(skip[t])                 # SNN_models.py:171
        outs.append(o)
    out = torch.stack(outs)
    loss = (out * torch.tensor(g_out)).sum() + (node.v * torch.tensor(g_v_last)).sum()
    loss.backward()
    return dict(out=out.detach().numpy(), h=torch.stack(hs).numpy(), v_last=node.v.detach().numpy(),
                g_x=x.grad.numpy(), g_w=node.w.grad.item() if kind == 'PLIF' else None,
                k=float(node.w.detach().sigmoid().item()) if kind == 'PLIF' else None,
                g_v_init=vi.grad.numpy() if vi is not None else None)


# --------------------------------------------------------------------------------------------------
def gen_loss_metric(ref):
    """PURE reference: Total_Loss (loss.py:110-135) and MeanDepthError (metrics.py:83-95), NaN = invalid pixel."""
    g = torch.Generator().manual_seed(7)
    out = {}
    for ci, (B, H, W, nan_frac) in enumerate([(1, 260, 346, 0.25), (2, 48, 64, 0.25), (3, 33, 47, 0.0), (1, 20, 30, 0.9)]):
        preds = [torch.randn(B, 1, H, W, generator=g, dtype=torch.float32).requires_grad_() for _ in range(4)]
        gt = 0.5 + 9.5 * torch.rand(B, 1, H, W, generator=g)
        gt[torch.rand(B, 1, H, W, generator=g) < nan_frac] = float('nan')
        spikes = [torch.randint(0, 3, (B, 8, H // 2, W // 2), generator=g).float() for _ in range(5)]
        for pen in (False, True):
            L = ref['loss'].Total_Loss(alpha=0.5, penalize_spikes=pen, beta=0.5)(preds, gt, spikes)
            grads = torch.autograd.grad(L, preds)
            L2 = rn.total_loss(preds, gt, spikes, penalize_spikes=pen, beta=0.5)
            assert torch.equal(L, L2), 'oracle loss != reference loss'
            tag = f'l{ci}_{"pen" if pen else "nopen"}_'
            out[tag + 'loss'] = L.detach().numpy()
            for i, gr in enumerate(grads):
                out[tag + f'gpred{i}'] = gr.numpy().astype(np.float32) if H < 100 else np.array(gr.double().abs().sum().item())
        mde = ref['metrics'].MeanDepthError(preds[0].detach(), gt)
        assert torch.equal(mde, rn.mean_depth_error(preds[0].detach(), gt))
        if H < 100:
            for i, p in enumerate(preds):
                out[f'l{ci}_pred{i}'] = p.detach().numpy()
            out[f'l{ci}_gt'] = gt.numpy()
            for i, s in enumerate(spikes):
                out[f'l{ci}_spk{i}'] = s.numpy().astype(np.uint8)
        else:
            out[f'l{ci}_seed_note'] = np.array('full-size case: inputs regenerated from torch.Generator().manual_seed(7) stream start')
            for i, p in enumerate(preds):
                out[f'l{ci}_pred{i}'] = p.detach().numpy()
            out[f'l{ci}_gt'] = gt.numpy()
            for i, s in enumerate(spikes):
                out[f'l{ci}_spk{i}'] = s.numpy().astype(np.uint8)
        out[f'l{ci}_mde'] = mde.numpy()
    out['n_cases'] = np.array(4)
    save('loss_metric.npz', **out)


# --------------------------------------------------------------------------------------------------
def gen_blocks(ref):
    """Reference blocks.py with the neuron stand-in: SEWResBlock(32) on 17x22 (IF and PLIF), NNConvUpsampling shapes."""
    B = ref['blocks']
    out = {}
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(12)
    for tag, use_plif in (('sew_if', False), ('sew_plif', True)):
        blk = B.SEWResBlock(32, connect_function='ADD', multiply_factor=10., use_plif=use_plif, tau=3.,
                            surrogate_function=sj.Sigmoid(SIGMOID_ALPHA))
        x = (torch.rand(2, 32, 17, 22, generator=g) < 0.3).float().requires_grad_()
        go = torch.randn(2, 32, 17, 22, generator=g)
        outs = []
        sj.reset_net(blk)
        for t in range(3):                     # stateful: 3 calls without reset
            outs.append(blk(x))
        y = torch.stack(outs)
        (y * go).sum().backward()
        out[tag + '_x'] = x.detach().numpy().astype(np.uint8)
        out[tag + '_go'] = go.numpy()
        out[tag + '_y'] = y.detach().numpy().astype(np.uint8)
        out[tag + '_gx'] = x.grad.numpy()
        for k, v in blk.state_dict().items():
            out[tag + '_w_' + k] = v.numpy()
        for k, p in blk.named_parameters():
            out[tag + '_g_' + k] = p.grad.numpy()
        out[tag + '_v_sn1'] = blk.sn1.v.detach().numpy()
        out[tag + '_v_sn2'] = blk.sn2.v.detach().numpy()
    for i, (cin, cout, k, insz, up) in enumerate([(8, 4, 5, (17, 22), (33, 44)), (4, 2, 5, (33, 44), (65, 87)),
                                                  (4, 1, 3, (33, 44), (260, 346)), (2, 1, 3, (7, 9), (20, 30))]):
        m = B.NNConvUpsampling(cin, cout, k, up, bias=(k == 3))
        x = torch.randn(1, cin, *insz, generator=g)
        y = m(x)
        assert tuple(y.shape[-2:]) == up
        out[f'up{i}_cfg'] = np.array(json.dumps(dict(cin=cin, cout=cout, k=k, up=up)))
        out[f'up{i}_x'] = x.numpy()
        out[f'up{i}_y'] = y.detach().numpy()
        for kk, v in m.state_dict().items():
            out[f'up{i}_w_' + kk] = v.numpy()
    m = B.MultiplyBy(10.)
    x = torch.randn(1000, generator=g)
    out['mul_x'] = x.numpy()
    out['mul_y'] = m(x).numpy()
    save('blocks.npz', **out)


# --------------------------------------------------------------------------------------------------
def synth_input(B, T, C, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.poisson(torch.full((B, T, C, 260, 346), 0.05), generator=g)


def synth_label(B, seed):
    g = torch.Generator().manual_seed(seed)
    gt = 0.5 + 9.5 * torch.rand(B, 1, 260, 346, generator=g)
    gt[torch.rand(B, 1, 260, 346, generator=g) < 0.25] = float('nan')
    return gt


def gen_model(ref, tag, make_ref, make_oracle, C, T, returns_spikes, seed=2021):
    """Full-resolution (260x346 is hard-wired in the reference: SNN_models.py:111-146), B=1.
    Weights = default init under torch.manual_seed(seed) (train.py:53) — regenerated, not stored; their sha256 is."""
    torch.manual_seed(seed)
    net = make_ref()
    torch.manual_seed(seed)
    orc = make_oracle()
    assert list(net.state_dict().keys()) == list(orc.state_dict().keys())
    sha0 = state_sha(net)          # before any forward: BatchNorm running stats (ANN) change afterwards
    assert sha0 == state_sha(orc), 'same seed must give same default init'
    x = synth_input(1, T, C, seed + 1)
    gt = synth_label(1, seed + 2)

    def run(n, is_ref):
        sj.reset_net(n)
        res = None
        for t in range(T):
            res = n(x[:, t:t + 1])
        depths, spikes = res if returns_spikes else (res, [])
        L = (ref['loss'].Total_Loss() if is_ref else (lambda p, g_, s: rn.total_loss(p, g_, s)))(depths, gt, spikes)
        mde = (ref['metrics'].MeanDepthError if is_ref else rn.mean_depth_error)(depths[0].detach(), gt)
        L.backward()
        return depths, spikes, L, mde

    d, s, L, mde = run(net, True)
    d2, s2, L2, mde2 = run(orc, False)
    assert all(torch.equal(a, b) for a, b in zip(d, d2)), 'oracle depths != reference'
    assert all(torch.equal(a, b) for a, b in zip(s, s2)), 'oracle spikes != reference'
    assert torch.equal(L, L2) and torch.equal(mde, mde2)
    for (k, p), (_, q) in zip(net.named_parameters(), orc.named_parameters()):
        assert torch.equal(p.grad, q.grad), f'oracle grad {k} != reference'
    out = dict(x=x.numpy().astype(np.uint8), gt=gt.numpy(), loss=L.detach().numpy(), mde=mde.numpy(),
               state_sha=np.array(sha0), seed=np.array(seed), T=np.array(T), C=np.array(C))
    for i, t in enumerate(d):
        out[f'depth{i + 1}'] = t.detach().numpy()
    for name, t in zip(('out_rconv', 'out_add4', 'out_add3', 'out_add2', 'out_add1'), s):
        out[name] = t.detach().numpy().astype(np.uint8)
    out['grad_names'] = np.array(json.dumps([k for k, _ in net.named_parameters()]))
    out['grad_l2'] = np.array([p.grad.double().norm().item() for p in net.parameters()])
    out['grad_sum'] = np.array([p.grad.double().sum().item() for p in net.parameters()])
    # membranes after the last step (state carried over T)
    vs = {k: m.v for k, m in net.named_modules() if isinstance(m, sj.BaseNode) and isinstance(m.v, torch.Tensor)}
    out['v_names'] = np.array(json.dumps(list(vs.keys())))
    out['v_sum'] = np.array([v.detach().double().sum().item() for v in vs.values()])
    out['v_sha'] = np.array(json.dumps([sha(v) for v in vs.values()]))
    if hasattr(net, 'calculate_firing_rates') and T == 1:
        sj.reset_net(net)
        with torch.no_grad():
            fr = net.calculate_firing_rates(x)
        out['rates'] = np.array(json.dumps({k: float(v) for k, v in fr.items()}))
    save(f'model_{tag}.npz', **out)


STAGE_NAMES = ('bottom', 'conv1', 'conv2', 'conv3', 'conv4', 'bottleneck.0', 'bottleneck.1', 'deconv4', 'deconv3', 'deconv2', 'deconv1')


def gen_stage_records(ref, tag, make_ref, C, T, seed=2021):
    """stages_<tag>.npz: the output of every spiking stage of the REFERENCE's own model (forward hooks on its modules) at every time
    step, bit-packed, on exactly the weights / inputs of model_<tag>.npz.  Lets the GPU suite hold each product stage against the
    reference TEACHER-FORCED (stage input = the reference's own tensor), i.e. without the chaotic cascade of a free-running comparison.
    Spiking stages emit 0/1 (one bit plane); the SEW blocks emit 0..3 (two planes)."""
    torch.manual_seed(seed)
    net = make_ref()
    x = synth_input(1, T, C, seed + 1)
    rec = {n: [] for n in STAGE_NAMES}
    mods = dict(net.named_modules())
    hooks = [mods[n].register_forward_hook(lambda m, i, o, n=n: rec[n].append(o.detach().clone())) for n in STAGE_NAMES]
    sj.reset_net(net)
    with torch.no_grad():
        for t in range(T):
            res = net(x[:, t:t + 1])
    for h in hooks:
        h.remove()
    depths, spikes = res if isinstance(res, tuple) else (res, None)
    z = np.load(os.path.join(HERE, f'model_{tag}.npz'))
    assert str(z['state_sha']) == state_sha_fresh(make_ref, seed), 'weights differ from model fixture'
    for i, d in enumerate(depths):
        assert np.array_equal(d.numpy(), z[f'depth{i + 1}']), 'this run does not reproduce the model fixture'
    out = dict(T=np.array(T), C=np.array(C), seed=np.array(seed), names=np.array(json.dumps(STAGE_NAMES)))
    for n in STAGE_NAMES:
        a = torch.stack(rec[n]).numpy()                         # [T, 1, C, H, W] float
        u = a.astype(np.uint8)
        assert np.array_equal(u.astype(np.float32), a) and u.max() <= 3
        out[f'{n}_shape'] = np.array(u.shape)
        out[f'{n}_p0'] = np.packbits(u & 1)
        if n.startswith('bottleneck'):
            out[f'{n}_p1'] = np.packbits((u >> 1) & 1)
        else:
            assert u.max() <= 1
    if spikes is not None:                                        # consistency with the model fixture's returned spike tensors
        add4 = rec['deconv4'][-1] + rec['conv3'][-1]
        assert np.array_equal(add4.numpy().astype(np.uint8), z['out_add4'])
        assert np.array_equal(rec['bottleneck.1'][-1].numpy().astype(np.uint8), z['out_rconv'])
    save(f'stages_{tag}.npz', **out)


def state_sha_fresh(make_ref, seed):
    torch.manual_seed(seed)
    return state_sha(make_ref())


def gen_voxelizer():
    """The reference's OWN voxeliser (datasets/MVSEC/utils.py:215-281), extracted from the file by name (the module itself
    cannot be imported: cv2 / h5py / skimage are absent) and run on synthetic event streams; pins oracle/np_voxelize.py."""
    import ast
    from oracle import np_voxelize as nv
    src = open('/root/reference/datasets/MVSEC/utils.py').read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'mvsecCumulateSpikesIntoFrames'][0]
    ns = dict(np=np, tqdm=lambda it: it, LIDAR_FPS=20, print=lambda *a, **k: None)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'utils.py', 'exec'), ns)
    ref_fn = ns['mvsecCumulateSpikesIntoFrames']
    rng = np.random.default_rng(77)
    out = {}
    for ci, (n_chunks, nfpdm, E) in enumerate([(3, 1, 4000), (2, 5, 6000), (2, 25, 5000), (4, 2, 3000)]):
        dur = n_chunks * 0.05
        t = np.sort(rng.uniform(1000.0, 1000.0 + dur * 1.1, E))
        ev = np.stack([rng.uniform(0, 345.99, E), rng.uniform(0, 259.99, E), t, rng.integers(0, 2, E).astype(np.float64)], 1)
        ev[: E // 50, 3] = -1.0                               # MVSEC stores OFF as -1 in some files: "else" branch
        # events exactly on frame boundaries (dropped by the strict inequalities)
        st, en = nv.frame_bounds(n_chunks, nfpdm)
        ev[10:10 + min(len(st), 20), 2] = ev[0, 2] + st[:20]
        ev = ev[np.argsort(ev[:, 2], kind='stable')]
        depth = np.zeros((n_chunks, 260, 346)); ts = ev[0, 2] + 0.05 * (1 + np.arange(n_chunks))
        frames_ref, _ = ref_fn(ev.copy(), depth, ts.copy(), num_frames_per_depth_map=nfpdm)
        mine = nv.cumulate_spikes_into_frames(ev, n_chunks, nfpdm)
        assert np.array_equal(frames_ref, mine), 'oracle voxeliser != reference'
        out[f'v{ci}_events'] = ev
        out[f'v{ci}_cfg'] = np.array([n_chunks, nfpdm])
        out[f'v{ci}_frames'] = frames_ref.astype(np.uint16)
    out['n_cases'] = np.array(4)
    save('voxelizer.npz', **out)


def main():
    torch.set_num_threads(8)
    stages_only = '--stages' in sys.argv
    if not stages_only:
        gen_voxelizer()
    ref = load_reference()
    S, A = ref['SNN_models'], ref['ANN_models']
    plif = dict(tau=3., v_threshold=1.0, v_reset=0.0, use_plif=True, multiply_factor=30.)   # gain 10 leaves the tau=3 PLIF net silent beyond conv1
    lif = dict(tau=3., v_threshold=1.0, v_reset=0.0, use_plif=False, multiply_factor=30.)
    mk_ss = lambda: S.StereoSpike(surrogate_function=sj.ATan(), detach_reset=True, v_threshold=1.0, v_reset=0., multiply_factor=10.)
    mk_plif = lambda: S.fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike(**plif)
    mk_lif = lambda: S.fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike(**lif)
    mk_mono = lambda: S.fromZero_feedforward_multiscale_tempo_monocular_SpikeFlowNetLike(**plif)
    if not stages_only:
        gen_neuron_kat()
        gen_loss_metric(ref)
        gen_blocks(ref)
        # config 3 network: StereoSpike, ATan outside / Sigmoid inside the bottleneck (train.py:118), gain 10 so neurons fire
        orc_ss = lambda: rn.build('StereoSpike', multiply_factor=10., surrogate_function=sj.ATan(), sigmoid_alpha=SIGMOID_ALPHA)
        gen_model(ref, 'stereospike_T1', mk_ss, orc_ss, 4, 1, True)
        gen_model(ref, 'stereospike_T5', mk_ss, orc_ss, 4, 5, True)
        # config 2 family: PLIF model (train.py:120) binocular T=1 and T=5, LIF variant, monocular
        gen_model(ref, 'plif_T1', mk_plif, lambda: rn.build('PLIFNet', sigmoid_alpha=SIGMOID_ALPHA, **plif), 4, 1, True)
        gen_model(ref, 'plif_T5', mk_plif, lambda: rn.build('PLIFNet', sigmoid_alpha=SIGMOID_ALPHA, **plif), 4, 5, True)
        gen_model(ref, 'lif_T1', mk_lif, lambda: rn.build('PLIFNet', sigmoid_alpha=SIGMOID_ALPHA, **lif), 4, 1, True)
        gen_model(ref, 'mono_plif_T1', mk_mono, lambda: rn.build('PLIFNetMono', sigmoid_alpha=SIGMOID_ALPHA, **plif), 2, 1, False)
        # config 1: the equivalent ANN (CPU plumbing case)
        gen_model(ref, 'ann_T1', lambda: A.StereoSpike_equivalentANN(), lambda: rn.build('ANN'), 4, 1, False)
    # per-stage records of the reference's own modules (teacher-forcing data for the GPU suite); `--stages` writes only these
    gen_stage_records(ref, 'stereospike_T1', mk_ss, 4, 1)
    gen_stage_records(ref, 'lif_T1', mk_lif, 4, 1)
    gen_stage_records(ref, 'mono_plif_T1', mk_mono, 2, 1)
    gen_stage_records(ref, 'plif_T1', mk_plif, 4, 1)
    gen_stage_records(ref, 'plif_T5', mk_plif, 4, 5)


if __name__ == '__main__':
    main()
