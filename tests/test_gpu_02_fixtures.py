"""TEACHER-FORCED parity against the REFERENCE's own tensors (tests/golden/stages_*.npz, blocks.npz, model_ann_T1.npz, written by
tests/golden/make_golden.py from hooks on /root/reference/network/SNN_models.py's modules): every product stage is fed the reference's
input of that stage at full resolution and must reproduce the reference's spikes; the heads + I-pool must reproduce the reference's depth
maps, Total_Loss and MeanDepthError; blocks.npz's SEWResBlock records (IF and PLIF) likewise incl. input / weight / dL/dw gradients; the
ANN fixture (no thresholds, not chaotic) free-running.
"""
import json
import os

import numpy as np
import pytest
import torch

from _models import DEV, pair, state_sha
from _pinned import pinned_parity, rel_l2
from _util import load_npz, synth_input, synth_label

pytestmark = pytest.mark.gpu

REPORT = {}
FLIP_FRAC = 1e-4           # fraction of a layer's neuron updates that may disagree with the reference's fixture


def _dump():
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_report_fixtures.json', 'w') as f:
        json.dump(REPORT, f, indent=1)


def _unpack(z, name):
    shape = tuple(int(v) for v in z[f'{name}_shape'])
    n = int(np.prod(shape))
    a = np.unpackbits(z[f'{name}_p0'])[:n].astype(np.float32)
    if f'{name}_p1' in z.files:
        a = a + 2.0 * np.unpackbits(z[f'{name}_p1'])[:n].astype(np.float32)
    return torch.from_numpy(a.reshape(shape))                       # [T, 1, C, H, W]


@pytest.mark.parametrize('tag,name', [('stereospike_T1', 'StereoSpike'), ('lif_T1', 'LIFNet'), ('mono_plif_T1', 'PLIFNetMono'),
                                      ('plif_T1', 'PLIFNet'), ('plif_T5', 'PLIFNet')])
def test_fixture_stages_teacher_forced(tag, name):
    from stereospike_amd.clock_driven import functional
    from stereospike_amd.fused import ipool
    from stereospike_amd.network.loss import Total_Loss
    from stereospike_amd.network.metrics import MeanDepthError
    zm, zs = load_npz(f'model_{tag}.npz'), load_npz(f'stages_{tag}.npz')
    T = int(zs['T'])
    orc, net = pair(name, 260, 346, seed=int(zm['seed']))
    assert state_sha(orc) == str(zm['state_sha']), 'seeded default init differs from the fixture generator'
    del orc
    ref = {n: _unpack(zs, n) for n in json.loads(str(zs['names']))}
    x = torch.tensor(zm['x'].astype(np.float32)).transpose(0, 1).contiguous()     # [T, 1, C, H, W]
    pm = dict(net.named_modules())
    cl = lambda t: t.to(DEV).permute(0, 1, 3, 4, 2).contiguous()                    # noqa: E731  [T, B, C, H, W] -> NHWC array
    rep = {}

    def mism(out_cl, want):
        return float((out_cl.permute(0, 1, 4, 2, 3).cpu() != want).float().mean())
    with torch.no_grad():
        functional.reset_net(net)
        prev = {'bottom': x, 'conv1': ref['bottom'], 'conv2': ref['conv1'], 'conv3': ref['conv2'], 'conv4': ref['conv3']}
        for st in ('bottom', 'conv1', 'conv2', 'conv3', 'conv4'):
            out = pm[st].forward_sequence_conv_cl(cl(prev[st]), spikes_in=st != 'bottom')
            rep[st] = mism(out, ref[st])
        out = pm['bottleneck.0'].forward_sequence_cl(cl(ref['conv4']), spikes_in=True)
        rep['bottleneck.0'] = mism(out, ref['bottleneck.0'])
        out = pm['bottleneck.1'].forward_sequence_cl(cl(ref['bottleneck.0']), spikes_in=True)
        rep['bottleneck.1'] = mism(out, ref['bottleneck.1'])
        cur, heads = ref['bottleneck.1'], []
        for lvl, skip in ((4, 'conv3'), (3, 'conv2'), (2, 'conv1'), (1, 'bottom')):
            want = ref[f'deconv{lvl}'] + ref[skip]                                   # out_addK of the reference
            out = pm[f'deconv{lvl}'].forward_sequence_cl(cl(cur), cl(ref[skip]), spikes_in=True)
            rep[f'deconv{lvl}'] = mism(out, want)
            heads.append(pm[f'predict_depth{lvl}'][0].forward_projected_cl(cl(want).flatten(0, 1)).view(T, 1, 1, 260, 346))
            cur = want
        depth_seq = ipool(torch.stack(heads), float(pm['predict_depth4'][1].scale_value), 0.0)
        depths = [depth_seq[T - 1, k] for k in (3, 2, 1, 0)]
        scale = float(np.abs(zm['depth1']).max())
        rep['depth_max_abs_rel'] = max(float(np.abs(d.cpu().numpy() - zm[f'depth{i + 1}']).max()) for i, d in enumerate(depths)) / scale
        gt = torch.tensor(zm['gt']).to(DEV)
        L = float(Total_Loss()(depths, gt, None))
        mde = float(MeanDepthError(depths[0], gt))
        rep['loss_rel'] = abs(L - float(zm['loss'])) / abs(float(zm['loss']))
        rep['mde_rel'] = abs(mde - float(zm['mde'])) / abs(float(zm['mde']))
    REPORT[f'fixture_stages_{tag}'] = rep
    _dump()
    for st in json.loads(str(zs['names'])):
        assert rep[st] <= FLIP_FRAC, (st, rep)
    assert rep['depth_max_abs_rel'] <= 1e-5 and rep['loss_rel'] <= 1e-5 and rep['mde_rel'] <= 1e-5, rep


def test_fixture_ann_T1():
    """BASELINE config 1 network (equivalent ANN: no thresholds, hence no chaos) at 260x346 on the reference's fixture input / weights:
    (a) with BatchNorm in eval mode, against the float64-conv oracle (forward, loss, every gradient) at fp32 tolerance; (b) in training
    mode (BatchNorm on the statistics of ONE sample, B = 1, as the reference ran it) free-running against the numbers the reference's
    own ANN_models.py produced on oneDNN.  The single-sample 1/std amplifies ANY fp32 conv rounding difference (MIOpen vs oneDNN vs
    float64: measured depth 7.4e-4, loss 4.7e-5, gradient norms 4.5e-3 — the same vs the float64 oracle), hence the wider bars of (b)."""
    from stereospike_amd.clock_driven import functional
    from stereospike_amd.network.loss import Total_Loss
    z = load_npz('model_ann_T1.npz')
    orc, net = pair('ANN', 260, 346, seed=int(z['seed']))
    assert state_sha(orc) == str(z['state_sha'])
    xc, gtc = torch.tensor(z['x'].astype(np.float32)), torch.tensor(z['gt'])
    # (a) BatchNorm in eval mode (running statistics: the well-conditioned form of the same graph) against the float64-conv oracle
    orc.eval(), net.eval()
    rep_p = pinned_parity(orc, net, xc, gtc, returns_spikes=False, is_ann=True)
    REPORT['pinned_full_ann_T1_bn_eval'] = rep_p
    assert rep_p['depth_max_abs_rel'] <= 1e-5 and rep_p['loss_rel'] <= 1e-5 and rep_p['grad_rel_l2_max'] <= 2e-3, rep_p
    net.train()
    net.zero_grad()
    x, gt = xc.to(DEV), gtc.to(DEV)
    functional.reset_net(net)
    d = net(x)
    L = Total_Loss()(d, gt, None)
    L.backward()
    scale = float(np.abs(z['depth1']).max())
    rep = dict(depth_max_abs_rel=max(float(np.abs(t.detach().cpu().numpy() - z[f'depth{i + 1}']).max()) for i, t in enumerate(d)) / scale,
               loss_rel=abs(float(L) - float(z['loss'])) / abs(float(z['loss'])))
    names = json.loads(str(z['grad_names']))
    g = dict(net.named_parameters())
    rep['grad_l2_rel'] = max(abs(float(g[k].grad.double().norm()) - float(z['grad_l2'][i])) / (float(z['grad_l2'][i]) + 1e-30)
                             for i, k in enumerate(names) if float(z['grad_l2'][i]) > 1e-6)
    REPORT['fixture_ann_T1'] = rep
    _dump()
    assert rep['depth_max_abs_rel'] <= 5e-3 and rep['loss_rel'] <= 1e-3 and rep['grad_l2_rel'] <= 3e-2, rep


@pytest.mark.parametrize('form', ['nchw', 'nhwc'])
@pytest.mark.parametrize('tag,use_plif', [('sew_if', False), ('sew_plif', True)])
def test_fixture_sew_blocks(tag, use_plif, form):
    """blocks.npz: the reference's own SEWResBlock(32) (blocks.py:135-181), 3 stateful calls without reset on one input, then backward:
    outputs, input gradient, weight gradients, PLIF dL/dw and final membranes against the product block (T = 3: run-time-T kernels)."""
    from stereospike_amd.clock_driven import functional, surrogate
    from stereospike_amd.network.blocks import SEWResBlock
    z = load_npz('blocks.npz')
    blk = SEWResBlock(32, connect_function='ADD', multiply_factor=10., use_plif=use_plif, tau=3., surrogate_function=surrogate.Sigmoid(4.0))
    blk.load_state_dict({k[len(tag) + 3:]: torch.tensor(z[k]) for k in z.files if k.startswith(tag + '_w_')})
    blk = blk.to(DEV)
    x = torch.tensor(z[tag + '_x'].astype(np.float32))
    go = torch.tensor(z[tag + '_go']).to(DEV)
    xs = x.unsqueeze(0).repeat(3, 1, 1, 1, 1).to(DEV)
    functional.reset_net(blk)
    if form == 'nhwc':
        xs = xs.permute(0, 1, 3, 4, 2).contiguous().requires_grad_()
        y = blk.forward_sequence_cl(xs, spikes_in=True).permute(0, 1, 4, 2, 3)
    else:
        xs.requires_grad_()
        y = blk.forward_sequence(xs)
    (y * go).sum().backward()
    gx = xs.grad.sum(0)
    if form == 'nhwc':
        gx = gx.permute(0, 3, 1, 2)
    rep = dict(spike_mismatch=float((y.detach().cpu() != torch.tensor(z[tag + '_y'].astype(np.float32))).float().mean()),
               gx_rel_l2=rel_l2(gx, torch.tensor(z[tag + '_gx'])),
               g_rel_l2={k: rel_l2(p.grad, torch.tensor(z[f'{tag}_g_{k}'])) for k, p in blk.named_parameters()},
               v_sn1=float((blk.sn1.v.detach().cpu() - torch.tensor(z[tag + '_v_sn1'])).abs().max()),
               v_sn2=float((blk.sn2.v.detach().cpu() - torch.tensor(z[tag + '_v_sn2'])).abs().max()))
    REPORT[f'fixture_{tag}_{form}'] = rep
    _dump()
    assert rep['spike_mismatch'] <= FLIP_FRAC, rep
    bar = 2e-3 + 500 * rep['spike_mismatch']                 # one flipped spike moves a 32-channel block's gradients by ~0.3 %
    assert rep['gx_rel_l2'] <= bar and max(rep['g_rel_l2'].values()) <= bar, rep
