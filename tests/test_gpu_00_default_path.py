"""End-to-end GPU parity of the SHIPPED DEFAULT configuration — the kernels bench.py prices — collected FIRST in `pytest -m gpu`.

TRAJECTORY-PINNED (tests/_pinned.py): the product runs freely (whole network on NHWC arrays, exact bf16x3 MFMA synapses, fused up-conv
kernels, packed spike tensors, forked + low-rank gradients, h recomputed in backward, fused loss); the float64-convolution oracle is pinned
to the product's spike trajectory layer by layer.  Every test here ASSERTS THE LAUNCH TAGS of the run (fused.TIMER): the compile-time-T
recompute / packed / forked / low-rank kernel forms must be the ones that ran — a silent fall-back to the run-time-T saved-h path fails
the test.  T = 5 (BASELINE configs 3 / 4) for all five model families, T = 10 (config 5), T = 1 (config 2), the
config-3 network at 260x346 with B = 1 (reference fixture input) and with the real config-3 batch B = 16; ONE case, labelled so, on the
run-time-T kernels (T = 7, odd sizes).

Bars, per kind of quantity (measured values: profiles/r03/parity_report_default_path*.json):
  * per layer <= 1e-4 of the neuron updates may disagree with the oracle's own Heaviside, each within MARGIN of its threshold;
  * depths / loss / MDE <= 1e-5 relative;
  * every weight TENSOR gradient of the composed backward <= 1e-4 relative L2;
  * every 0-dim PLIF w gradient within 2e-6 of its MAGNITUDE sum (float64, from hooks on the oracle's membranes) — the scalar is one
    cancelling sum over a layer with condition number up to ~1e3, so a bar relative to the cancelled value measures the conditioning of
    the layer, not the kernel.
MIOpen-backed non-default execution layouts: tests/test_gpu_zz_layouts.py (collected last).
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
REPORT_FILE = 'gpurun_out/parity_report_default_path.json'

MARGIN_DEFAULT = 5e-5      # |h - v_th| of a neuron on which product and float64-conv oracle may disagree: exact bf16x3 / fp32-GEMM synapses
FLIP_FRAC = 1e-4           # fraction of a layer's neuron updates that may disagree at all
TENSOR_GRAD_BAR = 1e-4     # relative L2 of a weight tensor's gradient (measured <= 3.0e-5 over every spiking case of this suite)
PLIF_W_BAR = 2e-6          # |dL/dw - float64 value| / magnitude sum (measured <= 1.6e-7; condition numbers of the scalar up to 8.7e4)


def _dump():
    os.makedirs('gpurun_out', exist_ok=True)
    with open(REPORT_FILE, 'w') as f:
        json.dump(REPORT, f, indent=1)


def check(tag, rep, margin=MARGIN_DEFAULT, depth_bar=1e-5, tensor_bar=TENSOR_GRAD_BAR, plif_bar=PLIF_W_BAR, flip_frac=FLIP_FRAC):
    REPORT[tag] = rep
    _dump()
    assert rep['flip_frac_max'] <= flip_frac, (tag, rep['layers'])
    assert rep['margin_max'] <= margin, (tag, rep['layers'])
    assert rep['depth_max_abs_rel'] <= depth_bar, (tag, rep['depth_max_abs_rel'])
    assert rep['loss_rel'] <= 1e-5 and rep['mde_rel'] <= 1e-5, (tag, rep['loss'], rep['mde'])
    assert rep['tensor_grad_rel_l2_max'] <= tensor_bar, (tag, {k: v for k, v in rep['grad_rel_l2'].items() if k not in rep['plif_w']})
    assert rep['plif_w_err_over_magnitude_max'] <= plif_bar, (tag, rep['plif_w'])


DEFAULT_PLAN = {
    # layer: (synapse forward, synapse backward) kernel forms of the shipped fp32 configuration, as the dispatch sites record them (net.plan())
    'bottom': ('dense_conv_s1_fwd6_mfma', 'g_x: none; g_w: dense_conv_s1_wgrad6_mfma'),
    'conv1': ('spike_conv_fwd3_mfma(packed in)', 'g_x: conv_s2_dgrad6_mfma; g_w: spike_conv_wgrad3_mfma'),
    'conv2': ('spike_conv_fwd3_mfma(packed in)', 'g_x: conv_s2_dgrad6_mfma; g_w: spike_conv_wgrad3_mfma'),
    'conv3': ('im2col(packed in)+exact_bf16x3_gemm', 'g_x: conv_s2_dgrad6_mfma; g_w: split3+exact_bf16x3_gemm'),
    'conv4': ('im2col(packed in)+exact_bf16x3_gemm', 'g_x: conv_s2_dgrad6_mfma; g_w: split3+exact_bf16x3_gemm'),
    **{f'bottleneck.{b}.conv{c}': ('im2col(packed in)+exact_bf16x3_gemm', 'g_x: winograd_f2x2_3x3+batched_gemm; g_w: split3+exact_bf16x3_gemm') for b in (0, 1) for c in (1, 2)},
    'predict_depth1': ('head_proj_packed_mfma+gather', 'g_x: lowrank_pair; g_w: head_wgrad_packed_mfma'),
    'predict_depth2': ('head_proj_packed_mfma+gather', 'g_x: lowrank_pair; g_w: head_wgrad_packed_mfma'),
    'predict_depth3': ('fp32_gemm+gather', 'g_x: lowrank_pair; g_w: library_gemm'),
    'predict_depth4': ('fp32_gemm+gather', 'g_x: lowrank_pair; g_w: library_gemm'),
}


def assert_default_plan(plan, full_resolution=False):
    """The dispatch plan of a default fp32 run (`assert_default_kernels` checks the plan as well as the launch tags): every encoder /
    bottleneck / head layer on the kernel form named above; the decoder stages by their GEOMETRY (fused.stage_plan, round 5): the sub-pixel forward on every
    stage but the small wide ones (deconv4 always; deconv3 on the 64x80 pyramid's 8x10 source map), the box-sum backward on deconv1 / deconv2 where the kernels'
    on-chip window holds the geometry, the per-tap g_P forms elsewhere."""
    for name, (fwd, bwd) in DEFAULT_PLAN.items():
        assert plan[name]['synapse_fwd'] == fwd and plan[name]['synapse_bwd'] == bwd, (name, plan[name])
    sub = (1, 2, 3) if full_resolution else (1, 2)
    for lvl in (1, 2, 3, 4):                                               # round 4: the sub-pixel (merged tap) implicit GEMM on the packed spikes
        want = 'upconv_sub_mfma(packed in)' if lvl in sub else 'exact_bf16x3_gemm+gather'
        assert plan[f'deconv{lvl}']['synapse_fwd'] == want, (lvl, plan[f'deconv{lvl}'])
    box = [lvl for lvl in (1, 2, 3, 4) if plan[f'deconv{lvl}']['synapse_bwd'].startswith('box: boxsum+dgrad6_mfma+wgrad3_mfma')]
    # (64x80 frames: deconv1's 32 source columns reach 84 horizontal ranges, more than the on-chip window's 76 — that stage runs the g_P forms there)
    assert set(box) == ({1, 2} if full_resolution else {2}), {lvl: plan[f'deconv{lvl}']['synapse_bwd'] for lvl in (1, 2, 3, 4)}
    for lvl in (1, 2, 3, 4):
        if lvl not in box:
            assert plan[f'deconv{lvl}']['synapse_bwd'] == ('g_x: adjoint+gemm6; g_w: spike_wgrad_mfma' if lvl >= 2 else 'g_x: adjoint+library_gemm; g_w: spike_wgrad_mfma'), plan[f'deconv{lvl}']
    assert len(plan) == 17 and all('neuron_fwd' in v and 'neuron_bwd' in v for k, v in plan.items() if not k.startswith('predict'))


def assert_default_kernels(tags, T, penalized=False, full_resolution=False):
    """The launch tags of one forward + backward of a 13-layer spiking network in the shipped default configuration at a compile-time T:
    packed-only outputs on the seven edges into exact-split / implicit-GEMM convs, packed skip operands, forked gradients on conv1..3, the four heads'
    gradients as rank-9 pairs (+ the full-resolution pair travelling on into the first encoder layer), nothing on the saved-h forms."""
    fwd = {k: v for k, v in tags.items() if k.startswith('neuron_fwd')}
    bwd = {k: v for k, v in tags.items() if k.startswith('neuron_bwd')}
    assert sum(fwd.values()) == 13 and sum(bwd.values()) == 13, tags
    assert not any(k.endswith('+h') for k in fwd) and not any('savedh' in k for k in bwd), ('saved-h (run-time-T) kernels ran', tags)
    # packed-only outputs: the 7 encoder / bottleneck edges, the SEW block's inner layer (with its packed skip) and — its prediction head reads packed
    # spikes (fused.PACKED_HEAD) — the two largest decoder stages (deconv2's other consumer, deconv1, reads packed spikes in its fused kernels)
    assert fwd.get('neuron_fwd_train+packed', 0) == 7 and fwd.get('neuron_fwd_train+skip+packed', 0) == 3, tags      # SEW inner layer, deconv2, deconv1
    # (round 4: deconv4 and deconv3 write a packed COPY beside their dense output — the head reads the dense tensor, the next stage's sub-pixel forward
    #  the packed one; the SEW block's last layer, the bottleneck output, stays dense only)
    # (64x80 pyramid: deconv3's 8x10 source map stays on GEMM + gather, so deconv4 writes no packed copy for it)
    n_copy = 2 if full_resolution else 1
    assert fwd.get('neuron_fwd_train', 0) == 0 and fwd.get('neuron_fwd_train+skip', 0) == 3 - n_copy and fwd.get('neuron_fwd_train+skip+pkcopy', 0) == n_copy, tags
    assert tags.get('spike_conv_fwd', 0) == 2, tags                       # conv1 / conv2 forward: the exact MFMA implicit GEMM on the packed spikes
    assert tags.get('dense_conv_s1_fwd', 0) == 1, tags                    # the first layer's forward: six-term MFMA implicit GEMM
    assert tags.get('conv_s2_dgrad', 0) == 4, tags                        # conv1 .. conv4 data gradient: six-term MFMA implicit GEMM (no MIOpen igemm_bwd)
    assert tags.get('dense_conv_s1_wgrad', 0) == 1, tags                  # the first layer's weight gradient: six-term MFMA contraction (the step's last MIOpen call is gone)
    if penalized:
        # Total_Loss(penalize_spikes=True): the full-resolution stage's output also gets a DENSE gradient (the penalty on the returned out_add1, through
        # fused.unpack_last_step), so its backward is the '+lr+sum' form and what travels on into the first encoder layer is a dense sum, not the pair
        assert bwd.get('neuron_bwd+lronly', 0) == 0 and bwd.get('neuron_bwd+lr', 0) == 0 and bwd.get('neuron_bwd+lr+sum', 0) == 4, tags
        assert bwd.get('neuron_bwd+fork', 0) == 4 and bwd.get('neuron_bwd', 0) == 5, tags
    else:
        assert bwd.get('neuron_bwd+lronly', 0) == 1 and bwd.get('neuron_bwd+lr', 0) == 1 and bwd.get('neuron_bwd+lr+sum', 0) == 3, tags
        assert bwd.get('neuron_bwd+fork', 0) == 3 and bwd.get('neuron_bwd', 0) == 5, tags
    assert tags.get('upconv_cl_fwd', 0) == 8 and tags.get('upconv_cl_bwd', 0) == 8, tags


FAMILIES = [('StereoSpike', 4), ('PLIFNet', 4), ('LIFNet', 4), ('PLIFNetMono', 2)]


@pytest.mark.parametrize('name,C', FAMILIES)
def test_pinned_parity_default_kernels_T5(name, C):
    """64x80 frames, B = 2, T = 5 with BPTT (membranes carried) — the four spiking families on the kernels of BASELINE configs 3 / 4."""
    H, W, T = 64, 80, 5
    orc, net = pair(name, H, W)
    x = synth_input(2, T, C, 77, H, W, lam=0.08)
    gt = synth_label(2, 78, H, W)
    rep = pinned_parity(orc, net, x, gt, returns_spikes=name != 'PLIFNetMono')
    assert len(rep['layers']) == 13 and len(rep['plif_w']) == {'StereoSpike': 0, 'LIFNet': 4}.get(name, 13)
    assert_default_kernels(rep['launch_tags'], T)
    assert_default_plan(rep['plan'])
    check(f'pinned_T5_{name}', rep)


@pytest.mark.parametrize('name,C,T', [('StereoSpike', 4, 10), ('PLIFNetMono', 2, 1), ('StereoSpike', 4, 1)])      # (PLIF at T = 10: tests/test_gpu_04_x16_parity.py)
def test_pinned_parity_default_kernels_T10_T1(name, C, T):
    """The other compile-time time-step counts BASELINE names: T = 10 (config 5) and T = 1 (config 2: monocular PLIF)."""
    H, W = 64, 80
    orc, net = pair(name, H, W)
    x = synth_input(2, T, C, 79, H, W, lam=0.12 if T == 1 else 0.08)
    gt = synth_label(2, 80, H, W)
    rep = pinned_parity(orc, net, x, gt, returns_spikes=name != 'PLIFNetMono')
    assert_default_kernels(rep['launch_tags'], T)
    check(f'pinned_T{T}_{name}', rep)


@pytest.mark.parametrize('name', ['StereoSpike', 'PLIFNet'])
def test_pinned_parity_penalize_spikes_T5(name):
    """Total_Loss(penalize_spikes=True, beta=20) at network level (/root/reference/network/loss.py:96-107,126-135; train.py:125): the loss reads the five
    RETURNED spike tensors, so a gradient enters out_rconv / out_add4..1 directly — two of them (out_add2, out_add1) are fused.unpack_last_step views of
    packed-only anchors that also carry a forked and a low-rank gradient.  Same pinned protocol and bars as the default loss; the default (packed /
    forked / low-rank) launch tags are asserted in the form this loss gives them."""
    H, W, T = 64, 80, 5
    orc, net = pair(name, H, W)
    x = synth_input(2, T, 4, 81, H, W, lam=0.08)
    gt = synth_label(2, 82, H, W)
    # beta = 20: on this untrained network the depth terms are ~9000 and their gradients dwarf the penalty's; at beta = 0.5 the penalty moves the bottleneck's
    # weight gradients by 8e-5 — BELOW the parity bar, i.e. invisible to it (measured, profiles/r04/) — at 20 by >= 30 x the bar (negative control below)
    rep = pinned_parity(orc, net, x, gt, penalize_spikes=True, beta=20.0)
    assert_default_kernels(rep['launch_tags'], T, penalized=True)
    check(f'pinned_T5_{name}_penalize_spikes', rep)
    g_pen = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    # negative control — the penalty really is in the loss and in the gradients: the same run without it gives a visibly smaller loss and
    # weight gradients that differ by far more than the parity bar
    rep0 = pinned_parity(orc, net, x, gt)
    assert rep['loss'][0] - rep0['loss'][0] > 10.0, (rep['loss'], rep0['loss'])     # beta / 2 * sum of five mean(s^2): 2.5 * beta at these densities
    moved = {k: rel_l2(g_pen[k], p.grad) for k, p in net.named_parameters()}
    assert moved['deconv1.0.up.1.weight'] > 10 * TENSOR_GRAD_BAR and moved['bottleneck.1.conv2.0.weight'] > 3 * TENSOR_GRAD_BAR, moved


def test_pinned_parity_ann():
    """BASELINE config 1's network (no thresholds: nothing to pin), T = 1: forward, loss and every gradient against the float64-conv oracle."""
    H, W = 64, 80
    orc, net = pair('ANN', H, W)
    rep = pinned_parity(orc, net, synth_input(2, 1, 4, 77, H, W, lam=0.08), synth_label(2, 78, H, W), returns_spikes=False, is_ann=True)
    assert len(rep['layers']) == 0
    # the ANN's BatchNorm layers run on batch statistics (training mode, as the reference): 1/std amplifies conv rounding (measured 1.2e-5)
    check('pinned_ANN', rep, depth_bar=1e-4, tensor_bar=2e-3)


def test_pinned_parity_runtime_T_kernels_odd_sizes():
    """RUN-TIME-T KERNELS (saved h, torch adds the forked gradients) — the one case that is NOT the benchmarked path, labelled so: T = 7,
    frame size with odd pyramid levels (50x70 -> 25x35 -> 13x18 -> 7x9 -> 4x5), B = 3."""
    orc, net = pair('PLIFNet', 50, 70)
    x = synth_input(3, 7, 4, 123, 50, 70, lam=0.08)
    gt = synth_label(3, 124, 50, 70)
    rep = pinned_parity(orc, net, x, gt)
    tags = rep['launch_tags']
    assert all(k.endswith('+h') for k in tags if k.startswith('neuron_fwd')) and all('savedh' in k for k in tags if k.startswith('neuron_bwd')), tags
    check('pinned_runtimeT_odd_50x70_T7_PLIFNet', rep)


def test_pinned_parity_full_resolution_stereospike_T5():
    """BASELINE config 3 network at 260x346, B = 1, T = 5 on the committed fixture's input and weights (model_stereospike_T5.npz)."""
    z = load_npz('model_stereospike_T5.npz')
    x = torch.tensor(z['x'].astype(np.float32))
    gt = torch.tensor(z['gt'])
    orc, net = pair('StereoSpike', 260, 346, seed=int(z['seed']))
    assert state_sha(orc) == str(z['state_sha'])
    rep = pinned_parity(orc, net, x, gt)
    assert_default_kernels(rep['launch_tags'], 5, full_resolution=True)
    assert_default_plan(rep['plan'], full_resolution=True)
    check('pinned_full_stereospike_T5', rep)
    # statistics of the free-running product against the reference's fixture (chaotic per neuron, stable in the mean)
    for nm, dens in zip(('out_rconv', 'out_add4', 'out_add3', 'out_add2', 'out_add1'), rep['product_spike_density']):
        assert abs(dens - float((z[nm] != 0).mean())) <= 1e-2, nm
    assert abs(rep['loss'][0] - float(z['loss'])) <= 0.1 * abs(float(z['loss']))


def test_pinned_parity_config3_step_B16_T5():
    """THE config-3 step: StereoSpike, B = 16, T = 5, 260x346 (bench.py's synthetic input distribution) — the launch shapes bench.py prices
    (2.3e8 updates in the bottom layer's launch, the fused up-conv kernels' persistent grids at the real frame count, split-K weight
    gradients over 1.4e6 rows).  The oracle walks the batch in chunks of 2 samples, one host process per chunk (tests/_pinned.py: only the loss
    couples the batch): 645 s sequentially on the GPU box, ~90 s this way."""
    if os.environ.get('SS_SKIP_B16_PARITY') == '1':
        pytest.skip('SS_SKIP_B16_PARITY=1')
    orc, net = pair('StereoSpike', 260, 346)
    x = synth_input(16, 5, 4, 2021)
    gt = synth_label(16, 2022)
    rep = pinned_parity(orc, net, x, gt, oracle_chunk=2, oracle_procs=True)
    assert_default_kernels(rep['launch_tags'], 5, full_resolution=True)
    assert_default_plan(rep['plan'], full_resolution=True)
    check('pinned_config3_B16_T5', rep)


def test_packed_spike_tensors_are_in_effect():
    """fused.PACK_SPIKES (default on; the pinned tests above assert its launch tags): 2-bit packed spike tensors between conv2 .. bottleneck and
    their consumers and on the full-resolution decoder stage's output (packed-only: the autograd output is a data-less anchor), packed skip operands
    for the decoder.  Here: the packed
    form really is in effect, and within ONE launch that writes both forms (bottom, conv1) unpack(packed) == the dense tensor bit for bit;
    with PACK_SPIKES off no packed tensor exists.  (Bit-equality of the consumers on packed vs dense input: tests/test_gpu_01_kernels.py.)"""
    from stereospike_amd import fused
    from stereospike_amd.clock_driven import functional
    from stereospike_amd.network.loss import Total_Loss
    H, W = 64, 80
    _, net = pair('StereoSpike', H, W)
    x = synth_input(2, 5, 4, 7, H, W, lam=0.08).to(DEV)
    gt = synth_label(2, 8, H, W).to(DEV)
    T0 = 5
    rec = {}
    orig = net.bottom[2].forward_sequence

    def spy(x_seq, *a, **kw):
        r = orig(x_seq, *a, **kw)
        r0 = r[0] if isinstance(r, tuple) else r
        rec['bottom_out'], rec['shape'], rec['anchor'] = r0.detach().clone(), x_seq.shape, not any(r0.stride())
        return r
    net.bottom[2].forward_sequence = spy
    functional.reset_net(net)
    d, s = net.forward_sequence(x)
    Total_Loss()(d, gt, s).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    for st in (net.bottom[2], net.conv1[2], net.conv2[2], net.conv3[2], net.conv4[2], net.bottleneck[0].sn1, net.bottleneck[0].sn2, net.bottleneck[1].sn1):
        assert st.last_packed is not None and st.last_packed.dtype == torch.int32                # packed only: every consumer reads the packed form
    assert net.bottleneck[1].sn2.last_packed is None                                            # dense only
    assert net.deconv4[2].last_packed is None and any(s[1].stride())                            # dense only HERE: deconv3's 8x10 source map stays on GEMM + gather (fused.stage_plan); at 260x346 deconv4 writes a packed copy
    # deconv3: dense (its head reads it) + a packed COPY for deconv2's sub-pixel forward / weight gradient (round 4), bit-identical contents
    pk3 = net.deconv3[2].last_packed
    assert pk3 is not None and pk3.dtype == torch.int32 and tuple(s[2].shape) == (2, 128, H // 4, W // 4) and any(s[2].stride())
    assert torch.equal(fused.unpack_dense(pk3[T0 - 1:T0], (1, 2, H // 4, W // 4, 128))[0].permute(0, 3, 1, 2), s[2])
    assert net.deconv2[2].last_packed is not None                                               # packed only: head 2 + deconv1's fused kernels read it
    # the full-resolution decoder stage: packed only (its prediction head reads the packed form); the model returns its last step unpacked
    pk1 = net.deconv1[2].last_packed
    assert pk1 is not None and pk1.dtype == torch.int32
    T, B = 5, 2
    last = fused.unpack_dense(pk1[T - 1:T], (1, B, H, W, 32))[0].permute(0, 3, 1, 2)
    assert tuple(s[-1].shape) == (B, 32, H, W) and torch.equal(s[-1], last) and 0.02 < float(last.mean()) < 2.0
    assert rec['anchor']                                                                        # bottom's dense output is a data-less anchor
    packed_only = fused.unpack_dense(net.bottom[2].last_packed, rec['shape'])
    # with conv1's forward back on MIOpen (which reads dense activations) bottom writes BOTH forms in one launch: unpack(packed) == dense
    with net.configured(SPIKE_CONV_FWD_MFMA=False):
        functional.reset_net(net)
        with torch.no_grad():
            net.forward_sequence(x)
        assert not rec['anchor'] and torch.equal(fused.unpack_dense(net.bottom[2].last_packed, rec['shape']), rec['bottom_out'])
        assert torch.equal(rec['bottom_out'], packed_only)
        assert net.plan()['conv1']['synapse_fwd'] == 'miopen'
    assert 0.02 < float(rec['bottom_out'].mean()) < 0.9
    with net.configured(PACK_SPIKES=False):
        functional.reset_net(net)
        with torch.no_grad():
            net.forward_sequence(x)
        assert net.conv3[2].last_packed is None and net.bottom[2].last_packed is None
    assert net.config.PACK_SPIKES and net.config.SPIKE_CONV_FWD_MFMA          # the context managers restored the network's own configuration


@pytest.mark.parametrize('name', ['StereoSpike', 'PLIFNet'])
def test_low_rank_head_gradients_are_in_effect_and_equal_the_dense_form(name):
    """fused.LOWRANK_HEAD_GRAD (default on; the pinned tests above assert its launch tags): the four prediction heads hand their input gradient to the
    stage's neuron backward as the rank-9 pair (g_P, W2) — the full-resolution pair travels on through the fused skip add into the first
    encoder layer's backward — instead of a GEMM + a C-channel tensor.  Here: the low-rank launches really happen (4 + 1 per backward), and
    every parameter gradient equals the dense form's up to the rounding of a 9-term sum done in another order."""
    from stereospike_amd import fused
    from stereospike_amd.clock_driven import functional
    from stereospike_amd.network.loss import Total_Loss
    H, W = 64, 80
    _, net = pair(name, H, W)
    x = synth_input(2, 5, 4, 7, H, W, lam=0.08).to(DEV)
    gt = synth_label(2, 8, H, W).to(DEV)

    def grads(on):
        fused.TIMER.clear()
        fused.TIMER.enabled = True
        try:
            with net.configured(LOWRANK_HEAD_GRAD=on):
                functional.reset_net(net)
                for p in net.parameters():
                    p.grad = None
                out = net.forward_sequence(x)
                d, s = out if isinstance(out, tuple) else (out, None)
                L = Total_Loss()(d, gt, s)
            L.backward()                  # outside the context on purpose: the backward dispatches from the configuration its forward captured
            torch.cuda.synchronize()
            tags = {k: v['launches'] for k, v in fused.TIMER.summary().items() if k.startswith('neuron_bwd')}
        finally:
            fused.TIMER.enabled = False
        return {n: p.grad.detach().double().clone() for n, p in net.named_parameters()}, tags
    g_on, t_on = grads(True)
    g_off, t_off = grads(False)
    if name == 'StereoSpike':      # under anomaly detection (it scans backward outputs for NaNs) the heads fall back to the dense form
        with torch.autograd.detect_anomaly(check_nan=True):
            _, t_an = grads(True)
        assert not any('lr' in k for k in t_an), t_an
    assert t_on.get('neuron_bwd+lronly', 0) == 1 and t_on.get('neuron_bwd+lr', 0) + t_on.get('neuron_bwd+lr+sum', 0) == 4, t_on
    assert not any('lr' in k for k in t_off), t_off
    for n in g_on:
        den = float(g_off[n].norm())
        # a PLIF layer's scalar dL/dw is ONE sum over all its neurons and steps with heavy cancellation: rounding differences of the
        # incoming gradient show up amplified there (measured 2.1e-5); the weight tensors agree to ~1e-6
        bar = 1e-4 if g_off[n].numel() == 1 else 2e-5
        assert float((g_on[n] - g_off[n]).norm()) <= bar * den + 1e-12, (n, float((g_on[n] - g_off[n]).norm()) / max(den, 1e-30))




def test_stage_plan_follows_the_geometry():
    """fused.stage_plan (round 5): which decoder stage takes the sub-pixel forward / the box-sum backward is decided ONCE per stage and
    input geometry from what the geometry admits and the measured thresholds — no channel-count tuples.  BASELINE's 260x346 pyramid: deconv1 .. deconv3 on the
    sub-pixel forward, deconv1 / deconv2 on the box-sum backward, deconv4 (17x22 source map, 512 channels) on GEMM + gather and the g_P forms; the parity suite's
    64x80 pyramid: deconv3's 8x10 source map drops out of the sub-pixel forward, deconv1's 32 source columns reach more horizontal ranges than the box kernels'
    on-chip window holds; a prediction head (k = 3) never takes either."""
    from stereospike_amd import fused
    from stereospike_amd.network import SNN_models as S
    dev = torch.device(DEV)

    def plans(size):
        net = S.StereoSpike(multiply_factor=10., input_size=size).to(dev)
        sz = S._pyramid(size)
        out = {}
        for lvl in (4, 3, 2, 1):
            up = getattr(net, f'deconv{lvl}')[0]
            h, w = sz[lvl]
            conv = up.up[1]
            out[lvl] = fused.stage_plan(up._tables(h, w, dev), conv.in_channels, conv.out_channels, 5, h, w, *sz[lvl - 1])
            assert fused.stage_plan(up._tables(h, w, dev), conv.in_channels, conv.out_channels, 5, h, w, *sz[lvl - 1]) is out[lvl]        # decided once, cached
            assert fused.stage_takes_packed_copy(up, h, w, dev) == out[lvl]['sub_fwd']
        head = net.predict_depth1[0]
        assert fused.stage_plan(head._tables(size[0], size[1], dev), 32, 1, 3, size[0], size[1], *size) == dict(sub_fwd=False, box_bwd=False)
        return out
    full = plans((260, 346))
    assert {l: p['sub_fwd'] for l, p in full.items()} == {4: False, 3: True, 2: True, 1: True}, full
    assert {l: p['box_bwd'] for l, p in full.items()} == {4: False, 3: False, 2: True, 1: True}, full
    small = plans((64, 80))
    assert {l: p['sub_fwd'] for l, p in small.items()} == {4: False, 3: False, 2: True, 1: True}, small
    assert {l: p['box_bwd'] for l, p in small.items()} == {4: False, 3: False, 2: True, 1: False}, small
