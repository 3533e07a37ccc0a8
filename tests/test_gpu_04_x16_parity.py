"""16-bit activation modes (BASELINE configs 2 / 5: bf16 / fp16 activations in HBM, fp32 membranes and neuron arithmetic) — END-TO-END parity of
the shipped path: every synapse on the engine's OWN single-term 16-bit-I/O kernels (round 5: EngineConfig.X16_OWN_KERNELS, include/ss_neuron.h ABI 9),
2-bit packed spikes between layers, against an oracle that narrows at the same points.

The reference is fp32-only; the modes are a build-side addition whose semantics are defined per kernel by oracle/np_x16.py and the float64 references of
tests/test_gpu_06_x16_kernels.py (bit-exact / narrowing-bound tests) and end to end by oracle/ref_network.py::float64_convs(narrow=...) +
tests/_pinned.py::narrowing_points: every synapse is evaluated in float64 from the weights the mode uses (rounded once to 16 bits where the mode does) and
its result rounded once to the storage format.  Same trajectory-pinned protocol as tests/test_gpu_00_default_path.py.

Bars — the forward of the own kernels is "exact products of the operands as stored, fp32 accumulation, one narrowing": it differs from the
once-rounded float64 value only where the fp32 sum lands on the other side of a 16-bit rounding boundary, and that flips a neuron only if its membrane
then sits within one 16-bit ulp of the threshold:
  * forward: <= 4 disagreeing neurons per layer (the MIOpen-under-autocast path of rounds 2 - 4: 949 / 234 in the network), every one of them within
    64 u of its threshold (u = 2^-8 bf16 / 2^-11 fp16); depths, loss, MDE <= 1e-5 relative as in the fp32 mode;
  * backward: every weight TENSOR within 1e-2 relative L2 of the oracle's autograd — what is left is the 16-bit storage of the activation gradients at
    every layer crossing (13 layers deep, two roundings each: ~sqrt(26) u / sqrt(3) = 5.7e-3 expected in bf16), the weight gradients themselves are fp32
    accumulations of exact products; a PLIF w gradient within u_b = 2^-8 of its magnitude sum.
fp16 activation gradients need a loss scale (engine.Trainer: torch.amp.GradScaler); here the largest power of four whose gradients are finite is used —
what the scaler converges to.  The legacy path (X16_OWN_KERNELS off) keeps one case with its own, wider, bars.
"""
import math

import pytest
import torch

from _models import DEV, pair
from _pinned import pinned_parity
from _util import synth_input, synth_label
import test_gpu_00_default_path as _t00
from test_gpu_00_default_path import check

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _own_report(monkeypatch):
    monkeypatch.setattr(_t00, 'REPORT_FILE', 'gpurun_out/parity_report_x16.json')
    monkeypatch.setattr(_t00, 'REPORT', {} if not hasattr(_own_report, 'rep') else _own_report.rep)
    _own_report.rep = _t00.REPORT


U = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}
U_BWD = 2.0 ** -8          # the backward of BOTH modes: see the module docstring


def _largest_finite_loss_scale(net, x, gt):
    """What torch.amp.GradScaler converges to (engine.Trainer): the largest power-of-four loss scale whose fp16 backward stays finite."""
    from stereospike_amd.clock_driven import functional
    from stereospike_amd.network.loss import Total_Loss
    xg, gg = x.to(DEV), gt.to(DEV)
    for e in range(8, -9, -2):
        net.zero_grad()
        functional.reset_net(net)
        with torch.autocast('cuda', dtype=torch.float16):
            out = net.forward_sequence(xg)
            d, s = out if isinstance(out, tuple) else (out, None)
            L = Total_Loss()(d, gg, s)
        (L * 2.0 ** e).backward()
        if all(bool(torch.isfinite(p.grad).all()) for p in net.parameters()):
            return 2.0 ** e
    raise AssertionError('no finite fp16 backward down to a loss scale of 2^-8')


def assert_x16_kernels(tags):
    """13 fused x16 neuron launches each way on the compile-time-T recompute forms (nothing on the saved-h path), forked gradients fused (dense second gradient
    or the prediction head's rank-9 pair)."""
    fwd = {k: v for k, v in tags.items() if k.startswith('neuron_fwd')}
    bwd = {k: v for k, v in tags.items() if k.startswith('neuron_bwd')}
    assert sum(fwd.values()) == 13 and sum(bwd.values()) == 13, tags
    assert not any(k.endswith('+h') for k in fwd) and not any('savedh' in k for k in bwd), ('saved-h (run-time-T) kernels ran', tags)
    fused2 = sum(v for k, v in bwd.items() if any(t in k for t in ('+fork', '+lr')))
    assert fused2 >= 6, tags        # conv1..3 + deconv4..2 (+ bottom / deconv1 through the low-rank pair)


def assert_own_plan(plan, dt):
    """No MIOpen synapse anywhere: the plan of the 16-bit mode names the engine's own kernel forms on every layer."""
    for name, d in plan.items():
        for key in ('synapse_fwd', 'synapse_bwd'):
            assert 'miopen' not in d.get(key, '').lower() and 'torch module' not in d.get(key, ''), (name, d)
    assert plan['bottom']['synapse_fwd'] == 'dense_conv_s1_fwd1_mfma_x16' and plan['bottom']['synapse_bwd'].endswith('dense_conv_s1_wgrad1_mfma_x16'), plan['bottom']
    for name in ('conv1', 'conv2'):
        assert plan[name]['synapse_fwd'] == 'spike_conv_fwd1_mfma_x16(packed in)' and 'conv_s2_dgrad1_mfma_x16' in plan[name]['synapse_bwd'] \
            and 'spike_conv_wgrad1_mfma_x16' in plan[name]['synapse_bwd'], plan[name]
    for name in ('conv3', 'conv4', 'bottleneck.0.conv1', 'bottleneck.0.conv2', 'bottleneck.1.conv1', 'bottleneck.1.conv2'):
        assert plan[name]['synapse_fwd'] == 'im2col(packed in)+gemm1_x16' and 'gemm1_x16(fp32 out)' in plan[name]['synapse_bwd'], plan[name]
    for name in ('deconv2', 'deconv1'):               # (64x80 pyramid: deconv3's 8x10 source map stays on GEMM + gather, fused.stage_plan; at 260x346 it is on the sub-pixel forward too)
        assert plan[name]['synapse_fwd'].startswith('upconv_sub_mfma_x16'), plan[name]
    for name in ('predict_depth1', 'predict_depth2'):
        assert plan[name]['synapse_fwd'] == 'head_proj_packed_mfma+gather', plan[name]
    for name in plan:
        if 'neuron_fwd' in plan[name] and name not in ('bottleneck.1.conv2', 'deconv4'):
            assert '+x16' in plan[name]['neuron_fwd'] and ('+packed' in plan[name]['neuron_fwd'] or '+pkcopy' in plan[name]['neuron_fwd']), (name, plan[name])


# T = 10 costs the float64 oracle 20 s per case: config 5's own combination (fp16 activations, StereoSpike) and the PLIF model in bf16; T = 5: all four
@pytest.mark.parametrize('dt,name,T', [(torch.bfloat16, 'StereoSpike', 5), (torch.bfloat16, 'PLIFNet', 5), (torch.float16, 'StereoSpike', 5), (torch.float16, 'PLIFNet', 5),
                                       (torch.float16, 'StereoSpike', 10), (torch.bfloat16, 'PLIFNet', 10)],
                         ids=['bf16-StereoSpike-5', 'bf16-PLIFNet-5', 'f16-StereoSpike-5', 'f16-PLIFNet-5', 'f16-StereoSpike-10', 'bf16-PLIFNet-10'])
def test_pinned_parity_16bit_activations(dt, name, T):
    H, W = 64, 80
    orc, net = pair(name, H, W)
    x = synth_input(2, T, 4, 81, H, W, lam=0.08)
    gt = synth_label(2, 82, H, W)
    assert net.config.X16_OWN_KERNELS
    scale = _largest_finite_loss_scale(net, x, gt) if dt == torch.float16 else 1.0
    for _ in range(3):
        rep = pinned_parity(orc, net, x, gt, amp_dtype=dt, loss_scale=scale)
        if all(math.isfinite(v) for v in rep['grad_rel_l2'].values()):
            break
        scale /= 4.0               # (GradScaler skips a step whose gradients overflow and backs off; so does the test)
    rep['loss_scale'] = scale
    assert_x16_kernels(rep['launch_tags'])
    assert_own_plan(rep['plan'], dt)
    u = U[dt]
    check(f'pinned_x16_{"bf16" if dt == torch.bfloat16 else "f16"}_T{T}_{name}', rep, margin=64 * u, tensor_bar=1e-2, plif_bar=U_BWD, flip_frac=u)
    assert max(v['flips'] for v in rep['layers'].values()) <= 4, rep['layers']


def test_pinned_parity_16bit_activations_legacy_miopen_path():
    """EngineConfig.X16_OWN_KERNELS off: the round-2 .. 4 path (encoder / bottleneck synapses = MIOpen convolutions under autocast, dense 16-bit spike tensors),
    kept as the A/B reference (profiles/r05/) — its own, wider bars: MIOpen's 16-bit convolutions differ from the once-rounded float64 value by one 16-bit ulp on a
    minority of elements (a disagreeing fraction <= u per layer), their weight gradients are rounded to 16 bits (tensors <= 8 u_b)."""
    H, W = 64, 80
    dt = torch.bfloat16
    orc, net = pair('StereoSpike', H, W)
    x = synth_input(2, 5, 4, 81, H, W, lam=0.08)
    gt = synth_label(2, 82, H, W)
    with net.configured(X16_OWN_KERNELS=False):
        rep = pinned_parity(orc, net, x, gt, amp_dtype=dt, x16_own=False)
        plan = net.plan()
    assert_x16_kernels(rep['launch_tags'])
    assert plan['conv1']['synapse_fwd'] == 'miopen', plan['conv1']
    check('pinned_x16legacy_bf16_T5_StereoSpike', rep, margin=64 * U[dt], tensor_bar=8 * U_BWD, plif_bar=U_BWD, flip_frac=U[dt])


def test_pinned_parity_config2_monocular_plif_T1_bf16():
    """BASELINE config 2's network and mode: monocular PLIF, T = 1, bf16 activations, B = 8."""
    H, W = 64, 80
    orc, net = pair('PLIFNetMono', H, W)
    x = synth_input(8, 1, 2, 83, H, W, lam=0.12)
    gt = synth_label(8, 84, H, W)
    rep = pinned_parity(orc, net, x, gt, returns_spikes=False, amp_dtype=torch.bfloat16)
    assert_x16_kernels(rep['launch_tags'])
    assert_own_plan(rep['plan'], torch.bfloat16)
    u = U[torch.bfloat16]
    check('pinned_x16_bf16_T1_PLIFNetMono_B8', rep, margin=64 * u, tensor_bar=1e-2, plif_bar=U_BWD, flip_frac=u)
    assert max(v['flips'] for v in rep['layers'].values()) <= 4, rep['layers']
