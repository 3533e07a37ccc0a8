"""16-bit activation modes (BASELINE configs 2 / 5: bf16 / fp16 activations in HBM, fp32 membranes and neuron arithmetic) — END-TO-END parity of
the x16 kernels' default path against an oracle that narrows at the same points.

The reference is fp32-only; the modes are a build-side addition whose semantics are defined per kernel by oracle/np_x16.py (bit-exact tests:
tests/test_gpu_01_kernels.py) and end to end by oracle/ref_network.py::float64_convs(narrow=...) + tests/_pinned.py::narrowing_points: every
synapse is evaluated in float64 from the weights the mode uses (rounded once to 16 bits where the mode does) and its result rounded once to
the storage format.  Same trajectory-pinned protocol as tests/test_gpu_00_default_path.py.

Bars — derived from the storage format's unit roundoff u = 2^-8 (bf16) / 2^-11 (fp16), not from a run:
  * forward: the product's 16-bit synapses are MIOpen's convolutions under torch.autocast (fp32 accumulation in an order, and with
    intermediate roundings, of the solver's choosing) and the build's own x16 up-conv kernels; what they store differs from the once-rounded
    float64 value by at most one 16-bit ulp on a minority of the elements.  Such a difference flips a neuron only if its membrane sits
    within gain x ulp(x) of the threshold: a disagreeing neuron must sit within 64 u of its threshold (gain <= 30, |x| <= 2), and with
    the membranes spread over a range of order one the disagreeing fraction of a layer is bounded by u itself (measured 4.5e-4 bf16 /
    1.6e-4 fp16: profiles/r03/parity_report_x16.json).  Depths, loss, MDE are fp32 quantities of identical (pinned) spike trains:
    <= 1e-5 relative, as in the fp32 mode;
  * backward — the SAME bars for both modes, u_b = 2^-8: every activation gradient is stored in 16 bits per layer crossing, every
    weight-gradient element is rounded to 16 bits once (autocast's convolution backward), the decoder's backward GEMMs take bf16 operands
    in BOTH modes (bf16 has the fp32 exponent range), and in the fp16 mode the activation gradients live between fp16's underflow and
    overflow thresholds — their dynamic range across the 13 layers (~2^20 at gain 10 - 30) exceeds what any loss scale can centre, so the
    fp16 backward is range-limited, not ulp-limited (measured 1.1e-2, the bf16 mode 1.9e-2): weight tensors <= 8 u_b relative L2
    (13 layers deep: sqrt(13) u_b expected), a PLIF w gradient within u_b of its magnitude sum (measured 1.1e-4).
fp16 activation gradients need a loss scale (engine.Trainer: torch.amp.GradScaler); here the largest power of four whose gradients are
finite is used — what the scaler converges to.
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
    """13 fused x16 neuron launches each way on the compile-time-T recompute forms (nothing on the saved-h path), forked gradients fused."""
    fwd = {k: v for k, v in tags.items() if k.startswith('neuron_fwd')}
    bwd = {k: v for k, v in tags.items() if k.startswith('neuron_bwd')}
    assert sum(fwd.values()) == 13 and sum(bwd.values()) == 13, tags
    assert not any(k.endswith('+h') for k in fwd) and not any('savedh' in k for k in bwd), ('saved-h (run-time-T) kernels ran', tags)
    assert bwd.get('neuron_bwd+fork', 0) + bwd.get('neuron_bwd+fork+sum', 0) >= 6, tags        # conv1..3 + deconv4..2 (dense head gradients)


# T = 10 costs the float64 oracle 20 s per case: config 5's own combination (fp16 activations, StereoSpike) and the PLIF model in bf16; T = 5: all four
@pytest.mark.parametrize('dt,name,T', [(torch.bfloat16, 'StereoSpike', 5), (torch.bfloat16, 'PLIFNet', 5), (torch.float16, 'StereoSpike', 5), (torch.float16, 'PLIFNet', 5),
                                       (torch.float16, 'StereoSpike', 10), (torch.bfloat16, 'PLIFNet', 10)],
                         ids=['bf16-StereoSpike-5', 'bf16-PLIFNet-5', 'f16-StereoSpike-5', 'f16-PLIFNet-5', 'f16-StereoSpike-10', 'bf16-PLIFNet-10'])
def test_pinned_parity_16bit_activations(dt, name, T):
    H, W = 64, 80
    orc, net = pair(name, H, W)
    x = synth_input(2, T, 4, 81, H, W, lam=0.08)
    gt = synth_label(2, 82, H, W)
    scale = _largest_finite_loss_scale(net, x, gt) if dt == torch.float16 else 1.0
    for _ in range(3):
        rep = pinned_parity(orc, net, x, gt, amp_dtype=dt, loss_scale=scale)
        if all(math.isfinite(v) for v in rep['grad_rel_l2'].values()):
            break
        # at the largest finite scale an fp16 gradient sits next to 65504: MIOpen's atomic split-K order can push one over in the next run.
        # GradScaler skips such a step and backs off; so does the test
        scale /= 4.0
    rep['loss_scale'] = scale
    assert_x16_kernels(rep['launch_tags'])
    u = U[dt]
    check(f'pinned_x16_{"bf16" if dt == torch.bfloat16 else "f16"}_T{T}_{name}', rep, margin=64 * u, tensor_bar=8 * U_BWD, plif_bar=U_BWD, flip_frac=u)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_pinned_parity_16bit_activations_on_the_fp32_kernels(dt):
    """EngineConfig.X16_OWN_CONVS (round 4; off by default — it costs a third of the mode's rate, profiles/r04/x16_own_convs_ab.md): the
    encoder / bottleneck synapses of the 16-bit modes on the fp32 mode's own kernels — fp32 master weights, exact products, fp32 weight gradients; only what
    is stored between layers is narrowed.  The oracle narrows accordingly (narrowing_points(own_convs=True)).  What is left in the backward is the 16-bit
    storage of the activation gradients: every weight tensor within 1e-2 (2.6 u_b) relative L2 of the oracle (measured 7.9e-3 bf16 / 3.7e-3 fp16; the
    autocast path: 1.7e-2 / 1.1e-2), and the forward disagrees on <= 4 neurons per layer (measured: 1 in the whole network; the autocast path 949 / 234)."""
    H, W = 64, 80
    orc, net = pair('StereoSpike', H, W)
    x = synth_input(2, 5, 4, 81, H, W, lam=0.08)
    gt = synth_label(2, 82, H, W)
    with net.configured(X16_OWN_CONVS=True):
        scale = _largest_finite_loss_scale(net, x, gt) if dt == torch.float16 else 1.0
        rep = pinned_parity(orc, net, x, gt, amp_dtype=dt, loss_scale=scale, x16_own_convs=True)
        plan = net.plan()
    rep['loss_scale'] = scale
    assert_x16_kernels(rep['launch_tags'])
    assert plan['conv1']['synapse_fwd'].startswith('spike_conv_fwd3_mfma') and plan['conv3']['synapse_fwd'].endswith('exact_bf16x3_gemm') \
        and plan['bottom']['synapse_fwd'] == 'dense_conv_s1_fwd6_mfma', plan
    u = U[dt]
    check(f'pinned_x16own_{"bf16" if dt == torch.bfloat16 else "f16"}_T5_StereoSpike', rep, margin=64 * u, tensor_bar=1e-2, plif_bar=U_BWD, flip_frac=u)
    assert sum(v['flips'] for v in rep['layers'].values()) <= 4 * len(rep['layers']), rep['layers']


def test_pinned_parity_config2_monocular_plif_T1_bf16():
    """BASELINE config 2's network and mode: monocular PLIF, T = 1, bf16 activations, B = 8."""
    H, W = 64, 80
    orc, net = pair('PLIFNetMono', H, W)
    x = synth_input(8, 1, 2, 83, H, W, lam=0.12)
    gt = synth_label(8, 84, H, W)
    rep = pinned_parity(orc, net, x, gt, returns_spikes=False, amp_dtype=torch.bfloat16)
    assert_x16_kernels(rep['launch_tags'])
    u = U[torch.bfloat16]
    check('pinned_x16_bf16_T1_PLIFNetMono_B8', rep, margin=64 * u, tensor_bar=8 * U_BWD, plif_bar=U_BWD, flip_frac=u)
