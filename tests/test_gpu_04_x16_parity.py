"""16-bit activation modes (BASELINE configs 2 / 5: bf16 / fp16 activations in HBM, fp32 membranes and neuron arithmetic) — END-TO-END parity of
the x16 kernels' default path against an oracle that narrows at the same points.

The reference is fp32-only; the modes are a build-side addition whose semantics are defined per kernel by oracle/np_x16.py (bit-exact tests:
tests/test_gpu_01_kernels.py) and end to end by oracle/ref_network.py::float64_convs(narrow=...) + tests/_pinned.py::narrowing_points: every
synapse is evaluated in float64 from the weights the mode uses (rounded once to 16 bits where the mode does) and its result rounded once to
the storage format.  Same trajectory-pinned protocol as tests/test_gpu_00_default_path.py.

Bars — derived from the storage format's unit roundoff u = 2^-8 (bf16) / 2^-11 (fp16), not from a run:
  * forward: the narrowed synapse outputs of product (fp32 accumulation, rounded) and oracle (float64, rounded) differ by at most one
    16-bit ulp on rare elements, so a disagreeing neuron sits within gain x ulp of its threshold: margin <= 32 u; still <= 1e-4 of a layer;
    depths, loss, MDE are fp32 quantities of identical spike trains: <= 1e-5 relative, as in the fp32 mode;
  * backward: every activation gradient is stored with relative error u per layer crossing and every weight-gradient element is rounded to
    16 bits once (autocast's convolution backward), errors of random sign: weight tensors <= 8 u relative L2 (13 layers deep: sqrt(13) u
    expected), a PLIF w gradient within u of its magnitude sum.
"""
import pytest
import torch

from _models import pair
from _pinned import pinned_parity
from _util import synth_input, synth_label
from test_gpu_00_default_path import check

pytestmark = pytest.mark.gpu
U = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def assert_x16_kernels(tags):
    """13 fused x16 neuron launches each way on the compile-time-T recompute forms (nothing on the saved-h path), forked gradients fused."""
    fwd = {k: v for k, v in tags.items() if k.startswith('neuron_fwd')}
    bwd = {k: v for k, v in tags.items() if k.startswith('neuron_bwd')}
    assert sum(fwd.values()) == 13 and sum(bwd.values()) == 13, tags
    assert not any(k.endswith('+h') for k in fwd) and not any('savedh' in k for k in bwd), ('saved-h (run-time-T) kernels ran', tags)
    assert bwd.get('neuron_bwd+fork', 0) + bwd.get('neuron_bwd+fork+sum', 0) >= 6, tags        # conv1..3 + deconv4..2 (dense head gradients)


@pytest.mark.parametrize('T', [5, 10])
@pytest.mark.parametrize('name', ['StereoSpike', 'PLIFNet'])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_pinned_parity_16bit_activations(dt, name, T):
    H, W = 64, 80
    orc, net = pair(name, H, W)
    x = synth_input(2, T, 4, 81, H, W, lam=0.08)
    gt = synth_label(2, 82, H, W)
    rep = pinned_parity(orc, net, x, gt, amp_dtype=dt, loss_scale=2.0 ** 12 if dt == torch.float16 else 1.0)
    assert_x16_kernels(rep['launch_tags'])
    u = U[dt]
    check(f'pinned_x16_{"bf16" if dt == torch.bfloat16 else "f16"}_T{T}_{name}', rep, margin=32 * u, tensor_bar=8 * u, plif_bar=u)


def test_pinned_parity_config2_monocular_plif_T1_bf16():
    """BASELINE config 2's network and mode: monocular PLIF, T = 1, bf16 activations, B = 8."""
    H, W = 64, 80
    orc, net = pair('PLIFNetMono', H, W)
    x = synth_input(8, 1, 2, 83, H, W, lam=0.12)
    gt = synth_label(8, 84, H, W)
    rep = pinned_parity(orc, net, x, gt, returns_spikes=False, amp_dtype=torch.bfloat16)
    assert_x16_kernels(rep['launch_tags'])
    u = U[torch.bfloat16]
    check('pinned_x16_bf16_T1_PLIFNetMono_B8', rep, margin=32 * u, tensor_bar=8 * u, plif_bar=u)
