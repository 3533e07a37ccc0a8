"""CPU suite, part 3: the host-side mirror of the reference interface (names, signatures, state_dict keys, state
protocol, drop-in import aliases) and the torch-level loss / metric (they contain no neuron arithmetic and run on
any device)."""
import inspect
import json

import numpy as np
import pytest
import torch

from _util import load_npz, ref_network as rn, sj


def test_state_dict_keys_match_the_reference():
    from stereospike_amd.network import SNN_models as S, ANN_models as A
    pairs = [
        (S.StereoSpike(multiply_factor=10.), rn.build('StereoSpike', multiply_factor=10.), 'stereospike_T1'),
        (S.fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike(tau=3., use_plif=True, multiply_factor=30.),
         rn.build('PLIFNet', tau=3., use_plif=True, multiply_factor=30.), 'plif_T1'),
        (S.fromZero_feedforward_multiscale_tempo_monocular_SpikeFlowNetLike(tau=3., use_plif=True, multiply_factor=30.),
         rn.build('PLIFNetMono', tau=3., use_plif=True, multiply_factor=30.), 'mono_plif_T1'),
        (A.StereoSpike_equivalentANN(), rn.build('ANN'), 'ann_T1'),
    ]
    for prod, orc, tag in pairs:
        ks = list(prod.state_dict().keys())
        assert ks == list(orc.state_dict().keys())
        z = load_npz(f'model_{tag}.npz')
        names = json.loads(str(z['grad_names']))           # named_parameters() of the REAL reference class
        assert [k for k, _ in prod.named_parameters()] == names
        for (k, a), (_, b) in zip(prod.state_dict().items(), orc.state_dict().items()):
            assert a.shape == b.shape, k
    assert pairs[0][0].count_trainable_params() == 18148708


def test_same_seed_gives_the_reference_default_init():
    """Module construction order matches the reference's, so torch.manual_seed(2021) reproduces its weights."""
    import hashlib
    from stereospike_amd.clock_driven import surrogate
    from stereospike_amd.network.SNN_models import StereoSpike
    z = load_npz('model_stereospike_T1.npz')
    torch.manual_seed(int(z['seed']))
    net = StereoSpike(surrogate_function=surrogate.ATan(), multiply_factor=10.)
    h = hashlib.sha256()
    for k, v in net.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    assert h.hexdigest() == str(z['state_sha'])


def test_constructor_signatures():
    from stereospike_amd.network import SNN_models as S, ANN_models as A, blocks as B
    from stereospike_amd.clock_driven import neuron, surrogate

    def names(f):
        return list(inspect.signature(f).parameters)[1:]
    assert names(S.StereoSpike.__init__)[:5] == ['surrogate_function', 'detach_reset', 'v_threshold', 'v_reset', 'multiply_factor']
    assert names(S.fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike.__init__)[:6] == \
        ['use_plif', 'detach_reset', 'tau', 'v_threshold', 'v_reset', 'multiply_factor']
    assert names(S.fromZero_feedforward_multiscale_tempo_monocular_SpikeFlowNetLike.__init__)[:7] == \
        ['use_plif', 'detach_reset', 'tau', 'v_threshold', 'v_reset', 'final_activation', 'multiply_factor']
    assert names(A.StereoSpike_equivalentANN.__init__)[:1] == ['activation_function']
    assert A.SteroSpike_equivalentANN is A.StereoSpike_equivalentANN        # the reference's mis-spelt import
    assert names(B.MultiplyBy.__init__) == ['scale_value', 'learnable']
    assert names(B.NNConvUpsampling.__init__) == ['in_channels', 'out_channels', 'kernel_size', 'up_size', 'bias']
    assert names(B.SEWResBlock.__init__) == ['in_channels', 'connect_function', 'v_threshold', 'v_reset',
                                             'surrogate_function', 'use_plif', 'tau', 'multiply_factor']
    assert names(neuron.IFNode.__init__) == ['v_threshold', 'v_reset', 'surrogate_function', 'detach_reset']
    assert names(neuron.LIFNode.__init__)[0] == 'tau' and names(neuron.ParametricLIFNode.__init__)[0] == 'init_tau'
    assert surrogate.ATan().alpha == 2.0 and surrogate.Sigmoid().alpha == 4.0
    blk = B.SEWResBlock(8, use_plif=True, tau=3., multiply_factor=10.)
    assert all(hasattr(blk, a) for a in ('conv1', 'sn1', 'conv2', 'sn2', 'connect_function'))
    assert abs(float(blk.sn1.w) + np.log(2.0)) < 1e-7
    up = B.NNConvUpsampling(4, 2, 5, (33, 44))
    assert up.up[0].size == (37, 48) and up.up[1].kernel_size == (5, 5) and up.up[1].bias is None


def test_state_protocol_and_quirks():
    from stereospike_amd.clock_driven import functional, neuron, surrogate, layer
    from stereospike_amd.network.SNN_models import StereoSpike, NeuromorphicNet
    net = StereoSpike(surrogate_function=surrogate.ATan(), v_threshold=0.5, v_reset=0.3, multiply_factor=10.)
    nodes = [m for m in net.modules() if isinstance(m, neuron.BaseNode)]
    assert len(nodes) == 14 and sum(isinstance(m, neuron.IFNode) for m in nodes) == 14
    # quirk kept (SNN_models.py:71-72): v_threshold / v_reset arguments are swallowed
    assert all(m.v_threshold == 1.0 and m.v_reset == 0.0 for m in nodes if m is not net.Ineurons)
    assert net.Ineurons.v_threshold == float('inf')
    # quirk kept (:105-106): bottleneck surrogate stays Sigmoid
    assert isinstance(net.bottleneck[0].sn1.surrogate_function, surrogate.Sigmoid)
    assert isinstance(net.bottom[2].surrogate_function, surrogate.ATan)
    assert all(m.detach_reset for m in nodes if m is not net.Ineurons) and not net.Ineurons.detach_reset
    assert net.max_test_accuracy == float('inf') and net.epoch == 0
    net.increment_epoch(); net.update_max_accuracy(0.5)
    assert net.epoch == 1 and net.get_max_accuracy() == 0.5
    state = net.get_network_state()
    assert len(state) == 14 and all(v == 0.0 for v in state)
    net.change_network_state([1.0] * 14)
    assert net.bottom[2].v == 1.0
    functional.reset_net(net)
    assert net.bottom[2].v == 0.0
    net.set_init_depths_potentials(torch.ones(1, 1, 260, 346))
    assert torch.is_tensor(net.Ineurons.v)
    net.detach()          # floats and tensors both fine
    assert isinstance(layer.Dropout(0.5), torch.nn.Module)
    assert issubclass(StereoSpike, NeuromorphicNet)


def test_install_dropin_aliases():
    import sys
    import stereospike_amd
    stereospike_amd.install_dropin()
    from spikingjelly.clock_driven import functional, surrogate, neuron, layer, rnn  # noqa: F401  (blocks.py:8)
    from network.SNN_models import StereoSpike, fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike  # noqa: F401
    from network.ANN_models import StereoSpike_equivalentANN, SteroSpike_equivalentANN  # noqa: F401
    from network.metrics import MeanDepthError, log_to_lin_depths, disparity_to_depth  # noqa: F401
    from network.loss import Total_Loss  # noqa: F401
    from network.blocks import SEWResBlock, NNConvUpsampling, MultiplyBy  # noqa: F401
    assert sys.modules['network.SNN_models'].StereoSpike is StereoSpike
    assert neuron.IFNode is stereospike_amd.clock_driven.neuron.IFNode


def test_product_loss_and_mde_vs_pure_reference_fixture():
    """Total_Loss / MeanDepthError (pure torch, no neuron arithmetic) vs what /root/reference/network/loss.py and
    metrics.py computed.  Tolerance 2e-6 relative: same sums, evaluated without the reference's masked gathers."""
    from stereospike_amd.network.loss import Total_Loss
    from stereospike_amd.network.metrics import MeanDepthError
    z = load_npz('loss_metric.npz')
    for ci in range(int(z['n_cases'])):
        preds = [torch.tensor(z[f'l{ci}_pred{i}'], requires_grad=True) for i in range(4)]
        gt = torch.tensor(z[f'l{ci}_gt'])
        spikes = [torch.tensor(z[f'l{ci}_spk{i}'].astype(np.float32)) for i in range(5)]
        for pen in (False, True):
            tag = f'l{ci}_{"pen" if pen else "nopen"}_'
            L = Total_Loss(alpha=0.5, penalize_spikes=pen, beta=0.5)(preds, gt, spikes)
            ref = float(z[tag + 'loss'])
            assert abs(float(L) - ref) <= 2e-6 * abs(ref), (ci, pen, float(L), ref)
            grads = torch.autograd.grad(L, preds)
            for i, g in enumerate(grads):
                r = z[tag + f'gpred{i}']
                if r.ndim == 0:
                    assert abs(float(g.double().abs().sum()) - float(r)) <= 1e-5 * float(r)
                else:
                    assert np.allclose(g.numpy(), r, rtol=1e-4, atol=1e-7 * np.abs(r).max() + 1e-12)
        mde = MeanDepthError(preds[0].detach(), gt)
        assert abs(float(mde) - float(z[f'l{ci}_mde'])) <= 2e-6 * float(z[f'l{ci}_mde'])
        assert not torch.isnan(L)


def test_blocks_upsampling_vs_reference_fixture():
    """NNConvUpsampling / MultiplyBy are plain torch modules: bit-exact against blocks.npz (reference blocks.py)."""
    from stereospike_amd.network.blocks import NNConvUpsampling, MultiplyBy
    z = load_npz('blocks.npz')
    for i in range(4):
        cfg = json.loads(str(z[f'up{i}_cfg']))
        m = NNConvUpsampling(cfg['cin'], cfg['cout'], cfg['k'], tuple(cfg['up']), bias=(cfg['k'] == 3))
        m.load_state_dict({k[len(f'up{i}_w_'):]: torch.tensor(z[k]) for k in z.files if k.startswith(f'up{i}_w_')})
        y = m(torch.tensor(z[f'up{i}_x']))
        assert tuple(y.shape[-2:]) == tuple(cfg['up'])
        assert np.allclose(y.detach().numpy(), z[f'up{i}_y'], rtol=0, atol=1e-6)
    assert np.array_equal(MultiplyBy(10.)(torch.tensor(z['mul_x'])).numpy(), z['mul_y'])


def test_pyramid_sizes():
    from stereospike_amd.network.SNN_models import _pyramid
    assert _pyramid((260, 346)) == [(260, 346), (130, 173), (65, 87), (33, 44), (17, 22)]


def test_gemm_tuning_record_is_wellformed_and_loader_is_inert_without_a_gpu():
    """The tracked TunableOp record: validator header for the image's library versions + one line per GEMM shape; the loader does
    nothing (returns None) when there is no HIP device."""
    import csv
    from stereospike_amd import gemm_tuning
    rows = list(csv.reader(open(gemm_tuning.SEED)))
    validators = {r[1]: r[2] for r in rows if r[0] == 'Validator'}
    assert {'PT_VERSION', 'HIPBLASLT_VERSION', 'ROCBLAS_VERSION', 'GCN_ARCH_NAME'} <= set(validators)
    assert validators['GCN_ARCH_NAME'].startswith('gfx950')
    ops = [r for r in rows if r[0] != 'Validator']
    assert len(ops) >= 25 and all(len(r) == 4 and r[0].startswith('Gemm') and float(r[3]) > 0 for r in ops)
    if not torch.cuda.is_available():
        assert gemm_tuning.enable(0) is None


def test_graph_helpers_refuse_cpu_loudly():
    """engine.GraphedInference / GraphedTrainer are HIP-graph features: on CPU tensors they fail with a clear message, never fall back."""
    from stereospike_amd.engine import GraphedInference, GraphedTrainer
    net = torch.nn.Linear(2, 2)
    with pytest.raises(AssertionError, match='MI355X'):
        GraphedInference(net, torch.zeros(1, 1, 4, 8, 8))
    with pytest.raises(AssertionError, match='MI355X'):
        GraphedTrainer(net)


def test_config1_equivalent_ann_runs_on_cpu_and_equals_the_oracle():
    """BASELINE.json configs[0]: StereoSpike_equivalentANN, monocular-sized plumbing case, batch 1, 64x64 synthetic voxels, CPU
    PyTorch.  The analog twin has no spiking state: on CPU tensors the product model is plain torch ops in the reference's order and
    must equal the oracle network (itself pinned to the reference's ANN_models.py by model_ann_T1.npz) bit for bit, forward and backward."""
    import torch
    from _util import ref_network as rn, sj, synth_input, synth_label
    from stereospike_amd.clock_driven import functional
    from stereospike_amd.network.ANN_models import StereoSpike_equivalentANN
    from stereospike_amd.network.loss import Total_Loss
    torch.manual_seed(2021)
    orc = rn.build('ANN', input_size=(64, 64))
    net = StereoSpike_equivalentANN(input_size=(64, 64))
    net.load_state_dict(orc.state_dict())
    x, gt = synth_input(1, 1, 4, 3, 64, 64, lam=0.1), synth_label(1, 4, 64, 64)
    sj.reset_net(orc)
    functional.reset_net(net)
    d0, d1 = orc(x), net(x)
    assert all(torch.equal(a, b) for a, b in zip(d0, d1))
    rn.total_loss(d0, gt).backward()
    Total_Loss()(d1, gt, None).backward()
    for (k, p), (_, q) in zip(net.named_parameters(), orc.named_parameters()):
        # the product's Total_Loss evaluates the same sums without boolean-index gathers (another fp32 order)
        assert float((p.grad - q.grad).norm() / (q.grad.norm() + 1e-30)) <= 1e-4, k
    # stateful second call (membrane of the read-out pool carried) and reset
    d0b, d1b = orc(x), net(x)
    assert all(torch.equal(a, b) for a, b in zip(d0b, d1b)) and not torch.equal(d1b[0], d1[0])


def test_lowrank_pair_travels_in_a_self_describing_buffer():
    """fused.lowrank_buffer / lowrank_of: the pair is found through any view of the anchor (no module-level registry: it lives exactly as
    long as the gradient tensor does); plain tensors and ordinary expanded scalars are not mistaken for anchors; an unknowing consumer sees
    NaNs; an anchor that reaches a layer of another shape RAISES; two interleaved producers do not disturb each other."""
    import gc
    import weakref
    from stereospike_amd import _lib, fused
    assert not hasattr(fused, '_LOWRANK')
    p, w = torch.randn(6, 9), torch.randn(9, 8)
    a = fused.lowrank_anchor((2, 3, 8), p, w)
    b, pb, wb = fused.lowrank_buffer((4, 5, 16), 'cpu')                 # a second, interleaved producer (another head / another network)
    pb.copy_(torch.randn(20, 9)), wb.copy_(torch.randn(9, 16))
    assert a.shape == (2, 3, 8) and not any(a.stride()) and bool(torch.isnan(a.contiguous()).all())
    for v in (a, a.view(6, 8), a.view(2, 3, 8).view_as(a), a.flatten(0, 1), a.reshape(1, 6, 8)):
        got = fused.lowrank_of(v)
        assert got is not None and torch.equal(got[0], p) and torch.equal(got[1], w)
    gb = fused.lowrank_of(b.flatten(0, 1))
    assert torch.equal(gb[0], pb) and torch.equal(gb[1], wb) and gb[0].data_ptr() == pb.data_ptr()      # views of the buffer, no copies
    assert fused.lowrank_of(torch.zeros(2, 3, 8)) is None and fused.lowrank_of(None) is None
    assert fused.lowrank_of(torch.zeros(1).expand(2, 3, 8)) is None            # an ordinary expanded scalar gradient
    assert fused.lowrank_of(torch.zeros(())) is None
    assert torch.equal(fused.lowrank_dense((p, w), (2, 3, 8)), (p @ w).view(2, 3, 8))
    with pytest.raises(_lib.SSNeuronError):
        fused.lowrank_of(a.view(6, 8)[:, :4])                                   # same storage, a shape the pair was not made for
    # a LEGITIMATE zero-stride gradient over a longer storage — d/dterms[0] of (stack(terms) * w).sum() is an expanded view of element 0
    # of a len(terms)-float storage — is not a pair and must come back as "dense", not raise
    terms = torch.arange(1.0, 8.0)
    legit = terms[0].expand(2, 3, 8)
    assert not any(legit.stride()) and legit.untyped_storage().nbytes() // 4 == 7 and fused.lowrank_of(legit) is None
    # lifetime = the gradient's lifetime: dropping the anchor frees the pair (nothing else holds it)
    probe = weakref.ref(a)
    del a, got, v
    gc.collect()
    assert probe() is None
