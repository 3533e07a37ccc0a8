"""The spiking encoder-decoder models of the reference, same class names / constructor signatures / forward I/O /
state_dict keys (/root/reference/network/SNN_models.py):

    NeuromorphicNet                                                    :11-60
    StereoSpike                                                        :63-248
    fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike        :251-435
    fromZero_feedforward_multiscale_tempo_monocular_SpikeFlowNetLike   :438-622

Execution is layer-by-layer over a [T, B, C, H, W] sequence: each conv runs once on the [T*B] batch
(PyTorch-ROCm), each neuron layer is one fused HIP launch that walks the T steps with the membrane in registers,
the decoder skip add / SEW add are kernel epilogues, and the four predict_depth heads charge the shared I-neuron
pool in one launch that keeps the reference's (t outer, head 4->1 inner) fp32 summation order.  Because the
network is purely feed-forward this is mathematically the reference's step-by-step evaluation
(`reset_net(net); for t: net(x[:, t:t+1])`).

`net(x)` is the reference's single-step call (consumes x[:, 0], membranes carried across calls until
functional.reset_net).  `net.forward_sequence(x)` consumes all T frames of x [B, T, C, H, W] in one pass and
returns what the last of T successive `net(...)` calls would return.
"""

import torch
import torch.nn as nn

from .. import config as _config
from ..clock_driven import layer, neuron, surrogate
from ..config import EngineConfig, current as _cfg
from ..fused import ipool
from . import blocks as _blocks
from .blocks import MultiplyBy, NNConvUpsampling, SEWResBlock, SpikingStage

_RATE_KEYS = ('out_bottom', 'out_conv1', 'out_conv2', 'out_conv3', 'out_conv4', 'out_rconv', 'out_combined',
              'out_deconv4', 'out_add4', 'out_deconv3', 'out_add3', 'out_deconv2', 'out_add2', 'out_deconv1',
              'out_add1')


def _pyramid(input_size):
    """Spatial sizes after bottom, conv1..conv4 (k5 s2 p2): (260,346) -> (130,173),(65,87),(33,44),(17,22)."""
    sizes = [tuple(input_size)]
    for _ in range(4):
        h, w = sizes[-1]
        sizes.append(((h + 4 - 5) // 2 + 1, (w + 4 - 5) // 2 + 1))
    return sizes


class NeuromorphicNet(nn.Module):
    def __init__(self, surrogate_function=None, detach_reset=True, v_threshold=1.0, v_reset=0.0):
        super().__init__()
        self.surrogate_fct = surrogate.Sigmoid() if surrogate_function is None else surrogate_function
        self.detach_rst = detach_reset
        self.v_th = v_threshold
        self.v_rst = v_reset
        self.max_test_accuracy = float('inf')
        self.epoch = 0
        # the engine configuration this network runs with (config.EngineConfig: which kernel form every layer takes).  Captured at construction from the
        # configuration in effect (the shipped default unless built inside `config.engine_config(...)` / handed `config=`); NOT part of state_dict
        self.config = _cfg()
        self._plan = {}

    # -- engine configuration (build-side addition; the reference has no such thing) -------------------
    def configured(self, **overrides):
        """Context manager: this network with `overrides` applied to its configuration, e.g. `with net.configured(BOX_BWD=False): ...`."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev, self.config = self.config, self.config.replace(**overrides)
            try:
                yield self
            finally:
                self.config = prev
        return cm()

    def plan(self, as_text=False):
        """The dispatch plan of the LAST forward (and, once it has run, backward) pass: {layer: {'synapse_fwd', 'neuron_fwd', 'synapse_bwd', 'neuron_bwd': kernel
        form}} for the 13 spiking layers and the 4 prediction heads, recorded at the dispatch sites themselves (not re-derived)."""
        import copy
        if not as_text:
            return copy.deepcopy(self._plan)
        keys = ('synapse_fwd', 'neuron_fwd', 'neuron_bwd', 'synapse_bwd')
        return '\n'.join(f'{name:22s} ' + ' | '.join(f'{k}: {d.get(k, "-")}' for k in keys) for name, d in self._plan.items())

    # -- state utilities (SNN_models.py:22-48) ------------------------------------------------------
    def detach(self):
        for m in self.modules():
            if isinstance(m, neuron.BaseNode):
                m.detach()
            elif isinstance(m, layer.Dropout) and m.mask is not None:
                m.mask.detach_()

    def get_network_state(self):
        return [m.v for m in self.modules() if hasattr(m, 'reset') and hasattr(m, 'v')]

    def change_network_state(self, new_state):
        it = iter(new_state)
        for m in self.modules():
            if hasattr(m, 'reset') and hasattr(m, 'v'):
                m.v = next(it)

    def set_output_potentials(self, new_pots):
        it = iter(new_pots)
        for m in self.modules():
            if isinstance(m, neuron.IFNode):
                m.v = next(it)

    def increment_epoch(self):
        self.epoch += 1

    def get_max_accuracy(self):
        return self.max_test_accuracy

    def update_max_accuracy(self, new_acc):
        self.max_test_accuracy = new_acc

    def count_trainable_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class _SpikingEncoderDecoder(NeuromorphicNet):
    """Topology shared by the three models; subclasses only choose the neuron factory and the input channels."""
    _returns_spikes = True

    def _build(self, in_channels, make_node, make_resblock, multiply_factor, ineurons, input_size=(260, 346)):
        sz = _pyramid(input_size)
        self.input_size = tuple(input_size)

        def stage(synapse):
            return SpikingStage(synapse, MultiplyBy(multiply_factor), make_node())

        self.bottom = stage(nn.Conv2d(in_channels, 32, kernel_size=5, stride=1, padding=2, bias=False))
        chans = (32, 64, 128, 256, 512)
        for i in range(1, 5):
            setattr(self, f'conv{i}', stage(nn.Conv2d(chans[i - 1], chans[i], kernel_size=5, stride=2, padding=2,
                                                      bias=False)))
        self.bottleneck = nn.Sequential(make_resblock(), make_resblock())
        for lvl in (4, 3, 2, 1):
            setattr(self, f'deconv{lvl}', stage(NNConvUpsampling(chans[lvl], chans[lvl - 1], kernel_size=5,
                                                                  up_size=sz[lvl - 1])))
        for lvl in (4, 3, 2, 1):
            setattr(self, f'predict_depth{lvl}', nn.Sequential(
                NNConvUpsampling(chans[lvl - 1], 1, kernel_size=3, up_size=sz[0], bias=True),
                MultiplyBy(multiply_factor)))
        self.Ineurons = ineurons

    # -- the engine --------------------------------------------------------------------------------------
    def _run(self, x_seq: torch.Tensor, count: bool = False):
        """x_seq [T, B, C, H, W] -> (depth_seq [T, 4, B, 1, H, W], last-step spike tensors, counters), under THIS network's configuration; the kernel
        form every layer takes is recorded into the dispatch plan (`plan()`), the backward adds its half when it runs."""
        self._plan.clear()
        with _config.use_config(self.config), _config.recording(self._plan):
            return self._run_impl(x_seq, count)

    def _run_impl(self, x_seq: torch.Tensor, count: bool = False):
        T, B = x_seq.shape[:2]          # x_seq may be a transposed view of [B, T, ...]: the NHWC path below copies it once, into its own layout
        cnt = {}

        def nnz(name):
            if not count:
                return None
            cnt[name] = torch.zeros(2, dtype=torch.int64, device=x_seq.device)
            return cnt[name]

        cl = _cfg().FUSE_UPCONV and _cfg().DECODER_CHANNELS_LAST
        enc_cl = cl and _cfg().ENCODER_CHANNELS_LAST and all(
            getattr(b, 'connect_function', 'ADD') == 'ADD' for b in self.bottleneck)
        if enc_cl:          # whole network on NHWC arrays [T, B, h, w, C]: no layout copies anywhere
            # every encoder output has two consumers (next conv + decoder skip): forked handles, gradients summed in the neuron backward
            fork = _cfg().FORK_OUTPUTS

            def two(r):
                return r if fork else (r, r)
            # 2-bit packed spike tensors (fused.PACK_SPIKES) next to / instead of the dense fp32 ones, per edge: the skip operands of
            # the decoder are read packed; a layer whose next synapse is an exact-split conv (reads packed through its im2col) writes
            # no dense output at all.  dense=1, packed-only=2, off=0.
            from .. import fused as _fused
            own16 = _fused.x16_mode(x_seq.device)          # 16-bit activation mode on the engine's own kernels (round 5): packed spikes there as well
            pk_on = _cfg().PACK_SPIKES and x_seq.is_cuda and ((x_seq.dtype == torch.float32 and not torch.is_autocast_enabled('cuda')) or own16 is not None)

            def mode(next_conv):
                if not pk_on:
                    return 0
                return 2 if (_fused.spike_conv_applies(next_conv, x_seq.device) or _fused.spike_conv_fwd_applies(next_conv, x_seq.device)) else 1

            def packed_in(conv, packed):      # hand the packed form only to a synapse that reads it (exact-split im2col; the operand copy of
                if packed is None:            # the MFMA weight gradient of conv1 / conv2, which still takes the dense form for MIOpen's forward)
                    return None
                if _fused.spike_conv_applies(conv, x_seq.device):
                    return packed
                if own16 is not None:
                    return packed if _fused.spike_conv_fwd_applies(conv, x_seq.device) else None
                wg = (_cfg().SPIKE_CONV_WGRAD_MFMA and getattr(conv, 'kernel_size', None) == (5, 5) and conv.stride == (2, 2)
                      and conv.in_channels in (32, 64))
                return packed if wg else None
            with _config.layer('bottom'):
                a, b = two(self.bottom.forward_sequence_conv_cl(x_seq.permute(0, 1, 3, 4, 2).contiguous(), nnz('bottom'), fork=fork,
                                                                pack=mode(self.conv1[0])))
            enc, enc_skip, enc_pk = [a], [b], [self.bottom[2].last_packed]
            for i in range(1, 4):
                st, nxt = getattr(self, f'conv{i}'), getattr(self, f'conv{i + 1}')
                with _config.layer(f'conv{i}'):
                    a, b = two(st.forward_sequence_conv_cl(enc[-1], nnz(f'conv{i}'), spikes_in=True, fork=fork,
                                                           x_packed=packed_in(st[0], enc_pk[-1]), pack=mode(nxt[0])))
                enc.append(a)
                enc_skip.append(b)
                enc_pk.append(st[2].last_packed)
            bn0, bn1 = self.bottleneck[0], self.bottleneck[1]
            with _config.layer('conv4'):
                enc.append(self.conv4.forward_sequence_conv_cl(enc[-1], nnz('conv4'), spikes_in=True, x_packed=packed_in(self.conv4[0], enc_pk[-1]),
                                                               pack=mode(bn0.conv1[0])))
            with _config.layer('bottleneck.0'):
                cur = bn0.forward_sequence_cl(enc[4], spikes_in=True, x_packed=self.conv4[2].last_packed, pack_out=mode(bn1.conv1[0]))   # enc[*], cur: spike tensors
            with _config.layer('bottleneck.1'):
                cur = bn1.forward_sequence_cl(cur, nnz('rconv'), spikes_in=True, x_packed=bn0.sn2.last_packed)
            spikes, heads = [cur.permute(0, 1, 4, 2, 3)], []
        else:
            x_seq = x_seq.contiguous()
            with _config.layer('bottom'):
                enc = [self.bottom.forward_sequence(x_seq, None, nnz('bottom'))]
            for i in range(1, 5):
                with _config.layer(f'conv{i}'):
                    enc.append(getattr(self, f'conv{i}').forward_sequence(enc[-1], None, nnz(f'conv{i}')))
            with _config.layer('bottleneck.0'):
                cur = self.bottleneck[0].forward_sequence(enc[4])
            with _config.layer('bottleneck.1'):
                cur = self.bottleneck[1].forward_sequence(cur, nnz('rconv'))
            if count and getattr(self.bottleneck[1], 'connect_function', 'ADD') != 'ADD':
                # the in-kernel counter saw sn2's output BEFORE the (unfused) connect function; the reference counts out_rconv itself
                cnt['rconv'] = torch.stack((cnt['rconv'][0], torch.count_nonzero(cur)))
            spikes, heads = [cur], []
            if cl:
                cur = cur.permute(0, 1, 3, 4, 2).contiguous()          # decoder runs on NHWC arrays [T, B, h, w, C]
        last_dense = {}                                                # index into `spikes` -> the last step of a packed-only stage output, unpacked
        prev_pk = None
        from .. import fused as _fused
        own16 = _fused.x16_mode(x_seq.device) if enc_cl else None
        plain32 = lambda t: t.dtype == torch.float32 and not torch.is_autocast_enabled('cuda')       # noqa: E731  (the fp32 mode)
        for lvl in (4, 3, 2, 1):
            stage, head = getattr(self, f'deconv{lvl}'), getattr(self, f'predict_depth{lvl}')
            if cl:
                skip = enc_skip[lvl - 1] if enc_cl else enc[lvl - 1].permute(0, 1, 3, 4, 2).contiguous()
                # the stage output feeds the next stage and its prediction head: forked handles again
                # the full-resolution stage feeds its prediction head only: when the head reads 2-bit packed spikes (fused.PACKED_HEAD) the stage
                # writes no dense output at all
                # (round 3, last: deconv2 as well — its other consumer, deconv1, reads packed spikes in its fused forward and weight-gradient kernels)
                C_out = stage[0].up[1].out_channels
                head_pk = bool(lvl in (1, 2) and enc_cl and _cfg().PACK_SPIKES and _cfg().PACKED_HEAD and _cfg().FORK_OUTPUTS and (plain32(cur) or own16 is not None)
                               and _fused._lib.head_packed_supported(C_out, 1, 3))
                hh, ww = stage[0].up[0].size[0] - 4, stage[0].up[0].size[1] - 4              # this stage's output = the next stage's input geometry
                if head_pk and lvl == 2:        # packed-only: its other consumer, deconv1, must read packed spikes too (a run-time form that cannot, unpacks)
                    head_pk = _cfg().PACKED_DECONV2 and _fused.stage_takes_packed_copy(self.deconv1[0], hh, ww, cur.device)
                # (round 4) a stage whose head needs the dense tensor still hands the next stage a packed COPY when that stage's sub-pixel forward reads one
                copy_pk = False
                if not head_pk and lvl > 1 and enc_cl and (plain32(cur) or own16 is not None):
                    copy_pk = _fused.stage_takes_packed_copy(getattr(self, f'deconv{lvl - 1}')[0], hh, ww, cur.device)
                with _config.layer(f'deconv{lvl}'):
                    r = stage.forward_sequence_cl(cur, skip, nnz(f'deconv{lvl}'), spikes_in=True, fork=_cfg().FORK_OUTPUTS,
                                                  skip_packed=enc_pk[lvl - 1] if enc_cl else None, pack=2 if head_pk else (1 if copy_pk else 0), x_packed=prev_pk)
                cur, cur_head = r if _cfg().FORK_OUTPUTS else (r, r)
                out_pk = stage[2].last_packed if head_pk else None     # None: the packed kernel form did not apply, the output is dense
                prev_pk = stage[2].last_packed if (head_pk or copy_pk) else None       # the next stage's input in packed form (None: dense only)
                spikes.append(cur.permute(0, 1, 4, 2, 3))              # logical [T, B, C, H, W] view
                if out_pk is not None:
                    last_dense[len(spikes) - 1] = _fused.unpack_last_step(cur, out_pk).permute(0, 3, 1, 2)
                # the head is the only consumer of its forked handle: its input gradient travels as a rank-9 pair into the stage's neuron backward.
                # The full-resolution stage has no other gradient, so there the pair itself travels on as dL/dskip — which only another fused
                # neuron layer (the forked NHWC encoder output) can take
                lr_ok = bool(_cfg().FORK_OUTPUTS) and (enc_cl or lvl != 1)
                with _config.layer(f'predict_depth{lvl}'):
                    pd = head[0].forward_projected_cl(cur_head.flatten(0, 1), lowrank_grad=lr_ok, x_packed=out_pk)  # [T*B, H, W, 1]: one channel, NHWC == NCHW
            else:
                with _config.layer(f'deconv{lvl}'):
                    cur = stage.forward_sequence(cur, enc[lvl - 1], nnz(f'deconv{lvl}'))
                spikes.append(cur)
                synapse = head[0].forward_projected if _cfg().FUSE_UPCONV else head[0]
                with _config.layer(f'predict_depth{lvl}'):
                    pd = synapse(cur.flatten(0, 1))
            heads.append(pd.view(T, B, 1, *self.input_size))
        # shared I-neuron pool: v += gain * head, heads charged in the order 4,3,2,1 every step (:172-188)
        gains = [h[1].scale_value for h in (self.predict_depth4, self.predict_depth3, self.predict_depth2,
                                            self.predict_depth1)]
        if any(isinstance(g, torch.Tensor) for g in gains) or len({float(g) for g in gains}) != 1:
            heads = [torch.mul(p, g) for p, g in zip(heads, gains)]
            gain = 1.0
        else:
            gain = float(gains[0])
        pool = self.Ineurons
        depth_seq = ipool(torch.stack(heads), gain, pool.v_reset, pool._v_init(heads[0][0]))
        pool.v = depth_seq[T - 1, 3]
        return depth_seq, [last_dense.get(i, s[T - 1]) if last_dense else s[T - 1] for i, s in enumerate(spikes)], cnt, (T, B)

    def forward_sequence(self, x: torch.Tensor, rates: dict = None):
        """x [B, T, C, H, W]: all T frames in one pass, membranes carried from their current state.
        rates: optional dict, filled with the 15 firing-rate entries of `calculate_firing_rates` over ALL T steps of this pass (0-dim
        device tensors, no host synchronisation) from the counters the fused neuron kernels accumulate anyway-loaded data with
        (wavefront reductions; BASELINE.json config 5) — the training step's own forward, not a second one."""
        depth_seq, spikes, cnt, (T, _) = self._run(x.transpose(0, 1), count=rates is not None)
        if rates is not None:
            rates.update(self._rates(cnt, T))
        # ONE select of the last step, then unbind: autograd's backward is one zero-filled [T, 4, ...] buffer + one stack instead of four
        # such buffers and three full-size additions (0.19 ms per step at config 3)
        last = depth_seq[T - 1].unbind(0)
        depths = [last[3], last[2], last[1], last[0]]                 # [depth1, depth2, depth3, depth4]
        return (depths, spikes) if self._returns_spikes else depths

    def forward(self, x: torch.Tensor):
        # x must be of shape [batch_size, num_frames_per_depth_map, C, H, W]; like the reference only frame 0 is read
        return self.forward_sequence(x[:, 0:1])

    def calculate_firing_rates(self, x: torch.Tensor):
        """Density count_nonzero / numel of the 14 named tensors, from the counters the fused kernels accumulate
        (wavefront reductions + integer atomics) instead of a second pass over every tensor."""
        _, _, cnt, (T, B) = self._run(x[:, 0:1].transpose(0, 1), count=True)
        return self._rates(cnt, T)

    def _rates(self, cnt, T):
        rates = {k: 0. for k in _RATE_KEYS}

        def numel(stage_name):
            node = self.bottleneck[1].sn2 if stage_name == 'rconv' else getattr(self, stage_name)[2]
            return T * (node.last_numel or node.v.numel())       # (last_numel: no need to materialise an unwritten membrane for its size)

        for name in ('bottom', 'conv1', 'conv2', 'conv3', 'conv4'):
            rates[f'out_{name}'] = cnt[name][0].float() / numel(name)
        rates['out_rconv'] = cnt['rconv'][1].float() / numel('rconv')
        for lvl in (4, 3, 2, 1):
            rates[f'out_deconv{lvl}'] = cnt[f'deconv{lvl}'][0].float() / numel(f'deconv{lvl}')
            rates[f'out_add{lvl}'] = cnt[f'deconv{lvl}'][1].float() / numel(f'deconv{lvl}')
        return rates

    def set_init_depths_potentials(self, depth_prior):
        self.Ineurons.v = depth_prior


class StereoSpike(_SpikingEncoderDecoder):
    """Baseline binocular model: IF neurons everywhere, all potentials reset before every prediction.

    As in the reference (SNN_models.py:71-72, :105-106) the `v_threshold` / `v_reset` arguments are accepted but
    not forwarded (1.0 / 0.0 are always used) and the bottleneck keeps SEWResBlock's default Sigmoid surrogate
    whatever `surrogate_function` is — quirks kept on purpose, they are part of numerical parity."""

    def __init__(self, surrogate_function=None, detach_reset=True, v_threshold=1.0, v_reset=0.0, multiply_factor=1.,
                 input_size=(260, 346), config: EngineConfig = None):
        super().__init__(surrogate_function=surrogate_function, detach_reset=detach_reset)
        if config is not None:
            self.config = config

        def node():
            return neuron.IFNode(v_threshold=self.v_th, v_reset=self.v_rst, surrogate_function=self.surrogate_fct,
                                 detach_reset=True)

        def resblock():
            return SEWResBlock(512, v_threshold=self.v_th, v_reset=self.v_rst, connect_function='ADD',
                               multiply_factor=multiply_factor)

        self._build(4, node, resblock, multiply_factor,
                    neuron.IFNode(v_threshold=float('inf'), v_reset=0.0, surrogate_function=self.surrogate_fct),
                    input_size)


class fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike(_SpikingEncoderDecoder):
    """LIF (ATan surrogate) or PLIF (library-default Sigmoid surrogate) variant; the bottleneck is always PLIF.
    The paper's setting: tau=3.0, multiply_factor=10.0, use_plif=True (train.py:120)."""
    _in_channels = 4

    def __init__(self, use_plif=False, detach_reset=True, tau=10., v_threshold=1.0, v_reset=0.0, multiply_factor=1.,
                 input_size=(260, 346), config: EngineConfig = None):
        super().__init__(detach_reset=detach_reset)
        self.is_cext_model = False
        if config is not None:
            self.config = config

        def node():
            if use_plif:
                return neuron.ParametricLIFNode(init_tau=tau, v_threshold=v_threshold, v_reset=v_reset,
                                                detach_reset=True)
            return neuron.LIFNode(tau=tau, v_threshold=v_threshold, v_reset=v_reset,
                                  surrogate_function=surrogate.ATan(), detach_reset=True)

        def resblock():
            return SEWResBlock(512, v_threshold=v_threshold, v_reset=v_reset, connect_function='ADD',
                               multiply_factor=multiply_factor, use_plif=True, tau=tau)

        self._build(self._in_channels, node, resblock, multiply_factor,
                    neuron.IFNode(v_threshold=float('inf'), v_reset=v_reset, surrogate_function=surrogate.ATan()),
                    input_size)


class fromZero_feedforward_multiscale_tempo_monocular_SpikeFlowNetLike(
        fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike):
    """One camera only (2 input channels); returns the depth list alone (SNN_models.py:566)."""
    _in_channels = 2
    _returns_spikes = False

    def __init__(self, use_plif=False, detach_reset=True, tau=10., v_threshold=1.0, v_reset=0.0,
                 final_activation=nn.Identity, multiply_factor=1., input_size=(260, 346), config: EngineConfig = None):
        super().__init__(use_plif=use_plif, detach_reset=detach_reset, tau=tau, v_threshold=v_threshold,
                         v_reset=v_reset, multiply_factor=multiply_factor, input_size=input_size, config=config)
        self.final_activation = final_activation
