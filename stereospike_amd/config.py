"""Engine configuration: ONE frozen object (23 switches since the round-5 pruning; per-stage kernel choices are made from the geometry, fused.stage_plan).

    cfg = EngineConfig.default()                       # the shipped configuration; SS_* environment variables only seed THIS object
    net = StereoSpike(..., config=cfg.replace(PACK_SPIKES=False))     # a network owns its configuration: two networks with different settings coexist
    with net.configured(BOX_BWD=False): ...            # temporarily another configuration for one network (tests, A/B measurements)
    with engine_config(ASSERT_EXACT_SPLIT=True): ...   # ambient configuration for code that calls the fused ops without a network (kernel tests)
    net.plan()                                         # which kernel form every layer ran, per direction, as recorded at dispatch time

How it reaches the kernels: `current()` is the configuration in effect (a network installs its own for the duration of its forward; otherwise the ambient
one); every autograd Function captures it at forward time (`ctx.cfg`) and dispatches its backward from that capture, so a configuration is fixed per
forward / backward pair no matter what is active when `loss.backward()` runs.  The knob names are the ones the round-1..3 documents use; the
modules that used to hold them (fused, network.blocks, network.loss) still answer READS of those names with the current value, and refuse assignments.
"""
import contextlib
import dataclasses
import os
import threading
from dataclasses import dataclass
from typing import Optional, Tuple


@dataclass(frozen=True)
class EngineConfig:
    # ---- neuron kernels -------------------------------------------------------------------------------------------------------------------------------
    RECOMPUTE_H: bool = True            # training keeps the layer INPUT and rebuilds h in the backward kernel (forward writes 8 instead of 12 B/update)
    PACK_SPIKES: bool = True            # 2-bit packed spike tensors on the edges whose consumers read them
    LOWRANK_HEAD_GRAD: bool = True      # a prediction head hands its input gradient over as the rank-9 pair (g_P, W2)   [SS_LOWRANK_HEAD_GRAD]
    FORK_OUTPUTS: bool = True           # two-consumer outputs as two handles; the neuron backward adds the two gradients on load
    LAZY_MEMBRANE: bool = True          # a T-step training pass on the packed kernel forms does not write the membrane after step T (the reference resets it before the next
                                        # pass, train.py:221): `node.v` recomputes it on first access from the layer input the backward keeps, WITHOUT autograd history — as after
                                        # net.detach().  False: written by every pass and differentiable (BPTT across un-reset, un-detached calls)   [SS_LAZY_MEMBRANE]
    # ---- network layout -------------------------------------------------------------------------------------------------------------------------------
    FUSE_UPCONV: bool = True            # NNConvUpsampling through the projected form (no up-sampled tensor); False: the reference's two-op form on MIOpen
    DECODER_CHANNELS_LAST: bool = True  # decoder in NHWC memory
    ENCODER_CHANNELS_LAST: bool = True  # encoder / bottleneck in NHWC memory as well
    FUSED_LOSS: bool = True             # per-scale loss terms from one statistics kernel + one stencil kernel
    X16_OWN_KERNELS: bool = True        # 16-bit activation modes (torch.autocast fp16 / bf16): every synapse on the engine's own single-term 16-bit-I/O kernels (weights rounded
                                        # once to the format, fp32 accumulation, fp32 weight gradients; include/ss_neuron.h ABI 9), packed spikes between layers.  False: the
                                        # round-2 .. 4 path — encoder / bottleneck synapses = MIOpen convolutions under autocast (A/B: profiles/r05/)   [SS_X16_OWN_KERNELS]
    # ---- decoder (which STAGE takes which form is decided from its geometry: fused.stage_plan) ------------------------------------------------------------
    EXACT_SPLIT_GEMM: bool = True       # synapses on spike inputs as exact bf16x3 GEMMs / MFMA kernels
    ASSERT_EXACT_SPLIT: bool = False    # tests: verify (with a host sync) that a "spike" input really is bf16-exact
    PACKED_HEAD: bool = True            # prediction heads 1 / 2 read 2-bit packed spikes   [SS_PACKED_HEAD]
    PACKED_DECONV2: bool = True         # deconv2's output packed-only   [SS_PACKED_DECONV2]
    SUB_FWD: bool = True                # decoder stages, spike input: sub-pixel (merged tap) implicit GEMM forward, ss_upconv_sub.hip [SS_SUB_FWD]
    BOX_BWD: bool = True                # decoder stages: the backward on the box-sum image (ss_upconv_box.hip), no g_P anywhere   [SS_BOX_BWD]
                                        # 16-bit modes: the box-sum image is ONE plane of the activation format — sums of up to 5 x 5 neighbouring g_out values rounded to
                                        # 8 (bf16) / 11 (fp16) significand bits before both contractions (the g_P forms kept them in fp32): the weight-gradient parity bar of
                                        # these modes is 1e-2 for that reason, and in fp16 a box sum can exceed 65504 under a large loss scale although every g_out element is
                                        # finite — GradScaler then sees inf gradients and skips the step (it halves the scale and recovers; fp16 + a FIXED large scale does
                                        # not).  BOX_BWD=False keeps the fp32-accumulating g_P forms.
    EXACT_WGRAD_MFMA: bool = True       # g_P forms: weight gradient on spike inputs as the exact bf16x3 MFMA contraction (ss_spike_wgrad_f32)
    GEMM6_DGRAD: bool = True            # g_P forms: data gradient of the wide stages on ss_gemm6_f32   [SS_GEMM6_DGRAD]
    # ---- encoder / bottleneck convolutions ----------------------------------------------------------------------------------------------------------------
    WINOGRAD_DGRAD: bool = True
    CONV_DGRAD_MFMA: bool = True        # [SS_CONV_DGRAD_MFMA]
    SPIKE_CONV_FWD_MFMA: bool = True    # [SS_CONV_FWD_MFMA]
    SPIKE_CONV_WGRAD_MFMA: bool = True  # [SS_CONV_WGRAD_MFMA]
    DENSE_CONV_S1_MFMA: bool = True     # [SS_CONV_S1_MFMA]
    DENSE_CONV_S1_WGRAD_MFMA: bool = True   # [SS_CONV_S1_WGRAD_MFMA]

    def replace(self, **kw) -> 'EngineConfig':
        unknown = set(kw) - {f.name for f in dataclasses.fields(self)}
        if unknown:
            raise TypeError(f'EngineConfig has no field(s) {sorted(unknown)}')
        return dataclasses.replace(self, **{k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()})

    @staticmethod
    def from_env(env=None) -> 'EngineConfig':
        """The shipped defaults with the SS_* tuning variables of `env` (default os.environ) applied — the ONLY place they are read."""
        e = os.environ if env is None else env

        def flag(name, default):
            return e.get(name, '1' if default else '0') == '1'

        return EngineConfig(
            LOWRANK_HEAD_GRAD=flag('SS_LOWRANK_HEAD_GRAD', True), PACKED_HEAD=flag('SS_PACKED_HEAD', True), PACKED_DECONV2=flag('SS_PACKED_DECONV2', True),
            BOX_BWD=flag('SS_BOX_BWD', True), X16_OWN_KERNELS=flag('SS_X16_OWN_KERNELS', True), SUB_FWD=flag('SS_SUB_FWD', True), GEMM6_DGRAD=flag('SS_GEMM6_DGRAD', True),
            CONV_DGRAD_MFMA=flag('SS_CONV_DGRAD_MFMA', True), SPIKE_CONV_FWD_MFMA=flag('SS_CONV_FWD_MFMA', True),
            SPIKE_CONV_WGRAD_MFMA=flag('SS_CONV_WGRAD_MFMA', True), DENSE_CONV_S1_MFMA=flag('SS_CONV_S1_MFMA', True),
            DENSE_CONV_S1_WGRAD_MFMA=flag('SS_CONV_S1_WGRAD_MFMA', True), LAZY_MEMBRANE=flag('SS_LAZY_MEMBRANE', True))

    @staticmethod
    def default() -> 'EngineConfig':
        global _DEFAULT
        if _DEFAULT is None:
            _DEFAULT = EngineConfig.from_env()
        return _DEFAULT


_DEFAULT: Optional[EngineConfig] = None
KNOBS = tuple(f.name for f in dataclasses.fields(EngineConfig))


class _State(threading.local):
    def __init__(self):
        self.cfg = None            # the configuration in effect (None: the default)
        self.plan = None           # dict the dispatch sites record into (a network's), or None
        self.layer = None          # name of the layer being dispatched


_S = _State()


def current() -> EngineConfig:
    return _S.cfg if _S.cfg is not None else EngineConfig.default()


@contextlib.contextmanager
def use_config(cfg: Optional[EngineConfig]):
    """Make `cfg` the configuration in effect inside the block (None: leave the current one)."""
    if cfg is None:
        yield current()
        return
    prev, _S.cfg = _S.cfg, cfg
    try:
        yield cfg
    finally:
        _S.cfg = prev


def engine_config(**overrides):
    """Context manager: the current configuration with `overrides` applied, for code that calls the fused ops directly."""
    return use_config(current().replace(**overrides))


# ---- dispatch plan -------------------------------------------------------------------------------------------------------------------------------------
@contextlib.contextmanager
def recording(plan: Optional[dict]):
    prev, _S.plan = _S.plan, plan
    try:
        yield plan
    finally:
        _S.plan = prev


@contextlib.contextmanager
def layer(name: str):
    prev, _S.layer = _S.layer, name
    try:
        yield
    finally:
        _S.layer = prev


@contextlib.contextmanager
def sublayer(suffix: str):
    """The current layer's name + `suffix` (the two halves of a SEW block)."""
    prev = _S.layer
    _S.layer = (prev or '') + suffix
    try:
        yield
    finally:
        _S.layer = prev


def site():
    """(plan dict, layer name) in effect — what an autograd Function captures at forward time so that its backward can record under the same layer."""
    return _S.plan, _S.layer


def note(key: str, form: str, where=None):
    """Record that the current layer's `key` ('synapse_fwd', 'synapse_bwd_x', 'synapse_bwd_w', 'neuron_fwd', 'neuron_bwd') ran as `form`.  where: a
    captured site() (backward); default: the site in effect."""
    plan, name = where if where is not None else (_S.plan, _S.layer)
    if plan is not None and name is not None:
        plan.setdefault(name, {})[key] = form


def guard_module(module_name: str, what: str):
    """Turn `module_name` into a module that answers reads of the knob names with the current value and REFUSES assignments to them (a stale
    `fused.PACK_SPIKES = False` must fail loudly, not silently do nothing)."""
    import sys
    import types
    mod = sys.modules[module_name]

    class _Guarded(types.ModuleType):
        def __getattr__(self, name):
            if name in KNOBS:
                return getattr(current(), name)
            raise AttributeError(f'module {module_name!r} has no attribute {name!r}')

        def __setattr__(self, name, value):
            if name in KNOBS:
                raise AttributeError(f'{what}.{name} is a field of stereospike_amd.config.EngineConfig, not a module attribute: use '
                                     f'`with net.configured({name}=...)`, `StereoSpike(..., config=cfg.replace({name}=...))` or '
                                     f'`with config.engine_config({name}=...)`')
            super().__setattr__(name, value)
    mod.__class__ = _Guarded
