"""Training / evaluation steps of the hot path as the reference's scripts perform them
(/root/reference/train.py:189-261, test.py:100-173), plus the synthetic MVSEC-shaped data of SURVEY.md §8(d)."""
import torch
import contextlib

from .clock_driven import functional
from .network.loss import Total_Loss
from .network.metrics import MeanDepthError


def synthetic_batch(B, T, C=4, H=260, W=346, seed=2021, device='cpu', lam=0.05, nan_frac=0.25):
    """Poisson(lam) event-count voxels [B, T, C, H, W] (integer counts stored as fp32, datasets/MVSEC/utils.py:263-274)
    and a metric depth label [B, 1, H, W] in [0.5, 10) m with a fixed fraction of NaN = invalid pixels
    (datasets/MVSEC/mvsec_dataset.py:144)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.poisson(torch.full((B, T, C, H, W), lam), generator=g)
    gt = 0.5 + 9.5 * torch.rand(B, 1, H, W, generator=g)
    gt[torch.rand(B, 1, H, W, generator=g) < nan_frac] = float('nan')
    return x.to(device), gt.to(device)


def set_deterministic(on: bool = True):
    """The reference's reproducibility switch (/root/reference/train.py:35-50: cudnn.deterministic = True, cudnn.benchmark = False,
    torch.use_deterministic_algorithms(True)) for the MI355X engine.  With it on, MIOpen is asked for deterministic solvers (no atomic
    split-K weight gradients) and runs without its timing-based find mode, the GEMM algorithm record (TunableOp) is not consulted, and
    torch refuses non-deterministic kernels; the engine's own kernels are deterministic by construction (integer counters, fixed-order
    reductions for dL/dk, loss sums and split-K weight gradients).  Two runs of a training step from the same state are then
    bit-identical (tests/test_gpu_03_model.py::test_deterministic_mode); slower than the default (measured in profiles/)."""
    torch.backends.cudnn.deterministic = bool(on)
    if on:
        torch.backends.cudnn.benchmark = False
    torch.use_deterministic_algorithms(bool(on), warn_only=False)
    if torch.cuda.is_available() and hasattr(torch.cuda, 'tunable'):
        if on:
            torch.cuda.tunable.enable(False)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class Trainer:
    """reset -> T-step forward -> Total_Loss -> backward (gradient all-reduce overlapped when DP) -> Adam -> detach."""

    def __init__(self, net, lr=2e-4, weight_decay=0.0, reducer=None, loss_module=None, amp_dtype=None, count_rates=False):
        """amp_dtype: None (fp32, the reference) | torch.bfloat16 | torch.float16 — 16-bit activations under torch.autocast with fp32
        membranes (BASELINE.json configs 2 / 5).  With float16 the activation GRADIENTS are fp16 as well: d loss / d pred is ~1e-6
        (Total_Loss normalises by ~1e6 valid pixels), below fp16's smallest normal 6.1e-5, so the loss is scaled by a dynamic
        torch.amp.GradScaler and the fp32 weight gradients are unscaled inside the fused Adam step (no host synchronisation).
        count_rates: firing rates of the 14 layers from the fused kernels' counters inside this very forward (config 5); the latest
        dict (device scalars) is `self.last_rates`."""
        self.net = net
        on_gpu = next(net.parameters()).is_cuda
        self.amp_dtype = amp_dtype
        self.scaler = torch.amp.GradScaler('cuda', init_scale=2.0 ** 16) if (amp_dtype == torch.float16 and on_gpu) else None
        self.count_rates = count_rates
        self.last_rates = None
        # one fused multi-tensor launch for the 21-34 parameter tensors instead of ~10 kernels per tensor
        self.opt = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=weight_decay, fused=on_gpu)
        self.sched = torch.optim.lr_scheduler.MultiStepLR(self.opt, milestones=[8, 42, 60], gamma=0.5)  # train.py:127
        self.loss_module = loss_module or Total_Loss(alpha=0.5, scale_weights=(1., 1., 1., 1.), penalize_spikes=False)
        self.reducer = reducer

    def step(self, x, label):
        net = self.net
        functional.reset_net(net)                              # train.py:221
        rates = {} if self.count_rates else None
        with (torch.autocast('cuda', dtype=self.amp_dtype) if self.amp_dtype is not None else _null()):
            out = net.forward_sequence(x, rates) if rates is not None else net.forward_sequence(x)
            pred, spks = out if isinstance(out, tuple) else (out, None)
            loss = self.loss_module(pred, label, spks)         # train.py:238
        self.last_rates = rates
        with torch.autocast('cuda', enabled=False) if loss.is_cuda else _null():
            (self.scaler.scale(loss) if self.scaler is not None else loss).backward()
        if self.reducer is not None:
            self.reducer.finish()
        if self.scaler is not None:
            self.scaler.step(self.opt)                         # fused Adam takes grad_scale / found_inf tensors: unscale + skip-on-inf on device
            self.scaler.update()
        else:
            self.opt.step()
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=True)
        net.detach()                                           # train.py:242
        return loss.detach(), pred[0].detach()

    @torch.no_grad()
    def evaluate(self, x, label):
        net = self.net
        functional.reset_net(net)
        out = net.forward_sequence(x)
        pred, spks = out if isinstance(out, tuple) else (out, None)
        return self.loss_module(pred, label, spks), MeanDepthError(pred[0], label)


class GraphedInference:
    """The reference's test-time loop body (test.py:140-150: reset_net -> forward, batch 1) captured ONCE into a HIP graph and replayed
    per sample.  At batch 1 the ~150 launches of a forward are launch-bound (2.45 ms per call for B = 1, T = 1..5, against 0.3 - 1.2 ms
    of GPU work; profiles/README.md); a replay costs one launch.  Semantics = reset before every call (what test.py does); for a
    membrane carried across calls use the eager path.

    net: an SNN model of stereospike_amd.network.SNN_models on a HIP device, in eval mode.  example_x: [B, T, C, H, W] on the same
    device; later inputs must have the same shape and dtype.  amp_dtype: optional torch.float16 / torch.bfloat16 activation mode."""

    def __init__(self, net, example_x, amp_dtype=None, warmup=3):
        assert example_x.is_cuda, 'GraphedInference needs the MI355X (HIP graphs)'
        self.net = net
        self.amp = dict(device_type='cuda', dtype=amp_dtype or torch.float32, enabled=amp_dtype is not None)
        self.static_x = example_x.clone()
        side = torch.cuda.Stream(device=example_x.device)
        side.wait_stream(torch.cuda.current_stream(example_x.device))
        with torch.cuda.stream(side):           # warm-up off the default stream: MIOpen find, gather tables, allocator pools
            for _ in range(warmup):
                self._forward()
        torch.cuda.current_stream(example_x.device).wait_stream(side)
        torch.cuda.synchronize(example_x.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = self._forward()

    def _forward(self):
        functional.reset_net(self.net)
        with torch.no_grad(), torch.autocast(**self.amp):
            return self.net.forward_sequence(self.static_x)

    def __call__(self, x):
        """Returns the model's outputs for x as STATIC tensors (overwritten by the next call; clone to keep)."""
        self.static_x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out


@contextlib.contextmanager
def _parameters_replaced(net, tensors):
    """Inside the block, the named Parameters of `net` are plain tensor attributes `tensors[name]`; restored on exit (public nn.Module API only —
    what torch's private stateless._reparametrize_module does)."""
    swapped, order = [], {}
    try:
        for name, t in tensors.items():
            mod_name, _, attr = name.rpartition('.')
            mod = net.get_submodule(mod_name) if mod_name else net
            order.setdefault(mod, list(mod._parameters.keys()))
            swapped.append((mod, attr, mod._parameters.pop(attr)))
            setattr(mod, attr, t)
        yield
    finally:
        for mod, attr, p in swapped:
            if attr in mod.__dict__:
                delattr(mod, attr)
            mod._parameters[attr] = p
        for mod, keys in order.items():          # the registration order (state_dict key order) as it was
            for k in keys:
                mod._parameters[k] = mod._parameters.pop(k)


class GraphedTrainer:
    """One whole training iteration — reset -> T-step forward -> Total_Loss -> backward -> Adam — captured ONCE into a HIP graph and
    replayed per batch (single GPU).  For small steps (e.g. BASELINE.json config 2: T = 1, B = 8) the ~900 launches of an iteration
    are host-bound: 12.9 ms per step against 6.3 ms of GPU work (profiles/README.md); a replay is one launch.  Large steps (config 3)
    are GPU-bound and gain nothing — use Trainer there.

    Same arithmetic as Trainer.step (the captured work IS that code): Adam with `capturable=True` and a tensor learning rate, so the
    MultiStepLR schedule (train.py:127) keeps working without a re-capture.  Inputs must keep the shape / dtype of the first batch.

    The captured forward runs on detached ALIASES of the parameters (same storage: `load_state_dict` / in-place updates are seen; moving the network to another
    device or replacing a Parameter object after construction is not) so that autograd graphs the caller still holds from earlier eager passes cannot drag the
    default stream into the capture (see `_iteration`); gradients land in the Parameters' `.grad` as usual."""

    def __init__(self, net, lr=2e-4, weight_decay=0.0, loss_module=None, amp_dtype=None, warmup=3):
        dev = next(net.parameters()).device
        assert dev.type == 'cuda', 'GraphedTrainer needs the MI355X (HIP graphs)'
        self.net, self.dev = net, dev
        self.opt = torch.optim.Adam(net.parameters(), lr=torch.tensor(lr, device=dev), weight_decay=weight_decay, fused=True,
                                    capturable=True)
        self.sched = torch.optim.lr_scheduler.MultiStepLR(self.opt, milestones=[8, 42, 60], gamma=0.5)
        self.loss_module = loss_module or Total_Loss(alpha=0.5, scale_weights=(1., 1., 1., 1.), penalize_spikes=False)
        self.amp = dict(device_type='cuda', dtype=amp_dtype or torch.float32, enabled=amp_dtype is not None)
        self.scaler = torch.amp.GradScaler('cuda', init_scale=2.0 ** 16) if amp_dtype == torch.float16 else None   # see Trainer
        self.warmup = warmup
        self.graph = None
        named = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
        self._params = [p for _, p in named]
        self._alias = {n: p.detach().requires_grad_() for n, p in named}     # same storage: Adam's in-place updates of p are what the next forward reads

    def _iteration(self):
        net = self.net
        functional.reset_net(net)
        # The forward runs on ALIASES of the parameters (detached views of the same storage that require grad), never on the Parameters themselves.  A
        # Parameter's AccumulateGrad node is stream-stateful (it belongs to the stream that was current when it was created) and lives for as long as ANY
        # autograd graph that reaches the parameter is alive — e.g. the caller still holds the loss / predictions of an eager iteration run on the default
        # stream.  The autograd engine synchronises every gradient it routes to such a node with the node's stream, so a surviving default-stream node drags the
        # default stream into the capture and hipStreamEndCapture crashed (the round-3 / round-4 "crash during capture" of config 2:
        # tools/r05/repro_graph.py, profiles/r05/repro_graph_*.log; neither loss.backward() nor autograd.grad on the Parameters avoids it).  The aliases are
        # leaves nothing outside this class can reach: their nodes are created inside this iteration, on this iteration's stream.
        with _parameters_replaced(net, self._alias):
            with torch.autocast(**self.amp):
                out = net.forward_sequence(self.static_x)
                pred, spks = out if isinstance(out, tuple) else (out, None)
                loss = self.loss_module(pred, self.static_gt, spks)
        grads = torch.autograd.grad(self.scaler.scale(loss) if self.scaler is not None else loss, list(self._alias.values()), allow_unused=True)
        for p, g in zip(self._params, grads):
            p.grad = g                           # static tensors of the graph's pool in a capture: every replay overwrites them
        if self.scaler is not None:
            self.scaler.step(self.opt)
            self.scaler.update()
        else:
            self.opt.step()
        return loss.detach(), pred[0].detach()

    def _capture(self, x, label):
        self.static_x, self.static_gt = x.clone(), label.clone()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        # the eager warm-up iterations (MIOpen find, GEMM record, allocator pools, Adam state allocation) must not TRAIN: parameters and
        # BatchNorm buffers are restored and the optimiser state they created is zeroed in place afterwards, so the first replay is
        # training step 1 exactly like the eager Trainer's first step
        snap = [t.detach().clone() for t in list(self.net.parameters()) + list(self.net.buffers())]
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.opt.zero_grad(set_to_none=True)
                self._iteration()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        with torch.no_grad():
            for t, s0 in zip(list(self.net.parameters()) + list(self.net.buffers()), snap):
                t.copy_(s0)
            for st in self.opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        if self.scaler is not None:
            # The scaler's state tensors must live OUTSIDE the capture: a fresh GradScaler would run its lazy initialisation (torch.full of
            # _scale / _growth_tracker) inside torch.cuda.graph, and every replay would then reset the loss scale to 2^16 — a static scale,
            # and a persistent overflow would skip every optimiser step for ever.  The warm-up iterations initialised them:
            # reset the values in place.
            if self.scaler._scale is None:          # warmup = 0: nothing initialised them yet — do it here, still outside the capture
                self.scaler._lazy_init_scale_growth_tracker(self.dev)
            self.scaler._scale.fill_(2.0 ** 16)
            self.scaler._growth_tracker.zero_()
        self.opt.zero_grad(set_to_none=True)    # gradients are (re)created inside the capture: every replay overwrites them
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = self._iteration()

    def step(self, x, label):
        """The first call runs `warmup` eager iterations on this batch and captures; it then replays once, like every later call.
        Returns (loss, final depth map) as STATIC tensors (overwritten by the next call)."""
        if self.graph is None:
            self._capture(x, label)
        # the aliases are views of the storage the Parameters had at construction: a Parameter moved / replaced since then would train against stale memory
        for (n, a), p in zip(self._alias.items(), self._params):
            if p.data_ptr() != a.data_ptr():
                raise RuntimeError(f'GraphedTrainer: the storage of parameter {n} changed after construction (net.to / net.half / p.data = ...): build a new GraphedTrainer')
        self.static_x.copy_(x, non_blocking=True)
        self.static_gt.copy_(label, non_blocking=True)
        self.graph.replay()
        functional.reset_net(self.net)          # python-side state only; the captured work starts from reset itself
        return self.static_out
