#!/usr/bin/env python3
"""bench.py — headline benchmark of BASELINE.json: train frames/sec of StereoSpike (binocular, T = 5,
260x346 voxels) on N MI355X, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full training iteration on one batch of synthetic input per GPU: reset_net -> T-step forward
(convs on PyTorch-ROCm, every neuron layer one fused HIP launch) -> Total_Loss -> backward (fused surrogate
backward; with N > 1 the RCCL gradient all-reduce overlapped) -> Adam step -> detach.  Inputs are resident in HBM
when the timed region starts.  Rank 0 prints ONE JSON line (contract in the task statement) carrying
  roofline     — the dominant fused neuron kernel (whichever of forward / backward took more of the timed region; roofline_fwd and
                 roofline_bwd carry both): algorithmic bytes / HIP-event time over its launches inside the timed region, vs the
                 8 TB/s HBM3E peak, with the counter-measured HBM bytes per launch as `traffic`;
  cpu_baseline — the oracle's unfused eager-PyTorch port of the same training step on the host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_T_PROCESS_START = time.perf_counter()


def launcher_command(gpus, argv, env):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: the command that re-executes this script as N ranks (one
    process per GPU, RCCL rendezvous on 127.0.0.1).  None when this process is already a rank (WORLD_SIZE set) or N == 1."""
    if gpus <= 1 or 'WORLD_SIZE' in env:
        return None
    port = env.get('MASTER_PORT', str(29500 + (os.getpid() % 499)))
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}', '--master-addr', '127.0.0.1',
            '--master-port', port, os.path.abspath(__file__)] + list(argv)


def _early_gpus(argv):
    for i, t in enumerate(argv):
        if t == '--gpus' and i + 1 < len(argv):
            return int(argv[i + 1])
        if t.startswith('--gpus='):
            return int(t.split('=', 1)[1])
    return 1


if __name__ == '__main__':
    _cmd = launcher_command(_early_gpus(sys.argv[1:]), sys.argv[1:], os.environ)
    if _cmd is not None:                                  # before torch / MIOpen initialise in this (parent) process
        import subprocess
        _env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        sys.exit(subprocess.call(_cmd, env=_env))

from stereospike_amd import miopen_cache  # noqa: E402
_skip_naive = os.environ.get('SS_MIOPEN_SKIP_NAIVE', '1') == '1'
if int(os.environ.get('WORLD_SIZE', '1')) > 1:      # before torch / MIOpen initialise: one MIOpen cache directory per rank
    miopen_cache.enable_per_rank(int(os.environ.get('LOCAL_RANK', '0')), skip_naive_solvers=_skip_naive)
else:
    miopen_cache.enable(skip_naive_solvers=_skip_naive)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='samples per GPU (config 3: 16)')
    ap.add_argument('--T', type=int, default=5)
    ap.add_argument('--model', default='StereoSpike', choices=['StereoSpike', 'PLIFNet', 'PLIFNetMono'],
                    help='StereoSpike (binocular IF, the headline); PLIFNet (binocular PLIF); PLIFNetMono = BASELINE.json configs[1]\'s monocular PLIF network '
                         '(fromZero_feedforward_multiscale_tempo_monocular_SpikeFlowNetLike, 2 input channels)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'f16'],
                    help='f32 = the headline config; bf16 / f16 = 16-bit activations under torch.autocast with fp32 membrane '
                         '(BASELINE.json configs 2 / 5) — reported as a separate line, never as the headline value')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--cpu-b16', type=int, default=1,
                    help='1 (default): also time ONE iteration of the CPU port at the bench batch size (SURVEY.md §8(d): B = 16; ~110 s on the GPU box host, '
                         'measured 0.145 frames/s in profiles/r02/) as `cpu_baseline_B16_n1`; 2: as `cpu_baseline_B16` with evaluation pass, warm-up and >= 1 timed iteration (~6 min); 0: skip')
    ap.add_argument('--sustained-seconds', type=float, default=15.0,
                    help='after the timed region (and the event-timer pass): this many seconds of back-to-back steps, reported as `sustained` '
                         '(clock / thermal steady state; also what an external GPU-busy sampler gets to see).  0: skip')
    ap.add_argument('--leg-steps', type=int, default=5,
                    help='steps of the SECOND pass that times every fused launch with HIP events (roofline legs); the headline pass runs with the timer off')
    ap.add_argument('--miopen-find', type=int, default=1,
                    help='torch.backends.cudnn.benchmark = MIOpen find mode: picks the fastest solver per conv (measured '
                         '73.8 vs 87.3 ms/step); the search costs ~3.5 min on a cold box, ~75 s with the in-tree find-db')
    ap.add_argument('--channels-last', type=int, default=0)
    ap.add_argument('--bucket-mb', type=float, default=20.0)
    ap.add_argument('--force-dp', action='store_true',
                    help='validation aid for a 1-GPU box: build the RCCL process group and the gradient reducer even with one rank')
    ap.add_argument('--decoder-nhwc', type=int, default=1, help='1: decoder kept in NHWC memory (one GEMM per stage)')
    ap.add_argument('--encoder-nhwc', type=int, default=1, help='1: encoder/bottleneck activations in NHWC as well')
    ap.add_argument('--recompute-h', type=int, default=1,
                    help='1 (default): neuron backward recomputes h from the layer input (forward writes 8 B/update); 0: forward saves h_seq')
    ap.add_argument('--exact-split', type=int, default=1,
                    help='1 (default): decoder forward projections of spike inputs as exact bf16x3 GEMMs on the bf16 MFMA path; 0: plain fp32 GEMM')
    ap.add_argument('--gemm-tuning', type=int, default=1,
                    help='1 (default): load the tracked TunableOp record (GEMM algorithm per shape) read-only; 2: tune unseen shapes and '
                         'write gpurun_out/tunableop_results.csv; 0: library defaults')
    ap.add_argument('--graph', type=int, default=0,
                    help='1: replay the whole training iteration as one HIP graph (engine.GraphedTrainer; single GPU, for small host-bound '
                         'steps such as config 2: --T 1 --batch 8); the per-kernel roofline legs are not available in this mode')
    ap.add_argument('--fork-outputs', type=int, default=1, help='1 (default): two-consumer spike tensors as forked handles, gradients summed in the neuron backward kernel')
    ap.add_argument('--count-rates', type=int, default=0,
                    help='1: firing-rate counters of all 14 layers inside the training step (BASELINE.json config 5), from the fused kernels')
    ap.add_argument('--sub-fwd', type=int, default=1, help='1 (default): decoder stages forward as the sub-pixel (merged tap) implicit GEMM (ss_upconv_sub.hip); 0: projection GEMM + gather kernel (A/B)')
    ap.add_argument('--x16-own', type=int, default=1, help='1 (default): the 16-bit activation modes (--dtype bf16 / f16) on the engine\'s own single-term kernels; 0: the round-2 .. 4 path '
                                                            '(encoder / bottleneck synapses = MIOpen convolutions under autocast) (A/B)')
    ap.add_argument('--pack-spikes', type=int, default=1, help='1 (default): 2-bit packed spike tensors on the edges whose consumers read them')
    ap.add_argument('--box-bwd', type=int, default=1, help='1 (default): decoder backward on the box-sum image (ss_upconv_box.hip); 0: the g_P forms of rounds 2 - 3 (A/B)')
    ap.add_argument('--fuse-upconv', type=int, default=1,
                    help='1: NN-upsample+conv as low-res projection + fused gather kernel; 0: two-op form on MIOpen')
    return ap.parse_args()


def build_net(model, device, config=None):
    from stereospike_amd.clock_driven import surrogate
    from stereospike_amd.network import SNN_models as S
    torch.manual_seed(2021)                                   # train.py:53
    if model == 'StereoSpike':
        net = S.StereoSpike(surrogate_function=surrogate.ATan(), detach_reset=True, v_threshold=1.0, v_reset=0.,
                            multiply_factor=10., config=config)   # gain 10 so neurons fire
    elif model == 'PLIFNetMono':
        net = S.fromZero_feedforward_multiscale_tempo_monocular_SpikeFlowNetLike(tau=3., v_threshold=1.0, v_reset=0.0,
                                                                                  use_plif=True, multiply_factor=30., config=config)
    else:
        net = S.fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike(tau=3., v_threshold=1.0, v_reset=0.0,
                                                                             use_plif=True, multiply_factor=30., config=config)
    return net.to(device)


def workload_label(model, T, batch, dtype, world, count_rates=False):
    """`config.workload`, derived from the arguments: the BASELINE.json index is named only when the arguments ARE that configuration
    (VERDICT r05 weak #8b); anything else is described as what it is."""
    cams = 'monocular' if model == 'PLIFNetMono' else 'binocular'
    neuron = 'IF' if model == 'StereoSpike' else 'PLIF'
    what = f'{model} ({cams}, {neuron}) T={T} 260x346, batch {batch} per GPU x {world} GPU(s), {dtype} activations'
    idx = None
    if model == 'PLIFNetMono' and T == 1 and batch == 8 and dtype == 'bf16' and world == 1:
        idx = 'BASELINE.json configs[1]'
    elif model == 'StereoSpike' and T == 5 and dtype == 'f32' and batch == 16 and world == 1:
        idx = 'BASELINE.json configs[2]'
    elif model == 'StereoSpike' and T == 5 and dtype == 'f32' and batch * world == 128 and world == 8:
        idx = 'BASELINE.json configs[3]'
    elif model == 'StereoSpike' and T == 5 and dtype == 'f32' and batch == 16:
        idx = f'BASELINE.json configs[3] at {world} of 8 GPUs (the per-GPU share, batch 16, is configs[2]\'s)'
    elif model == 'StereoSpike' and T == 10 and dtype == 'f16' and batch == 32 and count_rates:
        idx = 'BASELINE.json configs[4]' if world == 8 else f'the per-GPU share (batch 32) of BASELINE.json configs[4] (batch 256 over 8 GPUs) on {world} GPU(s)'
    return what + (f' = {idx}' if idx else ' (not a BASELINE.json configuration)') + \
        ': train step (reset, T-step fwd, Total_Loss, BPTT, Adam), fused LIF fwd + surrogate bwd' + (', firing-rate counters on' if count_rates else '')


_EVAL_REF = {}


def cpu_baseline(model, T, budget_s, B=1, min_iters=5, quick=False):
    """The oracle's eager port (oracle/ref_network.py on oracle/sj_clock_driven.py — the reference's op sequence)
    running the same training step on the host cores: B = 1, T = 5, 1 warm-up + >= 5 timed iterations (~budget_s seconds,
    at most 6x that).  Reported, not a target.  quick: ONE timed iteration, no evaluation pass, no warm-up (the B = 16 leg of the default
    run, SURVEY.md §8(d): ~2 min of host time on the GPU box)."""
    from oracle import ref_network as rn, sj_clock_driven as sj
    from stereospike_amd.engine import synthetic_batch
    torch.manual_seed(2021)
    if model == 'StereoSpike':
        net = rn.build('StereoSpike', multiply_factor=10., surrogate_function=sj.ATan())
    else:
        net = rn.build(model, tau=3., use_plif=True, multiply_factor=30.)
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)
    x, gt = synthetic_batch(B, T, C=2 if model == 'PLIFNetMono' else 4, seed=2021)
    eval_mde = eval_mde64 = None
    if not quick:
        # eval MDE on identical (seed-2021 default-init) weights and inputs, before any update: the "eval MDE" half of the metric
        with torch.no_grad():
            d_eval = rn.run_sequence(net, x)[0]
            eval_mde = float(rn.mean_depth_error(d_eval[0], gt))
            # ... and with every convolution evaluated in float64 and rounded once (oracle/ref_network.py::float64_convs): the yard-stick that does not
            # depend on a backend's fp32 summation order.  An untrained spiking network amplifies one threshold-straddling rounding difference into a
            # cascade (1 - 2 % of the last step's spikes, 0.4 % of the MDE: tools/diag_eval_mde.py, profiles/r03/diag_eval_mde.log), and the host's
            # fp32 convolution is itself one such backend — so the product's eval MDE is held against THIS value too, next to the eager fp32 one
            with rn.float64_convs(net):
                r64 = rn.run_sequence(net, x)
                d64, s64 = r64[0], (r64[1] if len(r64) > 1 else None)
                eval_mde64 = float(rn.mean_depth_error(d64[0], gt))
            _EVAL_REF['spikes'] = [t.clone() for t in s64] if isinstance(s64, (list, tuple)) else None     # the last step's spike tensors (rconv, out_add4..1)

    def step():
        out = rn.run_sequence(net, x)
        loss = rn.total_loss(out[0], gt, out[1])
        loss.backward()
        opt.step()
        opt.zero_grad()
        net.detach()

    if not quick:
        step()
    times, t0 = [], time.perf_counter()
    while True:
        t1 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t1)
        el = time.perf_counter() - t0
        if quick or (len(times) >= min_iters and el >= budget_s) or len(times) >= 20 or el > 6 * budget_s:
            break
    n = len(times)
    med = sorted(times)[n // 2]
    # the MEDIAN iteration prices the baseline (>= 5 timed iterations: single iterations on a shared 256-thread host vary 3x)
    how = 'ONE timed training iteration, no warm-up' if quick else f'median of {n} timed training iterations after 1 warm-up'
    return dict(value=B / med, unit='frames/s', cores=torch.get_num_threads(), kind='port', eval_mde_m=eval_mde, eval_mde_m_float64_convs=eval_mde64,
                iteration_s=dict(median=round(med, 3), min=round(min(times), 3), max=round(max(times), 3), n=n),
                sample=f'{model} {"monocular" if model == "PLIFNetMono" else "binocular"} T={T} 260x346 fp32, B={B}, {how}, eager unfused oracle port, torch {torch.__version__} CPU, '
                       f'{torch.get_num_threads()} threads of {os.cpu_count()} logical CPUs')


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` (spawns the ranks itself) '
                         f'or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`')
    assert torch.cuda.is_available(), 'bench.py needs an MI355X; there is no CPU fallback for the product path'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dp = world > 1 or a.force_dp
    if use_dp:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    torch.backends.cudnn.benchmark = bool(a.miopen_find)
    if a.gemm_tuning:
        from stereospike_amd import gemm_tuning
        if a.gemm_tuning == 2:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            out_csv = os.path.join(ROOT, 'gpurun_out', 'tunableop_results.csv')
            if not os.path.exists(out_csv) and os.path.exists(gemm_tuning.SEED):
                import shutil
                shutil.copy(gemm_tuning.SEED, out_csv)
            gemm_tuning.enable(local, tuning=True, filename=out_csv)
        else:
            gemm_tuning.enable(local)

    from stereospike_amd import _lib
    from stereospike_amd.dp import GradientAllReducer
    from stereospike_amd.engine import Trainer, synthetic_batch
    from stereospike_amd.fused import TIMER
    _lib.lib()                                                # fail loudly here if the HIP library is missing

    # the engine configuration of this run: the shipped default (config.EngineConfig.default(): SS_* variables seed it) with the command line's A/B
    # switches applied — handed to the network, which owns it; nothing global is mutated
    from stereospike_amd.config import EngineConfig
    ov = dict(FUSE_UPCONV=bool(a.fuse_upconv), FORK_OUTPUTS=bool(a.fork_outputs), RECOMPUTE_H=bool(a.recompute_h), EXACT_SPLIT_GEMM=bool(a.exact_split),
              SUB_FWD=bool(a.sub_fwd), PACK_SPIKES=bool(a.pack_spikes), X16_OWN_KERNELS=bool(a.x16_own),
              DECODER_CHANNELS_LAST=bool(a.decoder_nhwc), ENCODER_CHANNELS_LAST=bool(a.encoder_nhwc), BOX_BWD=bool(a.box_bwd))
    engine_cfg = EngineConfig.default().replace(**ov)
    net = build_net(a.model, dev, engine_cfg)
    if a.channels_last:
        net = net.to(memory_format=torch.channels_last)
    reducer = GradientAllReducer(net, bucket_bytes=int(a.bucket_mb * (1 << 20)), reduce_single_rank=a.force_dp) if use_dp else None
    amp_dtype = {'bf16': torch.bfloat16, 'f16': torch.float16}.get(a.dtype)
    # 16-bit modes: autocast inside Trainer.step; fp16 also scales the loss (torch.amp.GradScaler: fp16 activation gradients would underflow)
    trainer = Trainer(net, reducer=reducer, amp_dtype=amp_dtype, count_rates=bool(a.count_rates))
    if a.graph:
        assert not use_dp, '--graph is single-GPU'
        from stereospike_amd.engine import GraphedTrainer
        trainer = GraphedTrainer(net, amp_dtype=amp_dtype)
    in_ch = 2 if a.model == 'PLIFNetMono' else 4
    x, gt = synthetic_batch(a.batch, a.T, C=in_ch, seed=2021 + rank, device=dev)     # resident in HBM before timing

    def sync():
        if use_dp:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        t_w = time.perf_counter()
        trainer.step(x, gt)
        torch.cuda.synchronize()
        if rank == 0:
            print(f'[bench] warm-up step {i}: {time.perf_counter() - t_w:.2f} s (first steps include MIOpen kernel '
                  f'compilation when the in-tree cache is cold)', file=sys.stderr, flush=True)
    # ---- headline pass: EXACTLY a.steps steps, barrier + synchronize on both sides, the per-launch HIP-event timer OFF ----
    sync()
    TIMER.enabled = False
    TIMER.clear()
    t_region_start = time.perf_counter()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, pred = trainer.step(x, gt)
    sync()
    elapsed = time.perf_counter() - t0
    t_region_end = time.perf_counter()
    if use_dp:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- second, short pass for the roofline legs: the same step with a HIP-event pair around every fused launch (on the launch stream) ----
    # (under --graph the replayed launches cannot carry event pairs: the legs then run the SAME launches uncaptured, through an eager Trainer on the same
    # network, after the sustained leg — it steps the parameters outside the graph, so nothing of the graph is timed after it)
    leg_steps = max(1, min(a.leg_steps, a.steps))

    def legs_pass(tr):
        t_leg0 = time.perf_counter()
        TIMER.enabled = True
        TIMER.clear()
        for _ in range(leg_steps):
            tr.step(x, gt)
        sync()
        TIMER.enabled = False
        return time.perf_counter() - t_leg0
    leg_elapsed = legs_pass(trainer) if not a.graph else 0.0
    # ---- sustained leg: back-to-back steps for >= --sustained-seconds (the headline region is a sub-second burst after warm-up: this is what the chip holds
    # at steady clocks / temperature, and what an external SMI sampler can see).  Same step, timer off, barrier + synchronize on both sides, MAX over ranks.
    sustained = None
    if a.sustained_seconds > 0:
        per_step = elapsed / a.steps
        chunk = max(1, int(round(1.0 / max(per_step, 1e-4))))           # synchronise about once a second
        sync()
        t_s0 = time.perf_counter()
        n_sus = 0
        while True:
            for _ in range(chunk):
                trainer.step(x, gt)
            n_sus += chunk
            torch.cuda.synchronize()
            go = torch.tensor([1.0 if time.perf_counter() - t_s0 < a.sustained_seconds else 0.0], device=dev)
            if use_dp:
                dist.all_reduce(go, op=dist.ReduceOp.MIN)               # every rank runs the same number of steps
            if float(go.item()) == 0.0:
                break
        sync()
        sus_elapsed = time.perf_counter() - t_s0
        if use_dp:
            t = torch.tensor([sus_elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sus_elapsed = float(t.item())
        sustained = dict(steps=n_sus, seconds=round(sus_elapsed, 2), frames_per_s=round(a.batch * world * n_sus / sus_elapsed, 3),
                         ms_per_step=round(1e3 * sus_elapsed / n_sus, 3),
                         offsets_s=dict(start=round(t_s0 - _T_PROCESS_START, 2), end=round(t_s0 - _T_PROCESS_START + sus_elapsed, 2)))
    if a.graph:
        eager = Trainer(net, amp_dtype=amp_dtype, count_rates=bool(a.count_rates))
        for _ in range(2):
            eager.step(x, gt)
        sync()
        leg_elapsed = legs_pass(eager)
    summ = TIMER.summary()
    shapes = TIMER.summary(by_shape=True)

    if rank == 0:
        frames = a.batch * world * a.steps
        zero = dict(launches=0, bytes=0, ms=0.0, updates=0)
        # the dominant launch shape of the forward kernel: the largest layer (bottom: B x T x 32 x 260 x 346 updates)
        # (every forward form counts: since round 3 the bottom layer writes packed-only, tag 'neuron_fwd_train+packed'; at equal size the launch
        # group that took longer — the full-resolution decoder stage, which reads a packed skip and writes a dense output)
        fwd_shapes = {k: v for k, v in shapes.items() if k[0].startswith('neuron_fwd_train')}
        dom = max(fwd_shapes, key=lambda k: (k[1], fwd_shapes[k]['ms'])) if fwd_shapes else None
        fwd = fwd_shapes.get(dom, zero)
        fwd_tag = dom[0] if dom else 'neuron_fwd_train'
        fwd_pmc = {'neuron_fwd_train': 'neuron_fwd', 'neuron_fwd_train+packed': 'neuron_fwd_packed',
                   'neuron_fwd_train+skip+packed': 'neuron_fwd_skip_packed'}.get(fwd_tag, 'none')
        # backward: the launch group with the most updates; at equal size the forked form without the extra g_sum store (the bottom
        # layer: 16 B/update), which is also the variant the PMC passes measure (tools/pmc_target.py)
        # (since the prediction heads hand over their gradient as a rank-9 pair: the '+lr' form, 13.1 B/update at 32 channels)
        pref = {'neuron_bwd+lr': 3, 'neuron_bwd+fork': 2, 'neuron_bwd': 1, 'neuron_bwd+fork+sum': 0, 'neuron_bwd+lr+sum': 0, 'neuron_bwd+lronly': 0}
        bwd_shapes = {k: v for k, v in shapes.items() if k[0] in pref}
        bwd = bwd_shapes.get(max(bwd_shapes, key=lambda k: (k[1], pref.get(k[0], 0))), zero) if bwd_shapes else zero
        bwd_tag = max(bwd_shapes, key=lambda k: (k[1], pref.get(k[0], 0)))[0] if bwd_shapes else 'neuron_bwd'
        fwd_all = [v for k, v in summ.items() if k.startswith('neuron_fwd')]
        all_ms = sum(v['ms'] for k, v in summ.items() if k.startswith('neuron_'))
        all_bytes = sum(v['bytes'] for k, v in summ.items() if k.startswith('neuron_'))

        def inst(tags, name):
            # every launch of one kernel instantiation (all layer shapes): the numbers to hold against its row of the committed
            # `rocprofv3 --kernel-trace --stats` CSV (AverageNs), and the bandwidth over all of them
            ds = [summ[t] for t in tags if t in summ]
            n, ms, by = sum(d['launches'] for d in ds), sum(d['ms'] for d in ds), sum(d['bytes'] for d in ds)
            return dict(rocprof_kernel_name=name, launches=n, avg_launch_us=round(1e3 * ms / max(1, n), 2),
                        achieved_GBps=round(by / 1e9 / (ms / 1e3), 1) if ms > 0 else 0.0,
                        frac=round(by / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4) if ms > 0 else 0.0)

        def roof(d, which='neuron_fwd', pmc_key=None):
            ach = (d['bytes'] / 1e9) / (d['ms'] / 1e3) if d['ms'] > 0 else 0.0
            traffic = _pmc_traffic(pmc_key or which, a.dtype, a.T, a.batch)
            avg_us = 1e3 * d['ms'] / max(1, d['launches'])
            return dict(bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
                        traffic_source=_pmc_source(a.dtype) if traffic else None,
                        launches=d['launches'], avg_launch_us=round(avg_us, 2),
                        bytes_per_launch=int(d['bytes'] / max(1, d['launches'])),
                        # counter-measured HBM bytes (incl. the O(N) v_last write the per-update figure leaves out) over the same time
                        hbm_GBps_of_pmc_traffic=round(traffic / avg_us / 1e3, 1) if (traffic and avg_us > 0 and a.dtype == 'f32' and a.T == 5 and a.batch == 16) else None,
                        ms_per_step_all_launches_of_this_kernel=round(sum(v['ms'] for k, v in summ.items() if k.startswith(which + ('_train' if which == 'neuron_fwd' else ''))) / leg_steps, 3))
        from stereospike_amd.network.metrics import MeanDepthError
        from stereospike_amd import fused as _fused

        def bpu(d):
            return round(d['bytes'] / d['updates'], 1) if d['updates'] else 0
        recompute_h = bool(engine_cfg.RECOMPUTE_H)
        packed_on = bool(engine_cfg.PACK_SPIKES) and recompute_h and (a.dtype == 'f32' or bool(engine_cfg.X16_OWN_KERNELS))   # the packed forms run (2-bit packed outputs / skips)
        half = a.dtype != 'f32'
        dtc = {'f16': 1, 'bf16': 2}.get(a.dtype, 0)              # SS_DT_F16 / SS_DT_BF16: the DT template argument rocprof prints
        tf = lambda b: 'true' if b else 'false'                  # noqa: E731

        kd = 0 if a.model == 'StereoSpike' else 2                # SS_KIND_IF / SS_KIND_PLIF: the KIND template argument rocprof prints
        seg_T = a.T in (4, 5, 8, 10)                             # ss_neuron_bwd16_lr.hip: the step counts the segmented low-rank form is instantiated for

        def fwd_name(skip):
            if half:       # ss_neuron16_v2.hpp (round 6): <KIND, DT, T, SKIP, DENSE copy>
                return f'neuron_fwd16_pk8_kernel<{kd}, {dtc}, {a.T}, {tf(skip)}, false>'
            return (f'neuron_fwd_kernel<{kd}, {a.T}, {tf(skip)}, false, 4, true>' if packed_on
                    else f'neuron_fwd_kernel<{kd}, {a.T}, {tf(skip)}, {tf(not recompute_h)}, 4, false>')

        def bwd_name(tag):
            if half:
                v = 4 if a.T <= 5 else 2
                if '+lr' in tag and seg_T:   # <KIND, SG, DT, T, VEC, NSEG, G2, LR, WAVES, HAS_G1, SUM, PASS>: the fast pass (PASS 0); the exact pass behind it (PASS 1) returns at once
                    return f'neuron_bwd16_seg_kernel<{kd}, 0, {dtc}, {a.T}, 4, {2 if a.T > 5 else 1}, true, true, {3 if a.T > 5 else 4}, true, false, 0>'
                if '+lr' in tag:             # other step counts: the round-5 form <KIND, SG, DT, T, VEC, G2, LR>
                    return f'neuron_bwd16_rc_kernel<{kd}, 0, {dtc}, {a.T}, {v}, true, true>'
                return f'neuron_bwd16_rc_kernel<{kd}, 0, {dtc}, {a.T}, {v}, {tf("fork" in tag)}, false>'
            return (f'neuron_bwd_kernel<{kd}, 0, {a.T}, 4, true, true, true>' if '+lr' in tag else
                    f'neuron_bwd_kernel<{kd}, 0, {a.T}, 4, true, true, false>' if 'fork' in tag
                    else f'neuron_bwd_kernel<*, *, {a.T}, 4, {tf(recompute_h)}, false, false>')
        out = {
            'metric': f'train frames/sec (260x346xT={a.T} {"mono" if a.model == "PLIFNetMono" else "stereo"} voxels)', 'value': round(frames / elapsed, 3),
            'unit': 'frames/s', 'n_gpus': world, 'rccl_ranks': dist.get_world_size() if use_dp else 1, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(1e3 * elapsed / a.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            # where the timed region sits inside this process (seconds since interpreter start): a sampler that saw no GPU activity during most of the
            # run was looking at imports, warm-up or the CPU-baseline legs
            'timed_region_s': round(elapsed, 4),
            'timed_region_offsets_s': dict(start=round(t_region_start - _T_PROCESS_START, 2), end=round(t_region_end - _T_PROCESS_START, 2),
                                           legs_pass_end=round(t_region_end - _T_PROCESS_START + leg_elapsed, 2)),
            'per_kernel_legs': dict(steps=leg_steps, ms_per_step_with_event_timer=round(1e3 * leg_elapsed / leg_steps, 3),
                                    note='separate pass after the timed region: fused.TIMER (one HIP-event pair per fused launch) is OFF while `value` is measured'
                                         + ('; --graph: the same launches uncaptured (eager Trainer on the same network), after the sustained leg' if a.graph else '')),
            'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'sustained': sustained,
            'sustained_frames_per_s': sustained['frames_per_s'] if sustained else None,
            'sustained_over_value': round(sustained['frames_per_s'] / (frames / elapsed), 4) if sustained else None,
            'config': {'workload': workload_label(a.model, a.T, a.batch, a.dtype, world, bool(a.count_rates)),
                       'batch_per_gpu': a.batch, 'global_batch': a.batch * world, 'T': a.T,
                       'parallelism': f'dp{world}', 'fuse_upconv': bool(a.fuse_upconv), 'decoder_nhwc': bool(a.decoder_nhwc), 'encoder_nhwc': bool(a.encoder_nhwc), 'weights': f'default init, seed 2021, multiply_factor {10 if a.model == "StereoSpike" else 30}',
                       'input': 'Poisson(0.05) voxels, label 0.5+9.5U with 25% NaN'},
            'roofline_fwd': dict(kernel=f'{"neuron_fwd16_pk8_kernel" if half else "neuron_fwd_kernel"}<{"IF" if kd == 0 else "PLIF"},T={a.T},train> (fused gain+charge+fire+reset over T, {a.dtype} I/O, '
                                        f'{bpu(fwd)} B/update'
                                        f'{", + skip add (2-bit packed skip operand)" if "+skip" in fwd_tag and packed_on else ", + skip add" if "+skip" in fwd_tag else ""}'
                                        f'{", 2-bit packed output only" if "+packed" in fwd_tag else ""}) on its dominant launch shape: '
                                        f'{dom[1] if dom else 0} updates (launch tag {fwd_tag})',
                                 **roof(fwd, 'neuron_fwd', fwd_pmc),
                                 all_launches_of_this_instantiation=inst(
                                     ['neuron_fwd_train+skip', 'neuron_fwd_train+skip+packed'] if '+skip' in fwd_tag else ['neuron_fwd_train', 'neuron_fwd_train+packed'],
                                     fwd_name('+skip' in fwd_tag))),
            'roofline_bwd': dict(kernel=f'{bwd_name(bwd_tag).split("<")[0]} (fused surrogate backward over T, {a.dtype} I/O, {bpu(bwd)} B/update'
                                        f'{", h recomputed from the layer input" if recompute_h else ""}'
                                        f'{", second consumer gradient added on load" if "fork" in bwd_tag else ""}'
                                        f'{", second consumer (prediction head) gradient formed in registers from its rank-9 pair and added on load" if "+lr" in bwd_tag else ""}), largest launch shape',
                                 **roof(bwd, 'neuron_bwd', 'neuron_bwd_lr' if '+lr' in bwd_tag else None),
                                 all_launches_of_this_instantiation=inst(
                                     ['neuron_bwd+lr', 'neuron_bwd+lr+sum', 'neuron_bwd+lronly'] if '+lr' in bwd_tag else
                                     ['neuron_bwd+fork', 'neuron_bwd+fork+sum'] if 'fork' in bwd_tag else ['neuron_bwd'],
                                     bwd_name(bwd_tag))),
            'neuron_kernels_all_layers': dict(
                launches=sum(v['launches'] for k, v in summ.items() if k.startswith('neuron_')),
                ms_per_step=round(all_ms / leg_steps, 3), algorithmic_GB_per_step=round(all_bytes / leg_steps / 1e9, 3),
                achieved_GBps=round(all_bytes / 1e9 / (all_ms / 1e3), 1) if all_ms else 0.0,
                share_of_step=round((all_ms / leg_steps) / (1e3 * elapsed / a.steps), 4)),
            # HIP-event averages over ALL launches of each kernel instantiation, for comparison with the AverageNs column of
            # the committed `rocprofv3 --kernel-trace --stats` CSV (profiles/): events add a few us per launch
            'rocprof_check_avg_us': {
                name: (round(1e3 * sum(summ[t]['ms'] for t in tags if t in summ) / max(1, sum(summ[t]['launches'] for t in tags if t in summ)), 1)
                       if any(t in summ for t in tags) else None)
                for name, tags in (
                    (fwd_name(False), ['neuron_fwd_train', 'neuron_fwd_train+packed']),
                    (fwd_name(True), ['neuron_fwd_train+skip', 'neuron_fwd_train+skip+packed']),
                    ('neuron_bwd*_kernel<*>', [k for k in summ if k.startswith('neuron_bwd')]))},
            'other_fused_kernels_ms_per_step': {k: round(v['ms'] / leg_steps, 3) for k, v in summ.items()
                                                if not k.startswith('neuron_')},
            # the up-conv stages one by one (projection GEMM + gather, resp. adjoint + dgrad / wgrad GEMMs), keyed by output elements
            'upconv_by_stage_ms_per_step': {f'{k[0]}:{k[1]}': round(v['ms'] / leg_steps, 3) for k, v in sorted(shapes.items(), key=lambda kv: -kv[0][1])
                                            if k[0].startswith('upconv')},
            'roofline_upconv': _roof_upconv(shapes, a, engine_cfg),
            'roofline_upconv_bwd': _roof_upconv_bwd(shapes, a, engine_cfg),
            # which kernel form every layer ran, per direction, as recorded at the dispatch sites (net.plan())
            'plan': net.plan(),
            'peak_mem_GB': round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
            'final_loss': round(float(loss), 5), 'train_mde_m': round(float(MeanDepthError(pred, gt)), 5),
        }
        # `roofline` = the dominant kernel of the path: whichever of the two fused neuron kernels took more of the timed region
        if ('fork' in bwd_tag or '+lr' in bwd_tag) and bwd['ms'] > 0 and a.dtype == 'f32':
            # SURVEY.md §8(d) prices the backward at 12 B/update (g_out, x, g_x): THAT is `frac` / `achieved`.  The forked / low-rank form
            # also reads the second consumer's gradient (4 resp. 36 / C B/update that autograd's accumulation pass would otherwise move 3x): the rate
            # over the bytes the kernel really moves is kept beside it
            rb = out['roofline_bwd']
            rb['frac_by_bytes_moved'], rb['achieved_GBps_by_bytes_moved'], rb['bytes_moved_per_update'] = rb['frac'], rb['achieved'], bpu(bwd)
            rb['achieved'] = round(12 * bwd['updates'] / 1e9 / (bwd['ms'] / 1e3), 1)
            rb['frac'] = round(rb['achieved'] / HBM_PEAK_GBS, 4)
            rb['bytes_per_launch'] = int(12 * bwd['updates'] / max(1, bwd['launches']))
            rb['definition'] = 'SURVEY.md 8(d): 12 B per neuron update (read g_out 4, read x 4, write g_x 4) x updates of the launch / HIP-event time'
        dom_key = 'roofline_bwd' if out['roofline_bwd']['ms_per_step_all_launches_of_this_kernel'] >= \
            out['roofline_fwd']['ms_per_step_all_launches_of_this_kernel'] else 'roofline_fwd'
        out = {**{k: v for k, v in out.items() if k not in ('roofline_fwd', 'roofline_bwd')}, 'roofline': out[dom_key],
               'roofline_fwd': out['roofline_fwd'], 'roofline_bwd': out['roofline_bwd']}
        if world == 1 and not a.no_cpu_baseline and a.dtype == 'f32':
            # eval MDE of the product on the same weights (fresh seed-2021 net) and the same B = 1 input as the CPU port
            from stereospike_amd.clock_driven import functional as _F
            from stereospike_amd.engine import synthetic_batch as _sb
            net0 = build_net(a.model, dev, engine_cfg)
            x0, gt0 = _sb(1, a.T, C=in_ch, seed=2021, device=dev)
            with torch.no_grad():
                _F.reset_net(net0)
                r0 = net0.forward_sequence(x0)
                d0 = r0[0]
                s0 = [t.cpu() for t in r0[1]] if isinstance(r0[1], (list, tuple)) else None
                out['eval_mde_m'] = round(float(MeanDepthError(d0[0], gt0)), 5)
            del net0
            out['cpu_baseline'] = cpu_baseline(a.model, a.T, a.cpu_seconds)
            if s0 is not None and _EVAL_REF.get('spikes') is not None and len(s0) == len(_EVAL_REF['spikes']):
                # bit-exact spike masks, end to end and free-running: the last step's returned spike tensors (out_rconv, out_add4 .. out_add1) of the product
                # against the float64-conv oracle's on the same weights and input — the number of elements that differ
                out['eval_last_step_spikes_differing_vs_cpu'] = int(sum(int((a_ != b_).sum()) for a_, b_ in zip(s0, _EVAL_REF['spikes'])))
                out['eval_last_step_spikes_compared'] = int(sum(a_.numel() for a_ in s0))
            if a.cpu_b16:                                 # the CPU port at the bench's own batch size as well
                # (key says what it is: ONE un-warmed iteration with --cpu-b16 1 — VERDICT r04 weak #11; --cpu-b16 2 = warm-up + median)
                out['cpu_baseline_B16_n1' if a.cpu_b16 == 1 else 'cpu_baseline_B16'] = cpu_baseline(a.model, a.T, 120.0, B=a.batch, min_iters=1, quick=a.cpu_b16 == 1)
            # eval MDE "at matching Mean Depth Error".  `eval_mde_rel_diff_vs_cpu` keeps its round-1/2 meaning — against the eager fp32 port of the reference
            # (the reference's own fp32 arithmetic on this host) — and the float64-convolution oracle, the yard-stick that does not depend on a
            # backend's fp32 summation order (see cpu_baseline), has its own key
            out['eval_mde_rel_diff_vs_cpu'] = round(abs(out['eval_mde_m'] - out['cpu_baseline']['eval_mde_m']) / out['cpu_baseline']['eval_mde_m'], 7)
            out['eval_mde_rel_diff_vs_float64_conv_oracle'] = round(abs(out['eval_mde_m'] - out['cpu_baseline']['eval_mde_m_float64_convs'])
                                                                    / out['cpu_baseline']['eval_mde_m_float64_convs'], 7)
            out['speedup_vs_cpu'] = round(out['value'] / out['cpu_baseline']['value'], 1)
        line = json.dumps(out)
    if use_dp:
        dist.destroy_process_group()
    # RCCL writes its version banner with C stdio: block-buffered when stdout is a pipe or a file, i.e. it would come out at process exit,
    # AFTER the JSON line.  Every rank drains the C buffers now; rank 0 prints a moment later, so the JSON line is the last line of the job.
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        if use_dp and world > 1:
            time.sleep(1.0)
        print(line, flush=True)


def _roof_upconv(shapes, a, cfg):
    """The hand-written MFMA kernel of the decoder forward (ss_upconv_sub_fwd_f32 at deconv1, the largest stage) on the
    HBM roofline by its algorithmic bytes (input spikes + output + weights; P never leaves LDS) and on the bf16 MFMA roofline by the
    useful FLOPs of the minimal projection (3 exact bf16 terms); HIP-event time of its launches in the timed region (incl. the tiny
    weight-preparation launch)."""
    key = [k for k in shapes if k[0] == 'upconv_cl_fwd']
    if not key:
        return None
    k = max(key, key=lambda kk: kk[1])
    d = shapes[k]
    if not d['launches']:
        return None
    out_elems = k[1]                                   # NB * H * W * C_out of deconv1
    cout, cin = 32, 64
    src_px = a.batch * a.T * 130 * 173 if out_elems == a.batch * a.T * 260 * 346 * cout else None
    if src_px is None:
        return None
    nbytes = 4 * (src_px * cin + out_elems) + 2 * 3 * 25 * cin * cout
    flops = 2.0 * src_px * cin * 25 * cout * 3
    us = 1e3 * d['ms'] / d['launches']
    if cfg.SUB_FWD and a.dtype == 'f32':
        # round 4: the sub-pixel (merged tap) implicit GEMM on the packed input — 9 instead of 6.25 multiply-adds per output element and input channel
        # (1.44 x the projection's minimum), no P, no gather, no halo
        nbytes = src_px * cin // 4 + 4 * out_elems + 2 * 25 * 27 * 512 * (cin // 16) * (cout // 32)
        issued = 2.0 * (out_elems // cout) * 9 * cin * cout * 3
        return dict(kernel='upconv_sub_fwd_kernel<packed> (deconv1 forward: UpsamplingNearest2d + Conv2d(5) as ONE implicit GEMM over the output pixels with the '
                           'taps that read one source pixel merged — 3 x 3 taps per pixel class; 2-bit packed spikes in, fp32 out, exact bf16x3 products)',
                    bound='hbm', achieved=round(nbytes / us_of(d) / 1e3, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(nbytes / us_of(d) / 1e3 / HBM_PEAK_GBS, 4),
                    avg_launch_us=round(us_of(d), 1), bytes_per_launch=int(nbytes), traffic=_pmc_traffic('upconv_sub'),
                    mfma=dict(achieved_TFLOPs=round(flops / us_of(d) / 1e6, 1), peak_TFLOPs=2500.0, frac=round(flops / us_of(d) / 1e6 / 2500.0, 4),
                              issued_frac=round(issued / us_of(d) / 1e6 / 2500.0, 4),
                              note='frac: useful FLOPs of the minimal projection x 3 exact bf16 terms; issued_frac: what the kernel issues (9 merged taps per output pixel)'),
                    note='latency / issue bound at 3 workgroups per CU (profiles/r04/sub_fwd_ablations.log)',
                    fused=True)
    return None            # projection GEMM + gather (--sub-fwd 0): a library GEMM and an HBM-bound gather, no single kernel to price


def _roof_upconv_bwd(shapes, a, cfg):
    """The decoder backward of deconv1 (the largest stage; /root/reference/network/blocks.py:110-132 under autograd): every launch between the
    stage's incoming gradient and its two outgoing ones (weight preparation, adjoint + data gradient, adjoint + weight gradient, split-K
    reduce), timed with HIP events as one group.  HBM roofline by the stage's algorithmic bytes — g_y + x + g_x + weights, the per-tap tensor
    g_P (5.76 GB at config 3) is NOT algorithmic — and the bf16 MFMA roofline by the useful FLOPs (6 cross terms per MAC of the data gradient,
    3 exact terms per MAC of the weight gradient)."""
    key = [k for k in shapes if k[0] == 'upconv_cl_bwd']
    if not key:
        return None
    k = max(key, key=lambda kk: kk[1])
    d = shapes[k]
    cout, cin = 32, 64
    if not d['launches'] or k[1] != a.batch * a.T * 260 * 346 * cout:
        return None
    src_px = a.batch * a.T * 130 * 173
    nbytes = 4 * (k[1] + 2 * src_px * cin) + 2 * 4 * 25 * cin * cout
    macs = src_px * cin * 25 * cout
    flops = 2.0 * macs * (6 + 3)
    us = 1e3 * d['ms'] / d['launches']
    if a.dtype != 'f32':
        return None                                    # (the byte / term counts below are the fp32 mode's)
    box = bool(cfg.BOX_BWD)
    # counter-measured HBM bytes of the group (profiles/pmc_traffic.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes on tools/pmc_target.py)
    parts = [_pmc_traffic(kk) for kk in (('upconv_boxsum', 'upconv_box_dgrad', 'upconv_box_wgrad') if box else ())]
    traffic = int(sum(parts)) if parts and all(p is not None for p in parts) else None
    return dict(kernel=('upconv_boxsum_kernel + upconv_box_dgrad_kernel<32> + upconv_box_wgrad_kernel<2> (deconv1 backward on the box-sum image: one HBM-bound '
                        'box-sum launch, then both contractions as implicit GEMMs over its three bf16 planes; no g_P)') if box else
                       'upconv_cl_bwd_kernel + fp32 GEMM + spike_wgrad_kernel (deconv1 backward on the per-tap tensor g_P in HBM: --box-bwd 0)',
                bound='hbm', achieved=round(nbytes / us / 1e3, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4),
                avg_launch_us=round(us, 1), bytes_per_launch=int(nbytes), traffic=traffic,
                traffic_note=('sum of the three launches; the three bf16 planes of the box-sum image (1.55 GB at deconv1) are written once and read by both '
                              'contractions with their window halos — 4.8 x the stage\'s algorithmic I/O; round 3\'s g_P-on-chip pair moved 7.0 GB in 3.7 ms, this group 8.9 GB in 3.0 ms: '
                              'neither is HBM-bound, the contractions sit on the LDS / MFMA issue limits (profiles/r04/box_dgrad_ablations.log)') if box else None,
                mfma=dict(achieved_TFLOPs=round(flops / us / 1e6, 1), peak_TFLOPs=2500.0, frac=round(flops / us / 1e6 / 2500.0, 4),
                          note='6 bf16 cross terms per MAC (dense x dense data gradient) + 3 exact terms per MAC (spike x dense weight gradient)'),
                g_P_in_hbm=not box)


def us_of(d):
    return 1e3 * d['ms'] / d['launches']


def _pmc_file(dtype='f32'):
    # the counter passes are per mode: fp32 kernels -> profiles/pmc_traffic.json, 16-bit kernels -> profiles/pmc_traffic_x16.json (tools/pmc_target.py x16)
    return 'pmc_traffic.json' if dtype == 'f32' else 'pmc_traffic_x16.json'


def _pmc_source(dtype='f32'):
    """Where `traffic` comes from and whether it belongs to the kernels this process runs: profiles/pmc_traffic.json records the source hash compiled into the
    library the counter passes loaded; it is compared with the library loaded HERE."""
    from stereospike_amd import _lib
    src = {}
    try:
        src = json.load(open(os.path.join(ROOT, 'profiles', _pmc_file(dtype)))).get('source', {})
    except Exception:
        pass
    here = _lib.source_hash()
    return dict(file='profiles/' + _pmc_file(dtype), how='separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_target.py, committed; not measured in this run',
                measured_on_lib_source_hash=src.get('lib_source_hash'), measured_at_git_head=src.get('git_head'), loaded_lib_source_hash=here,
                tree_source_hash=_lib.tree_source_hash(), matches_loaded_library=(src.get('lib_source_hash') == here) if src.get('lib_source_hash') else None)


def _pmc_traffic(which='neuron_fwd', dtype='f32', T=5, batch=16):
    """HBM bytes per launch of the fused kernel at the dominant launch shape, from the rocprofv3 --pmc passes
    (profiles/pmc_traffic.json resp. pmc_traffic_x16.json, written by profiles/collect_pmc.sh with the guide's gfx950 correction: FETCH_SIZE doubled);
    None if not collected for this mode, or collected at another launch shape (the file records the T and batch of its target)."""
    p = os.path.join(ROOT, 'profiles', _pmc_file(dtype))
    if os.path.exists(p):
        try:
            j = json.load(open(p))
            shape = j.get('shape')
            if shape and (shape.get('T'), shape.get('batch'), shape.get('dtype', dtype)) != (T, batch, dtype):
                return None
            return j.get(which, {}).get('hbm_bytes_per_launch')
        except Exception:
            return None
    return None


if __name__ == '__main__':
    main()
