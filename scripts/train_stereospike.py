#!/usr/bin/env python3
"""Counterpart of the reference's train.py (/root/reference/train.py:56-358) for the MI355X engine: same model
construction, optimiser (Adam 2e-4, MultiStepLR [8,42,60] x0.5), per-iteration sequence (reset_net -> forward ->
Total_Loss -> backward -> step -> detach -> MDE), validation loop and best-MDE checkpoint
(results/checkpoints/stereospike.pth, same state_dict keys), with the module-level constants turned into flags and a
synthetic-data mode (the MVSEC download, h5py, cv2 are not available here — SURVEY.md §2 row 10).

  single GPU : python scripts/train_stereospike.py --epochs 1 --iters 20
  8 GPUs     : python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_stereospike.py
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def set_random_seed(seed, deterministic=False):   # train.py:35-53
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    if deterministic:                            # train.py:41-50 (cudnn.deterministic, benchmark off, use_deterministic_algorithms)
        from stereospike_amd.engine import set_deterministic
        set_deterministic(True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='StereoSpike', choices=['StereoSpike', 'PLIF', 'LIF'])
    ap.add_argument('--T', type=int, default=1, help='frames per label consumed step by step (reference: 1)')
    ap.add_argument('--batch', type=int, default=1, help='per-GPU batch (reference: 1)')
    ap.add_argument('--epochs', type=int, default=70)
    ap.add_argument('--iters', type=int, default=50, help='synthetic iterations per epoch')
    ap.add_argument('--lr', type=float, default=2e-4)
    ap.add_argument('--multiply-factor', type=float, default=10.)
    ap.add_argument('--penalize-spikes', action='store_true')
    ap.add_argument('--beta', type=float, default=1.)
    ap.add_argument('--out', default='results/checkpoints')
    ap.add_argument('--graph', type=int, default=-1,
                    help='replay each training iteration as one HIP graph (engine.GraphedTrainer): 1 on, 0 off, -1 (default) on for '
                         'single-GPU runs with batch * T <= 16, where an iteration is host-bound (the reference trains at batch 1)')
    ap.add_argument('--deterministic', action='store_true',
                    help='the reference\'s reproducibility switch (train.py:35-50): deterministic MIOpen solvers, no find mode, no TunableOp '
                         'record; two runs are bit-identical (slower)')
    a = ap.parse_args()

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (('WORLD_SIZE', 1), ('RANK', 0), ('LOCAL_RANK', 0)))
    torch.cuda.set_device(local)
    from stereospike_amd import gemm_tuning
    if not a.deterministic:
        gemm_tuning.enable(local)                              # tracked GEMM-algorithm record, read-only
    device = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', device_id=device)
    set_random_seed(2021, a.deterministic)

    from stereospike_amd.clock_driven import surrogate
    from stereospike_amd.dp import GradientAllReducer
    from stereospike_amd.engine import GraphedTrainer, Trainer, synthetic_batch
    from stereospike_amd.network.loss import Total_Loss
    from stereospike_amd.network.metrics import MeanDepthError
    from stereospike_amd.network.SNN_models import StereoSpike, fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike

    if a.model == 'StereoSpike':                 # train.py:118
        net = StereoSpike(surrogate_function=surrogate.ATan(), detach_reset=True, v_threshold=1.0, v_reset=0.,
                          multiply_factor=a.multiply_factor)
    else:                                        # train.py:120
        net = fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike(
            tau=3., v_threshold=1.0, v_reset=0.0, use_plif=(a.model == 'PLIF'), multiply_factor=a.multiply_factor)
    net = net.to(device)
    reducer = GradientAllReducer(net) if world > 1 else None
    loss_module = Total_Loss(alpha=0.5, scale_weights=(1., 1., 1., 1.), penalize_spikes=a.penalize_spikes, beta=a.beta)
    use_graph = (a.graph == 1 or (a.graph == -1 and a.batch * a.T <= 16)) and world == 1
    evaluator = Trainer(net, lr=a.lr, reducer=reducer, loss_module=loss_module)
    trainer = GraphedTrainer(net, lr=a.lr, loss_module=loss_module) if use_graph else evaluator
    os.makedirs(a.out, exist_ok=True)
    log = open(os.path.join(a.out, 'training_logs.txt'), 'w+') if rank == 0 else None

    for epoch in range(a.epochs):
        net.train()
        t0, run_loss, run_mde = time.time(), 0.0, 0.0
        for it in range(a.iters):
            x, label = synthetic_batch(a.batch, a.T, seed=1000 * epoch + it * world + rank, device=device)
            loss, pred = trainer.step(x, label)
            run_loss += float(loss) * x.size(0)
            run_mde += float(MeanDepthError(pred, label))
        frames = a.iters * a.batch * world
        msg = (f'Epoch: {epoch}, Training Loss: {run_loss / a.iters}, Training Mean Depth Error (m): '
               f'{run_mde / a.iters}, Time: {time.time() - t0}, frames/s: {frames / (time.time() - t0):.2f}\n')
        net.eval()
        vl, vm = 0.0, 0.0
        for it in range(max(1, a.iters // 10)):
            x, label = synthetic_batch(1, a.T, seed=10 ** 6 + it, device=device)
            l, m = evaluator.evaluate(x, label)
            vl += float(l)
            vm += float(m)
        n_val = max(1, a.iters // 10)
        msg += f'Epoch: {epoch}, Test Loss: {vl / n_val}, Test Mean Depth Error (m): {vm / n_val}\n'
        if rank == 0:
            print(msg)
            log.write(msg)
            if vm / n_val < net.get_max_accuracy():          # train.py:348-352
                torch.save(net.state_dict(), os.path.join(a.out, 'stereospike.pth'))
                net.update_max_accuracy(vm / n_val)
        net.increment_epoch()
        trainer.sched.step()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
