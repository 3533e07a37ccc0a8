#!/usr/bin/env python3
"""Counterpart of the reference's test.py (:100-186) and calculate_firing_rates.py (:92-149): evaluates a checkpoint
on (synthetic) test samples -> mean loss / MDE written to results/checkpoints/test_results.txt, and the per-layer
firing rates averaged over the set (from the counters the fused kernels accumulate)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--checkpoint', default='results/checkpoints/stereospike.pth')
    ap.add_argument('--samples', type=int, default=20)
    ap.add_argument('--T', type=int, default=1)
    ap.add_argument('--multiply-factor', type=float, default=10.)
    ap.add_argument('--out', default='results/checkpoints')
    ap.add_argument('--graph', type=int, default=1, help='1: replay the forward as one HIP graph per sample (engine.GraphedInference)')
    a = ap.parse_args()
    device = torch.device('cuda:0')
    from stereospike_amd import gemm_tuning
    gemm_tuning.enable(0)
    from stereospike_amd.clock_driven import functional, surrogate
    from stereospike_amd.engine import synthetic_batch
    from stereospike_amd.network.loss import Total_Loss
    from stereospike_amd.network.metrics import MeanDepthError
    from stereospike_amd.network.SNN_models import StereoSpike
    net = StereoSpike(surrogate_function=surrogate.ATan(), detach_reset=True, v_threshold=1.0, v_reset=0.,
                      multiply_factor=a.multiply_factor).to(device)
    if os.path.exists(a.checkpoint):
        net.load_state_dict(torch.load(a.checkpoint, map_location=device))      # test.py:84
    loss_module = Total_Loss(alpha=0.5, scale_weights=(1., 1., 1., 1.), penalize_spikes=False)
    net.eval()
    tot_loss, tot_mde, rates = 0.0, 0.0, None
    graphed = None
    with torch.no_grad():
        for i in range(a.samples):
            x, label = synthetic_batch(1, a.T, seed=10 ** 6 + i, device=device)
            if a.graph:
                if graphed is None:
                    from stereospike_amd.engine import GraphedInference
                    graphed = GraphedInference(net, x)                          # reset_net + forward captured once
                pred, spks = graphed(x)                                         # test.py:140,150
            else:
                functional.reset_net(net)                                       # test.py:140
                pred, spks = net.forward_sequence(x)                            # test.py:150
            tot_loss += float(loss_module(pred, label, spks))
            tot_mde += float(MeanDepthError(pred[0], label))
            functional.reset_net(net)
            fr = net.calculate_firing_rates(x)                                  # calculate_firing_rates.py:135
            rates = fr if rates is None else {k: rates[k] + fr[k] for k in fr}
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, 'test_results.txt'), 'w') as f:
        f.write(f'Mean Test Loss: {tot_loss / a.samples}\nMean Test MDE (m): {tot_mde / a.samples}\n')
        f.write(json.dumps({k: float(v) / a.samples for k, v in rates.items()}, indent=1) + '\n')
    print(open(os.path.join(a.out, 'test_results.txt')).read())


if __name__ == '__main__':
    main()
