#!/usr/bin/env python3
"""Counterpart of the reference's calculate_firing_rates.py (/root/reference/calculate_firing_rates.py:56-149): load the checkpoint
(:66), walk the (synthetic) test set with batch 1 (:92), reset_net before every sample (:125), accumulate the 15-key dict of
`net.calculate_firing_rates(test_chunks)` (:135-138; densities count_nonzero / numel, SNN_models.py:194-245), detach (:140), average
over the set (:143-144), print and log (:147-148: the reference hands the dict itself to file.write, a TypeError — the dict is written
as JSON here; :100 feeds the right camera as the left one — not reproduced).

On the MI355X the densities come from the counters the fused neuron kernels accumulate while they hold the spikes in registers
(per-lane count -> wavefront reduction -> per-workgroup partial -> fixed second pass; include/ss_neuron.h ss_neuron_fwd_ex), not from
a count_nonzero pass over 14 tensors.  `--T` > 1 reports the rates over all T steps of a stateful sequence (forward_sequence(x, rates)),
`--T 1` (default) is the reference's single-frame evaluation.

    python scripts/calculate_firing_rates.py --samples 20 [--checkpoint results/checkpoints/stereospike.pth]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='StereoSpike', choices=['StereoSpike', 'PLIF', 'LIF'])
    ap.add_argument('--checkpoint', default='results/checkpoints/stereospike.pth')
    ap.add_argument('--samples', type=int, default=20)
    ap.add_argument('--T', type=int, default=1)
    ap.add_argument('--multiply-factor', type=float, default=10.)
    ap.add_argument('--out', default='results/checkpoints')
    a = ap.parse_args()
    device = torch.device('cuda:0')
    from stereospike_amd import gemm_tuning
    gemm_tuning.enable(0)
    from stereospike_amd.clock_driven import functional, surrogate
    from stereospike_amd.engine import synthetic_batch
    from stereospike_amd.network.SNN_models import StereoSpike, fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike
    if a.model == 'StereoSpike':                                                   # calculate_firing_rates.py:63
        net = StereoSpike(surrogate_function=surrogate.ATan(), detach_reset=True, v_threshold=1.0, v_reset=0., multiply_factor=a.multiply_factor)
    else:
        net = fromZero_feedforward_multiscale_tempo_Matt_SpikeFlowNetLike(tau=3., v_threshold=1.0, v_reset=0.0, use_plif=(a.model == 'PLIF'),
                                                                         multiply_factor=a.multiply_factor)
    net = net.to(device)
    if os.path.exists(a.checkpoint):
        net.load_state_dict(torch.load(a.checkpoint, map_location=device))        # :66
    net.eval()
    firing_rates_dict = None
    with torch.no_grad():
        for i in range(a.samples):
            test_chunks, _ = synthetic_batch(1, a.T, seed=10 ** 6 + i, device=device)     # [1, T, 4, 260, 346]: left + right, 2 polarities each
            functional.reset_net(net)                                             # :125
            if a.T == 1:
                out_dict = net.calculate_firing_rates(test_chunks)                # :135
            else:
                out_dict = {}
                net.forward_sequence(test_chunks, out_dict)
            firing_rates_dict = out_dict if firing_rates_dict is None else {k: firing_rates_dict[k] + out_dict[k] for k in out_dict}
            net.detach()                                                          # :140
    firing_rates_dict = {k: float(v) / a.samples for k, v in firing_rates_dict.items()}   # :143-144 (one host sync, here)
    print(firing_rates_dict)
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, 'firing_rates.txt'), 'w') as logfile:
        logfile.write(json.dumps(firing_rates_dict, indent=1) + '\n')


if __name__ == '__main__':
    main()
