#!/usr/bin/env python3
"""GPU diagnostic: layer-by-layer spike mismatch of the product vs the oracle at 64x80."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from _util import ref_network as rn, sj, synth_input
from stereospike_amd.clock_driven import functional, surrogate
from stereospike_amd.network import SNN_models as S
torch.backends.cudnn.allow_tf32 = bool(int(os.environ.get('TF32', '1')))
torch.manual_seed(2021)
orc = rn.build('StereoSpike', multiply_factor=10., surrogate_function=sj.ATan(), input_size=(64, 80))
net = S.StereoSpike(surrogate_function=surrogate.ATan(), multiply_factor=10., input_size=(64, 80))
net.load_state_dict(orc.state_dict()); net.cuda()
x = synth_input(2, 1, 4, 77, 64, 80, lam=0.08)
rates = orc.calculate_firing_rates(x)
functional.reset_net(net)
with torch.no_grad():
    r2 = net.calculate_firing_rates(x.cuda())
for k in rates:
    print(f'{k:12s} oracle {float(rates[k]):.5f} product {float(r2[k]):.5f}')
# direct: bottom layer
sj.reset_net(orc); functional.reset_net(net)
with torch.no_grad():
    a = orc.bottom(x[:, 0]); b = net.bottom(x[:, 0].cuda()).cpu()
    print('bottom spike mismatch', float((a != b).float().mean()))
    ya = orc.bottom[0](x[:, 0]); yb = net.bottom[0](x[:, 0].cuda()).cpu()
    print('bottom conv rel err', float((ya - yb).abs().max() / ya.abs().max()))
    a1 = orc.conv1(a); b1 = net.conv1(a.cuda()).cpu()
    print('conv1 (same input) spike mismatch', float((a1 != b1).float().mean()))
    y1a = orc.conv1[0](a); y1b = net.conv1[0](a.cuda()).cpu()
    print('conv1 conv rel err', float((y1a - y1b).abs().max() / y1a.abs().max()))
