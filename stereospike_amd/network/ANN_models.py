"""The analog twin of StereoSpike (/root/reference/network/ANN_models.py:9-152): identical topology with biased
encoder convs, an activation and BatchNorm after every conv, and the same non-firing I-neuron read-out pool
(:111,:130-146).  Only that pool touches the neuron engine (one fused launch for the 4 heads); on CPU tensors (BASELINE config 1) the
whole model is plain torch."""
import torch
from torch import nn

from ..clock_driven import neuron, surrogate
from ..fused import ipool
from .blocks import NNConvUpsampling, ResBlock
from .SNN_models import _pyramid


class AnalogNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.max_test_accuracy = float('inf')
        self.epoch = 0

    def increment_epoch(self):
        self.epoch += 1

    def get_max_accuracy(self):
        return self.max_test_accuracy

    def update_max_accuracy(self, new_acc):
        self.max_test_accuracy = new_acc

    def count_trainable_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class StereoSpike_equivalentANN(AnalogNet):
    def __init__(self, activation_function=None, input_size=(260, 346)):
        super().__init__()
        act = nn.Sigmoid() if activation_function is None else activation_function
        sz = _pyramid(input_size)
        self.input_size = tuple(input_size)
        chans = (32, 64, 128, 256, 512)

        def stage(synapse, c):
            return nn.Sequential(synapse, act, nn.BatchNorm2d(c))

        self.bottom = stage(nn.Conv2d(4, 32, kernel_size=5, stride=1, padding=2, bias=True), 32)
        for i in range(1, 5):
            setattr(self, f'conv{i}', stage(nn.Conv2d(chans[i - 1], chans[i], kernel_size=5, stride=2, padding=2,
                                                      bias=True), chans[i]))
        self.bottleneck = nn.Sequential(
            ResBlock(512, connect_function='ADD', bias=True, activation_function=act),
            ResBlock(512, connect_function='ADD', bias=True, activation_function=act))
        for lvl in (4, 3, 2, 1):
            setattr(self, f'deconv{lvl}', stage(NNConvUpsampling(chans[lvl], chans[lvl - 1], kernel_size=5,
                                                                  up_size=sz[lvl - 1]), chans[lvl - 1]))
        for lvl in (4, 3, 2, 1):
            setattr(self, f'predict_depth{lvl}', nn.Sequential(
                NNConvUpsampling(chans[lvl - 1], 1, kernel_size=3, up_size=sz[0], bias=True)))
        self.Ineurons = neuron.IFNode(v_threshold=float('inf'), v_reset=0., surrogate_function=surrogate.ATan())

    def forward(self, x):
        frame = x[:, 0, :, :, :]
        enc = [self.bottom(frame)]
        for i in range(1, 5):
            enc.append(getattr(self, f'conv{i}')(enc[-1]))
        cur = self.bottleneck(enc[4])
        pool = self.Ineurons
        if not frame.is_cuda:
            # BASELINE.json config 1 ("CPU PyTorch ... plumbing, no GPU, no LIF state"): the analog twin has no spiking state and no
            # fused-kernel work — its read-out pool is four fp32 additions — so on CPU tensors it runs as plain torch ops in the
            # reference's own order (ANN_models.py:130-146: two-op up-convs, v = v + head for heads 4, 3, 2, 1).  This is NOT a fallback
            # of the HIP hot path (the spiking models still refuse CPU tensors).
            v, depths = pool.v, []
            for lvl in (4, 3, 2, 1):
                cur = getattr(self, f'deconv{lvl}')(cur) + enc[lvl - 1]
                v = v + getattr(self, f'predict_depth{lvl}')(cur)
                depths.append(v)
            pool.v = v
            return depths[::-1]
        heads = []
        for lvl in (4, 3, 2, 1):
            cur = getattr(self, f'deconv{lvl}')(cur) + enc[lvl - 1]
            heads.append(getattr(self, f'predict_depth{lvl}')[0].forward_projected(cur))
        depth = ipool(torch.stack(heads).unsqueeze(1), 1.0, pool.v_reset, pool._v_init(heads[0]))[0]
        pool.v = depth[3]
        return [depth[3], depth[2], depth[1], depth[0]]

    def set_init_depths_potentials(self, depth_prior):
        self.Ineurons.v = depth_prior


# the reference's own package __init__ and calculate_firing_rates.py import the class under a mis-spelt name
# (network/__init__.py:2, calculate_firing_rates.py:18,64); keep that import working too.
SteroSpike_equivalentANN = StereoSpike_equivalentANN
