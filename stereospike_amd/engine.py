"""Training / evaluation steps of the hot path as the reference's scripts perform them
(/root/reference/train.py:189-261, test.py:100-173), plus the synthetic MVSEC-shaped data of SURVEY.md §8(d)."""
import torch

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


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class Trainer:
    """reset -> T-step forward -> Total_Loss -> backward (gradient all-reduce overlapped when DP) -> Adam -> detach."""

    def __init__(self, net, lr=2e-4, weight_decay=0.0, reducer=None, loss_module=None):
        self.net = net
        on_gpu = next(net.parameters()).is_cuda
        # one fused multi-tensor launch for the 21-34 parameter tensors instead of ~10 kernels per tensor
        self.opt = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=weight_decay, fused=on_gpu)
        self.sched = torch.optim.lr_scheduler.MultiStepLR(self.opt, milestones=[8, 42, 60], gamma=0.5)  # train.py:127
        self.loss_module = loss_module or Total_Loss(alpha=0.5, scale_weights=(1., 1., 1., 1.), penalize_spikes=False)
        self.reducer = reducer

    def step(self, x, label):
        net = self.net
        functional.reset_net(net)                              # train.py:221
        out = net.forward_sequence(x)
        pred, spks = out if isinstance(out, tuple) else (out, None)
        loss = self.loss_module(pred, label, spks)             # train.py:238
        with torch.autocast('cuda', enabled=False) if loss.is_cuda else _null():
            loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
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
