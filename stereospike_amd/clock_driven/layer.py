"""`layer.Dropout` is only type-checked by the reference (SNN_models.py:26: `isinstance(m, layer.Dropout)` ->
`m.mask.detach_()`); no shipped model instantiates it.  Kept so that check has a class to test against."""
import torch
import torch.nn as nn


class Dropout(nn.Module):
    """Dropout whose mask is drawn once and kept until reset() (time-invariant mask across steps)."""

    def __init__(self, p: float = 0.5):
        super().__init__()
        assert 0. <= p < 1.
        self.p = p
        self.mask = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.training:
            return x
        if self.mask is None:
            self.mask = (torch.rand_like(x) > self.p).to(x) / (1. - self.p)
        return x * self.mask

    def reset(self):
        self.mask = None
