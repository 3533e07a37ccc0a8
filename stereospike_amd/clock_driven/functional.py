"""`functional.reset_net` (train.py:221,308; test.py:140; calculate_firing_rates.py:125)."""
import torch.nn as nn


def reset_net(net: nn.Module):
    """Every module that has a .reset() goes back to its initial state (v <- v_reset)."""
    for m in net.modules():
        if hasattr(m, 'reset'):
            m.reset()


def detach_net(net: nn.Module):
    """Cut BPTT at the current state (what NeuromorphicNet.detach does, SNN_models.py:22-27)."""
    for m in net.modules():
        if hasattr(m, 'detach'):
            m.detach()
