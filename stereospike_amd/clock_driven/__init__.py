"""Provider of the `spikingjelly.clock_driven` names the reference imports (network/blocks.py:8,
network/SNN_models.py:6, network/ANN_models.py:4, train.py:12-13), backed by the fused MI355X kernels.

`stereospike_amd.install_dropin()` registers this package under the upstream module names when the real
third-party package is absent, so `from spikingjelly.clock_driven import neuron, surrogate, functional` in a
user's train.py / test.py keeps working unchanged.
"""
from . import functional, layer, neuron, rnn, surrogate  # noqa: F401
