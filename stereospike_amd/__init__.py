"""stereospike_amd — MI355X-native StereoSpike forward/backward engine.

The spiking-neuron state update of every layer (gain -> IF/LIF/PLIF charge -> Heaviside fire -> hard reset ->
skip add, and its surrogate-gradient backward) is a hand-written HIP kernel for gfx950 behind the C ABI in
include/ss_neuron.h; the convolutions stay on PyTorch-ROCm.  The package mirrors the reference's Python
interface for that path:

    stereospike_amd.network.{SNN_models, ANN_models, blocks, loss, metrics}   <->  /root/reference/network/*
    stereospike_amd.clock_driven.{neuron, surrogate, functional, layer, rnn}  <->  spikingjelly.clock_driven.*

`install_dropin()` registers both under the names the reference's scripts import.
"""
import sys

__version__ = '0.1.0'


def install_dropin(force: bool = False):
    """Make `import network.SNN_models` and `from spikingjelly.clock_driven import neuron, surrogate, functional`
    (train.py:12-23, test.py, calculate_firing_rates.py) resolve to this package.  An already importable real
    `spikingjelly` / `network` is left alone unless force=True."""
    import importlib
    import importlib.util
    import types
    from . import clock_driven, network

    def absent(name):
        if name in sys.modules:
            return False
        try:
            return importlib.util.find_spec(name) is None
        except (ImportError, ValueError):
            return True

    if force or absent('spikingjelly'):
        root = types.ModuleType('spikingjelly')
        root.__path__ = []
        root.clock_driven = clock_driven
        sys.modules['spikingjelly'] = root
        sys.modules['spikingjelly.clock_driven'] = clock_driven
        for sub in ('neuron', 'surrogate', 'functional', 'layer', 'rnn'):
            sys.modules[f'spikingjelly.clock_driven.{sub}'] = getattr(clock_driven, sub)
    if force or absent('network'):
        sys.modules['network'] = network
        for sub in ('SNN_models', 'ANN_models', 'blocks', 'loss', 'metrics'):
            sys.modules[f'network.{sub}'] = importlib.import_module(f'{network.__name__}.{sub}')
