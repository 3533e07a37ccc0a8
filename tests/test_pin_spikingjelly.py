"""CPU suite: the pin-when-possible hook for the neuron oracle (tools/pin_spikingjelly.py).  With a real `spikingjelly` wheel
(clock_driven namespace) importable, every committed neuron KAT must be bit-identical to the real package — that converts the hot
path's "PARITY UNPINNED" status.  Without one (this image: not installed, no network) the test SKIPS with that message; the tool
itself is validated against the oracle's restatement, which must reproduce the KATs it generated."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import pin_spikingjelly as pin  # noqa: E402


def test_tool_replays_the_kats_through_the_oracle_restatement_bit_for_bit():
    from _util import sj
    n, bad = pin.check(sj.neuron, sj.surrogate)
    assert n == 60 and bad == [], bad


def test_neuron_kats_against_the_real_spikingjelly_if_importable():
    prov = pin.real_provider()
    if prov is None:
        pytest.skip('PARITY UNPINNED: spikingjelly (un-vendored dependency of the reference, requirements.txt:3) is not importable in '
                    'this image; run tools/pin_spikingjelly.py where a clock_driven-era wheel (<= 0.0.0.0.12) is installed')
    neuron, surrogate, ver = prov
    n, bad = pin.check(neuron, surrogate)
    assert bad == [], f'spikingjelly {ver}: {bad}'
