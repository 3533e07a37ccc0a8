import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from stereospike_amd import miopen_cache  # noqa: E402
# hermetic solver selection: a fresh find-db directory seeded from the TRACKED miopen_db/ only; the compiled-kernel cache (start-up time,
# not numerics) is shared with the in-tree directory bench.py uses
miopen_cache.enable_hermetic()

import torch  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no HIP device in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    # GPU runs exercise the shipped default configuration: GEMM algorithms from the tracked TunableOp record (read-only)
    import torch
    if torch.cuda.is_available():
        from stereospike_amd import gemm_tuning
        gemm_tuning.enable(0)
