"""NON-DEFAULT execution layouts of the same network, trajectory-pinned like tests/test_gpu_00_default_path.py — collected LAST: the layouts
whose synapses are MIOpen fp32 convolutions depend on which solver MIOpen's immediate mode picks on the box (the hermetic find-db of
tests/conftest.py seeds the benchmark's 260x346 configurations only), and they are not the shipped path.  Every MIOpen-backed layout runs
with MIOPEN_ENABLE_LOGGING_CMD-style evidence: on a bar violation the test re-runs the failing layout in a subprocess with MIOpen's command
logging on and leaves the log (which solver ran each convolution) under gpurun_out/ next to the parity report.
"""
import subprocess
import sys
import json
import os

import numpy as np
import pytest
import torch

from _models import DEV, pair, state_sha
from _pinned import pinned_parity, rel_l2
from _util import load_npz, synth_input, synth_label

pytestmark = pytest.mark.gpu
import test_gpu_00_default_path as _t00
from test_gpu_00_default_path import check, MARGIN_DEFAULT

@pytest.fixture(autouse=True)
def _own_report(monkeypatch):
    monkeypatch.setattr(_t00, 'REPORT_FILE', 'gpurun_out/parity_report_layouts.json')
    monkeypatch.setattr(_t00, 'REPORT', {} if not hasattr(_own_report, 'rep') else _own_report.rep)
    _own_report.rep = _t00.REPORT


MARGIN_MIOPEN = 5e-4       # layouts whose synapses are MIOpen fp32 convolutions (solver-dependent summation order, <= 1e-5 abs x gain 30)
EXACT = ('all_nhwc_exact_split', 'exact_split_dense_spikes', 'saved_h_no_fork')
LAYOUTS = ['all_nhwc_exact_split', 'exact_split_dense_spikes', 'saved_h_no_fork', 'all_nhwc', 'decoder_nhwc', 'nchw', 'two_op_miopen']


def _layout_config(layout):
    """The engine-configuration overrides (config.EngineConfig fields) of one execution layout."""
    ov = dict(FUSE_UPCONV=layout != 'two_op_miopen',
              DECODER_CHANNELS_LAST=layout in ('decoder_nhwc', 'all_nhwc') + EXACT,
              ENCODER_CHANNELS_LAST=layout in ('all_nhwc',) + EXACT,
              EXACT_SPLIT_GEMM=layout in EXACT,
              PACK_SPIKES=layout != 'exact_split_dense_spikes',
              ASSERT_EXACT_SPLIT=True)
    if layout == 'saved_h_no_fork':
        ov.update(RECOMPUTE_H=False, FORK_OUTPUTS=False)
    return ov


def _run(layout, T):
    H, W = 64, 80
    orc, net = pair('PLIFNet', H, W)
    net.config = net.config.replace(**_layout_config(layout))          # the network owns its configuration: nothing global is touched
    x = synth_input(2, T, 4, 77, H, W, lam=0.08)
    gt = synth_label(2, 78, H, W)
    return pinned_parity(orc, net, x, gt)


@pytest.mark.parametrize('layout,T', [(lay, 5) for lay in LAYOUTS] + [('nchw', 3)])        # (one run-time-T case; the default path has its own: test_gpu_00)
def test_pinned_parity_every_execution_layout(layout, T):
    """The same network through every execution variant of the synapses / neuron kernels: the shipped default; dense instead of packed
    spikes; saved-h backward without forked gradients; NHWC with plain fp32 GEMMs; NHWC decoder only; projected NCHW; the reference's
    two-op up-convs on MIOpen — at T = 5 (compile-time-T kernels), three of them also at T = 3 (run-time-T kernels)."""
    rep = _run(layout, T)
    try:
        check(f'pinned_layout_{layout}_T{T}', rep, MARGIN_DEFAULT if layout in EXACT else MARGIN_MIOPEN)
    except AssertionError:
        if layout not in EXACT:
            _dump_miopen_commands(layout, T)
        raise


def _dump_miopen_commands(layout, T):
    """Re-run one layout in a fresh process with MIOpen's command / solver logging on; stderr -> gpurun_out/miopen_cmd_<layout>_T<T>.log."""
    os.makedirs('gpurun_out', exist_ok=True)
    env = dict(os.environ, MIOPEN_ENABLE_LOGGING_CMD='1', MIOPEN_LOG_LEVEL='5', SS_LAYOUT=layout, SS_LAYOUT_T=str(T))
    with open(f'gpurun_out/miopen_cmd_{layout}_T{T}.log', 'w') as f:
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, stdout=f, stderr=subprocess.STDOUT, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


if __name__ == '__main__':            # the logging re-run of _dump_miopen_commands
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import conftest  # noqa: F401  (hermetic MIOpen user db, as under pytest)
    from stereospike_amd import gemm_tuning
    gemm_tuning.enable(0)
    lay, T = os.environ['SS_LAYOUT'], int(os.environ['SS_LAYOUT_T'])
    rep = _run(lay, T)
    print(json.dumps({k: rep[k] for k in ('grad_rel_l2', 'plif_w', 'layers', 'depth_max_abs_rel')}, indent=1))
