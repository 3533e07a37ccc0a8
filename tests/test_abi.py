"""CPU suite, part 2: the C-ABI library loads without a GPU and exports every symbol include/ss_neuron.h declares;
the product path never touches the oracle and fails loudly off-GPU.  (No compute calls here.)"""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'ss_neuron.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|long long)\s+(ss_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_the_expected_entry_points():
    assert _declared() == sorted(['ss_abi_version', 'ss_neuron_gk_ws_floats', 'ss_neuron_fwd_f32', 'ss_neuron_bwd_f32',
                                  'ss_ipool_fwd_f32', 'ss_ipool_bwd_f32', 'ss_upconv1_fwd_f32', 'ss_upconv1_bwd_f32', 'ss_upconv_cl_fwd_f32', 'ss_upconv_cl_bwd_f32', 'ss_neuron_fwd_x16', 'ss_neuron_bwd_x16', 'ss_voxelize_f64',
                                  'ss_loss_ws_doubles', 'ss_loss_stats_f32', 'ss_loss_grad_f32',
                                  'ss_neuron_bwd_rc_supported', 'ss_neuron_bwd_rc_f32', 'ss_neuron_bwd_rc_x16',
                                  'ss_im2col_cl_bf16', 'ss_split3_bf16', 'ss_neuron_bwd_fork_f32',
                                  'ss_upconv_cl_fwd_x16', 'ss_upconv_cl_bwd_x16', 'ss_upconv_cl_bwd_lowp', 'ss_upconv_cl_bwd_lowp_dt',
                                  # ABI 2
                                  'ss_neuron_bwd_fork_x16', 'ss_neuron_fwd_ex', 'ss_neuron_cnt_ws_words', 'ss_unpack_spikes',
                                  'ss_im2col_cl_bf16_packed',
                                  
                                  # ABI 3
                                  
                                  'ss_wino_dgrad_weights_f32', 'ss_wino_dgrad_input_f32',
                                  'ss_wino_dgrad_output_f32', 'ss_spike_wgrad_supported', 'ss_spike_wgrad_ws_floats', 'ss_spike_wgrad_f32',
                                  
                                  
                                  'ss_spike_conv_fwd_supported', 'ss_spike_conv_fwd_ws_floats',
                                  'ss_spike_conv_fwd_f32', 'ss_dense_conv_s1_fwd_supported', 'ss_dense_conv_s1_fwd_f32', 'ss_gemm6_supported', 'ss_gemm6_ws_floats',
                                  'ss_gemm6_f32', 'ss_gemm6_batched_f32', 'ss_spike_conv_wgrad_supported',
                                  'ss_spike_conv_wgrad_ws_floats', 'ss_spike_conv_wgrad_tr_ws_floats', 'ss_spike_conv_wgrad_f32',
                                  # ABI 4
                                  'ss_neuron_bwd_fork_lr_supported', 'ss_neuron_bwd_fork_lr_f32',
                                  # ABI 6
                                  'ss_conv_s2_dgrad_supported', 'ss_conv_s2_dgrad_ws_floats', 'ss_conv_s2_dgrad_f32',
                                  'ss_dense_conv_s1_wgrad_supported', 'ss_dense_conv_s1_wgrad_ws_floats', 'ss_dense_conv_s1_wgrad_f32',
                                  'ss_head_packed_supported', 'ss_head_wgrad_packed_ws_floats', 'ss_head_proj_packed_f32', 'ss_head_wgrad_packed_f32',
                                  # ABI 7
                                  'ss_upconv_box_elems', 'ss_upconv_boxsum_f32', 'ss_upconv_box_window', 'ss_upconv_box_dgrad_supported', 'ss_upconv_box_dgrad_ws_floats',
                                  'ss_upconv_box_dgrad_f32', 'ss_upconv_box_wgrad_supported', 'ss_upconv_box_wgrad_ws_floats', 'ss_upconv_box_wgrad_f32',
                                  # ABI 8
                                  'ss_wgrad_reduce3_f32', 'ss_spike_conv_fwd_wide_supported',
                                  'ss_upconv_sub_geometry', 'ss_upconv_sub_tall_geometry', 'ss_upconv_sub_supported', 'ss_upconv_sub_wm_elems', 'ss_upconv_sub_prep_f32', 'ss_upconv_sub_fwd_f32',
                                  # ABI 9
                                  'ss_upconv_box_tiles_supported', 'ss_neuron_bwd_fork_lr_x16_supported', 'ss_neuron_bwd_fork_lr_x16', 'ss_dense_conv_s1_fwd_x16',
                                  'ss_dense_conv_s1_wgrad_x16', 'ss_spike_conv_fwd_x16', 'ss_spike_conv_wgrad_x16', 'ss_conv_s2_dgrad_x16', 'ss_im2col_cl_packed_x16',
                                  'ss_im2col_cl_x16', 'ss_upconv_sub_prep_x16', 'ss_upconv_sub_fwd_x16', 'ss_upconv_box_planes_x16', 'ss_upconv_boxsum_x16',
                                  'ss_upconv_box_dgrad_x16', 'ss_upconv_box_wgrad_x16'])


def test_library_loads_and_exports_every_declared_symbol():
    from stereospike_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'stereospike_amd', 'csrc'), 'all'])
    L = _lib.lib()
    for name in _declared():
        assert hasattr(L, name), name
    assert sorted(_lib.EXPORTS) == _declared()
    assert L.ss_abi_version() == _lib.ABI_VERSION == 10
    assert L.ss_neuron_cnt_ws_words(1024) >= 2 * 4
    assert L.ss_neuron_gk_ws_floats() >= 2048
    out = subprocess.check_output(['nm', '-D', '--defined-only', _lib.LIB_PATH]).decode()
    exported = {l.split()[-1] for l in out.splitlines() if ' T ' in l}
    assert set(_declared()) <= exported
    # a gfx950 code object is embedded
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'gfx950' in blob


def test_argument_validation_without_a_gpu():
    """NULL / bad-size arguments are rejected before any HIP call (returns -22, never crashes)."""
    from stereospike_amd import _lib
    L = _lib.lib()
    assert L.ss_neuron_fwd_f32(None, None, None, None, None, None, None, 1, 4, 1.0, 0, 2.0, None, 1.0, 0.0, None) == -22
    assert L.ss_neuron_bwd_f32(None, None, None, None, None, None, None, None, 1, 4, 1.0, 0, 2.0, None, 1.0, 0.0, 0, 2.0, 1,
                               None) == -22
    assert L.ss_ipool_fwd_f32(None, 0, 0, None, None, 1, 4, 4, 1.0, 0.0, None) == -22
    assert L.ss_ipool_bwd_f32(None, None, None, 0, 0, None, 1, 4, 4, 1.0, None) == -22
    d = _lib.FwdDesc()
    assert L.ss_neuron_fwd_ex(None, None) == -22 and L.ss_neuron_fwd_ex(ctypes.byref(d), None) == -22      # NULL / size field not set
    d.size = ctypes.sizeof(_lib.FwdDesc)
    assert L.ss_neuron_fwd_ex(ctypes.byref(d), None) == -22                                                # no buffers
    assert L.ss_unpack_spikes(None, None, 16, 0, 0, 1, None) == -22
    assert L.ss_im2col_cl_bf16_packed(None, None, 1, 4, 4, 8, 3, 1, 1, 4, 4, None) == -22
    # ABI 7: the box-sum backward
    assert L.ss_upconv_boxsum_f32(None, None, None, None, 1, 32, 8, 8, 4, 4, None) == -22
    assert L.ss_upconv_box_dgrad_f32(None, None, None, None, None, 1, None, None, None, 1, 64, 32, 4, 4, 4, 4, None) == -22
    assert L.ss_upconv_box_wgrad_f32(None, None, None, None, None, None, 1, None, None, None, 1, 64, 32, 4, 4, 4, 4, 0, None) == -22
    assert L.ss_upconv_box_elems(2, 32, 10, 12) == 2 * 32 * 3 * 10 * 12 and L.ss_upconv_box_elems(2, 30, 10, 12) == 0
    assert L.ss_upconv_box_dgrad_supported(64, 32, 5, 15, 76) == 1 and L.ss_upconv_box_dgrad_supported(64, 32, 5, 16, 76) == 0
    assert L.ss_upconv_box_dgrad_supported(96, 32, 5, 12, 70) == 0 and L.ss_upconv_box_wgrad_supported(96, 40, 5, 12, 70) == 1
    assert L.ss_upconv_box_dgrad_ws_floats(64, 32) == 2 * 4 * 13 * 3072 // 4 and _lib.upconv_box_window() == (4, 15, 76)
    # ABI 8
    assert L.ss_wgrad_reduce3_f32(None, None, 4, 5, 128, 256, None) == -22
    assert L.ss_spike_conv_fwd_wide_supported(128, 256, 5, 2, 2) == 1 and L.ss_spike_conv_fwd_wide_supported(64, 128, 5, 2, 2) == 0
    assert _lib.upconv_sub_geometry() == dict(block_rows=16, block_cols=32, window_rows=20, window_cols=36, vrec_ints=88, hrec_ints=168, runs=3,
                                              tall_rows=64, tall_window_rows=68, trec_ints=328, narrow_cols=8, window_pixels=720)
    assert L.ss_upconv_sub_supported(64, 32, 5) == 1 and L.ss_upconv_sub_supported(72, 32, 5) == 0 and L.ss_upconv_sub_supported(64, 32, 3) == 0
    assert L.ss_upconv_sub_wm_elems(64, 32, 5, 5) == 25 * 1 * 4 * 27 * 512
    assert L.ss_upconv_sub_prep_f32(None, None, None, None, 64, 32, 5, 5, None) == -22
    assert L.ss_upconv_sub_fwd_f32(None, None, None, None, None, None, None, None, 1, 64, 32, 4, 4, 8, 8, 1, 1, 1, None, 0, 1, None) == -22


def test_product_fails_loudly_on_cpu_tensors():
    from stereospike_amd import _lib
    from stereospike_amd.clock_driven import neuron, surrogate
    from stereospike_amd.network.SNN_models import StereoSpike
    node = neuron.IFNode(surrogate_function=surrogate.ATan())
    with pytest.raises(_lib.SSNeuronError, match='no CPU fallback'):
        node(torch.zeros(2, 3))
    net = StereoSpike(multiply_factor=10., input_size=(32, 40))
    with pytest.raises(_lib.SSNeuronError):
        net(torch.zeros(1, 1, 4, 32, 40))


def test_product_fails_loudly_without_the_library(monkeypatch):
    from stereospike_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libss_neuron.so')
    with pytest.raises(_lib.SSNeuronError, match='not built'):
        _lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'stereospike_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(d, f)
                assert 'ss_ref_' not in src and 'libss_oracle' not in src, os.path.join(d, f)
    out = subprocess.check_output(['ldd', os.path.join(pkg, 'lib', 'libss_neuron.so')]).decode()
    assert 'oracle' not in out


def test_c_caller_compiles_and_links(tmp_path):
    """examples/c_caller.c — a plain C99 program using only include/ss_neuron.h — compiles and links against the built shared library
    (no torch, no C++): the boundary really is a C ABI.  (It is run on the MI355X by tests/test_gpu_01_kernels.py.)"""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, 'stereospike_amd', 'lib')
    if shutil.which('gcc') is None or not os.path.exists('/opt/rocm/lib'):
        pytest.skip('needs gcc and the ROCm runtime libraries')
    exe = str(tmp_path / 'c_caller')
    r = subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I' + os.path.join(root, 'include'), os.path.join(root, 'examples', 'c_caller.c'),
                        '-L' + lib_dir, '-lss_neuron', '-L/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath-link,/opt/rocm/lib', '-o', exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(exe)
