"""ctypes binding of libss_neuron.so (include/ss_neuron.h) — the only way the product reaches its kernels.

There is NO fallback: if the library is missing, or a tensor is not a contiguous fp32 HIP tensor, the call
raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

import torch  # imported BEFORE the CDLL so that libamdhip64.so.7 resolves to the runtime torch already loaded

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'lib', 'libss_neuron.so')
ABI_VERSION = 10

KIND_IF, KIND_LIF, KIND_PLIF = 0, 1, 2
SG_ATAN, SG_SIGMOID = 0, 1

_lib = None


class SSNeuronError(RuntimeError):
    pass


class FwdDesc(C.Structure):
    """struct ss_neuron_fwd_desc (include/ss_neuron.h)."""
    _fields_ = [('size', C.c_uint), ('act_dtype', C.c_int), ('x_seq', C.c_void_p), ('v_init', C.c_void_p), ('skip_seq', C.c_void_p),
                ('skip_packed', C.c_void_p), ('out_seq', C.c_void_p), ('out_packed', C.c_void_p), ('h_seq', C.c_void_p),
                ('v_last', C.c_void_p), ('nnz', C.c_void_p), ('cnt_ws', C.c_void_p), ('T', C.c_int), ('N', C.c_longlong),
                ('scale', C.c_float), ('kind', C.c_int), ('tau', C.c_float), ('k', C.c_void_p), ('v_th', C.c_float), ('v_reset', C.c_float)]


def lib():
    """Load (once) and return the C-ABI library.  Raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SSNeuronError(
            f'{LIB_PATH} not found: the HIP extension is not built.  Run `python -c "import __graft_entry__ as g; '
            f'g.build()"` or `make -C stereospike_amd/csrc` (hipcc --offload-arch=gfx950).  There is no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    p, i32, i64, f32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float
    L.ss_abi_version.restype = i32
    L.ss_source_hash.restype = C.c_char_p
    L.ss_neuron_gk_ws_floats.restype = i64
    L.ss_neuron_fwd_f32.argtypes = [p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, p]
    L.ss_neuron_bwd_f32.argtypes = [p, p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, i32, f32, i32, p]
    L.ss_neuron_bwd_rc_f32.argtypes = L.ss_neuron_bwd_f32.argtypes
    L.ss_neuron_bwd_rc_f32.restype = i32
    L.ss_neuron_bwd_fork_f32.argtypes = [p, p, p, p, p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, i32, f32, i32, p]
    L.ss_neuron_bwd_fork_f32.restype = i32
    L.ss_neuron_bwd_fork_lr_f32.argtypes = [p, p, p, i32, i32, p, p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, i32, f32, i32, p]
    L.ss_neuron_bwd_fork_lr_f32.restype = i32
    L.ss_neuron_bwd_fork_lr_supported.argtypes = [i32, i64, i32, i32]
    L.ss_neuron_bwd_fork_lr_supported.restype = i32
    L.ss_neuron_bwd_rc_supported.argtypes = [i32]
    L.ss_neuron_bwd_rc_supported.restype = i32
    L.ss_ipool_fwd_f32.argtypes = [p, i64, i64, p, p, i32, i32, i64, f32, f32, p]
    L.ss_ipool_bwd_f32.argtypes = [p, p, p, i64, i64, p, i32, i32, i64, f32, p]
    L.ss_upconv1_fwd_f32.argtypes = [p, p, p, p, p, i64, i32, i32, i32, i32, i32, p]
    L.ss_upconv1_bwd_f32.argtypes = [p, p, p, p, p, p, i64, i32, i32, i32, i32, i32, p]
    L.ss_neuron_fwd_x16.argtypes = [p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, i32, p]
    L.ss_neuron_bwd_x16.argtypes = [p, p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, i32, f32, i32, i32, p]
    L.ss_neuron_fwd_x16.restype = i32
    L.ss_neuron_bwd_x16.restype = i32
    L.ss_neuron_bwd_rc_x16.argtypes = L.ss_neuron_bwd_x16.argtypes
    L.ss_neuron_bwd_rc_x16.restype = i32
    L.ss_voxelize_f64.argtypes = [p, i64, p, p, i32, p, i32, i32, p]
    L.ss_voxelize_f64.restype = i32
    L.ss_loss_ws_doubles.restype = i64
    L.ss_im2col_cl_bf16.argtypes = [p, p, i64, i32, i32, i32, i32, i32, i32, i32, i32, p]
    L.ss_split3_bf16.argtypes = [p, p, i64, i32, p]
    L.ss_im2col_cl_bf16.restype = i32
    L.ss_split3_bf16.restype = i32
    L.ss_wgrad_reduce3_f32.argtypes = [p, p, i32, i32, i32, i32, p]
    L.ss_wgrad_reduce3_f32.restype = i32
    L.ss_loss_stats_f32.argtypes = [p, p, p, p, i64, i32, i32, p]
    L.ss_loss_grad_f32.argtypes = [p, p, p, p, p, i64, i32, i32, p]
    L.ss_loss_stats_f32.restype = i32
    L.ss_loss_grad_f32.restype = i32
    L.ss_upconv_cl_fwd_f32.argtypes = [p, p, p, p, p, i64, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_cl_bwd_f32.argtypes = [p, p, p, p, p, p, i64, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_cl_fwd_x16.argtypes = [p, p, p, p, p, i64, i32, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_cl_bwd_x16.argtypes = [p, p, p, p, p, p, i64, i32, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_cl_bwd_lowp.argtypes = [p, i32, p, p, p, p, p, i64, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_cl_bwd_lowp.restype = i32
    L.ss_upconv_cl_bwd_lowp_dt.argtypes = [p, i32, p, p, p, p, p, i32, i64, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_cl_bwd_lowp_dt.restype = i32
    L.ss_upconv_cl_fwd_x16.restype = i32
    L.ss_upconv_cl_bwd_x16.restype = i32
    L.ss_neuron_bwd_fork_x16.argtypes = [p, p, p, p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, i32, f32, i32, i32, p]
    L.ss_neuron_bwd_fork_x16.restype = i32
    L.ss_neuron_fwd_ex.argtypes = [C.POINTER(FwdDesc), p]
    L.ss_neuron_fwd_ex.restype = i32
    L.ss_neuron_cnt_ws_words.argtypes = [i64]
    L.ss_neuron_cnt_ws_words.restype = i64
    L.ss_unpack_spikes.argtypes = [p, p, i64, i32, i32, i32, p]
    L.ss_unpack_spikes.restype = i32
    L.ss_im2col_cl_bf16_packed.argtypes = [p, p, i64, i32, i32, i32, i32, i32, i32, i32, i32, p]
    L.ss_im2col_cl_bf16_packed.restype = i32
    L.ss_spike_conv_fwd_supported.argtypes = [i32, i32, i32, i32, i32]
    L.ss_spike_conv_fwd_supported.restype = i32
    L.ss_spike_conv_fwd_wide_supported.argtypes = [i32, i32, i32, i32, i32]
    L.ss_spike_conv_fwd_wide_supported.restype = i32
    L.ss_spike_conv_fwd_ws_floats.argtypes = [i32, i32]
    L.ss_spike_conv_fwd_ws_floats.restype = i64
    L.ss_spike_conv_fwd_f32.argtypes = [p, p, p, p, p, i64, i32, i32, i32, i32, p]
    L.ss_spike_conv_fwd_f32.restype = i32
    L.ss_dense_conv_s1_fwd_supported.argtypes = [i32, i32, i32, i32, i32]
    L.ss_dense_conv_s1_fwd_supported.restype = i32
    L.ss_dense_conv_s1_fwd_f32.argtypes = [p, p, p, i64, i32, i32, i32, i32, p]
    L.ss_dense_conv_s1_fwd_f32.restype = i32
    L.ss_conv_s2_dgrad_supported.argtypes = [i32, i32, i32, i32, i32]
    L.ss_conv_s2_dgrad_supported.restype = i32
    L.ss_conv_s2_dgrad_ws_floats.argtypes = [i32, i32]
    L.ss_conv_s2_dgrad_ws_floats.restype = i64
    L.ss_conv_s2_dgrad_f32.argtypes = [p, p, p, p, i64, i32, i32, i32, i32, p]
    L.ss_conv_s2_dgrad_f32.restype = i32
    L.ss_dense_conv_s1_wgrad_supported.argtypes = [i32, i32, i32, i32, i32]
    L.ss_dense_conv_s1_wgrad_supported.restype = i32
    L.ss_dense_conv_s1_wgrad_ws_floats.argtypes = [i32]
    L.ss_dense_conv_s1_wgrad_ws_floats.restype = i64
    L.ss_dense_conv_s1_wgrad_f32.argtypes = [p, p, p, p, i64, i32, i32, i32, i32, i32, p]
    L.ss_dense_conv_s1_wgrad_f32.restype = i32
    L.ss_head_packed_supported.argtypes = [i32, i32, i32]
    L.ss_head_packed_supported.restype = i32
    L.ss_head_wgrad_packed_ws_floats.argtypes = [i32]
    L.ss_head_wgrad_packed_ws_floats.restype = i64
    L.ss_head_proj_packed_f32.argtypes = [p, p, p, i64, i32, p]
    L.ss_head_proj_packed_f32.restype = i32
    L.ss_head_wgrad_packed_f32.argtypes = [p, p, p, p, i64, i32, i32, p]
    L.ss_head_wgrad_packed_f32.restype = i32
    L.ss_upconv_box_elems.argtypes = [i64, i32, i32, i32]
    L.ss_upconv_box_elems.restype = i64
    L.ss_upconv_boxsum_f32.argtypes = [p, p, p, p, i64, i32, i32, i32, i32, i32, p]
    L.ss_upconv_boxsum_f32.restype = i32
    L.ss_upconv_box_dgrad_supported.argtypes = [i32, i32, i32, i32, i32]
    L.ss_upconv_box_dgrad_supported.restype = i32
    L.ss_upconv_box_tiles_supported.argtypes = [i32, i32, i64]
    # ---- ABI 9: the 16-bit activation modes on the engine's own synapse kernels
    L.ss_neuron_bwd_fork_lr_x16_supported.argtypes = [i32, i64, i32, i32]
    L.ss_neuron_bwd_fork_lr_x16_supported.restype = i32
    L.ss_neuron_bwd_fork_lr_x16.argtypes = [p, p, p, i32, i32, p, p, p, p, p, p, p, p, i32, i64, f32, i32, f32, p, f32, f32, i32, f32, i32, i32, p]
    L.ss_neuron_bwd_fork_lr_x16.restype = i32
    L.ss_dense_conv_s1_fwd_x16.argtypes = [p, p, p, i64, i32, i32, i32, i32, i32, p]
    L.ss_dense_conv_s1_fwd_x16.restype = i32
    L.ss_dense_conv_s1_wgrad_x16.argtypes = [p, p, p, p, i64, i32, i32, i32, i32, i32, i32, p]
    L.ss_dense_conv_s1_wgrad_x16.restype = i32
    L.ss_spike_conv_fwd_x16.argtypes = [p, p, p, p, p, i64, i32, i32, i32, i32, i32, p]
    L.ss_spike_conv_fwd_x16.restype = i32
    L.ss_spike_conv_wgrad_x16.argtypes = [p, p, p, p, p, i64, i32, i32, i32, i32, i32, i32, p]
    L.ss_spike_conv_wgrad_x16.restype = i32
    L.ss_conv_s2_dgrad_x16.argtypes = [p, p, p, p, i64, i32, i32, i32, i32, i32, p]
    L.ss_conv_s2_dgrad_x16.restype = i32
    L.ss_im2col_cl_packed_x16.argtypes = [p, p, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, p]
    L.ss_im2col_cl_packed_x16.restype = i32
    L.ss_im2col_cl_x16.argtypes = [p, p, i64, i32, i32, i32, i32, i32, i32, i32, i32, p]
    L.ss_im2col_cl_x16.restype = i32
    L.ss_upconv_sub_prep_x16.argtypes = [p, p, p, p, i32, i32, i32, i32, i32, p]
    L.ss_upconv_sub_prep_x16.restype = i32
    L.ss_upconv_sub_fwd_x16.argtypes = [p, p, p, p, p, p, p, p, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, i32, i32, i32, p]
    L.ss_upconv_sub_fwd_x16.restype = i32
    L.ss_upconv_box_planes_x16.argtypes = [i32]
    L.ss_upconv_box_planes_x16.restype = i32
    L.ss_upconv_boxsum_x16.argtypes = [p, p, p, p, i64, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_boxsum_x16.restype = i32
    L.ss_upconv_box_dgrad_x16.argtypes = [p, p, p, p, p, i32, p, p, p, i64, i32, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_box_dgrad_x16.restype = i32
    L.ss_upconv_box_wgrad_x16.argtypes = [p, p, p, p, p, p, i32, p, p, p, i64, i32, i32, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_box_wgrad_x16.restype = i32
    L.ss_upconv_box_tiles_supported.restype = i32
    L.ss_upconv_box_dgrad_ws_floats.argtypes = [i32, i32]
    L.ss_upconv_box_dgrad_ws_floats.restype = i64
    L.ss_upconv_sub_geometry.argtypes = [C.POINTER(C.c_int)] * 6
    L.ss_upconv_sub_geometry.restype = i32
    L.ss_upconv_sub_supported.argtypes = [i32, i32, i32]
    L.ss_upconv_sub_supported.restype = i32
    L.ss_upconv_sub_wm_elems.argtypes = [i32, i32, i32, i32]
    L.ss_upconv_sub_wm_elems.restype = i64
    L.ss_upconv_sub_prep_f32.argtypes = [p, p, p, p, i32, i32, i32, i32, p]
    L.ss_upconv_sub_prep_f32.restype = i32
    L.ss_upconv_sub_fwd_f32.argtypes = [p, p, p, p, p, p, p, p, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, p, i32, i32, p]
    L.ss_upconv_sub_tall_geometry.argtypes = [C.POINTER(C.c_int)] * 4
    L.ss_upconv_sub_tall_geometry.restype = i32
    L.ss_upconv_sub_fwd_f32.restype = i32
    L.ss_upconv_box_window.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ss_upconv_box_window.restype = i32
    L.ss_upconv_box_dgrad_f32.argtypes = [p, p, p, p, p, i32, p, p, p, i64, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_box_dgrad_f32.restype = i32
    L.ss_upconv_box_wgrad_supported.argtypes = [i32, i32, i32, i32, i32]
    L.ss_upconv_box_wgrad_supported.restype = i32
    L.ss_upconv_box_wgrad_ws_floats.argtypes = [i32, i32, i64, i32, i32]
    L.ss_upconv_box_wgrad_ws_floats.restype = i64
    L.ss_upconv_box_wgrad_f32.argtypes = [p, p, p, p, p, p, i32, p, p, p, i64, i32, i32, i32, i32, i32, i32, i32, p]
    L.ss_upconv_box_wgrad_f32.restype = i32
    L.ss_gemm6_supported.argtypes = [i32, i32]
    L.ss_gemm6_supported.restype = i32
    L.ss_gemm6_ws_floats.argtypes = [i32, i32]
    L.ss_gemm6_ws_floats.restype = i64
    L.ss_gemm6_f32.argtypes = [p, p, p, p, i64, i32, i32, p]
    L.ss_gemm6_f32.restype = i32
    L.ss_gemm6_batched_f32.argtypes = [p, p, p, p, i32, i64, i32, i32, p]
    L.ss_gemm6_batched_f32.restype = i32
    L.ss_spike_conv_wgrad_supported.argtypes = [i32, i32, i32, i32, i32]
    L.ss_spike_conv_wgrad_supported.restype = i32
    L.ss_spike_conv_wgrad_ws_floats.argtypes = [i32, i32, i64, i32, i32]
    L.ss_spike_conv_wgrad_ws_floats.restype = i64
    L.ss_spike_conv_wgrad_tr_ws_floats.argtypes = [i32, i32]
    L.ss_spike_conv_wgrad_tr_ws_floats.restype = i64
    L.ss_spike_conv_wgrad_f32.argtypes = [p, p, p, p, p, i64, i32, i32, i32, i32, i32, p]
    L.ss_spike_conv_wgrad_f32.restype = i32
    L.ss_spike_wgrad_supported.argtypes = [i32, i32]
    L.ss_spike_wgrad_supported.restype = i32
    L.ss_spike_wgrad_ws_floats.argtypes = [i32, i32, i64]
    L.ss_spike_wgrad_ws_floats.restype = i64
    L.ss_spike_wgrad_f32.argtypes = [p, p, p, p, i64, i32, i32, i32, p]
    L.ss_spike_wgrad_f32.restype = i32
    L.ss_wino_dgrad_weights_f32.argtypes = [p, p, i32, i32, p]
    L.ss_wino_dgrad_input_f32.argtypes = [p, p, i64, i32, i32, i32, p]
    L.ss_wino_dgrad_output_f32.argtypes = [p, p, i64, i32, i32, i32, p]
    L.ss_wino_dgrad_weights_f32.restype = L.ss_wino_dgrad_input_f32.restype = L.ss_wino_dgrad_output_f32.restype = i32
    for f in (L.ss_neuron_fwd_f32, L.ss_neuron_bwd_f32, L.ss_ipool_fwd_f32, L.ss_ipool_bwd_f32,
              L.ss_upconv1_fwd_f32, L.ss_upconv1_bwd_f32, L.ss_upconv_cl_fwd_f32, L.ss_upconv_cl_bwd_f32):
        f.restype = i32
    if L.ss_abi_version() != ABI_VERSION:
        raise SSNeuronError(f'libss_neuron.so ABI {L.ss_abi_version()} != expected {ABI_VERSION}; rebuild')
    _lib = L
    return _lib


EXPORTS = ('ss_abi_version', 'ss_neuron_gk_ws_floats', 'ss_neuron_fwd_f32', 'ss_neuron_bwd_f32',
           'ss_ipool_fwd_f32', 'ss_ipool_bwd_f32', 'ss_upconv1_fwd_f32', 'ss_upconv1_bwd_f32',
           'ss_upconv_cl_fwd_f32', 'ss_upconv_cl_bwd_f32', 'ss_neuron_fwd_x16', 'ss_neuron_bwd_x16', 'ss_voxelize_f64',
           'ss_loss_ws_doubles', 'ss_loss_stats_f32', 'ss_loss_grad_f32', 'ss_neuron_bwd_rc_supported', 'ss_neuron_bwd_rc_f32', 'ss_neuron_bwd_rc_x16', 'ss_im2col_cl_bf16', 'ss_split3_bf16', 'ss_wgrad_reduce3_f32', 'ss_neuron_bwd_fork_f32', 'ss_upconv_cl_fwd_x16', 'ss_upconv_cl_bwd_x16', 'ss_upconv_cl_bwd_lowp', 'ss_upconv_cl_bwd_lowp_dt',
           'ss_neuron_bwd_fork_x16', 'ss_neuron_fwd_ex', 'ss_neuron_cnt_ws_words', 'ss_unpack_spikes', 'ss_im2col_cl_bf16_packed',
           
           
           'ss_wino_dgrad_weights_f32', 'ss_wino_dgrad_input_f32', 'ss_wino_dgrad_output_f32',
           'ss_spike_wgrad_supported', 'ss_spike_wgrad_ws_floats', 'ss_spike_wgrad_f32',
           
           
           
           'ss_upconv_sub_geometry', 'ss_upconv_sub_tall_geometry', 'ss_upconv_sub_supported', 'ss_upconv_sub_wm_elems', 'ss_upconv_sub_prep_f32', 'ss_upconv_sub_fwd_f32',
           'ss_spike_conv_fwd_supported', 'ss_spike_conv_fwd_wide_supported', 'ss_spike_conv_fwd_ws_floats', 'ss_spike_conv_fwd_f32',
           'ss_dense_conv_s1_fwd_supported', 'ss_dense_conv_s1_fwd_f32',
           'ss_conv_s2_dgrad_supported', 'ss_conv_s2_dgrad_ws_floats', 'ss_conv_s2_dgrad_f32',
           'ss_dense_conv_s1_wgrad_supported', 'ss_dense_conv_s1_wgrad_ws_floats', 'ss_dense_conv_s1_wgrad_f32',
           'ss_head_packed_supported', 'ss_head_wgrad_packed_ws_floats', 'ss_head_proj_packed_f32', 'ss_head_wgrad_packed_f32',
           'ss_upconv_box_elems', 'ss_upconv_boxsum_f32', 'ss_upconv_box_window', 'ss_upconv_box_dgrad_supported', 'ss_upconv_box_tiles_supported', 'ss_upconv_box_dgrad_ws_floats', 'ss_upconv_box_dgrad_f32',
           'ss_upconv_box_wgrad_supported', 'ss_upconv_box_wgrad_ws_floats', 'ss_upconv_box_wgrad_f32',
           'ss_gemm6_supported', 'ss_gemm6_ws_floats', 'ss_gemm6_f32', 'ss_gemm6_batched_f32',
           'ss_spike_conv_wgrad_supported', 'ss_spike_conv_wgrad_ws_floats', 'ss_spike_conv_wgrad_tr_ws_floats', 'ss_spike_conv_wgrad_f32',
           'ss_neuron_bwd_fork_lr_supported', 'ss_neuron_bwd_fork_lr_f32',
           # ABI 9
           'ss_neuron_bwd_fork_lr_x16_supported', 'ss_neuron_bwd_fork_lr_x16', 'ss_dense_conv_s1_fwd_x16', 'ss_dense_conv_s1_wgrad_x16', 'ss_spike_conv_fwd_x16',
           'ss_spike_conv_wgrad_x16', 'ss_conv_s2_dgrad_x16', 'ss_im2col_cl_packed_x16', 'ss_im2col_cl_x16', 'ss_upconv_sub_prep_x16', 'ss_upconv_sub_fwd_x16',
           'ss_upconv_box_planes_x16', 'ss_upconv_boxsum_x16', 'ss_upconv_box_dgrad_x16', 'ss_upconv_box_wgrad_x16')


def source_hash() -> str:
    """The source hash compiled into the loaded library (csrc/Makefile: sha256 over the .hip units, ss_common.hpp and include/ss_neuron.h, 16 hex digits)."""
    return lib().ss_source_hash().decode()


def tree_source_hash() -> str:
    """The same hash computed from the sources in THIS tree (what `make` would compile in): differs from source_hash() when the library is stale."""
    import hashlib
    import re
    csrc = os.path.join(_PKG, 'csrc')
    mk = open(os.path.join(csrc, 'Makefile')).read()
    units = re.search(r'^UNITS\s*:=\s*(.*)$', mk, re.M).group(1).split()
    # HDRS of the Makefile, in its order (headers next to the units, then $(ROOT)/include/...)
    hdrs = [os.path.join(os.path.dirname(_PKG), h[len('$(ROOT)/'):]) if h.startswith('$(ROOT)/') else os.path.join(csrc, h)
            for h in re.search(r'^HDRS\s*:=\s*(.*)$', mk, re.M).group(1).split()]
    hsh = hashlib.sha256()
    for f in [os.path.join(csrc, u + '.hip') for u in units] + hdrs:
        hsh.update(open(f, 'rb').read())
    return hsh.hexdigest()[:16]


def _ptr(t, name, numel=None):
    """Device pointer of a contiguous fp32 HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise SSNeuronError(f'{name}: expected a torch.Tensor, got {type(t)}')
    if not t.is_cuda:
        raise SSNeuronError(f'{name}: tensor is on {t.device}; the StereoSpike neuron engine runs on the MI355X only '
                            f'(no CPU fallback — use oracle/ for CPU checking)')
    if not t.is_contiguous():
        raise SSNeuronError(f'{name}: tensor must be contiguous')
    if numel is not None and t.numel() != numel:
        raise SSNeuronError(f'{name}: expected {numel} elements, got {t.numel()}')
    return C.c_void_p(t.data_ptr())


def _f32(t, name, numel=None):
    if t is not None and t.dtype != torch.float32:
        raise SSNeuronError(f'{name}: expected float32, got {t.dtype}')
    return _ptr(t, name, numel)


def _stream(ref):
    return C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)


def _check(rc, fn):
    if rc != 0:
        raise SSNeuronError(f'{fn} failed with code {rc} (-22 = invalid argument, -5 = HIP launch error)')


def _require_hip(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        where = t.device if isinstance(t, torch.Tensor) else type(t)
        raise SSNeuronError(f'{name} is on {where}: the StereoSpike neuron engine runs on the MI355X only — there is '
                            f'no CPU fallback (oracle/ holds the CPU checker used by the tests)')


def neuron_fwd(x_seq, v_init, skip_seq, out_seq, h_seq, v_last, nnz, T, N, scale, kind, tau, k, v_th, v_reset):
    _require_hip(x_seq, 'x_seq')
    if nnz is not None and (nnz.dtype != torch.int64 or nnz.numel() != 2):
        raise SSNeuronError('nnz must be an int64 tensor of 2 elements')
    with torch.cuda.device(x_seq.device):
        rc = lib().ss_neuron_fwd_f32(_f32(x_seq, 'x_seq', T * N), _f32(v_init, 'v_init', N),
                                     _f32(skip_seq, 'skip_seq', T * N), _f32(out_seq, 'out_seq', T * N),
                                     _f32(h_seq, 'h_seq', T * N), _f32(v_last, 'v_last', N), _ptr(nnz, 'nnz'),
                                     T, N, scale, kind, tau, _f32(k, 'k', 1), v_th, v_reset, _stream(x_seq))
    _check(rc, 'ss_neuron_fwd_f32')


def neuron_bwd(g_out_seq, g_v_last, h_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
               v_th, v_reset, surrogate, alpha, detach_reset):
    _require_hip(h_seq, 'h_seq')
    with torch.cuda.device(h_seq.device):
        rc = lib().ss_neuron_bwd_f32(_f32(g_out_seq, 'g_out_seq', T * N), _f32(g_v_last, 'g_v_last', N),
                                     _f32(h_seq, 'h_seq', T * N), _f32(v_init, 'v_init', N),
                                     _f32(g_x_seq, 'g_x_seq', T * N), _f32(g_v_init, 'g_v_init', N),
                                     _f32(g_k, 'g_k', 1), _f32(g_k_ws, 'g_k_ws'),
                                     T, N, scale, kind, tau, _f32(k, 'k', 1), v_th, v_reset, surrogate, alpha,
                                     int(bool(detach_reset)), _stream(h_seq))
    _check(rc, 'ss_neuron_bwd_f32')


def neuron_bwd_fork(g_out_seq, g_out2_seq, g_sum_seq, g_v_last, h_seq, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                    v_th, v_reset, surrogate, alpha, detach_reset):
    """Backward with an optional second output gradient added on load; exactly one of h_seq / x_seq is given."""
    _require_hip(g_out_seq, 'g_out_seq')
    with torch.cuda.device(g_out_seq.device):
        rc = lib().ss_neuron_bwd_fork_f32(_f32(g_out_seq, 'g_out_seq', T * N), _f32(g_out2_seq, 'g_out2_seq', T * N),
                                          _f32(g_sum_seq, 'g_sum_seq', T * N),
                                          _f32(g_v_last, 'g_v_last', N), _f32(h_seq, 'h_seq', T * N), _f32(x_seq, 'x_seq', T * N),
                                          _f32(v_init, 'v_init', N), _f32(g_x_seq, 'g_x_seq', T * N), _f32(g_v_init, 'g_v_init', N),
                                          _f32(g_k, 'g_k', 1), _f32(g_k_ws, 'g_k_ws'),
                                          T, N, scale, kind, tau, _f32(k, 'k', 1), v_th, v_reset, surrogate, alpha,
                                          int(bool(detach_reset)), _stream(g_out_seq))
    _check(rc, 'ss_neuron_bwd_fork_f32')


def neuron_bwd_fork_lr_supported(T, N, C, rank):
    return bool(lib().ss_neuron_bwd_fork_lr_supported(int(T), int(N), int(C), int(rank)))


def neuron_bwd_fork_lr(g_out_seq, lr_p, lr_w, g_sum_seq, g_v_last, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                       v_th, v_reset, surrogate, alpha, detach_reset):
    """Recompute-form backward whose second output gradient is the low-rank pair (lr_p [T, N / C, rank], lr_w [rank, C]) of a prediction
    head, formed in registers and added on load; g_out_seq (the dense first gradient) may be None."""
    _require_hip(x_seq, 'x_seq')
    rank, C = int(lr_w.shape[0]), int(lr_w.shape[1])
    with torch.cuda.device(x_seq.device):
        rc = lib().ss_neuron_bwd_fork_lr_f32(_f32(g_out_seq, 'g_out_seq', T * N), _f32(lr_p, 'lr_p', T * (N // C) * rank),
                                             _f32(lr_w, 'lr_w', rank * C), rank, C, _f32(g_sum_seq, 'g_sum_seq', T * N),
                                             _f32(g_v_last, 'g_v_last', N), _f32(x_seq, 'x_seq', T * N),
                                             _f32(v_init, 'v_init', N), _f32(g_x_seq, 'g_x_seq', T * N), _f32(g_v_init, 'g_v_init', N),
                                             _f32(g_k, 'g_k', 1), _f32(g_k_ws, 'g_k_ws'),
                                             T, N, scale, kind, tau, _f32(k, 'k', 1), v_th, v_reset, surrogate, alpha,
                                             int(bool(detach_reset)), _stream(x_seq))
    _check(rc, 'ss_neuron_bwd_fork_lr_f32')


def neuron_bwd_rc_supported(T):
    return bool(lib().ss_neuron_bwd_rc_supported(int(T)))


def neuron_bwd_rc(g_out_seq, g_v_last, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                  v_th, v_reset, surrogate, alpha, detach_reset):
    """Backward that recomputes h from the layer input (no saved h_seq); bit-identical to neuron_bwd."""
    _require_hip(x_seq, 'x_seq')
    with torch.cuda.device(x_seq.device):
        rc = lib().ss_neuron_bwd_rc_f32(_f32(g_out_seq, 'g_out_seq', T * N), _f32(g_v_last, 'g_v_last', N),
                                        _f32(x_seq, 'x_seq', T * N), _f32(v_init, 'v_init', N),
                                        _f32(g_x_seq, 'g_x_seq', T * N), _f32(g_v_init, 'g_v_init', N),
                                        _f32(g_k, 'g_k', 1), _f32(g_k_ws, 'g_k_ws'),
                                        T, N, scale, kind, tau, _f32(k, 'k', 1), v_th, v_reset, surrogate, alpha,
                                        int(bool(detach_reset)), _stream(x_seq))
    _check(rc, 'ss_neuron_bwd_rc_f32')


DT_CODE = {torch.float16: 1, torch.bfloat16: 2}


def _x16(t, name, numel, dtype):
    if t is not None and t.dtype != dtype:
        raise SSNeuronError(f'{name}: expected {dtype}, got {t.dtype}')
    return _ptr(t, name, numel)


def neuron_fwd_x16(x_seq, v_init, skip_seq, out_seq, h_seq, v_last, nnz, T, N, scale, kind, tau, k, v_th, v_reset):
    """16-bit activation I/O (x / skip / out in fp16 or bf16), fp32 membrane and h."""
    _require_hip(x_seq, 'x_seq')
    dt = x_seq.dtype
    if dt not in DT_CODE:
        raise SSNeuronError(f'x_seq: expected float16 or bfloat16, got {dt}')
    with torch.cuda.device(x_seq.device):
        rc = lib().ss_neuron_fwd_x16(_x16(x_seq, 'x_seq', T * N, dt), _f32(v_init, 'v_init', N),
                                     _x16(skip_seq, 'skip_seq', T * N, dt), _x16(out_seq, 'out_seq', T * N, dt),
                                     _f32(h_seq, 'h_seq', T * N), _f32(v_last, 'v_last', N), _ptr(nnz, 'nnz'),
                                     T, N, scale, kind, tau, _f32(k, 'k', 1), v_th, v_reset, DT_CODE[dt], _stream(x_seq))
    _check(rc, 'ss_neuron_fwd_x16')


def neuron_bwd_x16(g_out_seq, g_v_last, h_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                   v_th, v_reset, surrogate, alpha, detach_reset):
    _require_hip(h_seq, 'h_seq')
    dt = g_out_seq.dtype
    if dt not in DT_CODE:
        raise SSNeuronError(f'g_out_seq: expected float16 or bfloat16, got {dt}')
    with torch.cuda.device(h_seq.device):
        rc = lib().ss_neuron_bwd_x16(_x16(g_out_seq, 'g_out_seq', T * N, dt), _f32(g_v_last, 'g_v_last', N),
                                     _f32(h_seq, 'h_seq', T * N), _f32(v_init, 'v_init', N),
                                     _x16(g_x_seq, 'g_x_seq', T * N, dt), _f32(g_v_init, 'g_v_init', N),
                                     _f32(g_k, 'g_k', 1), _f32(g_k_ws, 'g_k_ws'), T, N, scale, kind, tau,
                                     _f32(k, 'k', 1), v_th, v_reset, surrogate, alpha, int(bool(detach_reset)),
                                     DT_CODE[dt], _stream(h_seq))
    _check(rc, 'ss_neuron_bwd_x16')


def neuron_bwd_rc_x16(g_out_seq, g_v_last, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                      v_th, v_reset, surrogate, alpha, detach_reset):
    """16-bit backward that recomputes h from the 16-bit layer input; bit-identical to neuron_bwd_x16."""
    _require_hip(x_seq, 'x_seq')
    dt = g_out_seq.dtype
    if dt not in DT_CODE:
        raise SSNeuronError(f'g_out_seq: expected float16 or bfloat16, got {dt}')
    with torch.cuda.device(x_seq.device):
        rc = lib().ss_neuron_bwd_rc_x16(_x16(g_out_seq, 'g_out_seq', T * N, dt), _f32(g_v_last, 'g_v_last', N),
                                        _x16(x_seq, 'x_seq', T * N, dt), _f32(v_init, 'v_init', N),
                                        _x16(g_x_seq, 'g_x_seq', T * N, dt), _f32(g_v_init, 'g_v_init', N),
                                        _f32(g_k, 'g_k', 1), _f32(g_k_ws, 'g_k_ws'), T, N, scale, kind, tau,
                                        _f32(k, 'k', 1), v_th, v_reset, surrogate, alpha, int(bool(detach_reset)),
                                        DT_CODE[dt], _stream(x_seq))
    _check(rc, 'ss_neuron_bwd_rc_x16')


def ipool_fwd(pd_seq, stride_t, stride_k, v_init, depth_seq, T, K, M, scale, v_reset):
    _require_hip(pd_seq, 'pd_seq')
    with torch.cuda.device(pd_seq.device):
        rc = lib().ss_ipool_fwd_f32(_f32(pd_seq, 'pd_seq'), stride_t, stride_k, _f32(v_init, 'v_init', M),
                                    _f32(depth_seq, 'depth_seq', T * K * M), T, K, M, scale, v_reset, _stream(pd_seq))
    _check(rc, 'ss_ipool_fwd_f32')


def ipool_bwd(g_depth_seq, g_v_last, g_pd_seq, stride_t, stride_k, g_v_init, T, K, M, scale):
    _require_hip(g_depth_seq, 'g_depth_seq')
    with torch.cuda.device(g_depth_seq.device):
        rc = lib().ss_ipool_bwd_f32(_f32(g_depth_seq, 'g_depth_seq', T * K * M), _f32(g_v_last, 'g_v_last', M),
                                    _f32(g_pd_seq, 'g_pd_seq'), stride_t, stride_k, _f32(g_v_init, 'g_v_init', M),
                                    T, K, M, scale, _stream(g_depth_seq))
    _check(rc, 'ss_ipool_bwd_f32')


def _i32(t, name, numel=None):
    if t.dtype != torch.int32:
        raise SSNeuronError(f'{name}: expected int32, got {t.dtype}')
    return _ptr(t, name, numel)


def upconv1_fwd(P, src_y, src_x, bias, out, NB, k, h, w, H, W):
    _require_hip(P, 'P')
    with torch.cuda.device(P.device):
        rc = lib().ss_upconv1_fwd_f32(_f32(P, 'P', NB * k * k * h * w), _i32(src_y, 'src_y', H + k - 1),
                                      _i32(src_x, 'src_x', W + k - 1), _f32(bias, 'bias', 1),
                                      _f32(out, 'out', NB * H * W), NB, k, h, w, H, W, _stream(P))
    _check(rc, 'ss_upconv1_fwd_f32')


def upconv1_bwd(g_out, y_lo, y_hi, x_lo, x_hi, g_P, NB, k, h, w, H, W):
    _require_hip(g_out, 'g_out')
    with torch.cuda.device(g_out.device):
        rc = lib().ss_upconv1_bwd_f32(_f32(g_out, 'g_out', NB * H * W), _i32(y_lo, 'y_lo', h), _i32(y_hi, 'y_hi', h),
                                      _i32(x_lo, 'x_lo', w), _i32(x_hi, 'x_hi', w),
                                      _f32(g_P, 'g_P', NB * k * k * h * w), NB, k, h, w, H, W, _stream(g_out))
    _check(rc, 'ss_upconv1_bwd_f32')


def upconv_cl_fwd(P, src_y, src_x, bias, out, NB, k, C, h, w, H, W):
    _require_hip(P, 'P')
    with torch.cuda.device(P.device):
        rc = lib().ss_upconv_cl_fwd_f32(_f32(P, 'P', NB * k * k * C * h * w), _i32(src_y, 'src_y', H + k - 1),
                                        _i32(src_x, 'src_x', W + k - 1), _f32(bias, 'bias', C),
                                        _f32(out, 'out', NB * H * W * C), NB, k, C, h, w, H, W, _stream(P))
    _check(rc, 'ss_upconv_cl_fwd_f32')


def upconv_cl_bwd(g_out, y_lo, y_hi, x_lo, x_hi, g_P, NB, k, C, h, w, H, W):
    _require_hip(g_out, 'g_out')
    with torch.cuda.device(g_out.device):
        rc = lib().ss_upconv_cl_bwd_f32(_f32(g_out, 'g_out', NB * H * W * C), _i32(y_lo, 'y_lo', h), _i32(y_hi, 'y_hi', h),
                                        _i32(x_lo, 'x_lo', w), _i32(x_hi, 'x_hi', w),
                                        _f32(g_P, 'g_P', NB * k * k * C * h * w), NB, k, C, h, w, H, W, _stream(g_out))
    _check(rc, 'ss_upconv_cl_bwd_f32')


def upconv_cl_fwd_x16(P, src_y, src_x, bias, out, NB, k, C, h, w, H, W):
    _require_hip(P, 'P')
    if out.dtype not in DT_CODE:
        raise SSNeuronError(f'out: expected float16 or bfloat16, got {out.dtype}')
    with torch.cuda.device(P.device):
        rc = lib().ss_upconv_cl_fwd_x16(_f32(P, 'P', NB * k * k * C * h * w), _i32(src_y, 'src_y', H + k - 1),
                                        _i32(src_x, 'src_x', W + k - 1), _f32(bias, 'bias', C),
                                        _x16(out, 'out', NB * H * W * C, out.dtype), NB, k, C, h, w, H, W, DT_CODE[out.dtype], _stream(P))
    _check(rc, 'ss_upconv_cl_fwd_x16')


def upconv_cl_bwd_x16(g_out, y_lo, y_hi, x_lo, x_hi, g_P, NB, k, C, h, w, H, W):
    _require_hip(g_out, 'g_out')
    if g_out.dtype not in DT_CODE:
        raise SSNeuronError(f'g_out: expected float16 or bfloat16, got {g_out.dtype}')
    with torch.cuda.device(g_out.device):
        rc = lib().ss_upconv_cl_bwd_x16(_x16(g_out, 'g_out', NB * H * W * C, g_out.dtype), _i32(y_lo, 'y_lo', h), _i32(y_hi, 'y_hi', h),
                                        _i32(x_lo, 'x_lo', w), _i32(x_hi, 'x_hi', w),
                                        _f32(g_P, 'g_P', NB * k * k * C * h * w), NB, k, C, h, w, H, W, DT_CODE[g_out.dtype], _stream(g_out))
    _check(rc, 'ss_upconv_cl_bwd_x16')


def upconv_cl_bwd_lowp(g_out, y_lo, y_hi, x_lo, x_hi, g_P, NB, k, C, h, w, H, W):
    """Adjoint gather with g_P written in g_P's dtype: bf16 (g_out fp32 / fp16 / bf16) or fp16 (g_out fp16: the fp16 activation mode's own format, ABI 10)."""
    _require_hip(g_out, 'g_out')
    code = 0 if g_out.dtype == torch.float32 else DT_CODE.get(g_out.dtype)
    if code is None:
        raise SSNeuronError(f'g_out: expected float32, float16 or bfloat16, got {g_out.dtype}')
    if g_P.dtype not in DT_CODE:
        raise SSNeuronError(f'g_P: expected bfloat16 or float16, got {g_P.dtype}')
    with torch.cuda.device(g_out.device):
        rc = lib().ss_upconv_cl_bwd_lowp_dt(_ptr(g_out, 'g_out', NB * H * W * C), code, _i32(y_lo, 'y_lo', h), _i32(y_hi, 'y_hi', h),
                                            _i32(x_lo, 'x_lo', w), _i32(x_hi, 'x_hi', w),
                                            _x16(g_P, 'g_P', NB * k * k * C * h * w, g_P.dtype), DT_CODE[g_P.dtype], NB, k, C, h, w, H, W, _stream(g_out))
    _check(rc, 'ss_upconv_cl_bwd_lowp_dt')


def voxelize(events, start, end, counts, H, W):
    _require_hip(events, 'events')
    for t, name, dt in ((events, 'events', torch.float64), (start, 'start', torch.float64), (end, 'end', torch.float64),
                        (counts, 'counts', torch.int32)):
        if t.dtype != dt:
            raise SSNeuronError(f'{name}: expected {dt}, got {t.dtype}')
    E, G = events.shape[0], start.numel()
    with torch.cuda.device(events.device):
        rc = lib().ss_voxelize_f64(_ptr(events, 'events', E * 4), E, _ptr(start, 'start'), _ptr(end, 'end', G), G,
                                   _ptr(counts, 'counts', G * 2 * H * W), H, W, _stream(events))
    _check(rc, 'ss_voxelize_f64')


def im2col_cl_bf16(x, A, NB, h, w, C, k, stride, pad, ho, wo):
    _require_hip(x, 'x')
    with torch.cuda.device(x.device):
        rc = lib().ss_im2col_cl_bf16(_f32(x, 'x', NB * h * w * C), _x16(A, 'A', NB * ho * wo * k * k * C, torch.bfloat16),
                                     NB, h, w, C, k, stride, pad, ho, wo, _stream(x))
    _check(rc, 'ss_im2col_cl_bf16')


def wgrad_reduce3(parts, g_w, S, k, Cin, Cout):
    """parts fp32 [S, k*k*Cin, 3, Cout] (split-K slices x bf16 terms) -> g_w fp32 [Cout, Cin, k, k] (ss_wgrad_reduce3_f32)."""
    with torch.cuda.device(parts.device):
        rc = lib().ss_wgrad_reduce3_f32(_f32(parts, 'parts', S * k * k * Cin * 3 * Cout), _f32(g_w, 'g_w', Cout * Cin * k * k), S, k, Cin, Cout, _stream(parts))
    _check(rc, 'ss_wgrad_reduce3_f32')


def split3_bf16(g, g3, M, N):
    _require_hip(g, 'g')
    with torch.cuda.device(g.device):
        rc = lib().ss_split3_bf16(_f32(g, 'g', M * N), _x16(g3, 'g3', 3 * M * N, torch.bfloat16), M, N, _stream(g))
    _check(rc, 'ss_split3_bf16')


def upconv_boxsum(g_out, bt, NB, Cout, H, W):
    """g_out [NB, H, W, Cout] fp32 -> the box-sum image of the stage as three bf16 planes (int16 tensor [NB, Cout / 8, 3, NVR, NHR, 8]); bt: fused.box_tables(...)."""
    _require_hip(g_out, 'g_out')
    NVR, NHR = bt['NVR'], bt['NHR']
    box = torch.empty((NB, Cout // 8, 3, NVR, NHR, 8), dtype=torch.int16, device=g_out.device)
    assert box.numel() == int(lib().ss_upconv_box_elems(NB, int(Cout), NVR, NHR))
    with torch.cuda.device(g_out.device):
        rc = lib().ss_upconv_boxsum_f32(_f32(g_out, 'g_out', NB * H * W * Cout), _i32(bt['vr'], 'vr', 2 * NVR), _i32(bt['hr'], 'hr', 2 * NHR),
                                        C.c_void_p(box.data_ptr()), NB, int(Cout), H, W, NVR, NHR, _stream(g_out))
    _check(rc, 'ss_upconv_boxsum_f32')
    return box


def upconv_sub_geometry():
    """dict(block_rows, block_cols, window_rows, window_cols, vrec_ints, hrec_ints, runs) of the sub-pixel forward kernel (ss_upconv_sub_geometry)."""
    v = [C.c_int(0) for _ in range(6)]
    runs = lib().ss_upconv_sub_geometry(*[C.byref(a) for a in v])
    t = [C.c_int(0) for _ in range(4)]
    cap = lib().ss_upconv_sub_tall_geometry(*[C.byref(a) for a in t])
    return dict(zip(('block_rows', 'block_cols', 'window_rows', 'window_cols', 'vrec_ints', 'hrec_ints'), (int(a.value) for a in v)), runs=int(runs),
                tall_rows=int(t[0].value), tall_window_rows=int(t[1].value), trec_ints=int(t[2].value), narrow_cols=int(t[3].value), window_pixels=int(cap))


def upconv_sub_supported(Cin, Cout, k):
    return bool(lib().ss_upconv_sub_supported(int(Cin), int(Cout), int(k)))


def upconv_sub_prep(weight, st, Cin, Cout):
    """weight [Cout, Cin, 5, 5] fp32 -> the merged-tap weight fragments of the sub-pixel forward (bf16 tensor; ss_upconv_sub_prep_f32)."""
    _require_hip(weight, 'weight')
    n = int(lib().ss_upconv_sub_wm_elems(int(Cin), int(Cout), int(st['NVC']), int(st['NHC'])))
    if n <= 0:
        raise SSNeuronError(f'ss_upconv_sub_wm_elems: unsupported C_in {Cin} / C_out {Cout}')
    wm = torch.empty(n, dtype=torch.bfloat16, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = lib().ss_upconv_sub_prep_f32(_f32(weight, 'weight', Cout * Cin * 25), _i32(st['vcls'], 'vcls', 8 * st['NVC']), _i32(st['hcls'], 'hcls', 8 * st['NHC']),
                                          _x16(wm, 'wm', n, torch.bfloat16), Cin, Cout, st['NVC'], st['NHC'], _stream(weight))
    _check(rc, 'ss_upconv_sub_prep_f32')
    return wm


def upconv_sub_fwd(x, x_packed, wm, st, out, NB, Cin, Cout, h, w):
    """out [NB, H, W, Cout] = Conv2d(5)(UpsamplingNearest2d(x)) of a spike input (dense fp32 NHWC x, or the 2-bit packed x_packed) in the sub-pixel form."""
    _require_hip(out, 'out')
    H, W = st['H'], st['W']
    counter = torch.empty(1, dtype=torch.int32, device=out.device)           # the launch's tile counter (zeroed by the entry point on the launch stream)
    with torch.cuda.device(out.device):
        rc = lib().ss_upconv_sub_fwd_f32(None if x_packed is not None else _f32(x, 'x', NB * h * w * Cin),
                                         _ptr(x_packed, 'x_packed', None if x_packed is None else NB * h * w * Cin // 16),
                                         _x16(wm, 'wm', int(lib().ss_upconv_sub_wm_elems(int(Cin), int(Cout), int(st['NVC']), int(st['NHC']))), torch.bfloat16),
                                         _i32(st['vblk'], 'vblk', st['NVB'] * st['vrec_ints']), _i32(st['hblk'], 'hblk', st['NHB'] * st['hrec_ints']),
                                         _i32(st['order'], 'order', st['NORD']), _i32(counter, 'counter', 1),
                                         _f32(out, 'out', NB * H * W * Cout), NB, Cin, Cout, h, w, H, W, st['NVB'], st['NHB'], st['NHC'],
                                         _i32(st['tblk'], 'tblk', st['NTB'] * st['trec_ints']) if st['NTB'] else None, st['NTB'], st['NORD'], _stream(out))
    _check(rc, 'ss_upconv_sub_fwd_f32')


def upconv_box_window():
    """(most source rows of a row tile, largest vertical id span of a tile, largest horizontal id span of 32 source columns) the box kernels hold on chip."""
    r, c = C.c_int(0), C.c_int(0)
    n = lib().ss_upconv_box_window(C.byref(r), C.byref(c))
    return int(n), int(r.value), int(c.value)


def _box_tiles_ok(bt, NB, h, w):
    """The launch entry points' own limits (tile tables in LDS, 32-bit pixel indices): an oversized geometry is 'not supported' — the caller falls back to the
    g_P forms — instead of raising in backward.  NB / h / w None: shape-only question."""
    return NB is None or bool(lib().ss_upconv_box_tiles_supported(int(bt['n_row_tiles']), int(w), int(NB) * int(h) * int(w)))


def upconv_box_dgrad_supported(Cin, Cout, k, bt, NB=None, h=None, w=None):
    return bool(lib().ss_upconv_box_dgrad_supported(int(Cin), int(Cout), int(k), int(bt['max_tile_rows']), int(bt['max_cols32']))) and _box_tiles_ok(bt, NB, h, w)


def upconv_box_wgrad_supported(Cin, Cout, k, bt, NB=None, h=None, w=None):
    return bool(lib().ss_upconv_box_wgrad_supported(int(Cin), int(Cout), int(k), int(bt['max_tile_rows']), int(bt['max_cols32']))) and _box_tiles_ok(bt, NB, h, w)


def upconv_box_dgrad(box, weight, bt, g_x, NB, Cin, Cout, h, w):
    """Decoder data gradient g_x [NB, h, w, Cin] from the box-sum planes and the Conv2d weight [Cout, Cin, 5, 5]: six-term bf16 MFMA implicit GEMM."""
    _require_hip(box, 'box')
    ws = torch.empty(int(lib().ss_upconv_box_dgrad_ws_floats(int(Cin), int(Cout))), dtype=torch.float32, device=box.device)
    with torch.cuda.device(box.device):
        rc = lib().ss_upconv_box_dgrad_f32(C.c_void_p(box.data_ptr()), _f32(weight, 'weight', Cout * Cin * 25), _i32(bt['vmap'], 'vmap', 5 * h),
                                           _i32(bt['hmap'], 'hmap', 5 * w), _i32(bt['tile_rows'], 'tile_rows', 4 * bt['n_row_tiles']), bt['n_row_tiles'], _i32(bt['tile_cols'], 'tile_cols'),
                                           _f32(g_x, 'g_x', NB * h * w * Cin), _f32(ws, 'ws'), NB, int(Cin), int(Cout), h, w, bt['NVR'], bt['NHR'], _stream(box))
    _check(rc, 'ss_upconv_box_dgrad_f32')


def upconv_box_wgrad(box, x, x_packed, bt, g_w, NB, Cin, Cout, h, w, accumulate=False):
    """Decoder weight gradient g_w [Cout, Cin, 5, 5] (+)= from the box-sum planes and the stage input x [NB, h, w, Cin] (fp32 spikes, or x_packed: the 2-bit
    packed form): exact bf16x3 MFMA contraction over the source pixels."""
    _require_hip(box, 'box')
    ws = torch.empty(int(lib().ss_upconv_box_wgrad_ws_floats(int(Cin), int(Cout), NB, h, w)), dtype=torch.float32, device=box.device)
    with torch.cuda.device(box.device):
        rc = lib().ss_upconv_box_wgrad_f32(C.c_void_p(box.data_ptr()), _f32(x, 'x', NB * h * w * Cin) if x_packed is None else None,
                                           _i32(x_packed, 'x_packed') if x_packed is not None else None, _i32(bt['vmap'], 'vmap', 5 * h),
                                           _i32(bt['hmap'], 'hmap', 5 * w), _i32(bt['tile_rows'], 'tile_rows', 4 * bt['n_row_tiles']), bt['n_row_tiles'], _i32(bt['tile_cols'], 'tile_cols'),
                                           _f32(g_w, 'g_w', Cout * Cin * 25), _f32(ws, 'ws'), NB, int(Cin), int(Cout), h, w, bt['NVR'], bt['NHR'],
                                           int(bool(accumulate)), _stream(box))
    _check(rc, 'ss_upconv_box_wgrad_f32')


def gemm6_supported(K, N):
    return bool(lib().ss_gemm6_supported(int(K), int(N)))


def gemm6(A, B, C_, R, K, N):
    """C [R, N] = A [R, K] @ B [K, N], dense fp32 operands, six bf16 cross terms on the matrix cores (fp32-product accuracy)."""
    _require_hip(A, 'A')
    ws = torch.empty(int(lib().ss_gemm6_ws_floats(int(K), int(N))), dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        rc = lib().ss_gemm6_f32(_f32(A, 'A', R * K), _f32(B, 'B', K * N), _f32(C_, 'C', R * N), _f32(ws, 'ws'), R, K, N, _stream(A))
    _check(rc, 'ss_gemm6_f32')


def gemm6_batched(A, B, C_, batch, R, K, N):
    """C [batch, R, N] = A [batch, R, K] @ B [batch, K, N] (contiguous), six bf16 cross terms per product."""
    _require_hip(A, 'A')
    ws = torch.empty(batch * int(lib().ss_gemm6_ws_floats(int(K), int(N))), dtype=torch.float32, device=A.device)
    with torch.cuda.device(A.device):
        rc = lib().ss_gemm6_batched_f32(_f32(A, 'A', batch * R * K), _f32(B, 'B', batch * K * N), _f32(C_, 'C', batch * R * N), _f32(ws, 'ws'),
                                        batch, R, K, N, _stream(A))
    _check(rc, 'ss_gemm6_batched_f32')


def spike_conv_fwd_supported(Cin, Cout, k, stride, pad):
    return bool(lib().ss_spike_conv_fwd_supported(int(Cin), int(Cout), int(k), int(stride), int(pad)))


def spike_conv_fwd_wide_supported(Cin, Cout, k, stride, pad):
    return bool(lib().ss_spike_conv_fwd_wide_supported(int(Cin), int(Cout), int(k), int(stride), int(pad)))


def spike_conv_fwd(x, x_packed, weight, out, NB, Cin, Cout, h, w):
    """out [NB, ho, wo, Cout] = conv2d(x, weight, stride 2, pad 2) on a spike input (dense fp32 NHWC x, or the 2-bit packed x_packed): exact
    bf16x3 implicit GEMM on the matrix cores, no im2col."""
    _require_hip(out, 'out')
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    ws = torch.empty(int(lib().ss_spike_conv_fwd_ws_floats(int(Cin), int(Cout))), dtype=torch.float32, device=out.device)
    with torch.cuda.device(out.device):
        rc = lib().ss_spike_conv_fwd_f32(None if x_packed is not None else _f32(x, 'x', NB * h * w * Cin),
                                         _ptr(x_packed, 'x_packed', None if x_packed is None else NB * h * w * Cin // 16),
                                         _f32(weight, 'weight', Cout * Cin * 25), _f32(out, 'out', NB * ho * wo * Cout), _f32(ws, 'ws'),
                                         NB, Cin, Cout, h, w, _stream(out))
    _check(rc, 'ss_spike_conv_fwd_f32')


def dense_conv_s1_fwd_supported(Cin, Cout, k, stride, pad):
    return bool(lib().ss_dense_conv_s1_fwd_supported(int(Cin), int(Cout), int(k), int(stride), int(pad)))


def dense_conv_s1_fwd(x, weight, out, NB, Cin, Cout, h, w):
    """out [NB, h, w, 32] = conv2d(x, weight, stride 1, pad 2), x [NB, h, w, Cin] dense fp32 (any values), Cin in (4, 2): six-term bf16 MFMA
    implicit GEMM, the whole weight in registers."""
    _require_hip(out, 'out')
    with torch.cuda.device(out.device):
        rc = lib().ss_dense_conv_s1_fwd_f32(_f32(x, 'x', NB * h * w * Cin), _f32(weight, 'weight', Cout * Cin * 25), _f32(out, 'out', NB * h * w * Cout),
                                            NB, Cin, Cout, h, w, _stream(out))
    _check(rc, 'ss_dense_conv_s1_fwd_f32')


def conv_s2_dgrad_supported(Cin, Cout, k, stride, pad):
    return bool(lib().ss_conv_s2_dgrad_supported(int(Cin), int(Cout), int(k), int(stride), int(pad)))


def conv_s2_dgrad(g, weight, g_x, NB, Cin, Cout, h, w):
    """g_x [NB, h, w, Cin] = data gradient of conv2d(x, weight, stride 2, pad 2) for the output gradient g [NB, ho, wo, Cout] (dense fp32 NHWC, any
    values): six-term bf16 MFMA implicit GEMM, every element of g_x written."""
    _require_hip(g, 'g')
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    ws = torch.empty(int(lib().ss_conv_s2_dgrad_ws_floats(int(Cin), int(Cout))), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        rc = lib().ss_conv_s2_dgrad_f32(_f32(g, 'g', NB * ho * wo * Cout), _f32(weight, 'weight', Cout * Cin * 25), _f32(g_x, 'g_x', NB * h * w * Cin),
                                        _f32(ws, 'ws'), NB, Cin, Cout, h, w, _stream(g))
    _check(rc, 'ss_conv_s2_dgrad_f32')


def dense_conv_s1_wgrad_supported(Cin, Cout, k, stride, pad):
    return bool(lib().ss_dense_conv_s1_wgrad_supported(int(Cin), int(Cout), int(k), int(stride), int(pad)))


def dense_conv_s1_wgrad(g, x, g_w, NB, Cin, Cout, h, w, accumulate=False):
    """g_w [32, Cin, 5, 5] (+)= weight gradient of conv2d(x, ., stride 1, pad 2): g [NB, h, w, 32], x [NB, h, w, Cin] dense fp32 NHWC (any values):
    six-term bf16 MFMA contraction over the pixels, deterministic."""
    _require_hip(g, 'g')
    ws = torch.empty(int(lib().ss_dense_conv_s1_wgrad_ws_floats(int(Cin))), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        rc = lib().ss_dense_conv_s1_wgrad_f32(_f32(g, 'g', NB * h * w * Cout), _f32(x, 'x', NB * h * w * Cin), _f32(g_w, 'g_w', Cout * Cin * 25), _f32(ws, 'ws'),
                                              NB, Cin, Cout, h, w, int(bool(accumulate)), _stream(g))
    _check(rc, 'ss_dense_conv_s1_wgrad_f32')


def head_packed_supported(Cin, Cout, k):
    return bool(lib().ss_head_packed_supported(int(Cin), int(Cout), int(k)))


def head_proj_packed(x_packed, Wt, P, rows, Cin):
    """P [rows, 9] = x [rows, Cin] @ Wt [Cin, 9] with x given as 2-bit packed spike codes (int32 words, 16 codes each): exact bf16x3 products
    on the matrix cores, fp32 accumulation."""
    _require_hip(P, 'P')
    with torch.cuda.device(P.device):
        rc = lib().ss_head_proj_packed_f32(_ptr(x_packed, 'x_packed', rows * Cin // 16), _f32(Wt, 'Wt', Cin * 9), _f32(P, 'P', rows * 9), rows, Cin, _stream(P))
    _check(rc, 'ss_head_proj_packed_f32')


def head_wgrad_packed(x_packed, g_P, g_Wt, rows, Cin, accumulate=False):
    """g_Wt [Cin, 9] (+)= x^T [Cin, rows] @ g_P [rows, 9] with x given as 2-bit packed spike codes; deterministic."""
    _require_hip(g_P, 'g_P')
    ws = torch.empty(int(lib().ss_head_wgrad_packed_ws_floats(int(Cin))), dtype=torch.float32, device=g_P.device)
    with torch.cuda.device(g_P.device):
        rc = lib().ss_head_wgrad_packed_f32(_ptr(x_packed, 'x_packed', rows * Cin // 16), _f32(g_P, 'g_P', rows * 9), _f32(g_Wt, 'g_Wt', Cin * 9),
                                            _f32(ws, 'ws'), rows, Cin, int(bool(accumulate)), _stream(g_P))
    _check(rc, 'ss_head_wgrad_packed_f32')


def spike_conv_wgrad_supported(Cin, Cout, k, stride, pad):
    return bool(lib().ss_spike_conv_wgrad_supported(int(Cin), int(Cout), int(k), int(stride), int(pad)))


def _spike_conv_wgrad_ws(Cin, Cout, NB, h, w, packed):
    """workspace floats of ss_spike_conv_wgrad_*: the packed-input form needs its partial sums only (ABI 10), not the first form's operand copies"""
    n = int(lib().ss_spike_conv_wgrad_tr_ws_floats(int(Cin), int(Cout))) if packed else 0
    return n if n > 0 else int(lib().ss_spike_conv_wgrad_ws_floats(int(Cin), int(Cout), int(NB), int(h), int(w)))


def spike_conv_wgrad(g, x, g_w, NB, Cin, Cout, h, w, accumulate=False, x_packed=None):
    """g_w [Cout, Cin, 5, 5] (+)= weight gradient of conv2d(x, ., stride 2, pad 2): g [NB, ho, wo, Cout] fp32 NHWC, x [NB, h, w, Cin] fp32 spikes."""
    _require_hip(g, 'g')
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    ws = torch.empty(_spike_conv_wgrad_ws(Cin, Cout, NB, h, w, x_packed is not None), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        rc = lib().ss_spike_conv_wgrad_f32(_f32(g, 'g', NB * ho * wo * Cout), None if x_packed is not None else _f32(x, 'x', NB * h * w * Cin),
                                           _ptr(x_packed, 'x_packed', None if x_packed is None else NB * h * w * Cin // 16),
                                           _f32(g_w, 'g_w', Cout * Cin * 25),
                                           _f32(ws, 'ws'), NB, Cin, Cout, h, w, int(bool(accumulate)), _stream(g))
    _check(rc, 'ss_spike_conv_wgrad_f32')


def spike_wgrad_supported(Cin, N):
    return bool(lib().ss_spike_wgrad_supported(int(Cin), int(N)))


def spike_wgrad(g, x, g_w, R, Cin, N, accumulate=False):
    """g_w [Cin, N] (+)= x^T @ g with x [R, Cin] fp32 spike counts (exact in bf16) and g [R, N] fp32: exact bf16x3 MFMA contraction."""
    _require_hip(g, 'g')
    ws = torch.empty(int(lib().ss_spike_wgrad_ws_floats(int(Cin), int(N), int(R))), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        rc = lib().ss_spike_wgrad_f32(_f32(g, 'g', R * N), _f32(x, 'x', R * Cin), _f32(g_w, 'g_w', Cin * N), _f32(ws, 'ws'), R, Cin, N,
                                      int(bool(accumulate)), _stream(g))
    _check(rc, 'ss_spike_wgrad_f32')


def wino_dgrad_weights(weight, U, Cout, Cin):
    """weight [Cout, Cin, 3, 3] fp32 (contiguous) -> U [16, Cout, Cin]: the transformed, flipped filters of the Winograd data gradient."""
    _require_hip(weight, 'weight')
    with torch.cuda.device(weight.device):
        rc = lib().ss_wino_dgrad_weights_f32(_f32(weight, 'weight', Cout * Cin * 9), _f32(U, 'U', 16 * Cout * Cin), Cout, Cin, _stream(weight))
    _check(rc, 'ss_wino_dgrad_weights_f32')


def wino_tiles(NB, H, W):
    return NB * ((H + 1) // 2) * ((W + 1) // 2)


def wino_dgrad_input(g, V, NB, H, W, C_):
    """g [NB, H, W, C] (NHWC array) -> V [16, T, C], T = wino_tiles(NB, H, W)."""
    _require_hip(g, 'g')
    with torch.cuda.device(g.device):
        rc = lib().ss_wino_dgrad_input_f32(_f32(g, 'g', NB * H * W * C_), _f32(V, 'V', 16 * wino_tiles(NB, H, W) * C_), NB, H, W, C_, _stream(g))
    _check(rc, 'ss_wino_dgrad_input_f32')


def wino_dgrad_output(M, g_in, NB, H, W, C_):
    """M [16, T, C] -> g_in [NB, H, W, C]."""
    _require_hip(M, 'M')
    with torch.cuda.device(M.device):
        rc = lib().ss_wino_dgrad_output_f32(_f32(M, 'M', 16 * wino_tiles(NB, H, W) * C_), _f32(g_in, 'g_in', NB * H * W * C_), NB, H, W, C_, _stream(M))
    _check(rc, 'ss_wino_dgrad_output_f32')


def loss_ws_doubles():
    return int(lib().ss_loss_ws_doubles())


def _f64(t, name, numel=None):
    if t.dtype != torch.float64:
        raise SSNeuronError(f'{name}: expected float64, got {t.dtype}')
    return _ptr(t, name, numel)


def loss_stats(pred, gt, sums, ws, B, H, W):
    _require_hip(pred, 'pred')
    with torch.cuda.device(pred.device):
        rc = lib().ss_loss_stats_f32(_f32(pred, 'pred', B * H * W), _f32(gt, 'gt', B * H * W), _f64(sums, 'sums', 5),
                                     _f64(ws, 'ws'), B, H, W, _stream(pred))
    _check(rc, 'ss_loss_stats_f32')


def loss_grad(pred, gt, sums, coef, g_pred, B, H, W):
    _require_hip(pred, 'pred')
    with torch.cuda.device(pred.device):
        rc = lib().ss_loss_grad_f32(_f32(pred, 'pred', B * H * W), _f32(gt, 'gt', B * H * W), _f64(sums, 'sums', 5),
                                    _f32(coef, 'coef', 2), _f32(g_pred, 'g_pred', B * H * W), B, H, W, _stream(pred))
    _check(rc, 'ss_loss_grad_f32')


def gk_ws_floats():
    return int(lib().ss_neuron_gk_ws_floats())


# ---- ABI 2: descriptor forward (packed spikes, counter partials), forked x16 backward, packed readers ---------------------------
def cnt_ws_words(N):
    return int(lib().ss_neuron_cnt_ws_words(int(N)))


def neuron_fwd_ex(x_seq, v_init, skip_seq, skip_packed, out_seq, out_packed, h_seq, v_last, nnz, cnt_ws, T, N, scale, kind, tau, k,
                  v_th, v_reset):
    """ss_neuron_fwd_ex: x_seq fp32 / fp16 / bf16 [T, N]; packed tensors are int32 [T, N/16] (2 bits per neuron)."""
    _require_hip(x_seq, 'x_seq')
    dt = x_seq.dtype
    code = 0 if dt == torch.float32 else DT_CODE.get(dt)
    if code is None:
        raise SSNeuronError(f'x_seq: expected float32, float16 or bfloat16, got {dt}')
    for t, name in ((skip_seq, 'skip_seq'), (out_seq, 'out_seq')):
        if t is not None and t.dtype != dt:
            raise SSNeuronError(f'{name}: expected {dt}, got {t.dtype}')
    for t, name in ((skip_packed, 'skip_packed'), (out_packed, 'out_packed')):
        if t is not None and (t.dtype != torch.int32 or t.numel() != T * (N // 16) or N % 16):
            raise SSNeuronError(f'{name}: expected int32 [T, N/16] with N % 16 == 0')
    if nnz is not None and (nnz.dtype != torch.int64 or nnz.numel() != 2):
        raise SSNeuronError('nnz must be an int64 tensor of 2 elements')
    if cnt_ws is not None and (cnt_ws.dtype != torch.int32 or cnt_ws.numel() < cnt_ws_words(N)):
        raise SSNeuronError(f'cnt_ws: expected int32 with >= {cnt_ws_words(N)} elements')
    def addr(t, name, numel=None, f32=False):          # validated device address (None -> NULL)
        (_f32 if f32 else _ptr)(t, name, numel)
        return None if t is None else t.data_ptr()
    d = FwdDesc(C.sizeof(FwdDesc), code, addr(x_seq, 'x_seq', T * N), addr(v_init, 'v_init', N, True), addr(skip_seq, 'skip_seq', T * N),
                addr(skip_packed, 'skip_packed'), addr(out_seq, 'out_seq', T * N), addr(out_packed, 'out_packed'),
                addr(h_seq, 'h_seq', T * N, True), addr(v_last, 'v_last', N, True), addr(nnz, 'nnz'), addr(cnt_ws, 'cnt_ws'),
                T, N, scale, kind, tau, addr(k, 'k', 1, True), v_th, v_reset)
    with torch.cuda.device(x_seq.device):
        rc = lib().ss_neuron_fwd_ex(C.byref(d), _stream(x_seq))
    _check(rc, 'ss_neuron_fwd_ex')


def neuron_bwd_fork_x16(g_out_seq, g_out2_seq, g_sum_seq, g_v_last, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                        v_th, v_reset, surrogate, alpha, detach_reset):
    _require_hip(x_seq, 'x_seq')
    dt = g_out_seq.dtype
    if dt not in DT_CODE:
        raise SSNeuronError(f'g_out_seq: expected float16 or bfloat16, got {dt}')
    with torch.cuda.device(x_seq.device):
        rc = lib().ss_neuron_bwd_fork_x16(_x16(g_out_seq, 'g_out_seq', T * N, dt), _x16(g_out2_seq, 'g_out2_seq', T * N, dt),
                                          _x16(g_sum_seq, 'g_sum_seq', T * N, dt), _f32(g_v_last, 'g_v_last', N),
                                          _x16(x_seq, 'x_seq', T * N, dt), _f32(v_init, 'v_init', N),
                                          _x16(g_x_seq, 'g_x_seq', T * N, dt), _f32(g_v_init, 'g_v_init', N),
                                          _f32(g_k, 'g_k', 1), _f32(g_k_ws, 'g_k_ws'), T, N, scale, kind, tau,
                                          _f32(k, 'k', 1), v_th, v_reset, surrogate, alpha, int(bool(detach_reset)),
                                          DT_CODE[dt], _stream(x_seq))
    _check(rc, 'ss_neuron_bwd_fork_x16')


def unpack_spikes(packed, out, n, row_len=0, copies=1):
    """packed int32 [n/16] -> out (fp32 / fp16 / bf16, n * copies elements)."""
    _require_hip(packed, 'packed')
    code = 0 if out.dtype == torch.float32 else DT_CODE.get(out.dtype)
    if code is None or packed.dtype != torch.int32:
        raise SSNeuronError('unpack_spikes: packed must be int32, out float32 / float16 / bfloat16')
    with torch.cuda.device(packed.device):
        rc = lib().ss_unpack_spikes(_ptr(packed, 'packed', n // 16), _ptr(out, 'out', n * copies), n, code, row_len, copies, _stream(packed))
    _check(rc, 'ss_unpack_spikes')


def im2col_cl_bf16_packed(x_packed, A, NB, h, w, C_, k, stride, pad, ho, wo):
    _require_hip(x_packed, 'x_packed')
    if x_packed.dtype != torch.int32:
        raise SSNeuronError('x_packed: expected int32')
    with torch.cuda.device(x_packed.device):
        rc = lib().ss_im2col_cl_bf16_packed(_ptr(x_packed, 'x_packed', NB * h * w * C_ // 16),
                                            _x16(A, 'A', NB * ho * wo * k * k * C_, torch.bfloat16), NB, h, w, C_, k, stride, pad, ho, wo,
                                            _stream(x_packed))
    _check(rc, 'ss_im2col_cl_bf16_packed')


def _dt_of(t, name):
    if t.dtype not in DT_CODE:
        raise SSNeuronError(f'{name}: expected float16 or bfloat16, got {t.dtype}')
    return t.dtype


def neuron_bwd_fork_lr_x16_supported(T, N, C_, rank):
    return bool(lib().ss_neuron_bwd_fork_lr_x16_supported(int(T), int(N), int(C_), int(rank)))


def neuron_bwd_fork_lr_x16(g_out_seq, lr_p, lr_w, g_sum_seq, g_v_last, x_seq, v_init, g_x_seq, g_v_init, g_k, g_k_ws, T, N, scale, kind, tau, k,
                           v_th, v_reset, surrogate, alpha, detach_reset):
    """ss_neuron_bwd_fork_lr_f32 on 16-bit activations: g_out_seq (nullable) / g_sum_seq (nullable) / x_seq / g_x_seq 16-bit, the pair fp32."""
    _require_hip(x_seq, 'x_seq')
    dt = _dt_of(x_seq, 'x_seq')
    rank, C_ = int(lr_w.shape[0]), int(lr_w.shape[1])
    with torch.cuda.device(x_seq.device):
        rc = lib().ss_neuron_bwd_fork_lr_x16(_x16(g_out_seq, 'g_out_seq', T * N, dt), _f32(lr_p, 'lr_p', T * (N // C_) * rank), _f32(lr_w, 'lr_w', rank * C_), rank, C_,
                                             _x16(g_sum_seq, 'g_sum_seq', T * N, dt), _f32(g_v_last, 'g_v_last', N), _x16(x_seq, 'x_seq', T * N, dt),
                                             _f32(v_init, 'v_init', N), _x16(g_x_seq, 'g_x_seq', T * N, dt), _f32(g_v_init, 'g_v_init', N),
                                             _f32(g_k, 'g_k', 1), _f32(g_k_ws, 'g_k_ws'), T, N, scale, kind, tau, _f32(k, 'k', 1), v_th, v_reset, surrogate, alpha,
                                             int(bool(detach_reset)), DT_CODE[dt], _stream(x_seq))
    _check(rc, 'ss_neuron_bwd_fork_lr_x16')


def dense_conv_s1_fwd_x16(x, weight, out, NB, Cin, Cout, h, w):
    """out [NB, h, w, 32] (fp16 / bf16) = conv2d(round(x), round(weight), stride 1, pad 2), x fp32 [NB, h, w, Cin]: one MFMA term, fp32 accumulation."""
    _require_hip(out, 'out')
    dt = _dt_of(out, 'out')
    with torch.cuda.device(out.device):
        rc = lib().ss_dense_conv_s1_fwd_x16(_f32(x, 'x', NB * h * w * Cin), _f32(weight, 'weight', Cout * Cin * 25), _x16(out, 'out', NB * h * w * Cout, dt),
                                            NB, Cin, Cout, h, w, DT_CODE[dt], _stream(out))
    _check(rc, 'ss_dense_conv_s1_fwd_x16')


def dense_conv_s1_wgrad_x16(g, x, g_w, NB, Cin, Cout, h, w, accumulate=False):
    """g_w [32, Cin, 5, 5] fp32 (+)= weight gradient of the first layer from a 16-bit output gradient g and the fp32 input x (rounded once to g's dtype)."""
    _require_hip(g, 'g')
    dt = _dt_of(g, 'g')
    ws = torch.empty(int(lib().ss_dense_conv_s1_wgrad_ws_floats(int(Cin))), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        rc = lib().ss_dense_conv_s1_wgrad_x16(_x16(g, 'g', NB * h * w * Cout, dt), _f32(x, 'x', NB * h * w * Cin), _f32(g_w, 'g_w', Cout * Cin * 25), _f32(ws, 'ws'),
                                              NB, Cin, Cout, h, w, int(bool(accumulate)), DT_CODE[dt], _stream(g))
    _check(rc, 'ss_dense_conv_s1_wgrad_x16')


def spike_conv_fwd_x16(x, x_packed, weight, out, NB, Cin, Cout, h, w):
    """out [NB, ho, wo, Cout] (fp16 / bf16) = conv2d(x, round(weight), stride 2, pad 2) on a spike input (dense 16-bit x of out's dtype, or the packed x_packed)."""
    _require_hip(out, 'out')
    dt = _dt_of(out, 'out')
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    ws = torch.empty(int(lib().ss_spike_conv_fwd_ws_floats(int(Cin), int(Cout))), dtype=torch.float32, device=out.device)
    with torch.cuda.device(out.device):
        rc = lib().ss_spike_conv_fwd_x16(None if x_packed is not None else _x16(x, 'x', NB * h * w * Cin, dt),
                                         _ptr(x_packed, 'x_packed', None if x_packed is None else NB * h * w * Cin // 16),
                                         _f32(weight, 'weight', Cout * Cin * 25), _x16(out, 'out', NB * ho * wo * Cout, dt), _f32(ws, 'ws'),
                                         NB, Cin, Cout, h, w, DT_CODE[dt], _stream(out))
    _check(rc, 'ss_spike_conv_fwd_x16')


def spike_conv_wgrad_x16(g, x, g_w, NB, Cin, Cout, h, w, accumulate=False, x_packed=None):
    """g_w [Cout, Cin, 5, 5] fp32 (+)= weight gradient of conv2d(x, ., stride 2, pad 2): g 16-bit NHWC, x 16-bit spikes of the same dtype or x_packed."""
    _require_hip(g, 'g')
    dt = _dt_of(g, 'g')
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    ws = torch.empty(_spike_conv_wgrad_ws(Cin, Cout, NB, h, w, x_packed is not None), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        rc = lib().ss_spike_conv_wgrad_x16(_x16(g, 'g', NB * ho * wo * Cout, dt), None if x_packed is not None else _x16(x, 'x', NB * h * w * Cin, dt),
                                           _ptr(x_packed, 'x_packed', None if x_packed is None else NB * h * w * Cin // 16),
                                           _f32(g_w, 'g_w', Cout * Cin * 25), _f32(ws, 'ws'), NB, Cin, Cout, h, w, int(bool(accumulate)), DT_CODE[dt], _stream(g))
    _check(rc, 'ss_spike_conv_wgrad_x16')


def conv_s2_dgrad_x16(g, weight, g_x, NB, Cin, Cout, h, w):
    """g_x [NB, h, w, Cin] (16-bit) = data gradient of conv2d(., round(weight), stride 2, pad 2) for the 16-bit output gradient g [NB, ho, wo, Cout]."""
    _require_hip(g, 'g')
    dt = _dt_of(g, 'g')
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    ws = torch.empty(int(lib().ss_conv_s2_dgrad_ws_floats(int(Cin), int(Cout))), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        rc = lib().ss_conv_s2_dgrad_x16(_x16(g, 'g', NB * ho * wo * Cout, dt), _f32(weight, 'weight', Cout * Cin * 25), _x16(g_x, 'g_x', NB * h * w * Cin, dt),
                                        _f32(ws, 'ws'), NB, Cin, Cout, h, w, DT_CODE[dt], _stream(g))
    _check(rc, 'ss_conv_s2_dgrad_x16')


def im2col_cl_packed_x16(x_packed, A, NB, h, w, C_, k, stride, pad, ho, wo):
    """Patch matrix A [NB*ho*wo, k*k*C] (fp16 / bf16) of a 2-bit packed spike tensor."""
    _require_hip(x_packed, 'x_packed')
    dt = _dt_of(A, 'A')
    if x_packed.dtype != torch.int32:
        raise SSNeuronError('x_packed: expected int32')
    with torch.cuda.device(x_packed.device):
        rc = lib().ss_im2col_cl_packed_x16(_ptr(x_packed, 'x_packed', NB * h * w * C_ // 16), _x16(A, 'A', NB * ho * wo * k * k * C_, dt),
                                           NB, h, w, C_, k, stride, pad, ho, wo, DT_CODE[dt], _stream(x_packed))
    _check(rc, 'ss_im2col_cl_packed_x16')


def im2col_cl_x16(x, A, NB, h, w, C_, k, stride, pad, ho, wo):
    """Patch matrix A [NB*ho*wo, k*k*C] of a dense 16-bit NHWC array x [NB, h, w, C] (same dtype)."""
    _require_hip(x, 'x')
    dt = _dt_of(x, 'x')
    with torch.cuda.device(x.device):
        rc = lib().ss_im2col_cl_x16(_x16(x, 'x', NB * h * w * C_, dt), _x16(A, 'A', NB * ho * wo * k * k * C_, dt), NB, h, w, C_, k, stride, pad, ho, wo, _stream(x))
    _check(rc, 'ss_im2col_cl_x16')


def upconv_sub_prep_x16(weight, st, Cin, Cout, dtype):
    """The merged-tap weight fragments of the sub-pixel forward as TWO terms of `dtype` (taps rounded once to it first)."""
    _require_hip(weight, 'weight')
    n = int(lib().ss_upconv_sub_wm_elems(int(Cin), int(Cout), int(st['NVC']), int(st['NHC'])))
    if n <= 0 or dtype not in DT_CODE:
        raise SSNeuronError(f'ss_upconv_sub_prep_x16: unsupported C_in {Cin} / C_out {Cout} / dtype {dtype}')
    wm = torch.empty(n, dtype=dtype, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = lib().ss_upconv_sub_prep_x16(_f32(weight, 'weight', Cout * Cin * 25), _i32(st['vcls'], 'vcls', 8 * st['NVC']), _i32(st['hcls'], 'hcls', 8 * st['NHC']),
                                          _x16(wm, 'wm', n, dtype), Cin, Cout, st['NVC'], st['NHC'], DT_CODE[dtype], _stream(weight))
    _check(rc, 'ss_upconv_sub_prep_x16')
    return wm


def upconv_sub_fwd_x16(x, x_packed, wm, st, out, NB, Cin, Cout, h, w):
    """out [NB, H, W, Cout] (fp16 / bf16) = Conv2d(5)(UpsamplingNearest2d(x)) of a spike input (dense 16-bit x of out's dtype, or x_packed) in the sub-pixel form."""
    _require_hip(out, 'out')
    dt = _dt_of(out, 'out')
    H, W = st['H'], st['W']
    counter = torch.empty(1, dtype=torch.int32, device=out.device)
    with torch.cuda.device(out.device):
        rc = lib().ss_upconv_sub_fwd_x16(None if x_packed is not None else _x16(x, 'x', NB * h * w * Cin, dt),
                                         _ptr(x_packed, 'x_packed', None if x_packed is None else NB * h * w * Cin // 16),
                                         _x16(wm, 'wm', int(lib().ss_upconv_sub_wm_elems(int(Cin), int(Cout), int(st['NVC']), int(st['NHC']))), dt),
                                         _i32(st['vblk'], 'vblk', st['NVB'] * st['vrec_ints']), _i32(st['hblk'], 'hblk', st['NHB'] * st['hrec_ints']),
                                         _i32(st['order'], 'order', st['NORD']), _i32(counter, 'counter', 1),
                                         _x16(out, 'out', NB * H * W * Cout, dt), NB, Cin, Cout, h, w, H, W, st['NVB'], st['NHB'], st['NHC'],
                                         _i32(st['tblk'], 'tblk', st['NTB'] * st['trec_ints']) if st['NTB'] else None, st['NTB'], st['NORD'], DT_CODE[dt], _stream(out))
    _check(rc, 'ss_upconv_sub_fwd_x16')


def upconv_box_planes_x16(dtype):
    return int(lib().ss_upconv_box_planes_x16(DT_CODE[dtype]))


def upconv_boxsum_x16(g_out, bt, NB, Cout, H, W):
    """g_out [NB, H, W, Cout] (fp16 / bf16) -> the box-sum image as upconv_box_planes_x16(dtype) planes of that dtype ([NB, Cout / 8, NP, NVR, NHR, 8])."""
    _require_hip(g_out, 'g_out')
    dt = _dt_of(g_out, 'g_out')
    NVR, NHR, NP = bt['NVR'], bt['NHR'], upconv_box_planes_x16(g_out.dtype)
    box = torch.empty((NB, Cout // 8, NP, NVR, NHR, 8), dtype=dt, device=g_out.device)
    with torch.cuda.device(g_out.device):
        rc = lib().ss_upconv_boxsum_x16(_x16(g_out, 'g_out', NB * H * W * Cout, dt), _i32(bt['vr'], 'vr', 2 * NVR), _i32(bt['hr'], 'hr', 2 * NHR),
                                        C.c_void_p(box.data_ptr()), NB, int(Cout), H, W, NVR, NHR, DT_CODE[dt], _stream(g_out))
    _check(rc, 'ss_upconv_boxsum_x16')
    return box


def upconv_box_dgrad_x16(box, weight, bt, g_x, NB, Cin, Cout, h, w):
    """Decoder data gradient g_x [NB, h, w, Cin] (box's dtype) from the 16-bit box-sum planes and the Conv2d weight (rounded once to that dtype)."""
    _require_hip(box, 'box')
    dt = _dt_of(box, 'box')
    ws = torch.empty(int(lib().ss_upconv_box_dgrad_ws_floats(int(Cin), int(Cout))), dtype=torch.float32, device=box.device)
    with torch.cuda.device(box.device):
        rc = lib().ss_upconv_box_dgrad_x16(C.c_void_p(box.data_ptr()), _f32(weight, 'weight', Cout * Cin * 25), _i32(bt['vmap'], 'vmap', 5 * h),
                                           _i32(bt['hmap'], 'hmap', 5 * w), _i32(bt['tile_rows'], 'tile_rows', 4 * bt['n_row_tiles']), bt['n_row_tiles'], _i32(bt['tile_cols'], 'tile_cols'),
                                           _x16(g_x, 'g_x', NB * h * w * Cin, dt), _f32(ws, 'ws'), NB, int(Cin), int(Cout), h, w, bt['NVR'], bt['NHR'], DT_CODE[dt], _stream(box))
    _check(rc, 'ss_upconv_box_dgrad_x16')


def upconv_box_wgrad_x16(box, x, x_packed, bt, g_w, NB, Cin, Cout, h, w, accumulate=False):
    """Decoder weight gradient g_w [Cout, Cin, 5, 5] fp32 (+)= from the 16-bit box-sum planes and the stage input (dense 16-bit spikes of box's dtype, or x_packed)."""
    _require_hip(box, 'box')
    dt = _dt_of(box, 'box')
    ws = torch.empty(int(lib().ss_upconv_box_wgrad_ws_floats(int(Cin), int(Cout), NB, h, w)), dtype=torch.float32, device=box.device)
    with torch.cuda.device(box.device):
        rc = lib().ss_upconv_box_wgrad_x16(C.c_void_p(box.data_ptr()), _x16(x, 'x', NB * h * w * Cin, dt) if x_packed is None else None,
                                           _i32(x_packed, 'x_packed') if x_packed is not None else None, _i32(bt['vmap'], 'vmap', 5 * h),
                                           _i32(bt['hmap'], 'hmap', 5 * w), _i32(bt['tile_rows'], 'tile_rows', 4 * bt['n_row_tiles']), bt['n_row_tiles'], _i32(bt['tile_cols'], 'tile_cols'),
                                           _f32(g_w, 'g_w', Cout * Cin * 25), _f32(ws, 'ws'), NB, int(Cin), int(Cout), h, w, bt['NVR'], bt['NHR'],
                                           int(bool(accumulate)), DT_CODE[dt], _stream(box))
    _check(rc, 'ss_upconv_box_wgrad_x16')
