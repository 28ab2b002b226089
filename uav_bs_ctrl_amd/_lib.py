"""ctypes binding of ``libuavgnn.so`` - the C-ABI declared in ``include/uavgnn.h``.

PyTorch only supplies device memory and the current HIP stream; every call passes raw pointers and sizes.
There is NO CPU fallback: if the shared library is missing, or a tensor is not on the GPU, the call raises.
"""
from __future__ import annotations

import ctypes
import os

import torch as th  # imported first on purpose: libuavgnn must bind to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# UAVGNN_LIB: another build of the same library (kernel A/Bs of the probes: a variant compiled with different -D switches)
LIB_PATH = os.environ.get("UAVGNN_LIB") or os.path.join(_HERE, "csrc", "libuavgnn.so")

_c_fp = ctypes.c_void_p   # const float* / float*
_c_ip = ctypes.c_void_p   # const int32_t*
_c_int = ctypes.c_int
_c_f32 = ctypes.c_float
_c_st = ctypes.c_void_p   # hipStream_t

# name -> (restype, argtypes); mirrors include/uavgnn.h one to one (checked by tests/test_cabi_symbols.py)
SIGNATURES = {
    "uavgnn_version": (_c_int, []),
    "uavgnn_strerror": (ctypes.c_char_p, [_c_int]),
    "uavgnn_gatv2_fwd": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_ip, _c_ip, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp,
                                  _c_fp, _c_fp, _c_int, _c_int, _c_f32, _c_fp, _c_int, _c_fp, _c_st]),
    "uavgnn_gatv2_fwd_valu": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_ip, _c_ip, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp,
                                       _c_fp, _c_fp, _c_int, _c_int, _c_f32, _c_fp, _c_int, _c_fp, _c_st]),
    "uavgnn_gatv2_fwd_mfma": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_ip, _c_ip, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp,
                                       _c_fp, _c_fp, _c_int, _c_int, _c_f32, _c_fp, _c_int, _c_fp, _c_st]),
    "uavgnn_gatv2_hetero_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "uavgnn_gatv2_hetero_fwd": (_c_int, [_c_fp, _c_int, _c_ip, _c_ip, _c_fp, _c_int, _c_ip, _c_fp, _c_int,
                                         ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _c_int, _c_int,
                                         _c_f32, _c_fp, _c_int, _c_fp, _c_fp, _c_st]),
    "uavgnn_gatv2_hetero_fwd_phases": (_c_int, [_c_fp, _c_int, _c_ip, _c_ip, _c_fp, _c_int, _c_ip, _c_fp, _c_int,
                                                ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _c_int,
                                                _c_int, _c_f32, _c_fp, _c_int, _c_fp, _c_fp, _c_int, _c_st]),
    "uavgnn_gatv2_hetero_image_bytes": (ctypes.c_size_t, []),
    "uavgnn_gatv2_hetero_prepare": (_c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _c_int, _c_int,
                                             _c_f32, _c_fp, _c_st]),
    "uavgnn_gatv2_hetero_fwd_image": (_c_int, [_c_fp, _c_int, _c_ip, _c_ip, _c_fp, _c_int, _c_ip, _c_fp, _c_int,
                                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _c_int,
                                               _c_int, _c_f32, _c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_int, _c_st]),
    "uavgnn_gatv2_hetero_fwd_rowmax": (_c_int, [_c_fp, _c_int, _c_ip, _c_ip, _c_fp, _c_int, _c_ip, _c_fp, _c_int,
                                                ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _c_int,
                                                _c_int, _c_f32, _c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_st]),
    "uavgnn_gatv2_bwd_workspace_bytes": (ctypes.c_size_t, [_c_int, _c_int]),
    "uavgnn_gatv2_bwd": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_ip, _c_ip, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp,
                                  _c_int, _c_int, _c_f32, _c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp,
                                  _c_fp, _c_fp, _c_fp, ctypes.c_void_p, ctypes.c_size_t, _c_st]),
    "uavgnn_gatv2_bwd_generic": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_ip, _c_ip, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp,
                                  _c_int, _c_int, _c_f32, _c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp,
                                  _c_fp, _c_fp, _c_fp, ctypes.c_void_p, ctypes.c_size_t, _c_st]),
    "uavgnn_talk_attn_fwd": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_ip, _c_ip,
                                      _c_int, _c_f32, _c_fp, _c_int, _c_fp, _c_fp, _c_int, _c_int, _c_st]),
    "uavgnn_talk_attn_bwd": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_ip, _c_ip,
                                      _c_ip, _c_ip, _c_ip, _c_int, _c_f32, _c_fp, _c_fp, _c_int, _c_fp, _c_int,
                                      _c_fp, _c_int, _c_fp, _c_int, _c_fp, _c_st]),
    "uavgnn_disc_comm_fwd": (_c_int, [_c_fp, _c_int, _c_fp, _c_fp, _c_int, _c_ip, _c_ip, _c_int, _c_f32, _c_fp, _c_int, _c_fp,
                                      _c_ip, _c_st]),
    "uavgnn_gumbel_noise": (_c_int, [_c_fp, ctypes.c_longlong, _c_int, _c_fp, _c_st]),
    "uavgnn_disc_comm_bwd": (_c_int, [_c_fp, _c_int, _c_fp, _c_ip, _c_int, _c_ip, _c_ip, _c_ip, _c_int, _c_f32, _c_fp,
                                      _c_int, _c_st]),
    "uavgnn_obs_degrees": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_ip, _c_ip, _c_st]),
    "uavgnn_obs_compact": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_ip, _c_ip, _c_fp, _c_fp,
                                    _c_st]),
    "uavgnn_talk_degrees": (_c_int, [_c_fp, _c_int, _c_int, _c_f32, _c_ip, _c_ip, _c_st]),
    "uavgnn_offsets_scan4": (_c_int, [_c_ip, _c_int, _c_ip, _c_ip, _c_int, _c_ip, _c_ip, _c_int, _c_ip, _c_ip, _c_int, _c_ip, _c_st]),
    "uavgnn_talk_compact": (_c_int, [_c_fp, _c_int, _c_int, _c_f32, _c_ip, _c_ip, _c_ip, _c_ip, _c_st]),
    "uavgnn_build_graph_small_max_agents": (_c_int, []),
    "uavgnn_build_graph_small": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_f32, _c_ip, _c_ip,
                                          _c_ip, _c_fp, _c_fp, _c_ip, _c_ip, _c_ip, _c_st]),
    "uavgnn_degree_order_workspace_bytes": (ctypes.c_size_t, [_c_int]),
    "uavgnn_degree_order": (_c_int, [_c_ip, _c_int, _c_ip, ctypes.c_void_p, ctypes.c_size_t, _c_st]),
    "uavgnn_csc_transpose_workspace_bytes": (ctypes.c_size_t, [_c_int]),
    "uavgnn_csc_transpose": (_c_int, [_c_ip, _c_ip, _c_int, _c_int, _c_ip, _c_ip, _c_ip, ctypes.c_void_p,
                                      ctypes.c_size_t, _c_st]),
    "uavgnn_csc_transpose_env": (_c_int, [_c_ip, _c_ip, _c_ip, _c_int, _c_int, _c_ip, _c_ip, _c_ip, _c_st]),
    "uavgnn_talk_attn_env_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "uavgnn_tarmac_msg_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int]),
    "uavgnn_tarmac_msg_weight_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int]),
    "uavgnn_tarmac_msg_planes_bytes": (ctypes.c_longlong, [_c_int, _c_int, _c_int]),
    "uavgnn_tarmac_msg_prepare": (_c_int, [_c_fp, _c_int, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_st]),
    "uavgnn_tarmac_msg_fwd_dbg": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_fp, _c_int, _c_int,
                                           _c_ip, _c_ip, _c_f32, _c_fp, _c_int, _c_fp, _c_fp, _c_int, _c_fp, _c_int, ctypes.c_void_p,
                                           _c_int, _c_st]),
    "uavgnn_tarmac_msg_fwd": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_fp, _c_int, _c_int,
                                       _c_ip, _c_ip, _c_f32, _c_fp, _c_int, _c_fp, _c_fp, _c_int, _c_fp, _c_int, ctypes.c_void_p,
                                       _c_st]),
    "uavgnn_talk_attn_env_fwd": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_ip, _c_ip,
                                          _c_ip, _c_int, _c_int, _c_f32, _c_fp, _c_int, _c_fp, _c_fp, _c_int, _c_int,
                                          _c_st]),
    "uavgnn_talk_attn_env_bwd": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_ip, _c_ip,
                                          _c_ip, _c_int, _c_int, _c_f32, _c_fp, _c_fp, _c_int, _c_fp, _c_int, _c_fp,
                                          _c_int, _c_fp, _c_int, _c_st]),
    "uavgnn_eps_greedy": (_c_int, [_c_fp, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp, _c_f32, ctypes.c_void_p, _c_st]),
    "uavgnn_eps_greedy_dev": (_c_int, [_c_fp, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp, _c_fp, ctypes.c_void_p, _c_st]),
    "uavgnn_colsum_acc": (_c_int, [_c_fp, ctypes.c_longlong, _c_int, _c_int, _c_fp, _c_int, _c_st]),
    "uavgnn_relu_bwd_colsum": (_c_int, [_c_fp, ctypes.c_longlong, _c_fp, ctypes.c_longlong, _c_fp, ctypes.c_longlong, _c_int, _c_int, _c_fp,
                               _c_int, _c_st]),
    "uavgnn_relu_bwd_colsum_rowmax": (_c_int, [_c_fp, ctypes.c_longlong, _c_fp, ctypes.c_longlong, _c_fp, ctypes.c_longlong, _c_int, _c_int,
                                               _c_fp, _c_int, _c_fp, _c_st]),
    "uavgnn_gemm_nt_x3_rowmax": (_c_int, [_c_fp, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_int, _c_fp, _c_fp, _c_int, _c_int, _c_fp, _c_st]),
    "uavgnn_col_absmax": (_c_int, [_c_fp, ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_fp, _c_st]),
    "uavgnn_gemm_tn_h2_supported": (_c_int, [ctypes.c_longlong, _c_int, _c_int]),
    "uavgnn_gemm_tn_h2_chunks": (_c_int, [ctypes.c_longlong, _c_int, _c_int]),
    "uavgnn_gemm_tn_h2": (_c_int, [_c_fp, ctypes.c_longlong, _c_int, _c_fp, ctypes.c_longlong, _c_int, ctypes.c_longlong, _c_fp, _c_fp, _c_fp,
                                   _c_int, _c_int, _c_st]),
    "uavgnn_gemm_h2_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "uavgnn_split_h2_bytes": (ctypes.c_longlong, [_c_int, _c_int]),
    "uavgnn_split_h2": (_c_int, [_c_fp, _c_int, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_st]),
    "uavgnn_gemm_nt_h2": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_fp, _c_fp, ctypes.c_void_p, _c_int, _c_fp, _c_fp,
                                   _c_int, _c_int, _c_st]),
    "uavgnn_gemm_nt_h2_rm2": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_fp, _c_fp, ctypes.c_void_p, _c_int, _c_fp, _c_fp,
                                       _c_int, _c_int, _c_st]),
    "uavgnn_env_state_dim": (_c_int, [_c_int, _c_int, _c_int]),
    "uavgnn_env_step": (_c_int, [ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double), _c_int] + [_c_fp] * 22 + [_c_st]),
    "uavgnn_adamw_polyak": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, ctypes.c_longlong, ctypes.c_longlong, _c_fp, ctypes.c_double,
                                     ctypes.c_double, _c_f32, _c_f32, _c_f32, _c_f32, _c_st]),
    "uavgnn_gru_cell_supported": (_c_int, [_c_int, _c_int]),
    "uavgnn_gru_cell_fwd": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_st]),
    "uavgnn_gru_cell_x3_supported": (_c_int, [_c_int, _c_int]),
    "uavgnn_gru_cell_x3_workspace_bytes": (ctypes.c_longlong, [_c_int, _c_int]),
    "uavgnn_gru_split_weights": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, ctypes.c_void_p, _c_st]),
    "uavgnn_gru_cell_fwd_x3": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, ctypes.c_void_p, _c_fp, _c_fp, _c_fp, _c_fp, _c_st]),
    "uavgnn_gru_cell_fwd_x3_cat": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, ctypes.c_void_p, _c_fp, _c_fp,
                                            _c_fp, _c_fp, _c_st]),
    "uavgnn_gru_cell_fwd_x3_opts": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, ctypes.c_void_p, _c_fp, _c_fp,
                                             _c_fp, _c_fp, _c_int, _c_st]),
    "uavgnn_gru_cell_h2_supported": (_c_int, [_c_int, _c_int]),
    "uavgnn_gru_cell_h2_workspace_bytes": (ctypes.c_longlong, [_c_int, _c_int]),
    "uavgnn_gru_split_weights_h2": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, ctypes.c_void_p, _c_st]),
    "uavgnn_gru_cell_fwd_h2": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_fp, ctypes.c_void_p, _c_fp,
                                        _c_fp, _c_fp, _c_fp, _c_st]),
    "uavgnn_row_absmax": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_fp, _c_st]),
    "uavgnn_tarmac_msg_rowmax_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int]),
    "uavgnn_tarmac_msg_fwd_rowmax": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_fp, _c_int, _c_int,
                                              _c_ip, _c_ip, _c_f32, _c_fp, _c_int, _c_fp, _c_fp, _c_int, _c_fp, _c_int, _c_fp, _c_st]),
    "uavgnn_gru_weight_tiles_bytes": (ctypes.c_longlong, [_c_int, _c_int]),
    "uavgnn_gru_split_weight_tiles": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, ctypes.c_void_p, _c_st]),
    "uavgnn_gru_cell_fwd_planes": (_c_int, [ctypes.c_void_p, _c_int, _c_fp, _c_int, _c_int, ctypes.c_void_p, _c_fp, _c_fp, _c_fp, _c_fp,
                                            _c_st]),
    "uavgnn_gru_cell_fwd_planes_opts": (_c_int, [ctypes.c_void_p, _c_int, _c_fp, _c_int, _c_int, ctypes.c_void_p, _c_fp, _c_fp, _c_fp,
                                                 _c_fp, _c_int, _c_st]),
    "uavgnn_gru_cell_bwd_workspace_bytes": (ctypes.c_longlong, [_c_int, _c_int]),
    "uavgnn_gru_split_weights_bwd": (_c_int, [_c_fp, _c_int, _c_fp, _c_int, ctypes.c_void_p, _c_st]),
    "uavgnn_gru_cell_bwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_fp, _c_fp, _c_fp, _c_int, _c_fp,
                                     _c_st]),
    "uavgnn_workspace_bytes": (ctypes.c_longlong, [_c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong]),
    "uavgnn_gemm_tn_x3_chunks": (_c_int, [ctypes.c_longlong, _c_int, _c_int]),
    "uavgnn_gemm_tn_x3": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, ctypes.c_longlong, _c_fp, _c_int, _c_int, _c_st]),
    "uavgnn_gemm_x3_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "uavgnn_head_supported": (_c_int, [_c_int, _c_int]),
    "uavgnn_head_fwd": (_c_int, [_c_fp, _c_int, _c_int, _c_int, _c_fp, _c_int, _c_fp, _c_int, _c_fp, _c_int, _c_st]),
    "uavgnn_split_bf16x3": (_c_int, [_c_fp, _c_int, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_st]),
    "uavgnn_gemm_nt_x3": (_c_int, [_c_fp, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_int, _c_fp, _c_fp, _c_int, _c_int, _c_st]),
    "uavgnn_gemm_nt_x3_cat": (_c_int, [_c_fp, _c_int, _c_int, _c_fp, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_int, _c_fp, _c_fp, _c_int,
                              _c_int, _c_st]),
    "uavgnn_gru_gates_bwd_fused": (_c_int, [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_fp, _c_fp, _c_fp, _c_st]),
    "uavgnn_gru_gates_bwd_fused_head": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_int, _c_int, _c_fp, _c_fp, _c_fp,
                                                 _c_st]),
    "uavgnn_gru_gates_bwd_sum_rows": (_c_int, [_c_int, _c_int]),
    "uavgnn_gru_gates_bwd_fused_sums": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_int, _c_int, _c_fp, _c_fp, _c_fp, _c_fp,
                                                 _c_st]),
    "uavgnn_gru_gates_bwd_fused_sums_rowmax": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_int, _c_int, _c_fp, _c_fp, _c_fp, _c_fp,
                                                        _c_fp, _c_st]),
    "uavgnn_gru_gates_fwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_fp, _c_st]),
    "uavgnn_gru_gates_bwd": (_c_int, [_c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_fp, _c_fp, _c_fp, _c_st]),
}

_LIB = None
UAVGNN_EINVAL = -1000
UAVGNN_EUNSUPPORTED = -1001


class UavGnnError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Loads libuavgnn.so once.  Fails loudly when it has not been built (``__graft_entry__.build()``)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise UavGnnError(f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; "
                              f"g.build()'`.  uav_bs_ctrl_amd has no CPU / eager fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _LIB = handle
        # Recorded vendor-GEMM solutions for the dense half of the path (uav_bs_ctrl_amd/tuned) are OPT-IN: PyTorch's
        # TunableOp switch is process-wide and would change GEMM selection for every other model of the host
        # application.  UAVGNN_TUNED_GEMM=1 or an explicit uav_bs_ctrl_amd.enable_tuned_gemms() turns it on
        # (bench.py does).
        if os.environ.get("UAVGNN_TUNED_GEMM", "0") == "1":
            from .tuned import enable_tuned_gemms
            enable_tuned_gemms()
    return _LIB


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().uavgnn_strerror(code).decode()
        raise UavGnnError(f"{what} failed with code {code}: {msg}")


def ptr(t: th.Tensor | None):
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def ptr_array(tensors):
    """Host array of device pointers (NULL for None) for the entry points that take a parameter block."""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


_RAW_STREAM = getattr(th._C, "_cuda_getCurrentRawStream", None)


def stream() -> int:
    """The HIP stream PyTorch is currently recording on (so launches order with torch ops and graph capture works).  Through the raw
    accessor where this PyTorch has it: ``torch.cuda.current_stream()`` builds a Stream object behind three layers of device-index
    helpers (2.7 us against 0.3 - and a rollout step asks ten times)."""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(th.cuda.current_device())
    return th.cuda.current_stream().cuda_stream


def require_gpu(*tensors: th.Tensor) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise UavGnnError("uav_bs_ctrl_amd ops run on the MI355X only: got a CPU tensor (no CPU fallback exists; "
                              "move the module and the graph to 'cuda').")


def f32c(t: th.Tensor) -> th.Tensor:
    """float32 + contiguous view of t (no copy when already so)."""
    if t.dtype != th.float32:
        raise UavGnnError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()
