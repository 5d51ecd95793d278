"""ctypes loader for ``libcoda_hip.so`` (the C ABI in ``include/*.h``).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C coda_neurips2023_amd/csrc`` and is the ONLY compute back end of this
package.  Loading fails loudly; nothing falls back to PyTorch or the CPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CODA_LIB_PATH: another build of the same library for A/B runs of development variants -- tools/ab_prio.sh; never a
# different implementation: there is none)
LIB_PATH = os.environ.get("CODA_LIB_PATH") or os.path.join(_HERE, "libcoda_hip.so")

CODA_OK = 0
CODA_EINVAL = -1
CODA_ENOSPC = -2
CODA_ELOST = -3

_c_void_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_float = ctypes.c_float
_c_size_t = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/coda_pointnet2.h one to one.
_P = _c_void_p
SIGNATURES = {
    "coda_version": (ctypes.c_char_p, []),
    "coda_get_distance_mode": (_c_int, []),
    "coda_furthest_point_sampling_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "coda_furthest_point_sampling_f32": (_c_int, [_P, _c_int, _c_int, _c_int, _P, _P, _c_size_t, _P]),
    "coda_furthest_point_sampling_opt_f32": (_c_int, [_P, _c_int, _c_int, _c_int, _P, _P, _c_size_t, _c_int, _c_int, _P]),
    "coda_fps_lost_partner_events": (ctypes.c_uint, [_c_int]),
    "coda_furthest_point_sampling_dbg_f32": (_c_int, [_P, _c_int, _c_int, _c_int, _P, _P, _c_size_t, _c_int, _c_int, _P]),
    "coda_gather_points_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P]),
    "coda_gather_points_grad_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P]),
    "coda_scatter_add_det_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "coda_gather_points_grad_det_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P, _c_size_t, _P]),
    "coda_group_points_grad_det_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _P, _c_size_t, _P]),
    "coda_ball_query_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    "coda_ball_query_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_float, _c_int, _P, _c_size_t, _P]),
    "coda_ball_query_opt_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_float, _c_int, _P, _c_size_t, _c_int, _c_int, _P]),
    "coda_group_points_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _P]),
    "coda_group_points_grad_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _P]),
    "coda_query_and_group_xyz_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _c_float, _c_int, _c_int, _P, _c_size_t, _P]),
    "coda_query_and_group_xyz_opt_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _c_float, _c_int, _c_int, _P, _c_size_t,
                                                  _c_int, _c_int, _P]),
    "coda_three_nn_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _P]),
    "coda_three_interpolate_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P]),
    "coda_three_interpolate_grad_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P]),
    "coda_three_nn_opt_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P]),
    "coda_three_interpolate_opt_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _P]),
    # include/coda_sa_mlp.h
    "coda_sa_col_stats_f32": (_c_int, [_P, _P, ctypes.c_longlong, _c_int, _P, _P, _P]),
    "coda_sa_bn_relu_apply_f32": (_c_int, [_P, _P, _P, _P, ctypes.c_longlong, _c_int, _P, _P]),
    "coda_sa_col_stats_pool_f32": (_c_int, [_P, ctypes.c_longlong, _c_int, _c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "coda_sa_bn_bwd_sparse_f32": (_c_int, [_P, _P, _P, _P, ctypes.c_longlong, _c_int, _c_int, _P, _P, _P, _P]),
    "coda_sa_relu_bn_bwd_stats_f32": (_c_int, [_P, _P, _P, _P, ctypes.c_longlong, _c_int, _P, _P]),
    "coda_sa_relu_bn_bwd_apply_f32": (_c_int, [_P, _P, _P, _P, ctypes.c_longlong, _c_int, _P, _P, _P, _P]),
    "coda_sa_bn_finalize_f32": (_c_int, [_P, ctypes.c_double, ctypes.c_double, _c_float, _P, _P, _P, _P, _P, _P, _c_int, _P]),
    "coda_sa_bn_bwd_coef_f32": (_c_int, [_P, ctypes.c_double, _P, _P, _P, _c_int, _P, _P, _c_int, _P]),
    "coda_sa_pool_select_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_int, _P]),
    "coda_sa_pool_bwd_stats_f32": (_c_int, [_P, _P, _P, _P, _P, ctypes.c_longlong, _c_int, _P, _P]),
    "coda_sa_compact_groups_f32": (_c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_int, ctypes.c_longlong,
                                             ctypes.c_longlong, _P]),
    # MFMA pipeline (csrc/sa_mfma.hip)
    "coda_sa_mfma_blocks": (_c_int, [_c_int]),
    "coda_sa_mfma_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int]),
    "coda_sa_pack_groups_f32": (_c_int, [_P, _P, _c_int, _P, _P, _P, _P, _P, _P, _P, _c_int, ctypes.c_longlong, _c_int, _P]),
    "coda_sa_l1_sums_f32": (_c_int, [_P, _P, _P, _c_int, _P]),
    "coda_sa_mfma_fwd_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_int, _c_int, _c_int, _P, _P, _P, _P,
                                      _P, _P, _P, _P, _c_int, _P]),
    "coda_sa_pool_finish_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_int, _c_int, _P]),
    "coda_sa_mfma_bwd_dx_f32": (_c_int, [_P, _P, _P, _P, _P, _c_int, _P, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_int,
                                         _c_int, _c_int, _P, _P, _c_int, _P]),
    "coda_sa_mfma_bwd_dw_f32": (_c_int, [_P, _P, _P, _P, _P, _c_int, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_int,
                                         _c_int, _c_int, _P, _P, _c_int, _P]),
    "coda_sa_l1_bwd_f32": (_c_int, [_P, _P, ctypes.c_double, _P, _P, _P, _P, _P, _P, _P, _c_int, _P]),
    # include/coda_token_ops.h
    "coda_tok_bn_stats_f32": (_c_int, [_P, _c_int, ctypes.c_longlong, _c_int, _P, _P]),
    "coda_tok_bn_parts": (_c_int, [_c_int, ctypes.c_longlong, _c_int]),
    "coda_tok_bn_finalize_f32": (_c_int, [_P, _c_int, _P, _P, _c_int, _c_int, ctypes.c_double, _c_float, _P, _P, _P]),
    "coda_tok_bn_act_f32": (_c_int, [_P, _P, _c_int, ctypes.c_longlong, _c_int, _c_int, _c_float, ctypes.c_uint64,
                                     _P, _P, _P]),
    "coda_tok_bn_act_bwd_stats_f32": (_c_int, [_P, _P, _P, _c_int, ctypes.c_longlong, _c_int, _c_int, _c_float,
                                               ctypes.c_uint64, _P, _P, _P]),
    "coda_tok_bn_bwd_finalize_f32": (_c_int, [_P, _c_int, _P, _P, _P, _c_int, _c_int, ctypes.c_double, _P, _P, _P, _P]),
    "coda_tok_bn_act_bwd_apply_f32": (_c_int, [_P, _P, _P, _P, _c_int, ctypes.c_longlong, _c_int, _c_int, _c_float,
                                               ctypes.c_uint64, _P, _P, _P]),
    "coda_tok_add_ln_fwd_f32": (_c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_int, _c_float, _c_float,
                                         ctypes.c_uint64, _P, _P, _P, _P, _P, _P, _P]),
    "coda_tok_add_ln_bwd_blocks": (_c_int, [ctypes.c_longlong, _c_int]),
    "coda_tok_add_ln_bwd_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_int, _c_float,
                                         ctypes.c_uint64, _P, _P, _P, _P, _P, _P]),
    "coda_tok_add_ln_fwd2_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_int, _c_float,
                                          _c_float, ctypes.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "coda_tok_add_ln_bwd2_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_longlong, _c_int, _c_float,
                                          ctypes.c_uint64, _P, _c_int, _P, _P, _c_int, _P, _P, _P, _P, _P]),
    "coda_tok_colsum_finalize_f32": (_c_int, [_P, _c_int, _c_int, _P, _P]),
    "coda_tok_colsum_blocks": (_c_int, [ctypes.c_longlong, _c_int]),
    "coda_tok_colsum_f32": (_c_int, [_P, _c_int, ctypes.c_longlong, _c_int, _P, _P, _P]),
    "coda_tok_bias_relu_dropout_fwd_f32": (_c_int, [_P, _P, ctypes.c_longlong, _c_int, _c_float, ctypes.c_uint64, _P,
                                                    _P, _P]),
    "coda_tok_bias_relu_dropout_bwd_blocks": (_c_int, [ctypes.c_longlong, _c_int]),
    "coda_tok_bias_relu_dropout_bwd_f32": (_c_int, [_P, _P, ctypes.c_longlong, _c_int, _c_float, _P, _P, _P, _P]),
    "coda_fourier_pos_embed_f32": (_c_int, [_P, _P, _P, _P, _c_int, _P, _c_int, _c_int, _c_int, _P]),
    # include/coda_align_loss.h
    "coda_align_loss_fwd_f32": (_c_int, [_P, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _P, _P, _P, _P,
                                         _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _P, _P]),
    "coda_align_loss_bwd_f32": (_c_int, [_P, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _P, _P, _P, _P,
                                         _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _P, _P]),
    "coda_align_rows_fwd_f32": (_c_int, [_P, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _P, _P, _c_int,
                                         _c_int, _c_int, _c_int, _P, _P, _P, _P]),
    "coda_align_rows_bwd_f32": (_c_int, [_P, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _P, _P, _P, _P, _P,
                                         _c_int, _c_int, _c_int, _c_int, _P, _P]),
    "coda_align_ce_f32": (_c_int, [_P, ctypes.c_longlong, _c_int, _c_int, _P, _P, _P, _P, _P, ctypes.c_longlong,
                                   ctypes.c_longlong, _P]),
    # include/coda_box_ops.h
    "coda_generalized_box3d_iou_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _P]),
    "coda_generalized_box3d_iou_devflag_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _P, _c_int, _c_int, _P]),
    "coda_box_decode_fwd_f32": (_c_int, [_P] * 9 + [_c_int] * 5 + [_P] * 10 + [_P]),
    "coda_box_decode_bwd_f32": (_c_int, [_P] * 7 + [_c_int] * 4 + [_P] * 8 + [_P] * 3 + [_P]),
    "coda_box_loss_fwd_f32": (_c_int, [_P] * 15 + [_c_int] * 6 + [_P, _P]),
    "coda_box_loss_bwd_f32": (_c_int, [_P] * 15 + [_c_int] * 6 + [_P] * 6 + [_P]),
    "coda_matcher_cost_f32": (_c_int, [_P] * 8 + [_c_float] * 4 + [_P] * 3 + [_c_int] * 5 + [_P, _c_int, _P]),
    "coda_hungarian_f32": (_c_int, [_P, _P, _P, _P, _c_int, _c_int, _c_int, _P]),
    "coda_grouped_gemm_tn_f32": (_c_int, [_P, _c_int, _P]),
    "coda_tok_colsum_finalize_grouped_f32": (_c_int, [_P, _c_int, _P]),
    # include/coda_stack.h
    "coda_decoder_stack_ws_floats": (ctypes.c_size_t, [_c_int] * 6),
    "coda_decoder_stack_bwd_ws_floats": (ctypes.c_size_t, [_c_int] * 6),
    "coda_decoder_stack_fwd_f32": (_c_int, [_P, _P]),
    "coda_decoder_stack_bwd_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    # include/coda_clip_crops.h
    "coda_project_box_rects_f64": (_c_int, [_P] * 16 + [_c_int, _c_int, _P]),
    "coda_crop_resize_f32": (_c_int, [_P] * 5 + [_c_int] * 6 + [_P]),
    # include/coda_clip_labels.h
    "coda_clip_weak_labels_f32": (_c_int, [_P, ctypes.c_longlong, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int,
                                           _c_int, _P]),
    "coda_pseudo_box_filter_f32": (_c_int, [_P] * 6 + [_c_float] * 3 + [_P, _P, _c_int, _c_int, _c_int, _P]),
    # include/coda_clip_tower.h
    "coda_vit_workspace_bytes": (ctypes.c_size_t, [_P, _c_int, _c_int]),
    "coda_vit_fwd": (_c_int, [_P, _P, _c_int, _P, _P, _P, ctypes.c_size_t, _P]),
    "coda_vit_attention_f16": (_c_int, [_P, _P, _c_int, _c_int, _c_int, _P]),
    "coda_vit_quickgelu_fused": (_c_int, [_c_int]),
    "coda_gemm_ex": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _P, ctypes.c_longlong, _P,
                              ctypes.c_longlong, _P, ctypes.c_longlong, _P, _c_float, _c_float, _P]),
    # include/coda_optim.h
    "coda_opt_chunk_elems": (_c_int, []),
    "coda_opt_grad_sumsq_f32": (_c_int, [_P, _P, _c_int, _P, _P]),
    "coda_opt_grad_scale_f32": (_c_int, [_P, _P, _c_int, _P, _c_float, _P, _P]),
    "coda_opt_pack_f32": (_c_int, [_P, _P, _c_int, _c_float, _P]),
    "coda_opt_adamw_f32": (_c_int, [_P, _P, _c_int, _c_float, _c_float, _c_float, _c_float, _c_float, _P]),
    # include/coda_eval.h
    "coda_box_point_count_f32": (_c_int, [_P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P]),
    "coda_nms_f32": (_c_int, [_P, _P, _P, _P, _P, _c_int, _c_int, _c_int, ctypes.c_double, _c_int, _P]),
    # include/coda_attention.h
    "coda_mha_fwd_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                  _c_int, _c_int, _c_float, _c_float, ctypes.c_uint64, _P, _P]),
    "coda_mha_bwd_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int,
                                  _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float,
                                  ctypes.c_uint64, _P, _P]),
    "coda_mha_bwd_parts_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int,
                                        _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float,
                                        ctypes.c_uint64, _P, _c_int, _P]),
    "coda_mha_fwd_opt_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                      _c_int, _c_int, _c_float, _c_float, ctypes.c_uint64, _P, _c_int, _P]),
    "coda_mha_bwd_ws_bytes": (_c_size_t, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "coda_mha_bwd_ws_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int,
                                     _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float,
                                     ctypes.c_uint64, _P, _P, _c_size_t, _c_int, _P]),
    "coda_mha_bwd_parts_opt_f32": (_c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int, _c_int,
                                            _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float,
                                            ctypes.c_uint64, _P, _c_int, _c_int, _P]),
    # include/coda_gemm.h
    "coda_gemm_f32": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _P, ctypes.c_longlong, _P, ctypes.c_longlong,
                               _P, ctypes.c_longlong, _P, _c_int, _P]),
    "coda_sgemm_f32": (_c_int, [_c_int, _c_int, _c_int, _c_int, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P,
                                ctypes.c_longlong, _P, _c_int, _P]),
    "coda_gemm_x3_split_f32": (_c_int, [_P, _c_int, _P]),
    "coda_gemm_x3_nt_f32": (_c_int, [_c_int, _c_int, _c_int, _P, ctypes.c_longlong, _P, _c_int, _c_int, _c_int, _P,
                                     ctypes.c_longlong, _P, _c_int, _P]),
    "coda_gemm_x3_tn_f32": (_c_int, [_c_int, _c_int, _c_int, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _c_int, _P]),
    "coda_sgemm_relu_dropout_f32": (_c_int, [_c_int, _c_int, _c_int, _c_int, _P, ctypes.c_longlong, _P, ctypes.c_longlong,
                                             _P, ctypes.c_longlong, _P, _c_float, ctypes.c_uint64, _P]),
    "coda_sgemm_relu_dropout_bwd_blocks": (_c_int, [_c_int]),
    "coda_sgemm_relu_dropout_bwd_f32": (_c_int, [_c_int, _c_int, _c_int, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P,
                                                 _c_float, _P, _P, _P]),
    "coda_mha_get_mfma_dtype": (_c_int, []),
    "coda_mha_timing_enable": (_c_int, [_c_int]),
    "coda_mha_timing_enable_kinds": (_c_int, [_c_int, ctypes.c_uint]),
    "coda_mha_timing_collect": (_c_int, [_P, _P, _P, _P, _c_int]),
}

_lib = None

# ---- per-call options --------------------------------------------------------------------------------------------------
# The C library has no mutable process-wide state: the arithmetic / kernel-selection switches are ARGUMENTS of its *_opt
# entry points (include/coda_pointnet2.h, coda_attention.h).  On the Python side the values travel in a THREAD-LOCAL
# record that the wrappers read when they issue a call (autograd functions capture them in their forward and hand them
# to their backward, which runs on another thread): two models -- or two threads -- with different options never
# interact.  ``options(...)`` scopes values to a ``with`` block; ``set_option`` makes one stick for the calling thread
# (tests, bench.py).  Every value defaults to "library default" (an environment variable read once by the library).
import contextlib as _contextlib
import threading as _threading

OPTION_DEFAULTS = {"distance_mode": -1, "fps_waves": 0, "bq_route": 0, "mfma_dtype": -1}
_tls = _threading.local()


def opt(name):
    return getattr(_tls, name, OPTION_DEFAULTS[name])


def set_option(name, value):
    if name not in OPTION_DEFAULTS:
        raise KeyError(name)
    setattr(_tls, name, int(value))


@_contextlib.contextmanager
def options(**values):
    saved = {k: opt(k) for k in values}
    try:
        for k, v in values.items():
            set_option(k, v)
        yield
    finally:
        for k, v in saved.items():
            set_option(k, v)


class CodaLibraryError(RuntimeError):
    pass


def load():
    """Load libcoda_hip.so once and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CodaLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path."
        )
    # The HIP runtime must be the one PyTorch-ROCm brought (torch bundles its own
    # libamdhip64): load torch first so that libcoda_hip.so binds to the SAME runtime and
    # its launches see torch's device context and streams.  Loading this library before
    # torch puts a second runtime in the process and every launch fails with
    # hipErrorNoDevice.
    import torch  # noqa: F401
    # (same for hipBLASLt, which coda_gemm_f32 calls: one copy per process, PyTorch's)
    for name in ("libamdhip64.so", "libhipblaslt.so"):
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", name)
        if os.path.exists(bundled):
            ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # stale build
            raise CodaLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def current_stream_handle():
    """Raw hipStream_t of torch's current stream on the current device (cheap: no Stream object)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def on_tensor_device(pick=lambda *a, **k: a[0]):
    """Decorator for the module entry points (``forward`` of the mirrors): the launchers below them take the raw
    stream of torch's CURRENT device, so when the tensor ``pick(*args, **kwargs)`` returns lives on another GPU
    the call runs under ``torch.cuda.device(that GPU)`` (autograd restores the forward's device for the backward
    nodes itself).  One integer comparison on the usual path."""
    import functools

    import torch

    def deco(fn):
        @functools.wraps(fn)
        def guarded(self, *args, **kwargs):
            t = pick(*args, **kwargs)
            if torch.is_tensor(t) and t.is_cuda and t.device.index != torch.cuda.current_device():
                with torch.cuda.device(t.device):
                    return fn(self, *args, **kwargs)
            return fn(self, *args, **kwargs)
        return guarded
    return deco


def check(status, what):
    """Turn a C-ABI status into a Python exception (the reference exit(-1)s)."""
    if status == CODA_OK:
        return
    if status == CODA_EINVAL:
        raise RuntimeError(f"{what}: invalid argument (CODA_EINVAL)")
    if status == CODA_ENOSPC:
        raise RuntimeError(f"{what}: workspace too small (CODA_ENOSPC)")
    if status == CODA_ELOST:
        word = load().coda_fps_lost_partner_events(1)  # acknowledged by raising
        raise RuntimeError(f"{what}: an earlier two-workgroup furthest-point-sampling launch lost its partner workgroup "
                           f"(scene {(word >> 16) & 0x7fff}, round {word & 0xffff}): the indices it returned are wrong "
                           "(CODA_ELOST)")
    raise RuntimeError(f"{what}: HIP kernel launch failed (hipError_t={status})")
