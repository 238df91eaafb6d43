"""ctypes binding of the C-ABI in include/sn_spmm.h (libsn_hip.so, built in-tree by __graft_entry__.build()).

There is deliberately NO fallback: if the shared library is missing or a launch fails this module raises.
The product never computes the hot path on the CPU or through torch.sparse.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsn_hip.so")

_vp, _i64, _i32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/sn_spmm.h declares (tests/test_boundary.py checks).
SIGNATURES = {
    "sn_abi_version": (C.c_int, []),
    "sn_status_string": (C.c_char_p, [C.c_int]),
    "sn_spmm_csr_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp]),
    "sn_spmm_csr_stats_workspace_bytes": (_sz, [_i64]),
    "sn_spmm_csr_stats_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "sn_spmm_bsr4_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp]),
    "sn_spmm_csr_elubwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64,
                                         _i32, _vp]),
    "sn_spmm_bsr4_elubwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64,
                                          _i32, _vp]),
    "sn_spmm_q3_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp]),
    "sn_spmm_q3_stats_workspace_bytes": (_sz, [_i64]),
    "sn_spmm_q3_stats_blocks": (_i32, []),
    "sn_spmm_q3_stats_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "sn_spmm_q3_elubwd_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp]),
    "sn_bsr4_to_q3_f32": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "sn_coo_to_csr_i32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp]),
    "sn_csr_transpose_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "sn_csr_transpose_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sn_scan_workspace_bytes": (_sz, [_i64]),
    "sn_bsr4_count": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "sn_bsr4_fill": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "sn_rb4_count": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "sn_rb4_fill": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "sn_spmm_rb4_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp]),
    "sn_spmm_rb4_elubwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "sn_spmm_rb4_absmax_blocks": (_i64, [_i64, _i32]),
    "sn_spmm_rb4_elubwd_absmax_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "sn_spmm_rb4_stats_workspace_bytes": (_sz, [_i64]),
    "sn_spmm_rb4_stats_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    "sn_spmm_csr_ring_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp]),
    "sn_spmm_csr_ring_elubwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "sn_spmm_csr_ring_absmax_blocks": (_i64, [_i64, _i32]),
    "sn_spmm_csr_ring_elubwd_absmax_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp,
                                                     _vp]),
    "sn_spmm_csr_ring_stats_workspace_bytes": (_sz, [_i64]),
    "sn_spmm_csr_ring_stats_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    "sn_spmm_csr_ring_half_window": (_i32, []),
    "sn_csr_band_i32": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "sn_blockdiag_concat_i32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "sn_blockdiag_concat_ragged_i32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "sn_validate_csr_i32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "sn_elu_into_f32": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "sn_elu_bwd_acc_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp]),
    "sn_colstats_workspace_bytes": (_sz, [_i64, _i32]),
    "sn_colstats_f32": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _sz, _vp]),
    "sn_wgrad_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sn_wgrad_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "sn_wgrad_seg_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "sn_wgrad_seg_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sn_wgrad_slabs_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sn_wgrad_bounded_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _sz, _vp, _i64, _vp, _i64, _vp]),
    "sn_wgrad_seg_bounded_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp, _i64, _vp,
                                           _i64, _vp]),
    "sn_wgrad_slabs_bounded_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz,
                                             _vp, _i64, _vp, _i64, _vp]),
    "sn_avg_prep_ragged_f32": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp]),
    "sn_bn_fold_seg_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _i64, _vp, _vp]),
    "sn_avg_bn_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _i64, _vp, _i64, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sn_linear_dgrad_absmax_blocks": (_i32, []),
    "sn_linear_dgrad_elu_absmax_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64,
                                                 _i32, _i32, _vp, _vp]),
    "sn_linear_dgrad_eluseg_absmax_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp,
                                                    _i64, _i64, _i32, _i32, _vp, _vp]),
    "sn_linear_dgrad_eluseg_ragged_absmax_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64,
                                                           _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "sn_spmm_q3_absmax_blocks": (_i64, [_i64, _i32]),
    "sn_spmm_q3_elubwd_absmax_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _i32,
                                               _vp, _vp]),
    "sn_wgrad_thin_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "sn_wgrad_thin_f32": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "sn_masked_smooth_l1_workspace_bytes": (_sz, [_i64, _i32]),
    "sn_masked_smooth_l1_fwd_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, C.c_double, _vp, _vp, _sz, _vp]),
    "sn_masked_smooth_l1_bwd_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, C.c_double, _vp, _vp, _i64, _vp]),
    "sn_pair_argmin_f32": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "sn_pair_ce_fwd_f32": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp]),
    "sn_pair_ce_bwd_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp]),
    "sn_pair_fused_workspace_bytes": (C.c_size_t, [_i64, _i64]),
    "sn_pair_fused_fwd_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i64, C.c_int32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "sn_pair_fused_bwd_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, C.c_int32, _vp, _i64, _vp, _i64, _vp, C.c_size_t, _vp]),
    "sn_gather_segments_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "sn_gather_segments_ragged_f32": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "sn_linear_thin_fwd_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "sn_bn_fold_f32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, C.c_double, C.c_double, _i32, _vp, _vp, _vp,
                                 _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sn_colstats_blocks": (C.c_int32, [_i64]),
    "sn_wgrad_bounded_enabled": (C.c_int32, []),
    "sn_gemm_variant": (C.c_int32, []),
    "sn_colstats_merge2_f64": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    "sn_colstats_partial_f32": (C.c_int, [_vp, _i64, _i64, C.c_int32, _vp, _vp]),
    "sn_bn_bwd_coeffs_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sn_segment_colsum_ragged_workspace_bytes": (_sz, [_i64, _i32]),
    "sn_segment_colsum_ragged_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _sz, _vp]),
    "sn_bcast_rows_ragged_f32": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp]),
    "sn_segment_colsum_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "sn_segment_colsum_f32": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _sz, _vp]),
    "sn_bcast_rows_f32": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "sn_elu_bwd_bcast_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _i32, _vp]),
    "sn_dirac_workspace_bytes": (_sz, [_i64, _i64]),
    "sn_dirac_bsr4_from_mesh": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sn_timing_enable": (C.c_int, [_i32]),
    "sn_timing_count": (_i64, []),
    "sn_timing_drain": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sn_laplacian_workspace_bytes": (_sz, [_i64, _i64]),
    "sn_laplacian_csr_from_mesh": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sn_linear_fwd_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "sn_linear_fwd_stats_blocks": (_i32, [_i64]),
    "sn_linear_fwd_tiles_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "sn_linear_fwd_segbias_tiles_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32,
                                                  _vp, _vp, _vp]),
    "sn_avg_stats_from_tiles_ragged_f32": (C.c_int, [_vp, _vp, _i32, _vp, _i64, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "sn_linear_fwd_segbias_ragged_tiles_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32,
                                                         _i32, _vp, _vp, _vp]),
    "sn_avg_stats_from_tiles_f32": (C.c_int, [_vp, _vp, _i32, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    "sn_colstats_into_f32": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _i64, _vp, _sz, _vp]),
    "sn_colstats_merge_f64": (C.c_int, [_vp, _i32, _i32, _vp, _i64, _i64, _vp]),
    "sn_linear_dgrad_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "sn_linear_dgrad_elu_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64,
                                          _i32, _i32, _vp]),
    "sn_linear_fwd_segbias_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp,
                                            _vp]),
    "sn_linear_dgrad_eluseg_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64,
                                             _i64, _i32, _i32, _vp]),
    "sn_avg_fwd_prep_f32": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "sn_avg_stats_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "sn_avg_stats_f32": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _sz, _vp]),
    "sn_seg_affine_f32": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _i32, _vp, _vp]),
    "sn_avg_bwd_gc_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "sn_avg_bwd_segvec_f32": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "sn_avg_bwd_segvec_ragged_f32": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "sn_linear_fwd_segbias_ragged_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32,
                                                   _i32, _vp, _vp]),
    "sn_linear_dgrad_eluseg_ragged_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp,
                                                    _i64, _i64, _i32, _i32, _vp]),
    "sn_affine_cols_acc_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp]),
    "sn_affine_cols_elu_bwd_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp]),
    # launch plans (csrc/sn_plan.hip): host objects, no device work of their own
    "sn_plan_create": (C.c_int, [C.POINTER(_vp)]),
    "sn_plan_destroy": (C.c_int, [_vp]),
    "sn_plan_lookup": (_i32, [C.c_char_p]),
    "sn_plan_entry_count": (_i32, []),
    "sn_plan_entry_name": (C.c_char_p, [_i32]),
    "sn_plan_entry_signature": (C.c_char_p, [_i32]),
    "sn_plan_length": (_i64, [_vp]),
    "sn_plan_add_call": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "sn_plan_add_memset": (C.c_int, [_vp, _i32, _i64, _i32, _i64, _i64, _i64]),
    "sn_plan_add_copy": (C.c_int, [_vp, _i32, _i64, _i64, _i32, _i64, _i64, _i64, _i64]),
    "sn_plan_run": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "sn_plan_instantiate": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "sn_plan_exec_launch": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "sn_plan_exec_destroy": (C.c_int, [_vp]),
}

_lib = None


class SnError(RuntimeError):
    """A C-ABI call returned non-zero (negative: SN_E_* argument error, positive: hipError_t)."""


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). surfacenetworks_amd has no CPU/eager fallback.")
        # The HIP runtime must be initialised BEFORE the library (whose load registers its code objects) is mapped: loaded
        # first, every later launch fails with hipErrorNoDevice (observed with ROCm 7.2 / torch 2.10: `build()` followed
        # by `smoke()` in one process).  On a box without a GPU there is nothing to initialise and only the symbols are used.
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().sn_status_string(status).decode()
        raise SnError(f"{what} failed with status {status}: {msg}")


# A launch-plan recorder (plans.py) while the launch list of a block is being taken down: every call() is then RECORDED, not
# executed (a dry run of the block's host code); None otherwise.
_recorder = None
_recorder_tid = 0                 # the thread whose calls are being recorded: every other thread (autograd's device threads, a
_record_lock = threading.RLock()  # loader thread) keeps LAUNCHING while a dry run is in progress; dry runs themselves are serialised


def recorder():
    """The recorder of a dry run in progress ON THIS THREAD, else None."""
    r = _recorder
    return r if r is not None and _recorder_tid == threading.get_ident() else None


def call(name: str, *args) -> None:
    if _recorder is not None:
        r = recorder()
        if r is not None:
            r.record_call(name, args)
            return
    check(getattr(load(), name)(*args), name)
