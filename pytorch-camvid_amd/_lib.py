"""ctypes binding of libcvk.so (include/cvk.h).  There is NO fallback: if the HIP library is missing or cannot be
loaded the product path raises — a silent eager/CPU path would void every parity claim."""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CVK_LIB_PATH") or os.path.join(_HERE, "lib", "libcvk.so")   # override: kernel experiments only
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "cvk.h")

CVK_STAT_ROWS = 64

c_f32p = ctypes.c_void_p      # device pointers are passed as integers
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_size = ctypes.c_size_t
c_float = ctypes.c_float
c_vp = ctypes.c_void_p


class View(ctypes.Structure):
    """cvk_view: strided NHWC view, strides in floats."""
    _fields_ = [("ptr", ctypes.c_void_p), ("sN", ctypes.c_int64), ("sY", ctypes.c_int64), ("sX", ctypes.c_int64)]


class ViewH(ctypes.Structure):
    """cvk_viewh: strided NHWC view of a bf16 / fp32 tensor, strides in elements."""
    _fields_ = [("ptr", ctypes.c_void_p), ("sN", ctypes.c_int64), ("sY", ctypes.c_int64), ("sX", ctypes.c_int64)]


# name -> (restype, argtypes); the list is checked against include/cvk.h by tests/test_abi.py
SIGNATURES = {
    "cvk_version": (c_int, []),
    "cvk_abi_hash": (ctypes.c_uint64, []),
    "cvk_last_error_string": (ctypes.c_char_p, []),
    "cvk_import_nchw": (c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_export_nchw": (c_int, [c_vp, c_int, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_zero_frame": (c_int, [View, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_pack_weight_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_pack_weight_dgrad": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_wgrad_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_wgrad": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_wino_weight_transform": (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    "cvk_conv3x3_wino_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_wino_gemm": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_wino_output": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_wino4_weight_transform": (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    "cvk_wino4_weight_transform_dgrad": (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    "cvk_conv3x3_wino4_ksplit": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_wino4_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_wino4_gemm": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_wino4_output": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_wino4f_weight_floats": (c_size, [c_int, c_int]),
    "cvk_wino4f_weight_transform": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_wino4f_weight_transform_batch": (c_int, [c_vp, c_int, c_vp]),
    "cvk_w2d_weight_transform_batch": (c_int, [c_vp, c_int, c_vp]),
    "cvk_wino4f_stat_partials": (c_int, [c_int, c_int, c_int]),
    "cvk_conv3x3_wino4f": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_wino4f_bnred": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "cvk_wino4h_weight_transform": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_wino4h": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_wino4h_bnred": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                         c_int, c_vp]),
    "cvk_thin_fwd_supported": (c_int, [c_int, c_int, c_int]),
    "cvk_thin_stat_partials": (c_int, [c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_thin_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_thin_wgrad_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_thin_wgrad_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_thin_wgrad": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_conv3x3_wgradp_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_wgradp": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_wgradp_plane_rows": (ctypes.c_long, [c_int, c_int, c_int]),
    "cvk_wgradp_planes": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_wgradp_gemm_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_wgradp_gemm": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_wgradp_zero_pads": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_wgradp_zero_pads_sm": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_wgradp_planes_sm": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_wgradp_gemm_sm": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_wgradp_zero_pads4": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bn_bwd_dx_e4p": (c_int, [View, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp,
                                  c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_wgradp_dy_slack": (c_int, [c_int]),
    "cvk_wgradp_gemm_sm_dy": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_conv3x3_wino4f_vplanes": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bn_bwd_dx_e6": (c_int, [View, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp,
                                 c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_tiles": (c_int, [c_int, c_int, c_int]),
    "cvk_w6_tiles": (c_int, [c_int, c_int, c_int]),
    "cvk_w6_stat_partials": (c_int, [c_int, c_int, c_int]),
    "cvk_conv3x3_w6_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_w6_weight_transform": (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    "cvk_w6_weight_transform_dgrad": (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    "cvk_w6_input_transform": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w6_gemm": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_w6_wgrad_ksplit": (c_int, [c_int, c_int, c_int]),
    "cvk_w6_dy_transform": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w6_gemm_tn": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_w6_wgrad_output": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w6_ksplit": (c_int, [c_int, c_int, c_int]),
    "cvk_w6_output": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_stat_partials": (c_int, [c_int, c_int, c_int]),
    "cvk_conv3x3_w2d_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_w2d_weight_transform": (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    "cvk_w2d_weight_transform_dgrad": (c_int, [c_vp, c_vp, c_int, c_int, c_vp]),
    "cvk_w2d_input_transform": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_gemm": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_wgrad_w2d_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_wgrad_w2d": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_w2d_tpad": (c_int, [c_int]),
    "cvk_split3_rows_pad": (c_int, [c_int, c_int]),
    "cvk_split3_planes": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_gemm_split3": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_input_transform_split3": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_dy_transform_both_split3": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_weight_transform_split3": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_output_plain": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_gemm_tn_split3_ksplit": (c_int, [c_int, c_int, c_int, c_int]),
    "cvk_w2d_gemm_tn_split3": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_amax_block_words": (c_int, []),
    "cvk_amax_block_value": (ctypes.c_uint, [c_vp, c_int]),
    "cvk_absmax_f32": (c_int, [c_vp, ctypes.c_long, c_int, c_int, c_vp, c_vp]),
    "cvk_split_scale_exponent": (c_int, [c_int, c_int, c_int, ctypes.c_uint]),
    "cvk_split_planes": (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_gemm_split": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_input_transform_split": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_dy_transform_both_split": (c_int, [c_int, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_weight_transform_split": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_gemm_tn_split": (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_wgrad_output_f": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_wgrad_ksplit": (c_int, [c_int, c_int, c_int]),
    "cvk_w2d_dy_transform": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_dy_transform_both": (c_int, [c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w6_dy_transform_both": (c_int, [c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_gemm_tn": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_wgrad_output": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_w2d_ksplit": (c_int, [c_int, c_int, c_int]),
    "cvk_w2d_output": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_w2d": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_conv3x3_wgrad_wino4_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_wgrad_wino4": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_bn_bwd_e_blocks": (c_int, [c_int, c_int, c_int]),
    "cvk_bn_bwd_dx_e": (c_int, [View, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp,
                                c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_wgrad_wino_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_wgrad_wino": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_bn_finalize_workspace_bytes": (c_size, [c_int, c_int]),
    "cvk_bn_finalize": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                c_float, c_float, c_vp, c_size, c_vp]),
    "cvk_bn_eval_params": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_float, c_vp]),
    "cvk_bn_relu_apply": (c_int, [c_vp, c_int, c_vp, c_vp, View, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bn_relu_apply_pool": (c_int, [c_vp, c_int, c_vp, c_vp, View, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bn_bwd_blocks": (c_int, [c_int]),
    "cvk_bn_bwd_reduce": (c_int, [View, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_colsum_finalize": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "cvk_colsum_finalize_batch": (c_int, [c_vp, c_int, c_vp]),            # jobs: host array of ColsumJob
    "cvk_bn_bwd_dx": (c_int, [View, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp,
                              c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bn_relu_apply_amax": (c_int, [c_vp, c_int, c_vp, c_vp, View, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "cvk_bn_relu_apply_pool_amax": (c_int, [c_vp, c_int, c_vp, c_vp, View, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "cvk_bn_bwd_dx_e_amax": (c_int, [c_int, View, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp,
                                     c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "cvk_bn_bwd_dx_amax": (c_int, [View, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp,
                                   c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "cvk_maxpool2x2_fwd": (c_int, [View, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_maxpool2x2_bwd": (c_int, [c_vp, View, c_vp, View, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_maxpool2x2_bwd_bnred_blocks": (c_int, [c_int, c_int, c_int, c_int]),
    "cvk_maxpool2x2_bwd_bnred": (c_int, [c_vp, View, c_vp, View, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "cvk_maxunpool2x2_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_maxunpool2x2_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_pool_code_to_index": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bilinear_up2_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bilinear_up2_bwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_ce_blocks": (c_int, [c_int]),
    "cvk_softmax_ce_fwd": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_softmax_ce_bwd": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_float, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_argmax_channels": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_vp]),
    "cvk_confusion_accumulate": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_preprocess_u8": (c_int, [c_vp, c_vp, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), c_vp]),
    "cvk_bf16s_rows_pad": (c_int, [c_int]),
    "cvk_bf16s_stat_partials": (c_int, [c_int, c_int, c_int]),
    "cvk_bf16s_stat_partials_c": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_pack_weight_fwd_bf16": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_pack_weight_dgrad_bf16": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_pack_weights_bf16_batch": (c_int, [c_vp, c_int, c_vp]),          # jobs: host array of PackJob
    "cvk_conv3x3_bf16s": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_bf16s_wg": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_bf16s_kernel": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "cvk_thin_bf16_mode": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_thin_bf16_stat_partials": (c_int, [c_int, c_int, c_int]),
    "cvk_thin_bf16_pack_elems": (c_size, [c_int]),
    "cvk_pack_weight_thin_bf16": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "cvk_conv3x3_thin_bf16": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bn_finalize_counts": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                       c_float, c_float, c_vp, c_size, c_vp]),
    "cvk_conv3x3_wgrad_bf16s_workspace_bytes": (c_size, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_wgrad_bf16s": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_size, c_vp]),
    "cvk_conv3x3_wgrad_bf16s_splits": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "cvk_conv3x3_wgrad_bf16s_slabs": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_size, c_vp]),
    "cvk_wgrad_reduce_bf16s_batch": (c_int, [c_vp, c_int, c_vp]),
    "cvk_import_nchw_bf16": (c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bn_relu_apply_bf16": (c_int, [c_vp, c_int, c_vp, c_vp, ViewH, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bn_bwd_blocks_bf16": (c_int, [c_int]),
    "cvk_bn_bwd_reduce_bf16": (c_int, [ViewH, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bn_bwd_dx_bf16": (c_int, [ViewH, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp,
                                   c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_maxpool2x2_bwd_bf16": (c_int, [c_vp, ViewH, ViewH, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_maxpool2x2_bwd_bnred_blocks_bf16": (c_int, [c_int, c_int, c_int, c_int]),
    "cvk_maxpool2x2_bwd_bnred_bf16": (c_int, [c_vp, ViewH, ViewH, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "cvk_maxunpool2x2_bwd_bf16": (c_int, [c_vp, ViewH, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bilinear_up2_fwd_bf16": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_bilinear_up2_bwd_bf16": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_zero_frame_bf16": (c_int, [ViewH, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "cvk_adamw_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_float, c_float, c_float, c_float, c_float, c_int, c_vp]),
}

_lib = None
_lock = threading.Lock()


class PackJob(ctypes.Structure):        # include/cvk.h cvk_pack_job
    _fields_ = [("w", ctypes.c_void_p), ("out", ctypes.c_void_p), ("Cout", ctypes.c_int), ("Cin", ctypes.c_int),
                ("Kpad", ctypes.c_int), ("dgrad", ctypes.c_int)]


class ColsumJob(ctypes.Structure):      # include/cvk.h cvk_colsum_job
    _fields_ = [("part", ctypes.c_void_p), ("out", ctypes.c_void_p), ("PB", ctypes.c_int), ("C", ctypes.c_int)]


class WtJob(ctypes.Structure):          # include/cvk.h cvk_wt_job
    _fields_ = [("w", ctypes.c_void_p), ("out", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("tile", ctypes.c_int),
                ("dgrad", ctypes.c_int)]


class WReduceJob(ctypes.Structure):     # include/cvk.h cvk_wreduce_job
    _fields_ = [("slabs", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("n", ctypes.c_ulonglong), ("splits", ctypes.c_int), ("pad", ctypes.c_int)]


PACK_BATCH_MAX = 48
COLSUM_BATCH_MAX = 64
WREDUCE_BATCH_MAX = 48
WT_BATCH_MAX = 48


class CvkError(RuntimeError):
    pass


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into lib/libcvk.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise CvkError("building libcvk.so failed (hipcc --offload-arch=gfx950); see output above")
    return LIB_PATH


def header_abi_hash(path=None):
    """First 64 bits of the SHA-256 of include/cvk.h — what csrc/Makefile stamps into the library as CVK_ABI_HASH."""
    import hashlib
    with open(path or HEADER, "rb") as f:
        return int(hashlib.sha256(f.read()).hexdigest()[:16], 16)


def check_abi(lib, header=None):
    """Refuse a library compiled against another include/cvk.h than the one this binding table was written for: entry points keep
    their names when argument lists change and ctypes does not check arity, so a stale libcvk.so would corrupt calls silently."""
    try:
        fn = lib.cvk_abi_hash
    except AttributeError:
        raise CvkError(f"{LIB_PATH} exports no cvk_abi_hash: it predates the ABI stamp; rebuild it (make -C {CSRC})") from None
    fn.restype = ctypes.c_uint64
    fn.argtypes = []
    got, want = fn(), header_abi_hash(header)
    if got != want:
        raise CvkError(f"{LIB_PATH} was built from another include/cvk.h (library stamp {got:016x}, header {want:016x}); "
                       f"rebuild it: make -C {CSRC}")


def load():
    """Load libcvk.so once.  Raises CvkError when it is missing — there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  torch's bundled libamdhip64 (SONAME libamdhip64.so.7) must be the one HIP runtime in-process
        if not os.path.exists(LIB_PATH):
            raise CvkError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  pytorch_camvid_amd has no CPU/eager fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        check_abi(lib)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = ABI drift between header and library
            fn.restype = res
            fn.argtypes = args
        if lib.cvk_version() != 100:
            raise CvkError(f"libcvk.so version {lib.cvk_version()} != 100 expected by the Python host side")
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().cvk_last_error_string()
        raise CvkError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
