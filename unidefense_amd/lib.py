"""ctypes binding of libunidefense_hip.so (the C ABI declared in include/unidefense_hip.h).

The library is REQUIRED: importing this module without the built ``.so`` raises — there is no
fallback path of any kind (no torch ops, no CPU code) behind the product's operators.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
from .config import cfg as _cfg  # noqa: E402

LIB_PATH = _cfg.lib_path or os.path.join(_HERE, "libunidefense_hip.so")   # cfg.lib_path: A/B of kernel builds


class UDLibraryError(RuntimeError):
    pass


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("N", "Hin", "Win", "Cin", "Hout", "Wout", "KH", "KW", "stride",
                                       "pad_t", "pad_l", "transposed")]


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_long), ("ldb", C.c_long), ("ldc", C.c_long),
                ("a_mode", C.c_int), ("b_mode", C.c_int), ("out_mode", C.c_int), ("split_k", C.c_int),
                ("batch", C.c_int), ("strideA", C.c_long), ("strideB", C.c_long), ("strideC", C.c_long),
                ("g", ConvGeom), ("stat_sum", C.c_void_p), ("stat_sumsq", C.c_void_p), ("half_mask", C.c_int), ("tile_cfg", C.c_int), ("slice_stride", C.c_long)]


class SplitItem(C.Structure):
    """ud_split_item of include/unidefense_hip.h (one weight matrix of ud_split_planes_h2t_multi's device table)"""
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("inv_scale", C.c_void_p),
                ("R", C.c_long), ("ld", C.c_long), ("panel", C.c_long), ("plane", C.c_long),
                ("C", C.c_int), ("amax_block0", C.c_int), ("amax_blocks", C.c_int), ("split_block0", C.c_int),
                ("split_bx", C.c_int), ("pad_", C.c_int)]


class LayoutItem(C.Structure):
    """ud_layout_item of include/unidefense_hip.h (one conv weight of ud_weight_layouts_multi's device table)"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("A", C.c_int), ("B", C.c_int), ("KH", C.c_int), ("KW", C.c_int),
                ("mode", C.c_int), ("block0", C.c_int)]


class GemmP3Desc(C.Structure):
    """ud_gemm_p3_desc: GEMM on pre-split bf16 planes (P32 layout), include/unidefense_hip.h."""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("a_panel", C.c_long), ("a_plane", C.c_long), ("b_panel", C.c_long), ("b_plane", C.c_long),
                ("a_npanel", C.c_int), ("b_npanel", C.c_int), ("ldc", C.c_long),
                ("a_mode", C.c_int), ("b_mode", C.c_int), ("out_mode", C.c_int), ("split_k", C.c_int),
                ("stat_sum", C.c_void_p), ("stat_sumsq", C.c_void_p), ("tile_cfg", C.c_int), ("slice_stride", C.c_long),
                ("prec", C.c_int), ("a_inv_scale", C.c_void_p), ("b_inv_scale", C.c_void_p),
                ("a_scale_stride", C.c_int), ("b_scale_stride", C.c_int), ("c_half", C.c_int)]


class BnRef(C.Structure):
    """ud_bn_ref: a deferred BatchNorm (fp64 sums + affine parameters) applied by the consuming kernel."""
    _fields_ = [("sum", C.c_void_p), ("sumsq", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("inv_count", C.c_double), ("unbias", C.c_double), ("eps", C.c_float), ("momentum", C.c_float),
                ("act", C.c_int), ("G", C.c_int), ("running_mean", C.c_void_p), ("running_var", C.c_void_p)]


class LossTail(C.Structure):
    """ud_loss_tail of include/unidefense_hip.h (the scalar tail of a pass's loss, ud_loss_tail_run)"""
    _fields_ = [("feat", C.c_void_p * 3), ("dfeat", C.c_void_p * 3), ("D", C.c_int * 3), ("nfeat", C.c_int),
                ("cls", C.c_void_p), ("tgt", C.c_void_p), ("dcls", C.c_void_p), ("N", C.c_int), ("C", C.c_int), ("R", C.c_int),
                ("F", C.c_int),
                ("fm", C.c_void_p), ("dfm", C.c_void_p), ("nfm", C.c_int), ("sm", C.c_void_p), ("dsm", C.c_void_p), ("nsm", C.c_int),
                ("spatial", C.c_void_p), ("dspatial", C.c_void_p), ("freq", C.c_void_p), ("dfreq", C.c_void_p),
                ("w_cls", C.c_float), ("w_fm", C.c_float), ("w_sm", C.c_float), ("w_trip", C.c_float), ("w_rec", C.c_float),
                ("w_freq", C.c_float), ("vals", C.c_void_p), ("ws", C.c_void_p)]


class WgradFold(C.Structure):
    """ud_wgrad_fold: one depthwise weight-gradient fold of ud_dwtile_wgrad_finalize_multi"""
    _fields_ = [("part", C.c_void_p), ("dwt", C.c_void_p), ("gate_alpha", C.c_void_p), ("nparts", C.c_int), ("K", C.c_int),
                ("C", C.c_int), ("gate_mode", C.c_int)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_long, C.c_float
_BN = C.POINTER(BnRef)

# name -> argtypes (the trailing stream argument included); every function returns int
_SIGNATURES = {
    "ud_gemm": [C.POINTER(GemmDesc), _P],
    "ud_gemm_p3": [C.POINTER(GemmP3Desc), _P],
    "ud_gemm_p3_pair": [C.POINTER(GemmP3Desc), C.POINTER(GemmP3Desc), _P],
    "ud_planes_from_half": [_P, _L, _I, _L, _P, _L, _P, _P],
    "ud_split_planes": [_P, _L, _I, _L, _P, _L, _L, _P],
    "ud_split_planes_h2": [_P, _L, _I, _L, _P, _L, _L, _P, _P],
    "ud_absmax": [_P, _L, _I, _L, _P, _P],
    "ud_split_planes_h2t": [_P, _L, _I, _L, _P, _L, _L, _P, _P, _P],
    "ud_gemm_set_path": [C.c_int],
    "ud_split_planes_h2t_multi": [_P, C.c_int, _P, C.c_int, C.c_int, _P],
    "ud_weight_layouts_multi": [_P, C.c_int, C.c_int, _P],
    "ud_im2col_planes": [_P, C.POINTER(ConvGeom), _P, _L, _L, _P, _P, _P],
    "ud_col2im": [_P, C.POINTER(ConvGeom), _P, _P],
    "ud_fft32_set_wave": [C.c_int],
    "ud_gemm_query_path": [C.POINTER(GemmDesc)],
    "ud_gemm_stats_slots": [C.POINTER(GemmDesc)],
    "ud_stat_slots_fold": [_P, _P, _I, _I, _P, _P, _P],
    "ud_reduce_ws_doubles": [_I, _I, _I],
    "ud_norm_stats": [_P, _I, _I, _I, _F, _P, _P, _P, _P, _F, _P, _P, _P],
    "ud_syncbn_combine": [_P, _I, _I, _L, _F, _F, _P, _P, _P, _P, _P],
    "ud_norm_apply_fwd": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P],
    "ud_norm_bwd": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P],
    "ud_norm_fused_ws_doubles": [_I, _I, _I],
    "ud_norm_fused_counters": [_I, _I, _I],
    "ud_norm_fwd_fused": [_P, _I, _I, _I, _P, _P, _I, _F, _P, _P, _P, _P, _F, _P, _P, _P, _P],
    "ud_norm_bwd_fused": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "ud_norm_bwd_apply": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _I, _P, _P],
    "ud_group_colsum": [_P, _I, _I, _I, _F, _P, _P, _P],
    "ud_group_coldot": [_P, _P, _I, _I, _I, _F, _P, _P, _P],
    "ud_dwconv_fwd": [_P, _P, _P] + [_I] * 10 + [_I, _P],
    "ud_dwconv_bwd_data": [_P, _P, _P, _P] + [_I] * 10 + [_I, _P],
    "ud_dw_weights_tapmajor": [_P, _I, _L, _P, _P],
    "ud_dwconv_bwd_weight_parts": [_I, _I],
    "ud_dwconv_bwd_weight": [_P, _P, _P, _P, _I] + [_I] * 10 + [_P],
    "ud_rfft2": [_P, _P, _I, _I, _I, _F, _F, _I, _P],
    "ud_irfft2": [_P, _P, _I, _I, _I, _F, _F, _I, _P],
    "ud_fc_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ud_fc_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ud_se_scale_fwd": [_P, _P, _P, _I, _I, _I, _P],
    "ud_se_scale_bwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "ud_sigmoid_grad_mul": [_P, _P, _L, _P],
    "ud_sfmix_blocks": [_I, _I, _I, _I],
    "ud_sfmix_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ud_sfmix_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ud_gate_mix_blocks": [_L],
    "ud_gate_mix_fwd": [_P, _P, _P, _P, _L, _P],
    "ud_gate_mix_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P],
    "ud_residual_fwd": [_P, _P, _P, _F, _P, _L, _L, _P],
    "ud_axpby": [_P, _F, _P, _F, _P, _L, _I, _P],
    "ud_mask_scale": [_P, _P, _F, _P, _L, _P],
    "ud_absdiff": [_P, _P, _P, _L, _P],
    "ud_bcast_rows": [_P, _F, _P, _I, _I, _I, _P],
    "ud_pix_to_planes": [_P, _P, _I, _I, _I, _I, _P],
    "ud_planes_to_pix": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ud_bilinear_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ud_bilinear_bwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ud_l1_chunks": [_L],
    "ud_l1_fwd": [_P, _P, _P, _P, _I, _L, _F, _P],
    "ud_l1_bwd": [_P, _P, _P, _F, _I, _P, _I, _L, _P],
    "ud_dynfilter_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ud_dynfilter_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "ud_avgpool_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ud_avgpool_bwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "ud_adaptive_avgpool_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ud_adaptive_avgpool_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ud_maxpool3s2_fwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ud_maxpool3s2_bwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ud_add_act_fwd": [_P, _P, _I, _P, _L, _P],
    "ud_relu_bwd": [_P, _P, _P, _L, _P],
    "ud_copy_cols": [_P, _P, _L, _I, _I, _I, _I, _P],
    "ud_aw_triplet": [_P, _I, _I, _I, _P, _P, _P, _P],
    "ud_loss_tail_ws_floats": [_I, _I, _I],
    "ud_loss_tail_run": [C.POINTER(LossTail), _P],
    "ud_conv_small_supported": [_I, _I, _I, _I],
    "ud_conv_small": [C.POINTER(ConvGeom), _P, _P, _P, _I, _P],
    "ud_conv_small_wgrad_supported": [_I, _I, _I, _I],
    "ud_conv_small_wgrad_ws_floats": [_I, _I],
    "ud_conv_small_wgrad": [C.POINTER(ConvGeom), _P, _P, _P, _P, _I, _P],
    "ud_gather2d": [_P, _P, _P, _P, _L, _I, _I, _P],
    "ud_blur5_reflect": [_P, _P, _L, _I, _I, _F, _F, _F, _P],
    "ud_amp_mix": [_P, _P, _P, _P, _L, _I, _I, _I, _P],
    "ud_efdm_ws_bytes": [_I, _I],
    "ud_efdm": [_P, _P, _P, _P, _I, _I, _I, _P, _L, _P],
    "ud_coral_moments": [_P, _P, _I, _I, _I, _P],
    "ud_affine3": [_P, _P, _P, _I, _I, _P],
    # fused MBConv path (csrc/fused.hip, csrc/fft.hip)
    "ud_fused_reduce_ws_doubles": [_I, _I, _I, _I, _I],
    "ud_colstats": [_P, _I, _I, _I, _P, _P, _P, _I, _P],
    "ud_colsum_bn": [_P, _BN, _I, _I, _I, _P, _P, _I, _P],
    "ud_coldot_bn": [_P, _P, _BN, _I, _I, _I, _P, _P, _I, _P],
    "ud_fc_fwd_d": [_P, _F, _P, _P, _P, _I, _I, _I, _P],
    "ud_se_scale_bn": [_P, _BN, _P, _P, _I, _I, _I, _I, _P, _P],
    "ud_colsum_bn_amax": [_P, _BN, _I, _I, _I, _P, _P, _P, _P],
    "ud_se_scale_bn_plane_half": [_P, _BN, _P, _P, _L, _P, _I, _I, _I, _P],
    "ud_se_scale_bn_planes": [_P, _BN, _P, _P, _L, _L, _P, _P, _I, _I, _I, _P],
    "ud_residual_bn": [_P, _BN, _P, _F, _P, _P, _I, _I, _I, _I, _P, _P],
    "ud_residual_bn_planes": [_P, _BN, _P, _F, _P, _P, _P, _P, _L, _L, _P, _I, _I, _I, _P, _P],
    "ud_normbwd_sums": [_P, _P, _P, _F, _BN, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    "ud_normbwd_apply_plane_half": [_P, _P, _P, _F, _BN, _I, _P, _P, _P, _P, _I, _I, _I, _P, _L, _P, _P, _P, _P],
    "ud_normbwd_apply_planes": [_P, _P, _P, _F, _BN, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _L, _L, _P, _P, _P, _P],
    "ud_normbwd_apply": [_P, _P, _P, _F, _BN, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P],
    "ud_pj_bwd_fused_ok": [_I, _I, _I],
    "ud_pj_bwd_fused_grid": [_I, _I, _I, _I],
    "ud_pj_fwd_fused_ok": [_I, _I, _I],
    "ud_pj_fwd_fused": [_P, _BN, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "ud_pj_bwd_fused_a": [_P, _P, _BN, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "ud_pj_bwd_fused_b": [_P, _P, _BN, _P, _P, _F, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    "ud_pw_bwd_fused_ok": [_I, _I],
    "ud_pw_bwd_fused_grid": [_L],
    "ud_pw_bwd_set_form": [_I],
    "ud_pw_bwd_fused": [_P, _P, _BN, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P],
    "ud_normbwd_apply_mix": [_P, _P, _BN, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    "ud_gate_grad_from_acc": [_P, _P, _P, _P],
    "ud_se_bwd_a": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "ud_se_bwd_b": [_P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P],
    "ud_se_scale_bwd_bn": [_P, _P, _BN, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "ud_bn_apply": [_P, _BN, _P, _I, _I, _I, _I, _P],
    "ud_dwconv_bwd_data_bn": [_P, _P, _I, _P, _P, _P, _BN, _P, _P, _P, _P] + [_I] * 10 + [_I, _P],
    "ud_dwconv_bwd_data_bn_ws_doubles": [_I, _I, _I, _I, _I],
    "ud_dwconv_bwd_data_ex": [_P, _P, _I, _P, _P, _P] + [_I] * 10 + [_I, _P],
    "ud_dwconv_bwd_weight_ex": [_P, _P, _P, _I, _P, _P, _I] + [_I] * 10 + [_I, _P],
    "ud_rfft2_ex": [_P, _P, _I, _I, _I, _F, _F, _BN, _P, _P, _I, _P, _P, _I, _P, _P],
    "ud_rfft2_ex_planes": [_P, _P, _L, _L, _P, _F, _P, _I, _I, _I, _F, _F, _BN, _P, _P, _I, _P, _P, _P, _P, _I, _P],
    "ud_irfft2_mix": [_P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _I, _P],
    "ud_rfft2_two_pass": [_P, _P, _P, _I, _I, _I, _F, _F, _BN, _P, _P, _I, _P, _P, _I, _P, _P],
    "ud_irfft2_two_pass": [_P, _P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _I, _P],
    "ud_fft2_two_pass_ws_floats": [_I, _I, _I],
    "ud_dwtile_ws_doubles": [_I, _I, _I, _I],
    "ud_dwtile": [_P, _BN, _P, _P] + [_I] * 10 + [_P, _I, _P, _P, _BN, _I, _P, _P, _P, _I, _I, _P],
    "ud_dwtile_wgrad_part_rows": [_I, _I, _I],
    "ud_dwtile_wgrad": [_P, _BN, _P, _P, _I, _P, _P, _L] + [_I] * 9 + [_I, _I, _P],
    "ud_dwtile_wgrad_finalize": [_P, _I, _I, _I, _P, _I, _P, _P],
    "ud_dwtile_wgrad_finalize_multi": [C.POINTER(WgradFold), _I, _P],
    "ud_rfft2_ex_plane_half": [_P, _P, _L, _P, _I, _I, _I, _F, _F, _BN, _P, _P, _I, _P, _P, _P, _P, _I, _P],
    "ud_irfft2_dwbwd": [_P, _I, _I, _I, _F, _F, _P, _P, _BN, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    "ud_dwtile_bwd": [_P, _P, _BN, _P, _P, _I, _P, _P, _P, _P, _L, _P, _P, _P] + [_I] * 8 + [_P],
    "ud_rfft2_planes_ws_floats": [_L, _I],
    "ud_rfft2_planes": [_P, _P, _P, _L, _I, _F, _P],
    "ud_rfft2_planes_adjoint": [_P, _P, _P, _L, _I, _F, _P],
    "ud_adamw_chunk_elems": [],
    "ud_adamw_multi": [_P, _P, _I, _P, _P, _I, C.c_double, C.c_double, C.c_double, _I, _I, _P, _P, _P, _P, _P],
    "ud_multi_add": [_P, _P, _P, _I, _P],
    "ud_gemm_get_path": [],
    "ud_sum_slices": [_P, _P, _I, _L, _L, _I, _P],
    "ud_xchg_bytes": [_I, _I, _I],
    "ud_xchg_create": [_I, _I, _I, _P, _P],
    "ud_xchg_open": [_P, _P],
    "ud_xchg_close": [_P],
    "ud_xchg_destroy": [_P],
    "ud_xchg_allreduce": [_P, _I, _P, _I, _I, _I, _I, _P, _P, _L, _P, _P],
}

# helpers that return a count rather than a status code
_COUNT_FUNCS = {"ud_pj_fwd_fused_ok", "ud_pj_bwd_fused_ok", "ud_pj_bwd_fused_grid", "ud_pw_bwd_fused_ok", "ud_pw_bwd_fused_grid", "ud_loss_tail_ws_floats", "ud_norm_fused_ws_doubles", "ud_norm_fused_counters", "ud_dwtile_wgrad", "ud_dwtile_bwd", "ud_fft32_set_wave", "ud_fft2_two_pass_ws_floats", "ud_dwtile_ws_doubles", "ud_dwtile_wgrad_part_rows", "ud_reduce_ws_doubles", "ud_gemm_query_path", "ud_gemm_get_path", "ud_gemm_stats_slots", "ud_adamw_chunk_elems", "ud_rfft2_planes_ws_floats", "ud_fused_reduce_ws_doubles", "ud_dwconv_bwd_data_bn_ws_doubles", "ud_dwconv_bwd_weight_parts", "ud_sfmix_blocks", "ud_gate_mix_blocks",
                "ud_l1_chunks", "ud_efdm_ws_bytes", "ud_conv_small_supported", "ud_conv_small_wgrad_supported",
                "ud_conv_small_wgrad_ws_floats", "ud_xchg_bytes"}
_LONG_FUNCS = {"ud_pj_bwd_fused_grid", "ud_pw_bwd_fused_grid", "ud_norm_fused_ws_doubles", "ud_fft2_two_pass_ws_floats", "ud_dwtile_ws_doubles", "ud_dwtile_wgrad_part_rows", "ud_xchg_bytes", "ud_efdm_ws_bytes", "ud_rfft2_planes_ws_floats", "ud_conv_small_wgrad_ws_floats", "ud_fused_reduce_ws_doubles",
               "ud_dwconv_bwd_data_bn_ws_doubles"}        # return a C long

EXPORTED = tuple(_SIGNATURES)

_lib = None


def load():
    """Load (once) and return the ctypes library; raises UDLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UDLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C unidefense_amd/csrc`. unidefense_amd has no fallback path.")
    # The library is linked against /opt/rocm's libamdhip64 while torch ships its own copy.  Loading it BEFORE torch
    # has initialised its HIP runtime leaves this library's kernels registered with a runtime that owns no device
    # (every launch then fails with hipErrorNoDevice) — measured with build() followed by smoke() in one process.
    # So: let torch bring the GPU up first (a no-op on a box without one, where only the symbol table is checked).
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise UDLibraryError(f"{LIB_PATH} does not export {name}; rebuild it")
        fn.argtypes = argtypes
        fn.restype = C.c_long if name in _LONG_FUNCS else C.c_int
    _lib = lib
    return lib


def check(status: int, name: str):
    if status != 0:
        raise UDLibraryError(f"{name} failed with status {status}"
                             + (" (invalid argument)" if status == -1000 else " (-hipError_t)"))


def call(name: str, *args):
    """Call a status-returning entry point and raise on failure."""
    fn = getattr(load(), name)
    st = fn(*args)
    if name in _COUNT_FUNCS:
        if st < 0:
            check(st, name)
        return st
    check(st, name)
    return 0
