"""ctypes binding of libglowtts_hip.so (the C-ABI declared in include/glowtts_hip.h).

The library is the product: there is NO fallback.  `lib()` raises if it is missing or does
not export a declared symbol.  PyTorch is used by callers only for device memory and streams.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GT_LIB") or os.path.join(_HERE, "libglowtts_hip.so")     # GT_LIB: dev (an experiment build, tools/exp_variant.py)

c_void_p, c_int, c_i64, c_size_t, c_float, c_u32 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                    ctypes.c_size_t, ctypes.c_float, ctypes.c_uint32)

# name -> (restype, argtypes); mirrors include/glowtts_hip.h one to one
PROTOTYPES = {
    "gt_version": (ctypes.c_char_p, []),
    "gt_mas_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                           c_int, c_int, c_int, c_i64, c_i64, c_void_p, c_size_t, c_void_p, c_void_p]),
    "gt_mas_lds_bytes": (c_size_t, [c_int, c_int]),
    "gt_mas_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gt_mas_lengths_from_mask_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                             c_i64, c_i64, c_void_p]),
    "gt_conv_gemm_bf16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                  c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_float, c_u32, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gt_conv_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_void_p]),
    "gt_conv_wgrad_ci_tile": (c_int, [c_int]),
    "gt_conv_wgrad_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "gt_weightnorm_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_int, c_int, c_void_p]),
    "gt_conv_wgrad_batched": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gt_weightnorm_bwd_batched": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "gt_adamw_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "gt_colsum": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "gt_squeeze_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gt_unsqueeze_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gt_flow_scalars": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "gt_flow_scalars_multi": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "gt_actnorm_ddi": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gt_actnorm_invconv_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gt_actnorm_invconv_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gt_actnorm_invconv_rev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gt_coupling_rev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gt_coupling_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "gt_coupling_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "gt_gate_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                            c_int, c_int, c_float, c_u32, c_void_p, c_void_p]),
    "gt_relu_drop_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "gt_rows_add_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gt_rows_f32_to_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "gt_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_u32, c_float, c_u32, c_int, c_void_p, c_void_p]),
    "gt_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_float, c_float, c_u32, c_float, c_u32, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "gt_layernorm_bwd_partial_rows": (c_int, [c_int]),
    "gt_layernorm_bwd_partials": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                          c_float, c_float, c_u32, c_float, c_u32, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                          c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gt_param_partials_reduce": (c_int, [c_void_p, c_void_p]),
    "gt_dds_bwd_partial_rows": (c_int, [c_int]),
    "gt_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                            c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_u32, c_void_p, c_void_p]),
    "gt_attn_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gt_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                            c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                            c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_u32, c_void_p, c_void_p]),
    "gt_embedding_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "gt_embedding_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "gt_rows_add_cond": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                 c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gt_rows_ctx_fill": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gt_step_inputs": (c_int, [c_void_p, c_void_p]),
    "gt_step_zero": (c_int, [c_void_p, c_void_p]),
    "gt_rows_utt_sum": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gt_logp_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "gt_prior_expand": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "gt_rows_from_bct": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gt_bct_from_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "gt_mle_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gt_duration_loss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gt_duration_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gt_prior_expand_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "gt_mle_sums": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gt_mle_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p]),
    "gt_mark": (c_int, [c_void_p, c_void_p]),
    "gt_length_mask": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "gt_wn_layer_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                                c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                c_int, c_int, c_int, c_float, c_u32, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gt_wn_layer_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_u32,
                                c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gt_wn_stack_fwd": (c_int, [c_void_p, c_void_p]),
    "gt_wn_stack_rows_per_workgroup": (c_int, [c_int]),
    "gt_cond_affine_grads": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gt_wn_stack_bwd": (c_int, [c_void_p, c_void_p]),
    "gt_wn_boundary_fwd": (c_int, [c_void_p, c_void_p]),
    "gt_wn_boundary_bwd": (c_int, [c_void_p, c_void_p]),
    "gt_boundary_param_partials": (c_int, []),
    "gt_boundary_param_reduce": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gt_rows_split3": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "gt_dds_sep_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "gt_dds_out_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_u32, c_void_p, c_void_p]),
    "gt_dds_out_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_u32, c_void_p, c_void_p]),
    "gt_dds_sep_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "gt_dds_dw_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "gt_convflow_pre_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gt_convflow_pre_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "gt_convflow_spline_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p]),
    "gt_convflow_spline_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p]),
    "gt_convflow_spline_partial_rows": (c_int, [c_int]),
    "gt_convflow_spline_partial_width": (c_int, []),
    "gt_convflow_spline_inv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gt_ea_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p]),
    "gt_ea_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "gt_sdp_mid_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gt_sdp_mid_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gt_nll_gauss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gt_nll_gauss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gt_rows_gather_tokens": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "gt_pack_conv_weights_multi": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "gt_pack_conv_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_void_p]),
}



class PackDesc(ctypes.Structure):
    """struct gt_pack_desc (include/glowtts_hip.h)"""
    _fields_ = [("v", c_void_p), ("g", c_void_p), ("pack_fwd", c_void_p), ("pack_dgrad", c_void_p), ("inv_norm", c_void_p),
                ("Cout", ctypes.c_int32), ("Cin", ctypes.c_int32), ("taps", ctypes.c_int32), ("Np_fwd", ctypes.c_int32),
                ("Kp_fwd", ctypes.c_int32), ("Np_dgrad", ctypes.c_int32), ("Kp_dgrad", ctypes.c_int32), ("gate", ctypes.c_int32),
                ("row_start", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class StepCopy(ctypes.Structure):
    """struct gt_step_copy (include/glowtts_hip.h)"""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("rows", ctypes.c_int32), ("src_words", ctypes.c_int32),
                ("dst_words", ctypes.c_int32), ("blk0", ctypes.c_int32)]


class StepCtx(ctypes.Structure):
    """struct gt_step_ctx (include/glowtts_hip.h)"""
    _fields_ = [("geo_src", c_void_p), ("geo_dst", c_void_p), ("rowbatch", c_void_p), ("rowframe", c_void_p), ("rowmask", c_void_p),
                ("rowutt", c_void_p), ("B", ctypes.c_int32), ("R", ctypes.c_int32), ("blk0", ctypes.c_int32), ("pad_", ctypes.c_int32)]


STEP_MAX_COPIES, STEP_MAX_CTX, STEP_MAX_B = 10, 3, 1024


class StepInputsArgs(ctypes.Structure):
    """struct gt_step_inputs_args (include/glowtts_hip.h)"""
    _fields_ = [("copy", StepCopy * STEP_MAX_COPIES), ("ctx", StepCtx * STEP_MAX_CTX), ("n_copy", ctypes.c_int32), ("n_ctx", ctypes.c_int32)]


class PartialsJob(ctypes.Structure):
    """struct gt_partials_job (include/glowtts_hip.h)"""
    _fields_ = [("partials", c_void_p), ("dst_a", c_void_p), ("dst_b", c_void_p), ("n_rows", ctypes.c_int32), ("Ca", ctypes.c_int32),
                ("Cb", ctypes.c_int32), ("pad_", ctypes.c_int32)]


PARTIALS_MAX = 32


class PartialsArgs(ctypes.Structure):
    """struct gt_partials_args (include/glowtts_hip.h)"""
    _fields_ = [("job", PartialsJob * PARTIALS_MAX), ("n_jobs", ctypes.c_int32)]


ZERO_MAX = 4


class StepZeroArgs(ctypes.Structure):
    """struct gt_step_zero_args (include/glowtts_hip.h)"""
    _fields_ = [("ptr", c_void_p * ZERO_MAX), ("bytes", ctypes.c_uint64 * ZERO_MAX), ("seed_word", c_void_p), ("seed_inc", ctypes.c_uint32),
                ("n", ctypes.c_int32)]


class WnStackFwdArgs(ctypes.Structure):
    """struct gt_wn_stack_fwd_args (include/glowtts_hip.h)"""
    _fields_ = [("x0", c_void_p), ("w_in", c_void_p * 4), ("b_in", c_void_p * 4), ("w_res", c_void_p * 4), ("b_res", c_void_p * 4),
                ("cond", c_void_p), ("ldc", c_int), ("row0", c_void_p), ("B", c_int), ("Tp", c_int), ("rowmask", c_void_p),
                ("acts", c_void_p), ("ldacts", c_int), ("gate_t", c_void_p * 4), ("gate_s", c_void_p * 4), ("x_out", c_void_p * 4),
                ("R", c_int), ("H", c_int), ("taps", c_int), ("n_layers", c_int), ("drop_p", c_float), ("drop_seed", c_u32),
                ("seed_dev", c_void_p), ("stamps", c_void_p), ("stamp_slot", c_int), ("stamp_base", c_void_p),
                ("aff_w", c_void_p), ("aff_b", c_void_p), ("aff_sig", c_void_p)]


class WnStackBwdArgs(ctypes.Structure):
    """struct gt_wn_stack_bwd_args (include/glowtts_hip.h)"""
    _fields_ = [("via_skip", c_void_p), ("ldvs", c_int), ("gate_t", c_void_p * 4), ("gate_s", c_void_p * 4), ("w_in_d", c_void_p * 4),
                ("w_res_d", c_void_p * 4), ("rowmask", c_void_p), ("dpre", c_void_p * 4), ("dpre_c", c_void_p * 4), ("dx", c_void_p * 4),
                ("R", c_int), ("H", c_int), ("taps", c_int), ("n_layers", c_int), ("drop_p", c_float), ("drop_seed", c_u32),
                ("seed_dev", c_void_p)]


class BoundaryFwdArgs(ctypes.Structure):
    """struct gt_boundary_fwd_args (include/glowtts_hip.h); pointer fields take tensor.data_ptr() or None"""
    _fields_ = [("acts", c_void_p), ("ldacts", c_int), ("w_skip", c_void_p), ("b_skip", c_void_p),
                ("w_end", c_void_p), ("b_end", c_void_p), ("ks_end", c_int), ("y", c_void_p), ("wn_out", c_void_p),
                ("logs_raw", c_void_p), ("z", c_void_p), ("logdet", c_void_p), ("rowutt", c_void_p), ("sigmoid_scale", c_int),
                ("x_in", c_void_p), ("an_logs", c_void_p), ("an_bias", c_void_p), ("w_ic", c_void_p), ("scal", c_void_p),
                ("len", c_void_p), ("B", c_int), ("y_next", c_void_p), ("y0_bf16", c_void_p), ("w_start", c_void_p),
                ("b_start", c_void_p), ("ks_start", c_int), ("h_next", c_void_p), ("rowmask", c_void_p),
                ("R", c_int), ("H", c_int), ("C", c_int), ("n_layers", c_int),
                ("y_bct", c_void_p), ("z_bct", c_void_p), ("T", c_int), ("rowbatch", c_void_p), ("rowframe", c_void_p),
                ("pf_ptr", c_void_p * 16), ("pf_bytes", c_u32 * 16)]


class BoundaryBwdArgs(ctypes.Structure):
    """struct gt_boundary_bwd_args (include/glowtts_hip.h)"""
    _fields_ = [("dh", c_void_p), ("w_start_d", c_void_p), ("ks_start_d", c_int), ("dx_in", c_void_p), ("x", c_void_p),
                ("an_logs", c_void_p), ("an_bias", c_void_p), ("w_ic", c_void_p), ("scal", c_void_p), ("len", c_void_p), ("B", c_int),
                ("d_an_logs", c_void_p), ("d_an_bias", c_void_p), ("d_w_ic", c_void_p),
                ("dz_in", c_void_p), ("logs_raw", c_void_p), ("y", c_void_p), ("dlogdet", c_void_p), ("rowutt", c_void_p),
                ("sigmoid_scale", c_int), ("dx_out", c_void_p), ("dout", c_void_p), ("w_end_d", c_void_p), ("ks_end_d", c_int),
                ("dwn_out", c_void_p), ("w_skip_d", c_void_p), ("ks_skip_d", c_int), ("via_skip", c_void_p), ("ldvs", c_int),
                ("rowmask", c_void_p), ("R", c_int), ("H", c_int), ("C", c_int), ("n_layers", c_int),
                ("dz_bct", c_void_p), ("dx_bct", c_void_p), ("T", c_int), ("rowbatch", c_void_p), ("rowframe", c_void_p),
                ("pg_partial", c_void_p), ("pf_ptr", c_void_p * 16), ("pf_bytes", c_u32 * 16)]


def fill_args(cls, **kw):
    """ctypes struct from keyword arguments: tensors become device pointers, None stays NULL, ints stay ints."""
    a = cls()
    for k, v in kw.items():
        if isinstance(v, (list, tuple)):                  # pointer arrays: tensors / None per entry
            arr = getattr(a, k)
            for i, t in enumerate(v):
                arr[i] = None if t is None else (t.data_ptr() if hasattr(t, "data_ptr") else t)
        else:
            setattr(a, k, v.data_ptr() if hasattr(v, "data_ptr") else v)
    return a


GT_TILE_AUTO, GT_TILE_64x64, GT_TILE_64x128, GT_TILE_128x64, GT_TILE_128x128, GT_TILE_256x64, GT_TILE_64x64_TAPS = 0, 1, 2, 3, 4, 5, 6
GT_DT_F32, GT_DT_I32, GT_DT_F16, GT_DT_BF16, GT_DT_U8 = 0, 1, 2, 3, 4
GT_ERRORS = {-1: "GT_E_INVAL", -2: "GT_E_UNSUPPORTED", -3: "GT_E_ALIGN", -4: "GT_E_LAUNCH"}

_LIB = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the HIP library; raise loudly if it is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: the HIP extension is not built.  Run "
                "`python glow-tts_amd/build.py` (or __graft_entry__.build()).  There is no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64: it must be in the process BEFORE our library is loaded,
        # otherwise the loader binds us to a second HIP runtime (/opt/rocm) that then reports
        # "no ROCm-capable device" next to torch's.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise HipLibraryMissing(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("GT_TRACE_CALLS"):
            L = _Traced(L, os.environ["GT_TRACE_CALLS"])
        _LIB = L
    return _LIB


class _Traced:
    """dev (GT_TRACE_CALLS=<file>): the name of every C-ABI entry is written to the file BEFORE the call and the device is
    synchronised after it — after a GPU memory fault (which kills the process) the file names the launch that faulted."""

    def __init__(self, L, path):
        self._L, self._f, self._n = L, open(path, "w"), 0

    def __getattr__(self, name):
        fn = getattr(self._L, name)
        if not name.startswith("gt_"):
            return fn

        def call(*a):
            import torch
            self._n += 1
            self._f.seek(0); self._f.write(f"{self._n} {name}".ljust(96) + "\n"); self._f.flush()
            rc = fn(*a)
            if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize()
            self._f.seek(0); self._f.write(f"{self._n} {name} returned".ljust(96) + "\n"); self._f.flush()
            return rc
        return call


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {GT_ERRORS.get(rc, rc)}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("glow_tts_amd ops run on the MI355X only (got a CPU tensor); "
                               "there is no CPU fallback — use oracle/ for CPU checking in tests")
