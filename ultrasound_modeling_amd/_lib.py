"""ctypes binding of libusseg_hip.so (the C ABI declared in include/usseg.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: the HIP runtime torch loads is the one this library must share)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("USSEG_LIB") or os.path.join(_HERE, "libusseg_hip.so")   # USSEG_LIB: a diagnostic build of the SAME sources (tools/diag_norm_variants.sh)

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class UssegError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [("B", c_i32), ("H", c_i32), ("W", c_i32), ("Cin", c_i32), ("Cout", c_i32), ("ldx", c_i32),
                ("ldy", c_i32), ("ksize", c_i32), ("dilation", c_i32), ("act", c_i32), ("alpha", c_f32),
                ("flags", c_i32)]


class ConvJob(C.Structure):
    _fields_ = [("desc", ConvDesc), ("x", c_vp), ("wp", c_vp), ("bias", c_vp), ("residual", c_vp), ("ldr", c_i32),
                ("reserved", c_i32), ("y", c_vp), ("scale", c_vp)]


class WgradBlock(C.Structure):
    _fields_ = [("dst", c_vp), ("sT", c_i64), ("sI", c_i64), ("sO", c_i64), ("i_off", c_i32), ("o_off", c_i32), ("ni", c_i32), ("no", c_i32)]


class WgradDst(C.Structure):
    _fields_ = [("nblocks", c_i32), ("reserved", c_i32), ("blk", WgradBlock * 4)]


P_WgradDst = C.POINTER(WgradDst)


class WgradJob(C.Structure):
    _fields_ = [("desc", ConvDesc), ("x", c_vp), ("dy", c_vp), ("dw", c_vp), ("dst", P_WgradDst)]


class AugSample(C.Structure):
    _fields_ = [("do_reduc", c_i32), ("nclip", c_i32), ("clip", (c_i32 * 4) * 2), ("do_shift", c_i32), ("shift_r", c_i32),
                ("shift_c", c_i32), ("shift_dir", c_i32), ("do_noise", c_i32), ("reserved", c_i32), ("seed", C.c_uint64)]


class AugDesc(C.Structure):
    _fields_ = [("B", c_i32), ("H", c_i32), ("W", c_i32), ("C", c_i32), ("Cphys", c_i32), ("num_classes", c_i32)]


class NormDesc(C.Structure):
    _fields_ = [("M", c_i64), ("C", c_i32), ("Cphys", c_i32), ("ldx", c_i32), ("ldy", c_i32), ("G", c_i32),
                ("mode", c_i32), ("eps", c_f32), ("act", c_i32), ("alpha", c_f32), ("lddx", c_i32)]


class CardinalDesc(C.Structure):
    _fields_ = [("B", c_i32), ("H", c_i32), ("W", c_i32), ("Cin", c_i32), ("P", c_i32), ("cv11", c_i32), ("cvkk", c_i32), ("Up", c_i32),
                ("Vp", c_i32), ("Oc", c_i32), ("ldx", c_i32), ("ldu", c_i32), ("ldv", c_i32), ("ldsc", c_i32), ("eps", c_f32), ("alpha", c_f32)]


class SplitAttnDesc(C.Structure):
    _fields_ = [("B", c_i32), ("HW", c_i32), ("P", c_i32), ("R", c_i32), ("Cg", c_i32), ("Hd", c_i32),
                ("ldy", c_i32), ("ldo", c_i32), ("Cy_phys", c_i32), ("Co_phys", c_i32), ("mult", c_f32),
                ("norm_mode", c_i32), ("eps", c_f32), ("act", c_i32), ("alpha", c_f32), ("use_sigmoid", c_i32)]


class SplitAttnParams(C.Structure):
    _fields_ = [(n, c_vp) for n in ("w1", "b1", "gamma", "beta", "mean", "var", "w2", "b2")]


class SplitAttnGrads(C.Structure):
    _fields_ = [(n, c_vp) for n in ("w1", "b1", "gamma", "beta", "w2", "b2")]


class GemmDesc(C.Structure):
    _fields_ = [("M", c_i32), ("N", c_i32), ("K", c_i32), ("ldx", c_i32), ("ldw", c_i32), ("ldy", c_i32), ("nb1", c_i32), ("nb2", c_i32),
                ("xs1", c_i64), ("xs2", c_i64), ("ws1", c_i64), ("ws2", c_i64), ("ys1", c_i64), ("ys2", c_i64), ("flags", c_i32)]


class FlashDesc(C.Structure):
    _fields_ = [("B", c_i32), ("N", c_i32), ("H", c_i32), ("head_dim", c_i32), ("ld_qkv", c_i32), ("ld_o", c_i32), ("scale", c_f32)]


class LossDesc(C.Structure):
    _fields_ = [("M", c_i64), ("HW", c_i32), ("C", c_i32), ("ldl", c_i32), ("lddl", c_i32), ("loss_kind", c_i32),
                ("label_smoothing", c_f32), ("clip_eps", c_f32), ("inv_global_batch", c_f32), ("quad_w", c_i32)]


ACT_NONE, ACT_LRELU, ACT_RELU, ACT_ELU, ACT_GELU = 0, 1, 2, 3, 4
OUT_F32, ACCUMULATE = 1, 2

P = C.POINTER
_PROTOS = {
    # name: (restype, argtypes)
    "usseg_last_error": (C.c_char_p, []),
    "usseg_version": (C.c_int, []),
    "usseg_conv2d_fwd": (C.c_int, [P(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "usseg_conv2d_fwd_affine": (C.c_int, [P(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "usseg_bn_fold_batched": (C.c_int, [c_vp, c_i32, c_vp]),
    "usseg_conv2d_dgrad": (C.c_int, [P(ConvDesc), c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "usseg_conv2d_wgrad": (C.c_int, [P(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "usseg_conv2d_wgrad_mapped": (C.c_int, [P(ConvDesc), c_vp, c_vp, P(WgradDst), c_vp, c_i64, c_vp]),
    "usseg_tconv2d_wgrad_mapped": (C.c_int, [P(ConvDesc), c_vp, c_vp, P(WgradDst), c_vp, c_i64, c_vp]),
    "usseg_conv2d_dgrad_branches": (C.c_int, [P(ConvDesc), c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "usseg_conv2d_wgrad_multi": (C.c_int, [c_i32, c_vp, c_vp, c_i64, c_vp]),
    "usseg_conv2d_fwd_multi": (C.c_int, [c_i32, c_vp, c_vp]),
    "usseg_conv2d_dgrad_multi": (C.c_int, [c_i32, c_vp, c_vp]),
    "usseg_tconv2d_fwd": (C.c_int, [P(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_tconv2d_dgrad": (C.c_int, [P(ConvDesc), c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "usseg_tconv2d_wgrad": (C.c_int, [P(ConvDesc), c_vp, c_vp, c_vp, c_vp]),
    "usseg_pack_weight": (C.c_int, [c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "usseg_pack_weights_batched": (C.c_int, [c_vp, c_i32, c_vp]),
    "usseg_pack_weights_flat": (C.c_int, [c_vp, c_vp, c_i32, c_vp]),
    "usseg_unpack_wgrad": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_i64, c_i64,
                                     c_f32, c_i32, c_vp]),
    "usseg_unpack_wgrad_batched": (C.c_int, [c_vp, c_i32, c_i32, c_vp]),
    "usseg_norm_act_fwd": (C.c_int, [P(NormDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "usseg_dropout_mask": (C.c_int, [c_vp, c_i64, c_i32, c_i32, C.c_uint64, c_f32, c_vp]),
    "usseg_dropout_mask_step": (C.c_int, [c_vp, c_i64, c_i32, c_i32, C.c_uint64, c_vp, c_f32, c_vp]),
    "usseg_norm_act_bwd": (C.c_int, [P(NormDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_bn_act_pool_fwd": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_f32, c_vp, c_i32, c_vp]),
    "usseg_bn_act_pool_bwd": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_f32,
                                        c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_reduce_ws_floats": (c_i64, []),
    "usseg_channel_stats": (C.c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "usseg_bn_finalize_stats": (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_bn_train_bwd_fix": (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp]),
    "usseg_act_fwd": (C.c_int, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp]),
    "usseg_act_bwd": (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp]),
    "usseg_act_bwd_colsum": (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "usseg_avgpool2_fwd": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "usseg_avgpool2_bwd": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "usseg_avgpool2_bwd_colsum": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "usseg_copy_channels": (C.c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp]),
    "usseg_cast_input": (C.c_int, [c_vp, c_i32, c_i64, c_i32, c_vp, c_i32, c_vp]),
    "usseg_cast_bf16_to_f32": (C.c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "usseg_splitattn_gap": (C.c_int, [P(SplitAttnDesc), c_vp, c_vp, c_vp, c_vp]),
    "usseg_splitattn_ws_floats": (c_i64, [P(SplitAttnDesc)]),
    "usseg_splitattn_mlp_fwd": (C.c_int, [P(SplitAttnDesc), c_vp, c_i32, c_i32, P(SplitAttnParams), c_vp, c_vp, c_vp]),
    "usseg_norm_act_fwd_gap": (C.c_int, [P(NormDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    # fused cardinal group + shortcut of a residual_S stage (ResNest.py:99-101,136-147): the ten Keras layers of a stage's first half
    # fused encoder stem (ResNest.py:39-47): three convs, two BatchNorms, the activations and the pool
    "usseg_stem_fwd": (C.c_int, [c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32,
                                 c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_conv3_dgrad_actbwd": (C.c_int, [c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_i32,
                                           c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_cardinal_supported": (c_i32, [P(CardinalDesc)]),
    "usseg_cardinal_fwd": (C.c_int, [P(CardinalDesc)] + [c_vp] * 21),
    # ... and its backward pass: (re-weighting + conv2_bn) backward -> grouped 3x3 backward-data -> conv1_bn backward | shortcut norm backward
    "usseg_cardinal_bwd": (C.c_int, [P(CardinalDesc), c_vp, c_i32, c_vp, c_i32] + [c_vp] * 12 + [c_f32, c_vp, c_vp, c_i32] + [c_vp] * 11),
    "usseg_accuracy": (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "usseg_norm_act_bwd_res": (C.c_int, [P(NormDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_norm_act_bwd_sa": (C.c_int, [P(NormDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_norm_act_bwd_pair": (C.c_int, [P(NormDesc)] + [c_vp] * 8 + [P(NormDesc)] + [c_vp] * 4 + [c_i32, c_vp, c_vp, c_f32] + [c_vp] * 6),
    "usseg_splitattn_apply_fwd": (C.c_int, [P(SplitAttnDesc), c_vp, c_vp, c_vp, c_vp]),
    "usseg_splitattn_apply_bwd_reduce": (C.c_int, [P(SplitAttnDesc), c_vp, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "usseg_splitattn_mlp_bwd_ws_floats": (c_i64, [P(SplitAttnDesc)]),
    "usseg_splitattn_bwd_fused": (C.c_int, [P(SplitAttnDesc), c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, P(SplitAttnParams), c_vp, c_vp, c_vp,
                                            P(SplitAttnGrads), c_vp, c_vp, c_vp]),
    "usseg_splitattn_mlp_bwd": (C.c_int, [P(SplitAttnDesc), c_vp, c_i32, c_i32, P(SplitAttnParams), c_vp, c_vp, c_vp, c_vp,
                                          P(SplitAttnGrads), c_vp, c_vp]),
    "usseg_splitattn_apply_bwd_dy": (C.c_int, [P(SplitAttnDesc), c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "usseg_softmax_loss_fwd_bwd": (C.c_int, [P(LossDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_head_quad_softmax_loss": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_vp]),
    "usseg_loss_from_probs": (C.c_int, [P(LossDesc), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_quad_bias_expand": (C.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "usseg_space_to_depth2": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "usseg_tconv_quad_fwd": (C.c_int, [P(ConvDesc), c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_tconv_quad_dgrad": (C.c_int, [P(ConvDesc), c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "usseg_tconv_quad_wgrad": (C.c_int, [P(ConvDesc), c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "usseg_quad_bias_fold": (C.c_int, [c_vp, c_i32, c_vp, c_vp]),
    "usseg_tconv_quad_unpack": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "usseg_quad_head_fold": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "usseg_label2vec": (C.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    "usseg_augment": (C.c_int, [P(AugDesc), c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_loss_cat_scale": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "usseg_patchify": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "usseg_patch_merge": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp]),
    "usseg_ln_wide_fwd": (C.c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_vp, c_i32, c_vp]),
    "usseg_ln_wide_bwd": (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_f32, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "usseg_window_attn_fwd": (C.c_int, [c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "usseg_window_attn_bwd_ws_floats": (c_i64, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "usseg_window_attn_bwd": (C.c_int, [c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "usseg_token_mean_fwd": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "usseg_token_mean_bwd": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "usseg_colsum": (C.c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "usseg_sumsq": (C.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "usseg_sum_f32": (C.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "usseg_sumsq_advance": (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_vp]),
    "usseg_reinject_hidden": (C.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "usseg_adam_clip_step": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_f32, c_vp, c_f32, c_f32, c_f32, c_vp]),
    "usseg_adam_advance": (C.c_int, [c_vp, c_vp, c_f32, c_f32, c_f32, c_vp]),
    "usseg_fill_f32": (C.c_int, [c_vp, c_i64, c_f32, c_vp]),
    "usseg_scale_f32": (C.c_int, [c_vp, c_i64, c_vp, c_f32, c_vp]),
    "usseg_gemm_nt_batched": (C.c_int, [P(GemmDesc), c_vp, c_vp, c_vp, c_vp]),
    "usseg_gemm_tn_batched": (C.c_int, [P(GemmDesc), c_vp, c_vp, c_vp, c_vp]),
    "usseg_flash_attn_fwd": (C.c_int, [P(FlashDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_flash_attn_bwd": (C.c_int, [P(FlashDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "usseg_softmax_rows_fwd": (C.c_int, [c_vp, c_i64, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "usseg_softmax_rows_bwd": (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp]),
    "usseg_transpose_batched": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_vp, c_vp]),
    "usseg_cast_f32_to_bf16_batched": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i64, c_i64, c_vp]),
    "usseg_defer_begin": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_i64]),
    "usseg_defer_flush": (C.c_int, [c_vp]),
    "usseg_defer_end": (C.c_int, [c_vp]),
    "usseg_prof_enable": (C.c_int, [c_i32, c_i32]),
    "usseg_prof_read": (C.c_int, [c_i32, P(C.c_double), P(c_i64)]),
    "usseg_prof_disable": (C.c_int, []),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib = None


def load() -> C.CDLL:
    """Load libusseg_hip.so (built by ``__graft_entry__.build()`` / ``make -C ultrasound_modeling_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UssegError(f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                         f"g.build()'). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().usseg_last_error().decode("utf-8", "replace")
        raise UssegError(f"{what}: usseg error {rc}: {msg}")
