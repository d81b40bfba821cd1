"""ctypes binding of libtok_gfx950.so (the C ABI declared in include/tok.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950) and
lives next to this package (``torchok_amd/lib``).  There is NO fallback: if the shared object
is missing, or a tensor is not on a HIP device, the product path raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TOK_LIB') or os.path.join(_HERE, 'lib', 'libtok_gfx950.so')   # TOK_LIB: A/B runs against another build

TOK_F32, TOK_F16, TOK_BF16 = 0, 1, 2
TOK_CE_LOSS_FLOATS = 2050


class ConvDesc(Structure):
    """Mirror of ``tok_conv_desc`` (include/tok.h)."""
    _fields_ = [('n', c_int32), ('h', c_int32), ('w', c_int32), ('c', c_int32),
                ('k', c_int32), ('r', c_int32), ('s', c_int32),
                ('p', c_int32), ('q', c_int32),
                ('stride', c_int32), ('pad', c_int32), ('s_pad', c_int32)]


class PackItem(Structure):
    """Mirror of ``tok_pack_item`` (include/tok.h)."""
    _fields_ = [('src', c_void_p), ('dst_fwd', c_void_p), ('dst_dgrad', c_void_p),
                ('k', c_int32), ('r', c_int32), ('s', c_int32), ('c', c_int32),
                ('k_pad', c_int32), ('s_pad', c_int32), ('c_pad', c_int32), ('block_start', c_int32)]


class BnFused(Structure):
    """Mirror of ``tok_bn_fused`` (include/tok.h)."""
    _fields_ = [('counters', c_void_p), ('count', c_int64), ('c_real', c_int32), ('param_accumulate', c_int32),
                ('momentum', c_float), ('eps', c_float), ('gamma', c_void_p), ('beta', c_void_p),
                ('running_mean', c_void_p), ('running_var', c_void_p), ('nbt', c_void_p), ('mean', c_void_p),
                ('rstd', c_void_p), ('scale', c_void_p), ('shift', c_void_p), ('dgamma', c_void_p),
                ('dbeta', c_void_p), ('coef', c_void_p)]


_P = c_void_p
_PD = POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol include/tok.h declares
PROTOTYPES = {
    'tok_last_error': (c_char_p, []),
    'tok_version': (c_int, []),
    'tok_nchw_to_nhwc_bf16': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    'tok_cast_f32_bf16': (c_int, [_P, _P, c_size_t, _P]),
    'tok_cast_bf16_f32': (c_int, [_P, _P, c_float, c_size_t, _P]),
    'tok_pack_weight_fwd': (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P]),
    'tok_pack_weight_dgrad': (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    'tok_pack_weight_both': (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, _P]),
    'tok_pack_item_blocks': (c_int, [POINTER(PackItem)]),
    'tok_pack_weights_batched': (c_int, [_P, c_int, c_int, _P]),
    'tok_conv_fwd_stat_rows': (c_int, [_PD]),
    'tok_conv_fwd': (c_int, [_PD, _P, _P, _P, _P, _P, _P]),
    'tok_conv_dgrad': (c_int, [_PD, _P, _P, _P, c_int, _P]),
    'tok_conv_dgrad_stat_rows': (c_int, [_PD]),
    'tok_conv_dgrad_bnstats': (c_int, [_PD, _P, _P, _P, c_int, _P, _P, _P, _P]),
    'tok_conv_fwd_bn': (c_int, [_PD, _P, _P, _P, _P, POINTER(BnFused), _P]),
    'tok_conv_dgrad_bn': (c_int, [_PD, _P, _P, _P, c_int, _P, _P, _P, POINTER(BnFused), _P]),
    'tok_conv_wgrad_ws_bytes': (c_size_t, [_PD]),
    'tok_conv_wgrad': (c_int, [_PD, _P, _P, _P, c_int, c_int, _P, c_size_t, c_int, _P]),
    'tok_conv_wgrad_bias_ok': (c_int, [_PD]),
    'tok_conv_wgrad_bias_ws_bytes': (c_size_t, [_PD]),
    'tok_conv_wgrad_bias': (c_int, [_PD, _P, _P, _P, c_int, c_int, _P, c_size_t, c_int, _P, c_int, _P]),
    'tok_bn_finalize': (c_int, [_P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P, _P, c_float, c_float,
                                _P, _P, _P, _P, _P]),
    'tok_bn_eval_coeffs': (c_int, [_P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P]),
    'tok_bn_stats_rows': (c_int, [c_int64, c_int]),
    'tok_bn_stats': (c_int, [_P, c_int64, c_int, _P, _P]),
    'tok_bn_act_fwd': (c_int, [_P, _P, _P, _P, c_int, _P, _P, c_int64, c_int, _P]),
    'tok_bn_act_fwd_colsum_rows': (c_int, [c_int64, c_int]),
    'tok_event_create': (_P, []),
    'tok_event_destroy': (c_int, [_P]),
    'tok_next_launch_event': (c_int, [_P]),
    'tok_stream_wait_event': (c_int, [_P, _P]),
    'tok_bn_act_fwd_colsum': (c_int, [_P, _P, _P, _P, c_int, _P, _P, c_int64, c_int, _P, _P]),
    'tok_bn_bwd_rows': (c_int, [c_int64, c_int]),
    'tok_bn_bwd_reduce': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int64, c_int, _P, _P]),
    'tok_bn_bwd_finalize': (c_int, [_P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    'tok_bn_bwd_apply': (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int64, c_int, _P]),
    'tok_maxpool3x3s2_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'tok_maxpool3x3s2_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'tok_avgpool2x2_fwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    'tok_avgpool2x2_bwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'tok_gap_fwd': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'tok_gap_bwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    'tok_global_pool_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'tok_global_pool_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'tok_colsum': (c_int, [_P, c_int64, c_int, c_int, _P, c_int, _P]),
    'tok_bn_gram_finalize': (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P]),
    'tok_conv_fwd_bn_apply': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    'tok_conv_dgrad_maskstore': (c_int, [POINTER(ConvDesc), _P, _P, _P, c_int, _P, _P, _P]),
    'tok_relu_mask_reduce': (c_int, [_P, _P, c_int64, c_int, _P, _P, _P]),
    'tok_bn3_bwd_prepare': (c_int, [_P, _P, _P, _P, _P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, c_int,
                                    _P, _P, _P, _P, _P]),
    'tok_bn3_bwd_prepare_ws_floats': (c_size_t, [c_int, c_int]),
    'tok_conv_dgrad_bias': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    'tok_conv_dgrad2_ok': (c_int, [POINTER(ConvDesc), POINTER(ConvDesc)]),
    'tok_conv_dgrad2': (c_int, [POINTER(ConvDesc), _P, _P, POINTER(ConvDesc), _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    'tok_subsample2_fwd': (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    'tok_subsample2_bwd': (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    'tok_conv_dgrad_subacc_ok': (c_int, [POINTER(ConvDesc)]),
    'tok_conv_dgrad_subacc': (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    'tok_softmax_ce_fwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int64, _P, _P, _P, _P]),
    'tok_softmax_ce_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int64, _P, _P]),
    'tok_softmax_ce_smooth_fwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int64, c_float, _P, _P, _P, _P]),
    'tok_softmax_ce_smooth_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int64, c_float, _P, _P]),
    'tok_upsample_ce_serves': (c_int, [c_int, c_int]),
    'tok_upsample_ce_fwd': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int64, _P, _P, _P, _P]),
    'tok_upsample_ce_bwd': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int64, _P, _P, _P, _P, c_int, _P]),
    'tok_dice_rows': (c_int, [c_int64]),
    'tok_dice_fwd': (c_int, [_P, _P, c_int64, c_int, c_int, c_int, c_float, c_float, c_int, _P, c_int, _P, _P, _P, _P]),
    'tok_dice_bwd': (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P, _P]),
    'tok_bce_logits_fwd': (c_int, [_P, _P, c_int64, c_int, c_int, c_float, c_int, _P, _P]),
    'tok_bce_logits_bwd': (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_float, c_int, _P, _P]),
    'tok_relevance_matrix_multilabel': (c_int, [_P, _P, c_int, c_int, c_int, _P, _P]),
    'tok_embed_reg_fwd': (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'tok_embed_reg_bwd': (c_int, [_P, _P, _P, c_float, c_int, c_int, c_int, c_int, _P, _P]),
    'tok_regression_loss_fwd': (c_int, [_P, _P, c_int64, c_int, c_float, c_int, _P, _P]),
    'tok_regression_loss_bwd': (c_int, [_P, _P, _P, c_int64, c_int, c_float, c_int, _P, _P]),
    'tok_confusion_update': (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, _P, _P]),
    'tok_cls_stats_update': (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, _P, _P]),
    'tok_l2norm_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    'tok_l2norm_bwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'tok_arcface_margin_fwd': (c_int, [_P, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, c_float, _P, _P]),
    'tok_arcface_margin_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, c_float, _P, _P]),
    'tok_relevance_matrix': (c_int, [_P, _P, c_int, c_int, _P, _P]),
    'tok_contrastive_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    'tok_contrastive_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, c_int, _P]),
    'tok_fuse_sum_relu_fwd': (c_int, [_P, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'tok_fuse_sum_affine_relu_fwd': (c_int, [_P, c_int, _P, _P, _P, c_int, _P, _P, _P, c_int, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'tok_fuse_sum_relu_bwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    'tok_bilinear_fwd': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    'tok_bilinear_bwd': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'tok_bilinear_bwd_multi': (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, _P]),
    'tok_bilinear_sum_stats_rows': (c_int, [c_int, c_int, c_int, c_int]),
    'tok_bilinear_sum_stats': (c_int, [_P, _P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    'tok_layernorm_fwd': (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_float, _P]),
    'tok_layernorm_bwd_rows': (c_int, [c_int64, c_int]),
    'tok_layernorm_bwd': (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, c_int, _P, c_int64, c_int, c_int, _P]),
    'tok_colsum_partial_rows': (c_int, [c_int64, c_int]),
    'tok_colsum_partial': (c_int, [_P, c_int64, c_int, _P, _P]),
    'tok_colsum_f32': (c_int, [_P, c_int64, c_int, _P, c_int, _P]),
    'tok_colsum_f32_pair': (c_int, [_P, _P, c_int64, c_int, _P, c_int, _P, c_int, _P]),
    'tok_act_fwd': (c_int, [c_int, _P, _P, c_size_t, _P]),
    'tok_act_bwd': (c_int, [c_int, _P, _P, _P, c_int, c_size_t, _P]),
    'tok_window_attn_fwd': (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'tok_window_attn_bwd_rows': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'tok_window_attn_bwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P,
                                    _P, _P, _P]),
    'tok_cpb_bias_fwd': (c_int, [_P, c_int, _P, c_int, c_int, _P, _P]),
    'tok_cpb_bias_bwd': (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, _P, _P]),
    'tok_patch_merge': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'tok_ntxent_fwd': (c_int, [_P, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    'tok_ntxent_bwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, _P, _P]),
    'tok_triplet_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_int, _P, _P, _P, _P]),
    'tok_triplet_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_int, _P, _P, _P, _P]),
    'tok_bn_relu_maxpool_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    'tok_bn_pool_bwd_reduce_pooled': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P]),
    'tok_bn_pool_bwd_reduce': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    'tok_bn_pool_bwd_apply': (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    'tok_conv_fwd_act': (c_int, [_PD, _P, _P, _P, _P, _P, c_int, _P]),
    'tok_conv_dgrad_act': (c_int, [_PD, _P, _P, _P, c_int, _P, _P]),
    'tok_mlp_serves': (c_int, [c_int64, c_int, c_int]),
    'tok_mlp_fwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    'tok_mlp_bwd_dx': (c_int, [_P, _P, _P, _P, _P, c_int, _P, c_int64, c_int, c_int, _P]),
    'tok_chan_gram': (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P, _P, _P]),
    'tok_chan_apply': (c_int, [_P, c_int, _P, c_int, c_float, c_int, c_int, c_int, _P, c_int, _P]),
    'tok_scale_rows_add': (c_int, [_P, _P, _P, c_int, _P, c_int, c_int64, c_int, _P]),
    'tok_pix_class_matmul': (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P]),
    'tok_class_pix_expand': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_int, c_int, _P]),
    'tok_weighted_pool_chunks': (c_int, [c_int]),
    'tok_weighted_pool': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P, _P, c_int, c_int, _P]),
    'tok_softmax_rows_f32': (c_int, [_P, c_int64, c_int, _P, _P]),
    'tok_softmax_rows_bwd_f32': (c_int, [_P, _P, c_int64, c_int, _P, _P]),
    'tok_softmax_cols_fwd': (c_int, [_P, c_int, c_int, c_int, c_int, c_float, _P, _P]),
    'tok_softmax_cols_bwd': (c_int, [_P, _P, c_int, c_int, c_int, c_float, _P, c_int, c_int, _P]),
    'tok_channel_scale': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'tok_dwconv3x3': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'tok_dwconv3x3_wgrad_blocks': (c_int, [c_int, c_int]),
    'tok_dwconv3x3_wgrad': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P]),
    'tok_sim_matrix': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int64, _P]),
    'tok_topk_rows': (c_int, [_P, c_int, c_int, c_int64, c_int, _P, _P, _P]),
    'tok_retrieval_nrel': (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, _P, _P]),
    'tok_retrieval_eval': (c_int, [c_int, _P, c_int, _P, _P, c_int, _P, _P, c_int, _P, _P, _P, _P, c_int, _P, _P]),
    'tok_rmsprop_step': (c_int, [_P, _P, _P, _P, _P, c_size_t, c_float, c_float, c_float, c_float, c_float, c_int, c_int, _P]),
    'tok_sgd_step': (c_int, [_P, _P, _P, _P, c_size_t, c_float, c_float, c_float, c_float,
                             c_int, c_int, c_int, _P]),
    'tok_adam_step': (c_int, [_P, _P, _P, _P, _P, c_size_t, c_float, c_float, c_float, c_float,
                              c_float, c_int, c_int64, c_int, _P]),
    'tok_adam_step_capturable': (c_int, [_P, _P, _P, _P, _P, c_size_t, c_float, c_float, c_float, c_float, c_float, c_int, _P,
                                         c_int, _P]),
    'tok_step_advance': (c_int, [_P, _P]),
    'tok_fill_f32': (c_int, [_P, c_float, c_size_t, _P]),
    'tok_scale_f32': (c_int, [_P, c_float, c_size_t, _P]),
}


class TokError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen the C-ABI library and attach prototypes.  Raises if it is not built."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise TokError(f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                       f'(hipcc --offload-arch=gfx950). torchok_amd has no CPU/eager fallback.')
    lib = ctypes.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


def lib():
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib


def check(status, what=''):
    if status != 0:
        msg = lib().tok_last_error()
        if isinstance(msg, bytes):
            msg = msg.decode()
        raise TokError(f'{what} failed ({status}): {msg}')
