"""ctypes signatures of the element-wise / stencil entry points of include/gigagan_amd.h."""
import ctypes as C

_P = C.c_void_p
_I = C.c_int32


def declare(L):
    L.gg_resample_nhwc_bf16.restype = C.c_int
    L.gg_resample_nhwc_bf16.argtypes = [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]
    _F = C.c_float
    L.gg_adamw_flat_f32.restype = C.c_int
    L.gg_adamw_flat_f32.argtypes = [_P, _P, _P, _P, _P, C.c_int64, _F, _F, _F, _F, _F, _F, _F, _F, _P]
    L.gg_ema_flat_f32.restype = C.c_int
    L.gg_ema_flat_f32.argtypes = [_P, _P, C.c_int64, _F, _P]
    L.gg_pack_weights.restype = C.c_int
    L.gg_pack_weights.argtypes = [_P, _P, _I, _P]
    L.gg_wgrad_finish.restype = C.c_int
    L.gg_wgrad_finish.argtypes = [_P, _P, _I, _I, _I, _I, _I, _F, _I, _P]
    L.gg_wgrad_finish_splits.restype = C.c_int
    L.gg_wgrad_finish_splits.argtypes = [_P, _P, _I, _I, _I, _I, _I, _F, _I, _I, C.c_int64, _P]
    L.gg_colsum_finish.restype = C.c_int
    L.gg_colsum_finish.argtypes = [_P, _P, _I, _I, _I, _F, _P]
    L.gg_modcoef_fwd.restype = C.c_int
    L.gg_modcoef_fwd.argtypes = [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]
    L.gg_modcoef_bwd.restype = C.c_int
    L.gg_modcoef_bwd.argtypes = [_P] * 11 + [_I] * 7 + [_F, _P]
    L.gg_softmax_fwd.restype = C.c_int
    L.gg_softmax_fwd.argtypes = [_P, _P, _P, C.c_int64, _I, _I, _I, _F, _P]
    L.gg_softmax_bwd.restype = C.c_int
    L.gg_softmax_bwd.argtypes = [_P, _P, _P, _P, C.c_int64, _I, _I, _I, _F, _P]
    L.gg_bias_act_bwd.restype = C.c_int
    L.gg_bias_act_bwd.argtypes = [_P, _P, _P, _P, C.c_int64, _I, _F, _P]
    L.gg_gelu.restype = C.c_int
    L.gg_gelu.argtypes = [_P, _P, _P, _P, _P, C.c_int64, _I, _P]
    L.gg_bias_act_bwd_partials.restype = C.c_int32
    L.gg_bias_act_bwd_partials.argtypes = [C.c_int64, _I]
    L.gg_softmax_bwd2.restype = C.c_int
    L.gg_softmax_bwd2.argtypes = [_P, _P, _P, _P, _P, _P, C.c_int64, _I, _I, _I, _F, _P]
    L.gg_modulate_fwd.restype = C.c_int
    L.gg_modulate_fwd.argtypes = [_P, _P, _P, _I, _I, _I, _P]
    L.gg_modulate_bwd.restype = C.c_int
    L.gg_modulate_bwd.argtypes = [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]
    L.gg_modmix_fwd.restype = C.c_int
    L.gg_modmix_fwd.argtypes = [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P]
    L.gg_modmix_bwd.restype = C.c_int
    L.gg_modmix_bwd.argtypes = [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]
    L.gg_attn_fwd.restype = C.c_int
    L.gg_attn_fwd.argtypes = [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P]
    L.gg_attn_bwd.restype = C.c_int
    L.gg_attn_bwd.argtypes = [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P]
    L.gg_rmsnorm_blocks.restype = C.c_int32
    L.gg_rmsnorm_blocks.argtypes = [C.c_int64]
    L.gg_rmsnorm_fwd.restype = C.c_int
    L.gg_rmsnorm_fwd.argtypes = [_P, _P, _P, C.c_int64, _I, _F, _P]
    L.gg_rmsnorm_bwd.restype = C.c_int
    L.gg_rmsnorm_bwd.argtypes = [_P, _P, _P, _P, _P, C.c_int64, _I, _F, _P]
    L.gg_rmsnorm_bwd2.restype = C.c_int
    L.gg_rmsnorm_bwd2.argtypes = [_P, _P, _P, _P, _P, _P, _P, C.c_int64, _I, _F, _P]
    L.gg_attn_bwd2.restype = C.c_int
    L.gg_attn_bwd2.argtypes = [_P] * 20 + [_I, _I, _I, _F, _F, _P]
