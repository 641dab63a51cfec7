"""ctypes binding of libvcvits_hip.so (the C ABI declared in include/vcvits_hip.h).

The library is the product path: there is NO CPU fallback.  `lib()` raises if the shared object
is missing; every launcher raises RuntimeError on a non-zero status.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
from . import tuning  # noqa: E402

LIB_PATH = tuning.text("VCVITS_HIP_LIB", "", "path of another build of libvcvits_hip.so (A/B builds)") or \
    os.path.join(_HERE, "csrc", "libvcvits_hip.so")

VCV_OK = 0
ACT_NONE, ACT_LEAKY, ACT_RELU, ACT_TANH, ACT_LOGCLAMP = 0, 1, 2, 3, 4
TF_NONE, TF_LEAKY, TF_DLEAKY, TF_DRELU, TF_DTANH, TF_DLOGCLAMP = 0, 1, 2, 3, 4, 5

_f32p = ctypes.c_void_p
_i32 = ctypes.c_int32


class VcvConvArgs(ctypes.Structure):
    _fields_ = [
        ("x", _f32p), ("w", _f32p), ("bias", _f32p), ("res", _f32p), ("mask", _f32p),
        ("xaux", _f32p), ("oaux", _f32p), ("y", _f32p),
        ("B", _i32), ("G", _i32), ("Cg", _i32), ("Mg", _i32),
        ("Tin", _i32), ("Tout", _i32), ("P", _i32),
        ("K", _i32),
        ("s", _i32), ("dj", _i32), ("off", _i32),
        ("os", _i32), ("oo", _i32),
        ("phases", _i32),
        ("Q", _i32),
        ("a_mode", _i32), ("in_tf", _i32), ("out_act", _i32), ("out_tf", _i32), ("accumulate", _i32),
        ("alpha", ctypes.c_float), ("slope", ctypes.c_float),
        ("io", _i32), ("ms", _i32), ("post_scale", ctypes.c_float),
    ]


class VcvPackJob(ctypes.Structure):
    """mirrors include/vcvits_hip.h"""
    _fields_ = [("w", ctypes.c_void_p), ("wp", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("kind", "M", "C", "K", "BM", "BKC", "JA", "nch", "nmt", "phases", "mode", "reserved")] + \
               [("total", ctypes.c_int64), ("block0", ctypes.c_int64)]


class VcvResPairArgs(ctypes.Structure):
    """mirrors include/vcvits_hip.h"""
    _fields_ = [("x", _f32p), ("wp", _f32p), ("b1", _f32p), ("b2", _f32p), ("y", _f32p)] + \
               [(n, _i32) for n in ("B", "C", "T", "K", "dil", "accumulate")] + \
               [("post_scale", ctypes.c_float), ("slope", ctypes.c_float)]


class VcvWgradArgs(ctypes.Structure):
    _fields_ = [
        ("a", _f32p), ("b", _f32p), ("aaux", _f32p), ("baux", _f32p), ("dw", _f32p),
        ("B", _i32), ("G", _i32), ("Cg", _i32), ("Mg", _i32),
        ("Ta", _i32), ("Tb", _i32), ("P", _i32),
        ("K", _i32),
        ("s", _i32), ("dj", _i32), ("off", _i32),
        ("a_tf", _i32), ("b_tf", _i32),
        ("transpose_out", _i32),
        ("alpha", ctypes.c_float), ("slope", ctypes.c_float),
        ("dbias", _f32p),
        ("slab", _f32p), ("slab_floats", ctypes.c_int64),
    ]


_lib = None


def lib():
    """Load the HIP library once; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libvcvits_hip.so not found at %s -- run `python -m vcvits_amd.build_ext` "
                "(there is no CPU fallback for the vcvits hot path)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.vcv_version.restype = ctypes.c_char_p
        for name in EXPORTS:
            fn = getattr(L, name)
            if name != "vcv_version":
                fn.restype = (ctypes.c_int64 if name in ("vcv_conv_dma_workspace", "vcv_wgrad_bf16_scratch", "vcv_wgrad_x3_scratch", "vcv_layernorm_c_bwd_scratch", "vcv_resblock_pair_supported")
                              else ctypes.c_void_p if name == "vcv_get_seed_offset_ptr" else ctypes.c_int)
                fn.argtypes = _ARGTYPES[name]
        _lib = L
    return _lib


# every symbol include/vcvits_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "vcv_version", "vcv_conv_gemm", "vcv_conv_wgrad", "vcv_conv_wgrad_takes_dma", "vcv_bias_grad",
    "vcv_weight_norm_fwd", "vcv_weight_norm_bwd", "vcv_spectral_norm_fwd", "vcv_spectral_norm_bwd", "vcv_avg3", "vcv_scale", "vcv_mask_mul",
    "vcv_reflect_pad_fwd", "vcv_reflect_pad_bwd", "vcv_avgpool4_fwd", "vcv_avgpool4_bwd",
    "vcv_loss_sum", "vcv_loss_grad", "vcv_adamw", "vcv_stft_mag_fwd", "vcv_stft_mag_bwd",
    "vcv_wn_gate_fwd", "vcv_wn_gate_bwd", "vcv_row_sum", "vcv_wn_res_skip_fwd", "vcv_wn_res_skip_bwd",
    "vcv_split_sample_fwd", "vcv_split_sample_bwd", "vcv_coupling", "vcv_layernorm_c_fwd",
    "vcv_layernorm_c_bwd", "vcv_rel_softmax_fwd", "vcv_rel_value_fwd", "vcv_rel_softmax_bwd",
    "vcv_kl_fwd", "vcv_kl_bwd", "vcv_nearest_fwd", "vcv_nearest_bwd", "vcv_nearest_raw_fwd", "vcv_nearest_raw_bwd", "vcv_slice_fwd", "vcv_slice_bwd",
    "vcv_dropout", "vcv_prof_begin", "vcv_prof_end", "vcv_prof_dump", "vcv_conv_m1_fwd", "vcv_conv_c1_fwd", "vcv_conv_c1_fwd_masked", "vcv_conv_c1_fwd_flip", "vcv_conv_c1_dgrad", "vcv_linear_t1_fwd", "vcv_linear_t1_dgrad", "vcv_linear_t1_wgrad", "vcv_thin_wgrad", "vcv_weight_flip_transpose",
    "vcv_act_grad", "vcv_act_grad_add", "vcv_weight_norm_many_fwd", "vcv_weight_norm_many_bwd", "vcv_loss_many_sum", "vcv_loss_many_grad", "vcv_conv_dma_workspace", "vcv_conv_dma", "vcv_conv_dma_plan", "vcv_conv_dma_run", "vcv_stft_complex_fwd", "vcv_istft", "vcv_grouped41_fwd", "vcv_grouped41_fwd_bf16", "vcv_grouped41_dgrad", "vcv_grouped41_dgrad_bf16", "vcv_grouped41_wgrad", "vcv_grouped41_wgrad_bf16",
    "vcv_prior_sample", "vcv_prof_bytes", "vcv_conv_bf16_plan", "vcv_conv_bf16_run", "vcv_wgrad_bf16_scratch", "vcv_wgrad_bf16", "vcv_act_grad_bias", "vcv_conv_pk_plan", "vcv_conv_pk_run",
    "vcv_conv_x3_plan", "vcv_conv_x3_run", "vcv_conv_x3_set_terms", "vcv_conv_x3_get_terms", "vcv_conv_x3_set_all", "vcv_wgrad_x3_scratch", "vcv_wgrad_x3", "vcv_rel_attn_supported", "vcv_rel_attn_fwd", "vcv_rel_attn_bwd", "vcv_rel_attn_bwd2", "vcv_set_deterministic", "vcv_get_deterministic", "vcv_prof_roof", "vcv_prof_pause", "vcv_conv_x3_pack_job", "vcv_conv_pk_pack_job", "vcv_conv_bf16_pack_job", "vcv_pack_many", "vcv_upload_table",
    "vcv_conv_bf16io_plan", "vcv_conv_bf16io_run", "vcv_cast_f32_x16", "vcv_cast_x16_f32", "vcv_conv_m1_x16_fwd",
    "vcv_prof_active", "vcv_set_seed_offset_ptr", "vcv_get_seed_offset_ptr", "vcv_pack_many_prepared", "vcv_adamw_dev", "vcv_set_words", "vcv_tuning_set", "vcv_tuning_get", "vcv_embedding_t_fwd", "vcv_embedding_t_fwd_checked", "vcv_embedding_t_bwd", "vcv_conv_x3_set_variant", "vcv_wgrad_bf16_set_force", "vcv_layernorm_c_bwd_scratch", "vcv_layernorm_c_bwd_ws", "vcv_resblock_pair_supported", "vcv_resblock_pair_pack", "vcv_resblock_pair_x16",
]


_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
_ARGTYPES = {
    "vcv_conv_gemm": [ctypes.POINTER(VcvConvArgs), _P],
    "vcv_conv_wgrad": [ctypes.POINTER(VcvWgradArgs), _P],
    "vcv_conv_wgrad_takes_dma": [ctypes.POINTER(VcvWgradArgs)],
    "vcv_bias_grad": [_P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "vcv_weight_norm_fwd": [_P, _P, _P, _P, _I, _I, _P],
    "vcv_weight_norm_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vcv_spectral_norm_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    "vcv_spectral_norm_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "vcv_avg3": [_P, _P, _P, _P, _L, _P],
    "vcv_scale": [_P, _P, _F, _L, _P],
    "vcv_mask_mul": [_P, _P, _P, _I, _I, _I, _P],
    "vcv_reflect_pad_fwd": [_P, _P, _I, _I, _I, _P],
    "vcv_reflect_pad_bwd": [_P, _P, _I, _I, _I, _P],
    "vcv_avgpool4_fwd": [_P, _P, _I, _I, _P],
    "vcv_avgpool4_bwd": [_P, _P, _I, _I, _P],
    "vcv_loss_sum": [_P, _P, _F, _I, _F, _P, _L, _P],
    "vcv_loss_grad": [_P, _P, _F, _I, _F, _P, _P, _I, _L, _P],
    "vcv_adamw": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P],
    "vcv_stft_mag_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_stft_mag_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_wn_gate_fwd": [_P, _P, _I, _I, _P, _I, _I, _I, _P],
    "vcv_wn_gate_bwd": [_P, _P, _I, _I, _P, _P, _I, _I, _I, _P],
    "vcv_row_sum": [_P, _P, _I, _I, _I, _I, _I, _P],
    "vcv_wn_res_skip_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vcv_wn_res_skip_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vcv_split_sample_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vcv_split_sample_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vcv_coupling": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "vcv_layernorm_c_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    "vcv_layernorm_c_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vcv_layernorm_c_bwd_scratch": [_I, _I, _I],
    "vcv_resblock_pair_supported": [_I, _I, _I, _I],
    "vcv_resblock_pair_pack": [_P, _P, _P, _I, _I, _P],
    "vcv_resblock_pair_x16": [ctypes.POINTER(VcvResPairArgs), _P],
    "vcv_layernorm_c_bwd_ws": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _L, _P],
    "vcv_rel_softmax_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, ctypes.c_uint64, _P],
    "vcv_dropout": [_P, _P, _L, _F, ctypes.c_uint64, _P],
    "vcv_rel_value_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vcv_rel_softmax_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "vcv_rel_attn_supported": [_I, _I, _I, _I, _I],
    "vcv_rel_attn_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, ctypes.c_uint64, _I, _P],
    "vcv_rel_attn_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, ctypes.c_uint64, _I, _P],
    "vcv_rel_attn_bwd2": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, ctypes.c_uint64, _I, _P],
    "vcv_set_deterministic": [_I],
    "vcv_get_deterministic": [],
    "vcv_kl_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vcv_kl_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "vcv_nearest_fwd": [_P, _P, _I, _I, _I, _P],
    "vcv_nearest_bwd": [_P, _P, _I, _I, _I, _P],
    "vcv_nearest_raw_fwd": [_P, _P, _I, _I, _I, _P, _P],
    "vcv_nearest_raw_bwd": [_P, _P, _I, _I, _I, _P, _P],
    "vcv_slice_fwd": [_P, _P, _I, _P, _I, _I, _I, _I, _P],
    "vcv_slice_bwd": [_P, _P, _I, _P, _I, _I, _I, _I, _P],
    "vcv_conv_m1_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_conv_c1_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_conv_c1_fwd_masked": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_conv_c1_fwd_flip": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "vcv_linear_t1_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "vcv_linear_t1_dgrad": [_P, _P, _P, _I, _I, _I, _P],
    "vcv_linear_t1_wgrad": [_P, _P, _P, _I, _I, _I, _P],
    "vcv_conv_c1_dgrad": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "vcv_thin_wgrad": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P],
    "vcv_weight_flip_transpose": [_P, _P, _I, _I, _I, _P],
    "vcv_grouped41_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_grouped41_fwd_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_grouped41_dgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_grouped41_dgrad_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_grouped41_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_grouped41_wgrad_bf16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "vcv_act_grad": [_P, _P, _P, _I, _F, _L, _P],
    "vcv_act_grad_add": [_P, _P, _P, _P, _I, _F, _L, _P],
    "vcv_act_grad_bias": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "vcv_stft_complex_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "vcv_istft": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vcv_loss_many_sum": [_P, _I, _I, _F, _I, _P, _P],
    "vcv_loss_many_grad": [_P, _I, _I, _F, _I, _P, _P, _P],
    "vcv_weight_norm_many_fwd": [_P, _I, _I, _P, _P, _P],
    "vcv_weight_norm_many_bwd": [_P, _I, _I, _P, _P],
    "vcv_conv_dma_workspace": [ctypes.POINTER(VcvConvArgs)],
    "vcv_conv_dma": [ctypes.POINTER(VcvConvArgs), _P, _I, _P],
    "vcv_conv_dma_plan": [ctypes.POINTER(VcvConvArgs), _I, ctypes.POINTER(ctypes.c_int64)],
    "vcv_conv_dma_run": [ctypes.POINTER(VcvConvArgs), _P, _P, _I, _I, _P],
    "vcv_prior_sample": [_P, _P, _P, _P, _L, _F, _P],
    "vcv_conv_bf16_plan": [ctypes.POINTER(VcvConvArgs), _I, ctypes.POINTER(ctypes.c_int64)],
    "vcv_conv_bf16_run": [ctypes.POINTER(VcvConvArgs), _P, _P, _I, _I, _P],
    "vcv_conv_pk_plan": [ctypes.POINTER(VcvConvArgs), _I, ctypes.POINTER(ctypes.c_int64)],
    "vcv_conv_pk_run": [ctypes.POINTER(VcvConvArgs), _P, _P, _I, _I, _P],
    "vcv_conv_x3_plan": [ctypes.POINTER(VcvConvArgs), _I, ctypes.POINTER(ctypes.c_int64)],
    "vcv_conv_x3_run": [ctypes.POINTER(VcvConvArgs), _P, _P, _I, _I, _P],
    "vcv_conv_x3_set_terms": [_I],
    "vcv_conv_x3_get_terms": [],
    "vcv_conv_x3_set_all": [_I],
    "vcv_conv_x3_set_variant": [_I, _I, _I],
    "vcv_wgrad_bf16_set_force": [_I, _I],
    "vcv_wgrad_bf16_scratch": [ctypes.POINTER(VcvWgradArgs)],
    "vcv_wgrad_x3_scratch": [ctypes.POINTER(VcvWgradArgs)],
    "vcv_wgrad_x3": [ctypes.POINTER(VcvWgradArgs), _P, _L, _P],
    "vcv_wgrad_bf16": [ctypes.POINTER(VcvWgradArgs), _P, _L, _P],
    "vcv_prof_begin": [_I],
    "vcv_prof_end": [ctypes.POINTER(ctypes.c_double), _I],
    "vcv_prof_dump": [ctypes.c_char_p],
    "vcv_prof_bytes": [ctypes.POINTER(ctypes.c_double), _I],
    "vcv_prof_roof": [ctypes.POINTER(ctypes.c_double), _I],
    "vcv_prof_pause": [_I],
    "vcv_conv_x3_pack_job": [ctypes.POINTER(VcvConvArgs), _I, ctypes.POINTER(VcvPackJob)],
    "vcv_conv_pk_pack_job": [ctypes.POINTER(VcvConvArgs), _I, ctypes.POINTER(VcvPackJob)],
    "vcv_conv_bf16_pack_job": [ctypes.POINTER(VcvConvArgs), _I, ctypes.POINTER(VcvPackJob)],
    "vcv_pack_many": [ctypes.POINTER(VcvPackJob), _I, _P, _P],
    "vcv_upload_table": [_P, _P, _L, _P],
    "vcv_pack_many_prepared": [ctypes.POINTER(VcvPackJob), _I, _P, _P],
    "vcv_tuning_set": [ctypes.c_char_p, _I],
    "vcv_tuning_get": [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)],
    "vcv_embedding_t_fwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "vcv_embedding_t_fwd_checked": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "vcv_embedding_t_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "vcv_set_words": [_P, _I, _I, _I, _I, _I, _P],
    "vcv_adamw_dev": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _P, _I, _P],
    "vcv_conv_bf16io_plan": [ctypes.POINTER(VcvConvArgs), _I, ctypes.POINTER(ctypes.c_int64)],
    "vcv_conv_bf16io_run": [ctypes.POINTER(VcvConvArgs), _P, _P, _I, _I, _P],
    "vcv_prof_active": [],
    "vcv_set_seed_offset_ptr": [_P],
    "vcv_get_seed_offset_ptr": [],
    "vcv_cast_f32_x16": [_P, _P, _L, _I, _P],
    "vcv_cast_x16_f32": [_P, _P, _L, _I, _P],
    "vcv_conv_m1_x16_fwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
}


class Capture:
    """State of ONE launch sequence being recorded into a HIP graph (light/graphed.py sets CAPTURE[0] for its duration).

    What a recorded launch bakes is an ADDRESS.  The sequence is valid on replay only while everything behind those
    addresses is still there, so this object makes that true by construction instead of by convention:
      * `keep`     -- objects that must live as long as the graph (device tables, host arrays);
      * `tables`   -- device tables whose contents are known while recording and never change between replays (they hold
                      addresses of the graph's own tensors).  They are NOT uploaded by a recorded copy node -- a node
                      that re-reads pageable host memory at every replay -- but once, eagerly, right after the capture
                      (`flush`).  That only works for memory NO recorded launch ever writes: a block of the graph's pool
                      is re-written at every replay by whichever earlier tensor of the sequence lived in it before the
                      table did (the first version of this scheme took tables from the pool and faulted on its first
                      replay for exactly that reason).  So tables are cut from `table_arena`, allocated before the capture
                      begins, outside the pool; when it is full the caller falls back to a recorded copy node;
      * `external` -- every tensor handed to a launcher (`ptr`) whose storage was allocated BEFORE the capture began, i.e.
                      outside the graph's private pool (parameters, optimizer buffers, cached constant tables, ...).  The
                      graph holds a reference to each, so none of them can be freed -- and its address recycled -- under
                      the graph, whatever the eager code does with its caches afterwards.
    VCVITS_CHECK_PTRS=1 prints every distinct external tensor with the call site that handed it over (the audit the
    round-4 memory fault called for: an address baked into a captured launch that no longer exists)."""
    _next_id = [1]
    LOG = tuning.flag("VCVITS_CHECK_PTRS", False, "graph capture: list every external tensor a recorded launch bakes, audit the device tables")

    TABLE_BYTES = int(tuning.number("VCVITS_GRAPH_TABLE_MB", 16.0, "device-table arena of a recorded launch sequence, MB") * (1 << 20))

    def __init__(self, device=None):
        self.id = Capture._next_id[0]
        Capture._next_id[0] += 1
        self.keep, self.pending, self.external = [], [], {}
        self.table_arena, self.table_off = None, 0
        if device is not None and torch.cuda.is_available():
            self.table_arena = torch.zeros(self.TABLE_BYTES, device=device, dtype=torch.uint8)
        self.log = Capture.LOG
        self._starts, self._ends = [], []
        if device is not None and torch.cuda.is_available():
            segs = sorted((s["address"], s["address"] + s["total_size"]) for s in torch.cuda.memory_snapshot()
                          if s.get("device", 0) == (device.index if device.index is not None else torch.cuda.current_device()))
            self._starts = [a for a, _ in segs]
            self._ends = [b for _, b in segs]

    # (list protocol of the round-4 `keep` list: ops appends host arrays / tables to the capture)
    def append(self, obj):
        self.keep.append(obj)

    def extend(self, objs):
        self.keep.extend(objs)

    def is_external(self, addr):
        import bisect
        i = bisect.bisect_right(self._starts, addr) - 1
        return i >= 0 and addr < self._ends[i]

    def note(self, t):
        a = t.data_ptr()
        if a in self.external or not self.is_external(a):
            return
        self.external[a] = t
        if self.log:
            import sys
            import traceback
            fr = [f for f in traceback.extract_stack(limit=12)[:-2] if "torch/" not in f.filename][-4:]
            sys.stderr.write("vcvits_amd[capture %d]: external tensor %s %s @0x%x  <- %s\n" % (
                self.id, tuple(t.shape), str(t.dtype).replace("torch.", ""), a,
                " <- ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(fr))))

    def table(self, host_bytes, dtype, shape):
        """A device tensor (`dtype`, `shape`) cut from the table arena that will hold `host_bytes` (a contiguous numpy array
        or a ctypes array) whenever the graph runs -- or None when the arena is full (caller: recorded copy node)."""
        import numpy as np
        n = host_bytes.nbytes if isinstance(host_bytes, np.ndarray) else ctypes.sizeof(host_bytes)
        need = 1
        for d in shape:
            need *= int(d)
        need *= torch.empty((), dtype=dtype).element_size()
        if self.table_arena is None or need < n or self.table_off + need > self.table_arena.numel():
            return None
        view = self.table_arena[self.table_off:self.table_off + need].view(dtype).view(tuple(shape))
        self.pending.append((view, host_bytes))
        self.table_off = (self.table_off + need + 255) & ~255
        return view

    def check_tables(self):
        """VCVITS_CHECK_PTRS: every 8-byte word of the pending device tables that looks like a device address must lie
        inside a live allocator segment (the graph's pool included)."""
        import bisect
        import sys
        import numpy as np
        segs = sorted((s["address"], s["address"] + s["total_size"]) for s in torch.cuda.memory_snapshot())
        starts = [a for a, _ in segs]
        lo, hi = (segs[0][0], segs[-1][1]) if segs else (0, 0)
        bad = 0
        for ti, (dev_tensor, host) in enumerate(self.pending):
            raw = np.frombuffer(host, dtype=np.uint8) if not isinstance(host, np.ndarray) else host.reshape(-1).view(np.uint8)
            words = raw[:raw.size // 8 * 8].view(np.int64)
            for wi, w in enumerate(words.tolist()):
                if w < (1 << 40) or w >= (1 << 48):
                    continue  # (not an address: a count, an offset, packed ints)
                i = bisect.bisect_right(starts, w) - 1
                if not (i >= 0 and w < segs[i][1]):
                    bad += 1
                    sys.stderr.write("vcvits_amd[capture %d]: table %d (%s, %d bytes) word %d = 0x%x is outside every live "
                                     "segment [0x%x, 0x%x)\n" % (self.id, ti, tuple(dev_tensor.shape), raw.size, wi, w, lo, hi))
        sys.stderr.write("vcvits_amd[capture %d]: %d device tables checked, %d stray addresses\n" % (self.id, len(self.pending), bad))

    def flush(self):
        import numpy as np
        if self.log:
            self.check_tables()
        for dev_tensor, host in self.pending:
            src = np.frombuffer(host, dtype=np.uint8) if not isinstance(host, np.ndarray) else host.reshape(-1).view(np.uint8)
            dst = dev_tensor.reshape(-1).view(torch.uint8)
            dst[:src.size].copy_(torch.from_numpy(src.copy()))
        self.pending = []


# the capture in progress on this process's launch thread (None: eager)
CAPTURE = [None]


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be fp32/int, contiguous, on GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("vcvits_amd: tensor is not on the GPU; the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("vcvits_amd: non-contiguous tensor passed to a HIP launcher")
    if CAPTURE[0] is not None:
        CAPTURE[0].note(t)
    return ctypes.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current HIP stream of the current device as a void*.  torch.cuda.current_stream() builds a Stream object
    through four python layers (~10 us; 900 launches per step made it 8 ms of a host-bound 80 ms step); the two C
    entry points behind it return the same handle in well under a microsecond."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_GET_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# launcher calls that went through check() since import (bench.py reports the per-step count: one call = one kernel launch,
# two for the launchers that add a finishing / fill pass)
CALLS = [0]


def check(status, what):
    CALLS[0] += 1
    if status != VCV_OK:
        raise RuntimeError("vcvits_hip: %s failed with status %d" % (what, status))
