"""16-bit activations in HBM (inference decoder): casts, the bf16-io convolutions, the fused ResBlock pair, the 1-channel
output conv.

Part of `vcvits_amd.ops` (the package re-exports every name: `from vcvits_amd import ops; ops.conv1d(...)`).  Everything here
runs on the GPU through libvcvits_hip.so; there is no CPU fallback."""
import ctypes

import torch

from .. import tuning
from .._lib import (ACT_NONE, TF_LEAKY, TF_NONE, VcvConvArgs, check, lib, ptr, stream)
from .core import (LAUNCH_COUNTS, _f32c, _rows, conv_out_len)
from .conv import (_common, _launch_conv, convT_out_len)
from .weights import (_stable_entry)


# ---- 16-bit activations in HBM (inference decoder; no autograd) ---------------------------------------------------------
# Storage kinds: torch.bfloat16 (an MFMA operand as it is: the tensor between the two convs of a ResBlock pair, stored
# after the leaky-ReLU its consumer would apply) and torch.float16 (the residual stream: 11 significand bits, so the
# re-rounding at every residual add stays far below the operand rounding; the reference's autocast stores fp16 too).
_KIND = {torch.bfloat16: 1, torch.float16: 2}


def _x16(t, what):
    if t is None:
        return None
    if t.dtype not in _KIND or not t.is_contiguous():
        raise RuntimeError("vcvits_amd: %s must be a contiguous bf16 / fp16 tensor" % what)
    return t


def _io_bits(x, y):
    return 3 | (4 if x.dtype == torch.float16 else 0) | (8 if y.dtype == torch.float16 else 0)


def cast_x16(x, dtype=torch.bfloat16):
    """fp32 -> bf16 / fp16 (round to nearest even; fp16 clamps to its finite range), same shape."""
    x = _f32c(x)
    y = torch.empty(x.shape, device=x.device, dtype=dtype)
    check(lib().vcv_cast_f32_x16(ptr(x), ptr(y), x.numel(), _KIND[dtype], stream()), "vcv_cast_f32_x16")
    return y


def cast_f32(x):
    x = _x16(x, "x")
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib().vcv_cast_x16_f32(ptr(x), ptr(y), x.numel(), _KIND[x.dtype], stream()), "vcv_cast_x16_f32")
    return y


def conv_forward_x16(x, w, bias=None, stride=1, pad=0, dil=1, in_leaky=False, out_act=ACT_NONE, slope=0.1, res=None,
                     out=None, accumulate=False, post_scale=0.0, out_dtype=torch.bfloat16):
    """conv_forward over 16-bit activations: x is bf16 / fp16 [B, C, T]; res / out share one 16-bit dtype (out_dtype when
    `out` is created here); w / bias fp32.  With `out` given and accumulate=True the result is added onto it; post_scale
    multiplies (conv + bias + res) first (0 = none)."""
    x, res, out = _x16(x, "x"), _x16(res, "res"), _x16(out, "out")
    B, C, Tin, P = _rows(x)
    M, Cg, K = w.shape[0], w.shape[1], w.shape[2]
    if Cg != C:
        raise RuntimeError("conv_forward_x16: channel mismatch")
    Tout = conv_out_len(Tin, K, stride, pad, dil)
    if out is None:
        out = torch.empty((B, M, Tout) if x.dim() == 3 else (B, M, Tout, P), device=x.device, dtype=out_dtype)
    if res is not None and res.dtype != out.dtype:
        raise RuntimeError("conv_forward_x16: res and out must share their storage type")
    a = VcvConvArgs()
    a.x, a.w, a.y = ptr(x), ptr(w), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, 1, Cg, M
    a.Tin, a.Tout, a.P, a.K = Tin, Tout, P, K
    a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q, a.a_mode = stride, dil, -pad, 1, 0, 1, Tout, 0
    _common(a, bias=_f32c(bias), res=res, in_tf=TF_LEAKY if in_leaky else TF_NONE, out_act=out_act, slope=slope,
            accumulate=accumulate)
    a.io, a.post_scale = _io_bits(x, out), float(post_scale)
    _launch_conv(a)
    return out


_CONVT_MERGED = [tuning.flag("VCVITS_CONVT_MERGED", True, "16-bit transposed convs: all output phases as rows of one launch")]


def convT_forward_x16(x, w, bias=None, stride=1, pad=0, in_leaky=False, slope=0.1, out_dtype=torch.bfloat16):
    """convT_forward over 16-bit activations (x, result: bf16 / fp16; w [Cin, Cout, K] / bias fp32)."""
    x = _x16(x, "x")
    B, C, Tin, P = _rows(x)
    Cin, M, K = w.shape
    if Cin != C:
        raise RuntimeError("convT_forward_x16: channel mismatch")
    Tout = convT_out_len(Tin, K, stride, pad)
    out = torch.empty((B, M, Tout), device=x.device, dtype=out_dtype)
    a = VcvConvArgs()
    a.x, a.w, a.y = ptr(x), ptr(w), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, 1, C, M
    a.Tin, a.Tout, a.P, a.K = Tin, Tout, P, K
    a.a_mode = 1
    if stride == 1:
        a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q = 1, -1, pad, 1, 0, 1, Tout
    elif K % stride == 0 and _CONVT_MERGED[0]:
        # all `stride` output phases as rows (cout, phase) of ONE launch: one staged input span feeds every phase and the
        # epilogue writes runs of consecutive samples (VcvConvArgs.ms)
        a.Mg, a.K, a.ms = M * stride, K // stride, stride
        a.s, a.dj, a.off, a.os, a.oo, a.phases = 1, -1, 0, stride, -pad, 1
        a.Q = (Tout - 1 + pad) // stride + 1
    else:
        a.s, a.dj, a.off, a.os, a.oo, a.phases = 1, -1, 0, stride, -pad, stride
        a.Q = (Tout - 1 + pad) // stride + 1
    _common(a, bias=_f32c(bias), in_tf=TF_LEAKY if in_leaky else TF_NONE, slope=slope)
    a.io = _io_bits(x, out)
    _launch_conv(a)
    return out


# ---- one conv pair of a ResBlock1 as ONE launch (resblock_pair.hip) ------------------------------------------------------------
_PAIR_FUSED = [tuning.flag("VCVITS_PAIR_FUSED", True, "inference: a ResBlock conv pair as one launch where the fused kernel takes it")]


def resblock_pair_supported(x, w1, w2, dil):
    """True when the fused pair kernel takes (x fp16 [B, C, T], two [C, C, K] convs, c1's dilation)."""
    if not _PAIR_FUSED[0] or x.dtype != torch.float16 or x.dim() != 3 or not x.is_contiguous():
        return False
    C, K = w1.shape[0], w1.shape[2]
    if tuple(w1.shape) != (C, C, K) or tuple(w2.shape) != (C, C, K) or x.shape[1] != C:
        return False
    return lib().vcv_resblock_pair_supported(C, K, int(dil), x.shape[2]) > 0


def resblock_pair_x16(x, w1, b1, w2, b2, dil, slope=0.1, out=None, accumulate=False, post_scale=0.0):
    """out = conv2(leaky(conv1(leaky(x); w1, dil) + b1); w2) + b2 + x over fp16 activations in ONE launch (the intermediate,
    rounded to bf16 exactly as the two-launch path stores it, stays in LDS).  With `out` given and accumulate=True:
    out += post_scale * result (a block's last pair: the stage mean).  modules.ResBlock1.forward_x16 is the caller."""
    from .._lib import VcvResPairArgs
    x = _x16(x, "x")
    B, C, T = x.shape
    K = w1.shape[2]
    w1, w2, b1, b2 = _f32c(w1), _f32c(w2), _f32c(b1), _f32c(b2)
    # the packed weights live with the cached weight-norm buffer / parameter region that holds w1 (as the conv packs do:
    # dropped when those weights change); weights outside any such buffer are packed per call (an address alone can be recycled)
    ent = _stable_entry(w1.data_ptr())
    if ent is not None and "dirty" in ent:
        ent = None  # (a parameter region keys its packs on tensor versions: not worth it for a 45 KB pack)
    key = ("pair", w1.data_ptr(), w2.data_ptr(), K)
    wp = ent["packs"].get(key) if ent is not None else None
    if wp is None:
        nbytes = lib().vcv_resblock_pair_supported(C, K, int(dil), T)
        wp = torch.empty((nbytes // 4,), device=x.device, dtype=torch.float32)
        check(lib().vcv_resblock_pair_pack(ptr(w1), ptr(w2), ptr(wp), C, K, stream()), "vcv_resblock_pair_pack")
        if ent is not None:
            ent["packs"][key] = wp
    if out is None:
        out = torch.empty_like(x)
        accumulate = False
    out = _x16(out, "out")
    a = VcvResPairArgs()
    a.x, a.wp, a.b1, a.b2, a.y = ptr(x), ptr(wp), ptr(b1), ptr(b2), ptr(out)
    a.B, a.C, a.T, a.K, a.dil, a.accumulate = B, C, T, K, int(dil), 1 if accumulate else 0
    a.post_scale, a.slope = float(post_scale), float(slope)
    check(lib().vcv_resblock_pair_x16(ctypes.byref(a), stream()), "vcv_resblock_pair_x16")
    LAUNCH_COUNTS["pair_fused"] = LAUNCH_COUNTS.get("pair_fused", 0) + 1
    return out


def conv_m1_x16(x, w, bias=None, pad=0, in_leaky=False, out_act=ACT_NONE, slope=0.1):
    """One-output-channel conv (stride 1, dilation 1) over a 16-bit [B, C, T] input -> fp32 [B, 1, Tout]."""
    x = _x16(x, "x")
    B, C, Tin = x.shape
    K = w.shape[2]
    Tout = conv_out_len(Tin, K, 1, pad, 1)
    y = torch.empty((B, 1, Tout), device=x.device, dtype=torch.float32)
    check(lib().vcv_conv_m1_x16_fwd(ptr(x), _KIND[x.dtype], ptr(_f32c(w)), ptr(_f32c(bias)), ptr(y), B, C, Tin, Tout, K, 1, pad,
                                    1 if in_leaky else 0, out_act, slope, stream()), "vcv_conv_m1_x16_fwd")
    return y
