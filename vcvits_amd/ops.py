"""Launch wrappers + autograd Functions over the C ABI of libvcvits_hip.so.

Everything here runs on the GPU through hand-written HIP kernels; torch is used for buffer
allocation, the current stream and the autograd graph only.  There is no CPU fallback: calling
any op with CPU tensors raises.
"""
import ctypes

import torch

from . import _lib
from ._lib import (ACT_LEAKY, ACT_LOGCLAMP, ACT_NONE, ACT_RELU, ACT_TANH, TF_DLEAKY, TF_DRELU,
                   TF_LEAKY, TF_NONE, VcvConvArgs, VcvWgradArgs, check, lib, ptr, stream)

TF_DTANH = 4

_ACT_TO_DTF = {ACT_NONE: TF_NONE, ACT_LEAKY: TF_DLEAKY, ACT_RELU: TF_DRELU, ACT_TANH: TF_DTANH}


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise RuntimeError("vcvits_amd: fp32 tensors expected, got %s" % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def _rows(t):
    """[B, C, T] -> (B, C, T, 1);  [B, C, H, P] -> (B, C, H, P)."""
    if t.dim() == 3:
        return t.shape[0], t.shape[1], t.shape[2], 1
    if t.dim() == 4:
        return tuple(t.shape)
    raise RuntimeError("vcvits_amd: expected [B,C,T] or [B,C,H,P], got %s" % (tuple(t.shape),))


def conv_out_len(tin, k, stride, pad, dil):
    return (tin + 2 * pad - dil * (k - 1) - 1) // stride + 1


def _launch_conv(a):
    check(lib().vcv_conv_gemm(ctypes.byref(a), stream()), "vcv_conv_gemm")


def _launch_wgrad(a):
    check(lib().vcv_conv_wgrad(ctypes.byref(a), stream()), "vcv_conv_wgrad")


def _common(a, *, in_tf=TF_NONE, xaux=None, out_act=ACT_NONE, out_tf=TF_NONE, oaux=None, res=None,
            mask=None, bias=None, accumulate=False, alpha=1.0, slope=0.1):
    a.bias, a.res, a.mask = ptr(bias), ptr(res), ptr(mask)
    a.xaux, a.oaux = ptr(xaux), ptr(oaux)
    a.in_tf, a.out_act, a.out_tf = in_tf, out_act, out_tf
    a.accumulate = 1 if accumulate else 0
    a.alpha, a.slope = alpha, slope


# ---------------------------------------------------------------------------------------------
# raw launches (no autograd)
# ---------------------------------------------------------------------------------------------
def conv_forward(x, w, bias=None, stride=1, pad=0, dil=1, groups=1, out=None, **kw):
    """F.conv1d / period F.conv2d((k,1)) forward.  x: [B,C,T] or [B,C,H,P]; w: [M, C/groups, K]
    (a Conv2d weight [M, C/g, K, 1] is the same memory)."""
    B, C, Tin, P = _rows(x)
    M, Cg, K = w.shape[0], w.shape[1], w.shape[2]
    if Cg * groups != C or M % groups:
        raise RuntimeError("conv_forward: channel mismatch")
    Tout = conv_out_len(Tin, K, stride, pad, dil)
    if out is None:
        shape = (B, M, Tout) if x.dim() == 3 else (B, M, Tout, P)
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    a = VcvConvArgs()
    a.x, a.w, a.y = ptr(x), ptr(w), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, groups, Cg, M // groups
    a.Tin, a.Tout, a.P, a.K = Tin, Tout, P, K
    a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q, a.a_mode = stride, dil, -pad, 1, 0, 1, Tout, 0
    _common(a, bias=bias, **kw)
    _launch_conv(a)
    return out


def conv_dgrad(dy, w, x_shape, stride=1, pad=0, dil=1, groups=1, out=None, **kw):
    """Data gradient of conv_forward: dy [B,M,Tout(,P)] -> dx of shape x_shape."""
    B, M, Tout, P = _rows(dy)
    C, Tin = x_shape[1], x_shape[2]
    Cg, K = w.shape[1], w.shape[2]
    if out is None:
        out = torch.empty(tuple(x_shape), device=dy.device, dtype=torch.float32)
    a = VcvConvArgs()
    a.x, a.w, a.y = ptr(dy), ptr(w), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, groups, M // groups, Cg
    a.Tin, a.Tout, a.P, a.K = Tout, Tin, P, K
    a.a_mode = 1
    if stride == 1:
        a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q = 1, -dil, pad, 1, 0, 1, Tin
    else:
        if dil != 1:
            raise RuntimeError("conv_dgrad: stride > 1 needs dilation 1")
        a.s, a.dj, a.off, a.os, a.oo, a.phases = 1, -1, 0, stride, -pad, stride
        a.Q = (Tin - 1 + pad) // stride + 1
    _common(a, **kw)
    _launch_conv(a)
    return out


def conv_wgrad(dy, x, w_shape, stride=1, pad=0, dil=1, groups=1, out=None, a_tf=TF_NONE, aaux=None,
               b_tf=TF_NONE, baux=None, alpha=1.0, slope=0.1):
    """Weight gradient of conv_forward (accumulates onto `out` when given, else onto zeros)."""
    B, M, Tout, P = _rows(dy)
    _, C, Tin, _ = _rows(x)
    Cg, K = w_shape[1], w_shape[2]
    if out is None:
        out = torch.zeros(tuple(w_shape), device=dy.device, dtype=torch.float32)
    a = VcvWgradArgs()
    a.a, a.b, a.aaux, a.baux, a.dw = ptr(dy), ptr(x), ptr(aaux), ptr(baux), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, groups, Cg, M // groups
    a.Ta, a.Tb, a.P, a.K = Tout, Tin, P, K
    a.s, a.dj, a.off = stride, dil, -pad
    a.a_tf, a.b_tf, a.transpose_out, a.alpha, a.slope = a_tf, b_tf, 0, alpha, slope
    _launch_wgrad(a)
    return out


def convT_out_len(tin, k, stride, pad):
    return (tin - 1) * stride - 2 * pad + k


def convT_forward(x, w, bias=None, stride=1, pad=0, out=None, **kw):
    """F.conv_transpose1d forward (groups=1, dilation 1, output_padding 0).  w: [Cin, Cout, K]."""
    B, C, Tin, P = _rows(x)
    Cin, M, K = w.shape
    if Cin != C:
        raise RuntimeError("convT_forward: channel mismatch")
    Tout = convT_out_len(Tin, K, stride, pad)
    if out is None:
        out = torch.empty((B, M, Tout), device=x.device, dtype=torch.float32)
    a = VcvConvArgs()
    a.x, a.w, a.y = ptr(x), ptr(w), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, 1, C, M
    a.Tin, a.Tout, a.P, a.K = Tin, Tout, P, K
    a.a_mode = 1
    if stride == 1:
        a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q = 1, -1, pad, 1, 0, 1, Tout
    else:
        a.s, a.dj, a.off, a.os, a.oo, a.phases = 1, -1, 0, stride, -pad, stride
        a.Q = (Tout - 1 + pad) // stride + 1
    _common(a, bias=bias, **kw)
    _launch_conv(a)
    return out


def convT_dgrad(dy, w, x_shape, stride=1, pad=0, out=None, **kw):
    B, M, Tout, P = _rows(dy)
    Cin, Cout, K = w.shape
    Tin = x_shape[2]
    if out is None:
        out = torch.empty(tuple(x_shape), device=dy.device, dtype=torch.float32)
    a = VcvConvArgs()
    a.x, a.w, a.y = ptr(dy), ptr(w), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, 1, Cout, Cin
    a.Tin, a.Tout, a.P, a.K = Tout, Tin, P, K
    a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q, a.a_mode = stride, 1, -pad, 1, 0, 1, Tin, 0
    _common(a, **kw)
    _launch_conv(a)
    return out


def convT_wgrad(dy, x, w_shape, stride=1, pad=0, out=None, a_tf=TF_NONE, aaux=None, b_tf=TF_NONE,
                baux=None, alpha=1.0, slope=0.1):
    """dW[ci,co,k] of conv_transpose1d: `a` = x (un-shifted), `b` = dy (shifted)."""
    B, Cin, Tin, P = _rows(x)
    _, Cout, Tout, _ = _rows(dy)
    K = w_shape[2]
    if out is None:
        out = torch.zeros(tuple(w_shape), device=dy.device, dtype=torch.float32)
    a = VcvWgradArgs()
    a.a, a.b, a.aaux, a.baux, a.dw = ptr(x), ptr(dy), ptr(aaux), ptr(baux), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, 1, Cout, Cin
    a.Ta, a.Tb, a.P, a.K = Tin, Tout, P, K
    a.s, a.dj, a.off = stride, 1, -pad
    a.a_tf, a.b_tf, a.transpose_out, a.alpha, a.slope = a_tf, b_tf, 0, alpha, slope
    _launch_wgrad(a)
    return out


def bias_grad(dy, aux=None, tf=TF_NONE, slope=0.1):
    B, C = dy.shape[0], dy.shape[1]
    T = dy.numel() // (B * C)
    out = torch.empty((C,), device=dy.device, dtype=torch.float32)
    check(lib().vcv_bias_grad(ptr(dy), ptr(aux), ptr(out), B, C, T, tf, ctypes.c_float(slope),
                              stream()), "vcv_bias_grad")
    return out


# ---------------------------------------------------------------------------------------------
# autograd
# ---------------------------------------------------------------------------------------------
class _ConvFn(torch.autograd.Function):
    """y = act(conv(in_act(x), w) + bias) + res      (act and res are mutually exclusive)."""

    @staticmethod
    def forward(ctx, x, w, bias, res, stride, pad, dil, groups, in_leaky, out_act, slope, transposed):
        x, w = _f32c(x), _f32c(w)
        bias, res = _f32c(bias), _f32c(res)
        if out_act != ACT_NONE and res is not None:
            raise RuntimeError("conv: out_act and res cannot be combined")
        kw = dict(bias=bias, res=res, in_tf=TF_LEAKY if in_leaky else TF_NONE, out_act=out_act,
                  slope=slope)
        if transposed:
            y = convT_forward(x, w, stride=stride, pad=pad, **kw)
        else:
            w3 = w.view(w.shape[0], w.shape[1], w.shape[2])
            y = conv_forward(x, w3, stride=stride, pad=pad, dil=dil, groups=groups, **kw)
        ctx.cfg = (stride, pad, dil, groups, in_leaky, out_act, slope, transposed)
        ctx.has_bias, ctx.has_res = bias is not None, res is not None
        ctx.save_for_backward(x, w, y if out_act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, pad, dil, groups, in_leaky, out_act, slope, transposed = ctx.cfg
        x, w, y = ctx.saved_tensors
        dy = _f32c(dy)
        dtf = _ACT_TO_DTF[out_act]
        dx = dw = db = dres = None
        w3 = w.view(w.shape[0], w.shape[1], w.shape[2])
        if ctx.needs_input_grad[0]:
            kw = dict(in_tf=dtf, xaux=y, slope=slope)
            if in_leaky:
                kw.update(out_tf=TF_DLEAKY, oaux=x)
            if transposed:
                dx = convT_dgrad(dy, w3, x.shape, stride=stride, pad=pad, **kw)
            else:
                dx = conv_dgrad(dy, w3, x.shape, stride=stride, pad=pad, dil=dil, groups=groups, **kw)
        if ctx.needs_input_grad[1]:
            b_tf = TF_LEAKY if in_leaky else TF_NONE
            if transposed:
                dw = convT_wgrad(dy, x, w3.shape, stride=stride, pad=pad, a_tf=b_tf, b_tf=dtf,
                                 baux=y, slope=slope)
            else:
                dw = conv_wgrad(dy, x, w3.shape, stride=stride, pad=pad, dil=dil, groups=groups,
                                a_tf=dtf, aaux=y, b_tf=b_tf, slope=slope)
            dw = dw.view(w.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = bias_grad(dy, aux=y, tf=dtf, slope=slope)
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres, None, None, None, None, None, None, None, None


def conv1d(x, w, bias=None, stride=1, pad=0, dil=1, groups=1, in_leaky=False, out_act=ACT_NONE,
           slope=0.1, res=None):
    """Conv1d on [B,C,T] or the (k,1) Conv2d of the period discriminators on [B,C,H,P]."""
    return _ConvFn.apply(x, w, bias, res, stride, pad, dil, groups, in_leaky, out_act, slope, False)


def conv_transpose1d(x, w, bias=None, stride=1, pad=0, in_leaky=False, out_act=ACT_NONE, slope=0.1):
    return _ConvFn.apply(x, w, bias, None, stride, pad, 1, 1, in_leaky, out_act, slope, True)


# ---------------------------------------------------------------------------------------------
# weight norm
# ---------------------------------------------------------------------------------------------
class _WeightNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, g):
        v, g = _f32c(v), _f32c(g)
        R = v.shape[0]
        C = v.numel() // R
        w = torch.empty_like(v)
        norm = torch.empty((R,), device=v.device, dtype=torch.float32)
        check(lib().vcv_weight_norm_fwd(ptr(v), ptr(g), ptr(w), ptr(norm), R, C, stream()),
              "vcv_weight_norm_fwd")
        ctx.save_for_backward(v, g, norm)
        return w

    @staticmethod
    def backward(ctx, dw):
        v, g, norm = ctx.saved_tensors
        dw = _f32c(dw)
        R = v.shape[0]
        C = v.numel() // R
        dv = torch.empty_like(v)
        dg = torch.empty_like(g)
        check(lib().vcv_weight_norm_bwd(ptr(dw), ptr(v), ptr(g), ptr(norm), ptr(dv), ptr(dg), R, C,
                                        stream()), "vcv_weight_norm_bwd")
        return dv, dg


def weight_norm(v, g):
    """w = g * v / ||v|| with the norm over all dims but 0 (torch.nn.utils.weight_norm, dim=0)."""
    return _WeightNormFn.apply(v, g)


# ---------------------------------------------------------------------------------------------
# streaming helpers
# ---------------------------------------------------------------------------------------------
def scale(x, alpha):
    x = _f32c(x)
    y = torch.empty_like(x)
    check(lib().vcv_scale(ptr(x), ptr(y), alpha, x.numel(), stream()), "vcv_scale")
    return y


class _Avg3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        a, b, c = _f32c(a), _f32c(b), _f32c(c)
        y = torch.empty_like(a)
        check(lib().vcv_avg3(ptr(a), ptr(b), ptr(c), ptr(y), a.numel(), stream()), "vcv_avg3")
        return y

    @staticmethod
    def backward(ctx, dy):
        d = scale(dy, 1.0 / 3.0)
        return d, d, d


def avg3(a, b, c):
    return _Avg3Fn.apply(a, b, c)


class _MaskMulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        x, mask = _f32c(x), _f32c(mask)
        B, C, T = x.shape
        y = torch.empty_like(x)
        check(lib().vcv_mask_mul(ptr(x), ptr(mask), ptr(y), B, C, T, stream()), "vcv_mask_mul")
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _f32c(dy)
        B, C, T = dy.shape
        dx = torch.empty_like(dy)
        check(lib().vcv_mask_mul(ptr(dy), ptr(mask), ptr(dx), B, C, T, stream()), "vcv_mask_mul")
        return dx, None


def mask_mul(x, mask):
    """x [B,C,T] * mask [B,1,T] (mask carries no gradient)."""
    return _MaskMulFn.apply(x, mask)


class _ReflectPadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_pad):
        x = _f32c(x)
        T = x.shape[-1]
        R = x.numel() // T
        y = torch.empty(x.shape[:-1] + (T + n_pad,), device=x.device, dtype=torch.float32)
        check(lib().vcv_reflect_pad_fwd(ptr(x), ptr(y), R, T, T + n_pad, stream()), "vcv_reflect_pad_fwd")
        ctx.T, ctx.n_pad = T, n_pad
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        T = ctx.T
        R = dy.numel() // (T + ctx.n_pad)
        dx = torch.empty(dy.shape[:-1] + (T,), device=dy.device, dtype=torch.float32)
        check(lib().vcv_reflect_pad_bwd(ptr(dy), ptr(dx), R, T, T + ctx.n_pad, stream()),
              "vcv_reflect_pad_bwd")
        return dx, None


def reflect_pad_right(x, n_pad):
    return _ReflectPadFn.apply(x, n_pad)


class _AvgPool4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32c(x)
        T = x.shape[-1]
        R = x.numel() // T
        y = torch.empty(x.shape[:-1] + (T // 2 + 1,), device=x.device, dtype=torch.float32)
        check(lib().vcv_avgpool4_fwd(ptr(x), ptr(y), R, T, stream()), "vcv_avgpool4_fwd")
        ctx.T = T
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        T = ctx.T
        R = dy.numel() // (T // 2 + 1)
        dx = torch.empty(dy.shape[:-1] + (T,), device=dy.device, dtype=torch.float32)
        check(lib().vcv_avgpool4_bwd(ptr(dy), ptr(dx), R, T, stream()), "vcv_avgpool4_bwd")
        return dx


def avgpool4(x):
    """AvgPool1d(kernel_size=4, stride=2, padding=2)."""
    return _AvgPool4Fn.apply(x)


# ---------------------------------------------------------------------------------------------
# losses:  sum_i scale_i * sum f(a_i, b_i)   as ONE autograd node over many tensors
# ---------------------------------------------------------------------------------------------
class _LossSumFn(torch.autograd.Function):
    """mode 0: |a-b| (b carries no grad), mode 1: (a-target)^2.  `scales[i]` multiplies term i."""

    @staticmethod
    def forward(ctx, mode, target, scales, n_a, *tensors):
        a_list = [_f32c(t) for t in tensors[:n_a]]
        b_list = [_f32c(t) for t in tensors[n_a:]] if mode == 0 else [None] * n_a
        out = torch.zeros((), device=a_list[0].device, dtype=torch.float32)
        for a, b, sc in zip(a_list, b_list, scales):
            check(lib().vcv_loss_sum(ptr(a), ptr(b), target, mode, sc, ptr(out), a.numel(), stream()),
                  "vcv_loss_sum")
        ctx.mode, ctx.target, ctx.scales, ctx.n_a = mode, target, scales, n_a
        ctx.save_for_backward(*a_list, *[b for b in b_list if b is not None])
        return out

    @staticmethod
    def backward(ctx, gout):
        n_a = ctx.n_a
        saved = ctx.saved_tensors
        a_list = saved[:n_a]
        b_list = saved[n_a:] if ctx.mode == 0 else [None] * n_a
        gout = _f32c(gout)
        grads = []
        for i, (a, b, sc) in enumerate(zip(a_list, b_list, ctx.scales)):
            if not ctx.needs_input_grad[4 + i]:
                grads.append(None)
                continue
            da = torch.empty_like(a)
            check(lib().vcv_loss_grad(ptr(a), ptr(b), ctx.target, ctx.mode, sc, ptr(gout), ptr(da), 0,
                                      a.numel(), stream()), "vcv_loss_grad")
            grads.append(da)
        grads += [None] * (len(saved) - n_a)
        return (None, None, None, None, *grads)


def l1_mean_sum(a_list, b_list, weight=1.0):
    """weight * sum_i mean|a_i - b_i|  (feature_loss: weight 2; mel loss: weight c_mel)."""
    scales = [weight / a.numel() for a in a_list]
    return _LossSumFn.apply(0, 0.0, scales, len(a_list), *a_list, *b_list)


def sq_mean_sum(a_list, target, weight=1.0):
    """weight * sum_i mean((a_i - target)^2)   (LSGAN terms of losses.py:14-38)."""
    scales = [weight / a.numel() for a in a_list]
    return _LossSumFn.apply(1, float(target), scales, len(a_list), *a_list)


# ---------------------------------------------------------------------------------------------
# STFT magnitude
# ---------------------------------------------------------------------------------------------
_stft_tables = {}


def _stft_consts(device, n_fft):
    key = (str(device), n_fft)
    if key not in _stft_tables:
        import numpy as np
        k = np.arange(n_fft // 2, dtype=np.float64)
        ang = 2.0 * np.pi * k / n_fft
        tw = np.stack([np.cos(ang), -np.sin(ang)], axis=1).astype(np.float32)
        n = np.arange(n_fft, dtype=np.float64)
        win = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)  # periodic Hann
        _stft_tables[key] = (torch.from_numpy(win).to(device), torch.from_numpy(tw).to(device).contiguous())
    return _stft_tables[key]


class _StftMagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, n_fft, hop, pad, reflect, eps):
        y = _f32c(y)
        B, T = y.shape
        win, tw = _stft_consts(y.device, n_fft)
        F_ = (T + 2 * pad - n_fft) // hop + 1
        mag = torch.empty((B, n_fft // 2 + 1, F_), device=y.device, dtype=torch.float32)
        check(lib().vcv_stft_mag_fwd(ptr(y), ptr(win), ptr(tw), ptr(mag), B, T, n_fft, hop, pad,
                                     1 if reflect else 0, eps, stream()), "vcv_stft_mag_fwd")
        ctx.cfg = (n_fft, hop, pad, reflect, eps)
        ctx.save_for_backward(y)
        return mag

    @staticmethod
    def backward(ctx, dmag):
        (y,) = ctx.saved_tensors
        n_fft, hop, pad, reflect, eps = ctx.cfg
        dmag = _f32c(dmag)
        B, T = y.shape
        win, tw = _stft_consts(y.device, n_fft)
        dy = torch.empty_like(y)
        check(lib().vcv_stft_mag_bwd(ptr(y), ptr(win), ptr(tw), ptr(dmag), ptr(dy), B, T, n_fft, hop, pad,
                                     1 if reflect else 0, eps, stream()), "vcv_stft_mag_bwd")
        return dy, None, None, None, None, None


def stft_mag(y, n_fft=2048, hop=512, pad=768, reflect=False, eps=1e-6):
    """sqrt(|STFT|^2 + eps) of y [B, T] -> [B, n_fft/2+1, frames] (Hann window, center=False)."""
    return _StftMagFn.apply(y, n_fft, hop, pad, reflect, eps)


class _MelLogFn(torch.autograd.Function):
    """log(clamp(M @ spec, clamp)) as a 1x1 conv with the log-clamp fused in the epilogue."""

    @staticmethod
    def forward(ctx, spec, melmat, clamp):
        spec, melmat = _f32c(spec), _f32c(melmat)
        w = melmat.view(melmat.shape[0], melmat.shape[1], 1)
        y = conv_forward(spec, w, out_act=ACT_LOGCLAMP, slope=clamp)
        ctx.clamp = clamp
        ctx.save_for_backward(w, y)
        ctx.xshape = spec.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        w, y = ctx.saved_tensors
        dy = _f32c(dy)
        dx = conv_dgrad(dy, w, ctx.xshape, in_tf=_lib.TF_DLOGCLAMP, xaux=y, slope=ctx.clamp)
        return dx, None, None


def mel_log(spec, melmat, clamp=1e-5):
    return _MelLogFn.apply(spec, melmat, clamp)


# ---------------------------------------------------------------------------------------------
# AdamW on flat buffers
# ---------------------------------------------------------------------------------------------
def adamw_step(p, g, m, v, lr, betas, eps, weight_decay, step):
    check(lib().vcv_adamw(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps,
                          weight_decay, step, stream()), "vcv_adamw")
