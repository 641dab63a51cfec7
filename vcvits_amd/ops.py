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
