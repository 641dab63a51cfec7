"""Streaming kernels with autograd: scale, avg3, mask multiply, reflect pad, AvgPool1d(4, 2, 2) and the batched loss terms.

Part of `vcvits_amd.ops` (the package re-exports every name: `from vcvits_amd import ops; ops.conv1d(...)`).  Everything here
runs on the GPU through libvcvits_hip.so; there is no CPU fallback."""
import torch

from .._lib import (check, lib, ptr, stream)
from .core import (_f32c, _upload_table)


# ---------------------------------------------------------------------------------------------
# streaming helpers
# ---------------------------------------------------------------------------------------------
def scale(x, alpha):
    x = _f32c(x)
    y = torch.empty_like(x)
    check(lib().vcv_scale(ptr(x), ptr(y), alpha, x.numel(), stream()), "vcv_scale")
    return y


class _ScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return scale(x, alpha)

    @staticmethod
    def backward(ctx, dy):
        return scale(_f32c(dy), ctx.alpha), None


def scale_grad(x, alpha):
    """alpha * x with autograd (scale() is the raw launch)."""
    return _ScaleFn.apply(x, float(alpha))


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
class _LossTermsFn(torch.autograd.Function):
    """terms[i] = scale_i * sum f(a_i, b_i) as ONE autograd node -- and one launch each way -- over many tensors.
    mode 0: |a-b| (b carries no grad), mode 1: (a-target)^2."""

    @staticmethod
    def forward(ctx, mode, target, scales, n_a, *tensors):
        import struct
        import numpy as np
        a_list = [_f32c(t) for t in tensors[:n_a]]
        b_list = [_f32c(t) for t in tensors[n_a:]] if mode == 0 else [None] * n_a
        dev = a_list[0].device
        out = torch.zeros((n_a,), device=dev, dtype=torch.float32)
        tab = np.zeros((n_a, 6), dtype=np.int64)
        blk = off = 0
        for i, (a, b, sc) in enumerate(zip(a_list, b_list, scales)):
            if mode == 0 and b.numel() != a.numel():
                raise RuntimeError("loss terms: shape mismatch")
            n = a.numel()
            tab[i] = (a.data_ptr(), b.data_ptr() if b is not None else 0, n, blk,
                      struct.unpack("<i", struct.pack("<f", sc))[0], off)
            blk += max(1, min((n + 2047) // 2048, 512))
            off += n
        tab_dev = _upload_table(tab, dev)
        check(lib().vcv_loss_many_sum(ptr(tab_dev), n_a, blk, target, mode, ptr(out), stream()), "vcv_loss_many_sum")
        ctx.mode, ctx.target, ctx.n_a, ctx.blocks, ctx.total = mode, target, n_a, blk, off
        ctx.offs = [int(o) for o in tab[:, 5]]
        ctx.save_for_backward(tab_dev, *a_list, *[b for b in b_list if b is not None])
        return out

    @staticmethod
    def backward(ctx, gout):
        n_a = ctx.n_a
        tab_dev = ctx.saved_tensors[0]
        a_list = ctx.saved_tensors[1:1 + n_a]
        gout = _f32c(gout)
        dabuf = torch.empty((ctx.total,), device=gout.device, dtype=torch.float32)
        check(lib().vcv_loss_many_grad(ptr(tab_dev), n_a, ctx.blocks, ctx.target, ctx.mode, ptr(gout), ptr(dabuf),
                                       stream()), "vcv_loss_many_grad")
        grads = [dabuf[o:o + a.numel()].view(a.shape) if ctx.needs_input_grad[4 + i] else None
                 for i, (a, o) in enumerate(zip(a_list, ctx.offs))]
        grads += [None] * (len(ctx.saved_tensors) - 1 - n_a)
        return (None, None, None, None, *grads)


def l1_mean_terms(a_list, b_list, weight=1.0):
    """[weight * mean|a_i - b_i|]_i as a vector (feature_loss: weight 2; mel loss: weight c_mel)."""
    scales = [weight / a.numel() for a in a_list]
    return _LossTermsFn.apply(0, 0.0, scales, len(a_list), *a_list, *b_list)


def sq_mean_terms(a_list, target, weight=1.0):
    """[weight * mean((a_i - target)^2)]_i   (LSGAN terms of losses.py:14-38)."""
    scales = [weight / a.numel() for a in a_list]
    return _LossTermsFn.apply(1, float(target), scales, len(a_list), *a_list)


def l1_mean_sum(a_list, b_list, weight=1.0):
    return l1_mean_terms(a_list, b_list, weight).sum()


def sq_mean_sum(a_list, target, weight=1.0):
    return sq_mean_terms(a_list, target, weight).sum()
