"""Relative-position multi-head attention: the batched-GEMM form and the fused kernels.

Part of `vcvits_amd.ops` (the package re-exports every name: `from vcvits_amd import ops; ops.conv1d(...)`).  Everything here
runs on the GPU through libvcvits_hip.so; there is no CPU fallback."""
import torch

from .. import tuning
from .._lib import (VcvConvArgs, check, lib, ptr, stream)
from .core import (LAUNCH_COUNTS, _COMPUTE, _f32c)
from .conv import (_common, _launch_conv)
from .blocks import (DROPOUT_TRACE, next_seed)


# ---------------------------------------------------------------------------------------------
# relative-position attention: QK^T and P.V on the MFMA GEMM kernel, banded softmax in between
# ---------------------------------------------------------------------------------------------
def _bgemm(x, w, out, G, Cg, Mg, T, a_mode, alpha=1.0, res=None):
    """out[g*Mg + m, t] = alpha * sum_c A_g(m, c) * x[g*Cg + c, t] (+ res): one grouped 1x1 'conv'
    per (batch, head) pair on vcv_conv_gemm."""
    a = VcvConvArgs()
    a.x, a.w, a.y = ptr(x), ptr(w), ptr(out)
    a.B, a.G, a.Cg, a.Mg = 1, G, Cg, Mg
    a.Tin, a.Tout, a.P, a.K = T, T, 1, 1
    a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q, a.a_mode = 1, 1, 0, 1, 0, 1, T, a_mode
    _common(a, alpha=alpha, res=res)
    _launch_conv(a)
    return out


class _RelAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, embk, embv, mask, n_heads, window, pdrop, seed):
        q, k, v, embk, embv, mask = (_f32c(t) for t in (q, k, v, embk, embv, mask))
        B, C, T = q.shape
        H = n_heads
        dk = C // H
        G = B * H
        qscale = 1.0 / (dk ** 0.5)
        dev = q.device
        S = torch.empty((G, T, T), device=dev, dtype=torch.float32)
        _bgemm(k, q, S, G, dk, T, T, a_mode=1, alpha=qscale)
        P = S  # softmax in place
        Pd = torch.empty_like(S) if pdrop > 0 else None
        Pt = torch.empty_like(S)
        check(lib().vcv_rel_softmax_fwd(ptr(S), ptr(q), ptr(embk), ptr(mask), ptr(P), ptr(Pd), ptr(Pt), B, H,
                                        dk, T, window, qscale, pdrop, seed, stream()), "vcv_rel_softmax_fwd")
        if Pd is None:
            Pd = P
        out = torch.empty_like(q)
        _bgemm(Pt, v, out, G, T, dk, T, a_mode=0)
        check(lib().vcv_rel_value_fwd(ptr(Pd), ptr(embv), ptr(out), B, H, dk, T, window, stream()),
              "vcv_rel_value_fwd")
        ctx.cfg = (H, window, qscale)
        ctx.save_for_backward(q, k, v, embk, embv, mask, P, Pd)
        attn = Pd.view(B, H, T, T)
        ctx.mark_non_differentiable(attn)
        return out, attn

    @staticmethod
    def backward(ctx, dout, _dattn):
        q, k, v, embk, embv, mask, P, Pd = ctx.saved_tensors
        H, window, qscale = ctx.cfg
        dout = _f32c(dout)
        B, C, T = q.shape
        dk = C // H
        G = B * H
        dv = torch.empty_like(v)
        _bgemm(Pd, dout, dv, G, T, dk, T, a_mode=0)
        dP = torch.empty_like(P)
        _bgemm(v, dout, dP, G, dk, T, T, a_mode=1)
        dSt = torch.empty_like(P)
        dqband = torch.empty_like(q)
        dembk = torch.empty_like(embk)
        dembv = torch.empty_like(embv)
        check(lib().vcv_rel_softmax_bwd(ptr(P), ptr(Pd), ptr(dP), ptr(dout), ptr(q), ptr(embk), ptr(embv),
                                        ptr(mask), ptr(dSt), ptr(dqband), ptr(dembk), ptr(dembv), B, H, dk, T,
                                        window, qscale, stream()), "vcv_rel_softmax_bwd")
        dS = dP
        dq = torch.empty_like(q)
        _bgemm(dSt, k, dq, G, T, dk, T, a_mode=0, alpha=qscale, res=dqband)
        dkk = torch.empty_like(k)
        _bgemm(dS, q, dkk, G, T, dk, T, a_mode=0, alpha=qscale)
        return dq, dkk, dv, dembk, dembv, None, None, None, None, None


class _RelAttnFusedFn(torch.autograd.Function):
    """The whole attention of one layer as ONE launch forward and two backward (attention.hip): both contractions on the
    matrix cores straight from the [B, C, T] activations, softmax / band terms / mask fill / dropout in between on the
    LDS tile.  The probabilities are written only for the backward pass or when `attn` is asked for; the dropped
    probabilities are never stored (the backward pass regenerates the mask from the seed)."""

    @staticmethod
    def forward(ctx, q, k, v, embk, embv, mask, n_heads, window, pdrop, seed, want_attn):
        q, k, v, embk, embv, mask = (_f32c(t) for t in (q, k, v, embk, embv, mask))
        B, C, T = q.shape
        H = n_heads
        dk = C // H
        G = B * H
        qscale = 1.0 / (dk ** 0.5)
        dev = q.device
        need_p = any(ctx.needs_input_grad[:5])
        bf = 1 if _COMPUTE[0] == "bf16" else 0
        P = torch.empty((G, T, T), device=dev, dtype=torch.float32) if (need_p or (want_attn and pdrop == 0)) else None
        Pd = torch.empty((G, T, T), device=dev, dtype=torch.float32) if (want_attn and pdrop > 0) else None
        out = torch.empty_like(q)
        check(lib().vcv_rel_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(embk), ptr(embv), ptr(mask), ptr(out), ptr(P), ptr(Pd),
                                     B, H, dk, T, window, qscale, pdrop, seed, bf, stream()), "vcv_rel_attn_fwd")
        LAUNCH_COUNTS["attn_fused"] += 1
        ctx.cfg = (H, window, qscale, pdrop, seed, bf)
        ctx.save_for_backward(q, k, v, embk, embv, mask, P, out)
        attn = None
        if want_attn:
            attn = (Pd if pdrop > 0 else P).view(B, H, T, T)
            ctx.mark_non_differentiable(attn)
        return out, attn

    @staticmethod
    def backward(ctx, dout, _dattn):
        q, k, v, embk, embv, mask, P, out = ctx.saved_tensors
        H, window, qscale, pdrop, seed, bf = ctx.cfg
        dout = _f32c(dout)
        B, C, T = q.shape
        dk = C // H
        # workspace: dS [G, T, T] + the per-(head, query tile) partial tables of the two table gradients
        dS = torch.empty((P.numel() + B * H * ((T + 31) // 32) * 2 * embk.shape[-2] * dk,), device=P.device, dtype=torch.float32)
        dq, dkk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dembk, dembv = torch.empty_like(embk), torch.empty_like(embv)
        # (the forward's output rides along: sum_j dPd Pd = sum_d dO out lets the row pass form dS tile by tile)
        check(lib().vcv_rel_attn_bwd2(ptr(q), ptr(k), ptr(v), ptr(embk), ptr(embv), ptr(mask), ptr(P), ptr(out), ptr(dout),
                                      ptr(dS), ptr(dq), ptr(dkk), ptr(dv), ptr(dembk), ptr(dembv), B, H, dk, T, window, qscale,
                                      pdrop, seed, bf, stream()), "vcv_rel_attn_bwd2")
        return dq, dkk, dv, dembk, dembv, None, None, None, None, None, None


# the fused attention kernels (attention.hip) take every shape they support; VCVITS_ATTN_FUSED=0 keeps the unfused path
_ATTN_FUSED = [tuning.flag("VCVITS_ATTN_FUSED", True, "relative-position attention as the fused MFMA kernels (0: the three-launch form)")]


def rel_attention(q, k, v, emb_rel_k, emb_rel_v, mask, n_heads, window, pdrop=0.0, training=False, want_attn=True):
    """Self-attention with shared-head windowed relative embeddings; mask [B,T] (key/query
    validity).  Returns (out [B,C,T], attn [B,H,T,T] -- None when want_attn is False on the fused path)."""
    p = float(pdrop) if training else 0.0
    seed = next_seed() if p > 0 else 0
    B, C, T = q.shape
    if p > 0 and DROPOUT_TRACE[0] is not None:
        DROPOUT_TRACE[0].append(("attn", p, seed, (B * n_heads, T, T)))
    if _ATTN_FUSED[0] and lib().vcv_rel_attn_supported(B, n_heads, C // n_heads, T, window) == 0:
        return _RelAttnFusedFn.apply(q, k, v, emb_rel_k, emb_rel_v, mask, n_heads, window, p, seed, bool(want_attn))
    return _RelAttnFn.apply(q, k, v, emb_rel_k, emb_rel_v, mask, n_heads, window, p, seed)
