"""STFT magnitude / complex STFT / inverse STFT and the log-mel projection.

Part of `vcvits_amd.ops` (the package re-exports every name: `from vcvits_amd import ops; ops.conv1d(...)`).  Everything here
runs on the GPU through libvcvits_hip.so; there is no CPU fallback."""
import torch

from .. import _lib
from .._lib import (ACT_LOGCLAMP, check, lib, ptr, stream)
from .core import (_f32c)
from .conv import (_from_bt, _to_bt, conv_dgrad, conv_forward)


# ---------------------------------------------------------------------------------------------
# STFT magnitude
# ---------------------------------------------------------------------------------------------
_stft_tables = {}


def _stft_consts(device, n_fft, win_length=None):
    """(window [n_fft], twiddle [n_fft/2] (cos, -sin)) on `device`.  The window is the periodic Hann window of `win_length`
    samples, zero-padded on both sides to n_fft when shorter -- what torch.stft / torch.istft do with it
    (mel_processing.py:66-68 passes win_size as win_length)."""
    win_length = n_fft if win_length is None else int(win_length)
    if not 0 < win_length <= n_fft:
        raise ValueError("STFT: 0 < win_length <= n_fft required (torch.stft's own rule)")
    key = (str(device), n_fft, win_length)
    if key not in _stft_tables:
        import numpy as np
        k = np.arange(n_fft // 2, dtype=np.float64)
        ang = 2.0 * np.pi * k / n_fft
        tw = np.stack([np.cos(ang), -np.sin(ang)], axis=1).astype(np.float32)
        n = np.arange(win_length, dtype=np.float64)
        win = np.zeros(n_fft, dtype=np.float32)
        left = (n_fft - win_length) // 2
        win[left:left + win_length] = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)).astype(np.float32)  # periodic Hann
        _stft_tables[key] = (torch.from_numpy(win).to(device), torch.from_numpy(tw).to(device).contiguous())
    return _stft_tables[key]


def _check_n_fft(n_fft):
    if n_fft < 16 or n_fft > 4096 or n_fft & 1:
        raise NotImplementedError("STFT kernels: even n_fft in [16, 4096] (both reference configs: 2048)")


class _StftMagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, n_fft, hop, pad, reflect, eps, win_length=None):
        y = _f32c(y)
        B, T = y.shape
        _check_n_fft(n_fft)
        win, tw = _stft_consts(y.device, n_fft, win_length)
        F_ = (T + 2 * pad - n_fft) // hop + 1
        mag = torch.empty((B, n_fft // 2 + 1, F_), device=y.device, dtype=torch.float32)
        check(lib().vcv_stft_mag_fwd(ptr(y), ptr(win), ptr(tw), ptr(mag), B, T, n_fft, hop, pad,
                                     1 if reflect else 0, eps, stream()), "vcv_stft_mag_fwd")
        ctx.cfg = (n_fft, hop, pad, reflect, eps, win_length)
        ctx.save_for_backward(y)
        return mag

    @staticmethod
    def backward(ctx, dmag):
        (y,) = ctx.saved_tensors
        n_fft, hop, pad, reflect, eps, win_length = ctx.cfg
        dmag = _f32c(dmag)
        B, T = y.shape
        win, tw = _stft_consts(y.device, n_fft, win_length)
        dy = torch.empty_like(y)
        check(lib().vcv_stft_mag_bwd(ptr(y), ptr(win), ptr(tw), ptr(dmag), ptr(dy), B, T, n_fft, hop, pad,
                                     1 if reflect else 0, eps, stream()), "vcv_stft_mag_bwd")
        return dy, None, None, None, None, None, None


def stft_mag(y, n_fft=2048, hop=512, pad=768, reflect=False, eps=1e-6, win_length=None):
    """sqrt(|STFT|^2 + eps) of y [B, T] -> [B, n_fft/2+1, frames] (Hann window of win_length <= n_fft, center=False).
    n_fft = 2048 runs the kernels tuned for the reference configs; any other power of two in [64, 4096] the generic
    radix-2 kernels, any other even size in [16, 4096] a direct DFT (stft_generic.hip)."""
    return _StftMagFn.apply(y, n_fft, hop, pad, reflect, eps, win_length)


def stft_complex(y, n_fft=2048, hop=512, pad=768, reflect=False, win_length=None):
    """Complex STFT of y [B, T] -> complex64 [B, n_fft/2+1, frames] (no autograd: the reference runs the
    source pipeline under inference_mode, vcvits.py:61-62)."""
    y = _f32c(y.detach())
    B, T = y.shape
    _check_n_fft(n_fft)
    win, tw = _stft_consts(y.device, n_fft, win_length)
    F_ = (T + 2 * pad - n_fft) // hop + 1
    out = torch.empty((B, n_fft // 2 + 1, F_, 2), device=y.device, dtype=torch.float32)
    check(lib().vcv_stft_complex_fwd(ptr(y), ptr(win), ptr(tw), ptr(out), B, T, n_fft, hop, pad, 1 if reflect else 0,
                                     stream()), "vcv_stft_complex_fwd")
    return torch.view_as_complex(out)


def istft(spec, n_fft=2048, hop=512, center=True, win_length=None):
    """Inverse STFT of complex64 [B, n_fft/2+1, F] -> [B, hop*(F-1)] (torch.istft, Hann window of win_length <= n_fft)."""
    s = torch.view_as_real(spec.detach()).contiguous()
    B, _, F_, _ = s.shape
    _check_n_fft(n_fft)
    win, tw = _stft_consts(s.device, n_fft, win_length)
    L = n_fft + hop * (F_ - 1)
    ola = torch.empty((B, L), device=s.device, dtype=torch.float32)
    tout = hop * (F_ - 1) if center else L
    out = torch.empty((B, tout), device=s.device, dtype=torch.float32)
    check(lib().vcv_istft(ptr(s), ptr(win), ptr(tw), ptr(ola), ptr(out), B, F_, n_fft, hop, 1 if center else 0,
                          stream()), "vcv_istft")
    return out


class _MelLogFn(torch.autograd.Function):
    """log(clamp(M @ spec, clamp)) as a 1x1 conv with the log-clamp fused in the epilogue.  Short segments (the training
    step's 32 frames) fold the batch into the column dimension, as the convs of short sequences do: one batch element's
    32 columns cannot fill a GEMM tile (the launch ran at 1.7 TFLOP/s on the generic kernel)."""

    @staticmethod
    def forward(ctx, spec, melmat, clamp):
        spec, melmat = _f32c(spec), _f32c(melmat)
        w = melmat.view(melmat.shape[0], melmat.shape[1], 1)
        ctx.bt = spec.dim() == 3 and spec.shape[0] > 1 and spec.shape[2] <= 64
        if ctx.bt:
            yf = conv_forward(_to_bt(spec), w, out_act=ACT_LOGCLAMP, slope=clamp)  # [1, n_mel, T, B]
            y = _from_bt(yf)
        else:
            y = yf = conv_forward(spec, w, out_act=ACT_LOGCLAMP, slope=clamp)
        ctx.clamp = clamp
        ctx.save_for_backward(w, yf)
        ctx.xshape = spec.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        w, yf = ctx.saved_tensors
        dy = _f32c(dy)
        # the log-clamp derivative as its own streaming pass over the [n_mel, frames] gradient (a few hundred KB), then a
        # PLAIN data gradient: the packed-weight kernels take no derivative mask on their input, so the fused form fell to the
        # register-staged kernel (76 us for 0.27 GFLOP in the step; the split-operand kernel does the same GEMM in ~10)
        dyf = _to_bt(dy) if ctx.bt else dy
        dye = torch.empty_like(dyf)
        check(lib().vcv_act_grad(ptr(dyf), ptr(yf), ptr(dye), _lib.TF_DLOGCLAMP, ctx.clamp, dyf.numel(), stream()), "vcv_act_grad")
        if ctx.bt:
            B, C, T = ctx.xshape
            dx = _from_bt(conv_dgrad(dye, w, (1, C, T, B)))
        else:
            dx = conv_dgrad(dye, w, ctx.xshape)
        return dx, None, None


def mel_log(spec, melmat, clamp=1e-5):
    return _MelLogFn.apply(spec, melmat, clamp)
