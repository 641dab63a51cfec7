"""Shared state of the launch wrappers: switches (kernel families, compute dtype, deterministic mode), launch counters,
the capture hook and table uploads, layout helpers and the gradient sinks the optimizer registers.

Part of `vcvits_amd.ops` (the package re-exports every name: `from vcvits_amd import ops; ops.conv1d(...)`).  Everything here
runs on the GPU through libvcvits_hip.so; there is no CPU fallback."""
import ctypes

import torch

from .. import _lib, tuning
from .._lib import (ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_TANH, TF_DLEAKY, TF_DRELU, TF_NONE, _GET_DEVICE, check, lib,
                    ptr, stream)


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


_USE_DMA = [tuning.flag("VCVITS_CONV_DMA", True, "conv launches may go to the packed-weight kernel families (0: register-staged kernel only)")]


# fp32 launches try the channel-innermost packed kernel (vcv_conv_pk_*) before the LDS-DMA kernel
_USE_PK = [tuning.flag("VCVITS_CONV_PK", True, "fp32 launches try the packed fp32-input MFMA kernel before the LDS-DMA kernel")]


# fp32 launches try the split-operand kernel first (vcv_conv_x3_*: fp32 operands as three exact bf16 terms each, nine -- or
# six -- bf16 MFMA products per fp32 product, fp32 accumulate: fp32 results at 1.8-2.7 x the fp32 MFMA peak)
_USE_X3 = [tuning.flag("VCVITS_CONV_X3", True, "fp32 launches on the bf16 pipe by exact operand splitting (ops.set_f32_split)")]


# the weight gradient in the same arithmetic (vcv_wgrad_x3: wgrad_bf16.hip with three term planes and producer waves);
# the library takes the shapes where it beats the fp32 kernel (wgrad_dma.hip) and declines the rest
_USE_X3_WGRAD = [tuning.flag("VCVITS_WGRAD_X3", True, "... weight gradients too, where the split kernel is ahead")]


def set_f32_split(on, terms=None, all_shapes=None, wgrad=None):
    """fp32 GEMM-shaped launches on the bf16 matrix pipe by exact operand splitting (True, default) or on fp32-input MFMAs
    (False: bit-for-bit an fmaf chain).  terms: 6 (default: the three products below 2^-24 of the fp32 product left out) or
    9 (all bf16 products); all_shapes: take every eligible launch, not only the shapes where the split kernel is faster;
    wgrad: weight gradients in the same arithmetic too."""
    _USE_X3[0] = bool(on)
    if terms is not None:
        check(lib().vcv_conv_x3_set_terms(int(terms)), "vcv_conv_x3_set_terms")
    if all_shapes is not None:
        check(lib().vcv_conv_x3_set_all(1 if all_shapes else 0), "vcv_conv_x3_set_all")
    if wgrad is not None:
        _USE_X3_WGRAD[0] = bool(wgrad)


# Arithmetic of the GEMM-shaped kernels: "f32" (fp32-input MFMA, exact fp32) or "bf16" (operands rounded to bf16 on
# their way into the matrix cores, fp32 accumulate; activations, master weights, losses and the optimizer stay fp32 --
# the reference's AMP recipe, configs/base.json:18 / train.py:104-106, with bf16 in place of fp16).
_COMPUTE = ["f32"]


# which kernel family each GEMM-shaped launch went to (tests assert that the bf16 path really ran)
LAUNCH_COUNTS = {"bf16": 0, "bf16io": 0, "x3": 0, "pk": 0, "dma": 0, "gemm": 0, "wgrad_bf16": 0, "wgrad_x3": 0, "wgrad": 0,
                 "attn_fused": 0}


# bf16 mode stores the conv <-> conv activations of the decoder's inference pass in bf16 in HBM (vcv_conv_bf16io_*: what the
# reference's fp16 autocast does to every conv output, train.py:104-106); VCVITS_BF16_ACT=0 / set_bf16_activations(False)
# keeps them fp32 (operands still rounded on their way into the matrix cores)
_BF16_ACT = [tuning.flag("VCVITS_BF16_ACT", True, "bf16 mode: no-grad decoder passes keep 16-bit activations in HBM (ops.set_bf16_activations)")]


# bf16 mode: DiscriminatorS's grouped k = 41 forward on the bf16 matrix pipe (VCVITS_GROUPED_BF16=0: fp32-input MFMA)
_GROUPED_BF16 = [tuning.flag("VCVITS_GROUPED_BF16", True, "bf16 mode: DiscriminatorS's grouped k41 layers on the bf16 matrix pipe")]


def set_bf16_activations(on):
    _BF16_ACT[0] = bool(on)


# CAPTURING[0] is the _lib.Capture of the launch sequence being recorded into a HIP graph (vcvits_amd/light/graphed.py), None
# in eager execution.  While it is set: weights derived from parameters are made INSIDE the sequence (caches filled by eager
# passes are not consulted; entries made by this capture are, so the discriminators' weights are normalised and packed once
# per recorded batch as in the eager loop), device tables are filled once after the capture instead of by recorded copy
# nodes, and every tensor from outside the graph's pool that a launcher is handed is held by the graph (_lib.Capture).
CAPTURING = _lib.CAPTURE


def _upload_table(tab, dev):
    """int64 host table (numpy) -> device tensor.  Eager: through a pinned staging copy on the current stream.  While a
    launch sequence is being recorded (CAPTURING): the tensor is allocated now (its address is what the recorded launches
    bake) and filled ONCE, right after the capture (Capture.flush) -- its contents are addresses of the graph's own tensors
    and never change between replays.  It is cut from the capture's table arena, which lives OUTSIDE the graph's pool: a pool
    block is re-written at every replay by the earlier tensors of the sequence that shared it.  (Round 4 recorded a copy
    node from the numpy array instead, re-read at every replay; that remains the fallback when the arena is full.)"""
    cap = CAPTURING[0]
    if cap is None:
        return torch.from_numpy(tab).pin_memory().to(dev, non_blocking=True)
    import numpy as np
    tab = np.ascontiguousarray(tab)
    out = cap.table(tab, torch.int64, tab.shape)
    if out is None:  # (table arena full: round 4's form -- a recorded copy node re-reading the host array)
        out = torch.empty(tab.shape, device=dev, dtype=torch.int64)
        check(lib().vcv_upload_table(ptr(out), ctypes.c_void_p(tab.ctypes.data), tab.nbytes, stream()), "vcv_upload_table")
        cap.append(tab)
    return out


def bf16_activations():
    """True when no-grad decoder passes keep their intermediate activations in bf16."""
    return _BF16_ACT[0] and _COMPUTE[0] == "bf16"


def set_compute_dtype(name):
    if name not in ("f32", "bf16"):
        raise ValueError("compute dtype must be 'f32' or 'bf16'")
    _COMPUTE[0] = name


def compute_dtype():
    return _COMPUTE[0]


_FAMILIES = {}


_FAMILY_KEY = {"vcv_conv_bf16_run": "bf16", "vcv_conv_x3_run": "x3", "vcv_conv_pk_run": "pk", "vcv_conv_bf16io_run": "bf16io"}


_DEVS = {}


def _cur_dev():
    """torch.device of the current GPU (cached objects; the index through the C entry point when torch has it)."""
    i = _GET_DEVICE() if _GET_DEVICE is not None else torch.cuda.current_device()
    d = _DEVS.get(i)
    if d is None:
        d = _DEVS[i] = torch.device("cuda", i)
    return d


# Combine of the split weight-gradient reductions: True = per-workgroup slabs added in a fixed order (bit-reproducible),
# False (default: ~2 % faster per step) = fp32 atomics (order varies from run to run).  The bf16 kernel always uses slabs.
_DETERMINISTIC = [tuning.flag("VCVITS_DETERMINISTIC", False, "bit-reproducible gradients run to run (ops.set_deterministic; also read by the library)")]


def set_deterministic(on):
    """Bit-reproducible gradients run to run: the MFMA weight-gradient kernels combine their split reductions through
    slabs added in a fixed order, bias gradients are summed by one workgroup per channel instead of inside the
    weight-gradient launch, and the library's other split reductions (thin / grouped / register-staged weight gradients,
    activation-derivative bias sums, the one-output-channel forward) run unsplit (vcv_set_deterministic).  Covers the
    GAN step of the vocoder workload (tests/test_determinism_gpu.py: two identical steps, gradients bit for bit); the
    full model's LayerNorm-parameter and relative-position-table gradients still meet in fp32 atomics."""
    _DETERMINISTIC[0] = bool(on)
    check(lib().vcv_set_deterministic(1 if on else 0), "vcv_set_deterministic")


# Gradient sinks: an optimizer that owns a flat gradient buffer registers, per parameter, the view its
# gradient lives in.  The backward of the ops below then adds a leaf parameter's gradient straight into that
# view from the producing kernel and returns None to autograd -- no temporary, no accumulation launch.
# (autograd still fires the parameter's post-accumulate hooks for a None gradient, so the data-parallel
# bucket accounting needs nothing extra; `notify` is for owners that do not use those hooks.)
_GRAD_SINKS = {}


def register_grad_sink(param, grad_view, notify=None):
    import weakref
    _GRAD_SINKS[param.data_ptr()] = (grad_view, notify, weakref.ref(param))


def unregister_grad_sink(param):
    e = _GRAD_SINKS.get(param.data_ptr())
    if e is not None and e[2]() is param:
        del _GRAD_SINKS[param.data_ptr()]


def clear_grad_sinks():
    _GRAD_SINKS.clear()


def _sunk(sink, grad):
    """After a kernel added `grad` into its sink: run the notify and hand autograd nothing."""
    if sink is None:
        return grad
    if sink[1] is not None:
        sink[1]()
    return None


def _sink(t):
    if not _GRAD_SINKS or t is None or not t.requires_grad or not t.is_leaf:
        return None
    e = _GRAD_SINKS.get(t.data_ptr())
    return e if e is not None and e[2]() is t else None  # identity: a recycled address is not the parameter
