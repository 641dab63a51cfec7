"""Convolution launches (forward, data gradient, weight gradient, transposed forms, bias gradient) over VcvConvArgs /
VcvWgradArgs, the weight-gradient arena, feature-map taps and the autograd Functions conv1d / conv_transpose1d.

Part of `vcvits_amd.ops` (the package re-exports every name: `from vcvits_amd import ops; ops.conv1d(...)`).  Everything here
runs on the GPU through libvcvits_hip.so; there is no CPU fallback."""
import ctypes

import torch

from .._lib import (ACT_LEAKY, ACT_NONE, ACT_TANH, TF_DLEAKY, TF_LEAKY, TF_NONE, VcvConvArgs, VcvWgradArgs, check,
                    lib, ptr, stream)
from .core import (LAUNCH_COUNTS, _ACT_TO_DTF, _COMPUTE, _DETERMINISTIC, _FAMILIES, _FAMILY_KEY, _GROUPED_BF16,
                   _USE_DMA, _USE_PK, _USE_X3, _USE_X3_WGRAD, _cur_dev, _f32c, _rows, _sink, _sunk, conv_out_len)
from .weights import (_PACK_JOBS, _stable_entry)
from .. import tuning


def _launch_conv(a, flip_w=None, wt=None):
    """Forward-type launches go to the packed-weight kernels when one is eligible -- the bf16-operand kernel
    (vcv_conv_bf16_*) under set_compute_dtype("bf16"), else the fp32 LDS-DMA kernel (vcv_conv_dma_*) -- everything else to
    the register-staged fp32 kernel.  flip_w: original [C, M, K] weight of a stride-1 data gradient (the pack flips it;
    the register path needs the explicit flipped copy in a.w).  Packed weights of tensors inside a cached weight-norm
    buffer (see _WeightNormManyFn) are kept with that buffer and reused until its parameters change."""
    if _USE_DMA[0] and (a.a_mode == 0 or (a.a_mode == 1 and (a.phases > 1 or a.ms > 1))):
        L = lib()
        if flip_w is not None:
            saved = a.w
            a.w = ptr(flip_w)
        flip = 1 if flip_w is not None else 0
        plan = (ctypes.c_int64 * 3)()
        fkey = (_COMPUTE[0], _USE_X3[0], _USE_PK[0], a.io)
        families = _FAMILIES.get(fkey)
        if families is None and a.io != 0:  # bf16 activations: one family reads / writes them
            families = _FAMILIES[fkey] = ((L.vcv_conv_bf16io_plan, L.vcv_conv_bf16io_run, "vcv_conv_bf16io_run"),)
        if families is None:  # (built once per switch setting: this function runs ~450 times per step)
            families = ((L.vcv_conv_bf16_plan, L.vcv_conv_bf16_run, "vcv_conv_bf16_run"),) if _COMPUTE[0] == "bf16" else ()
            if _USE_X3[0] and _COMPUTE[0] == "f32":
                families += ((L.vcv_conv_x3_plan, L.vcv_conv_x3_run, "vcv_conv_x3_run"),)
            if _USE_PK[0]:
                families += ((L.vcv_conv_pk_plan, L.vcv_conv_pk_run, "vcv_conv_pk_run"),)
            families += ((L.vcv_conv_dma_plan, L.vcv_conv_dma_run, "vcv_conv_dma_run"),)
            _FAMILIES[fkey] = families
        for plan_fn, run_fn, name in families:
            if plan_fn(ctypes.byref(a), flip, plan) != 0:
                continue
            dev = _cur_dev()
            ent = _stable_entry(a.w)
            wver = 0
            if ent is not None and "dirty" in ent:
                # a parameter region: the pack is valid for the weight tensor's version it was made from (an in-place write
                # that did not go through the optimizer bumps it); callers that do not hand the tensor over pack per use
                wtt = wt if wt is not None else flip_w
                if wtt is None:
                    ent = None
                else:
                    wver = wtt._version
            packs = ent["packs"] if ent is not None else None
            key = (a.w, plan[0], plan[2]) if wver == 0 else (a.w, plan[0], plan[2], wver)
            pack = packs.get(key) if packs is not None else None
            valid = 1 if pack is not None else 0
            if pack is None:
                pack = torch.empty((plan[0],), device=dev, dtype=torch.float32)
                if packs is not None:
                    packs[key] = pack
                    # remember the job: the next time this tree's weights are re-normalised all of its packs are made
                    # in one launch (_replay_packs)
                    if len(_PACK_JOBS) > 64:
                        _PACK_JOBS.clear()
                    jobs = _PACK_JOBS.get(ent["key"])
                    if jobs is None or jobs["shapes"] != ent["shapes"]:  # (a recycled address set is another module's)
                        jobs = _PACK_JOBS[ent["key"]] = {"shapes": ent["shapes"], "jobs": {}}
                    # (a region's job remembers the tensor version it was recorded at: the optimizer's raw update leaves
                    # versions alone, so the replay registers the pack under the version the next use will ask for)
                    jobs["jobs"][(a.w - ent["lo"], int(plan[0]), int(plan[2]), name)] = (bytes(a), flip, wver)
            scratch = torch.empty((plan[1],), device=dev, dtype=torch.float32) if plan[1] > 0 else None
            check(run_fn(ctypes.byref(a), ptr(pack), ptr(scratch), flip, valid, stream()), name)
            LAUNCH_COUNTS[_FAMILY_KEY.get(name, "dma")] += 1
            return
        if flip_w is not None:
            a.w = saved
    if a.io != 0:
        raise RuntimeError("vcvits_amd: no kernel takes this launch with bf16 activations (shape outside vcv_conv_bf16io_*)")
    LAUNCH_COUNTS["gemm"] += 1
    check(lib().vcv_conv_gemm(ctypes.byref(a), stream()), "vcv_conv_gemm")


def _launch_wgrad(a):
    if _DETERMINISTIC[0] and a.G == 1:
        nw = a.Mg * a.Cg * a.K
        n = min(nw * 512, max(nw * 4, 24 << 20))
        slab = torch.empty((n,), device=_cur_dev(), dtype=torch.float32)
        a.slab, a.slab_floats = ptr(slab), n
    if _COMPUTE[0] == "bf16" or (_USE_X3[0] and _USE_X3_WGRAD[0]):
        L = lib()
        bf = _COMPUTE[0] == "bf16"
        n = (L.vcv_wgrad_bf16_scratch if bf else L.vcv_wgrad_x3_scratch)(ctypes.byref(a))
        if n > 0:
            scratch = torch.empty((n,), device=_cur_dev(), dtype=torch.float32)
            if bf:
                check(L.vcv_wgrad_bf16(ctypes.byref(a), ptr(scratch), n, stream()), "vcv_wgrad_bf16")
            else:
                check(L.vcv_wgrad_x3(ctypes.byref(a), ptr(scratch), n, stream()), "vcv_wgrad_x3")
            LAUNCH_COUNTS["wgrad_bf16" if bf else "wgrad_x3"] += 1
            return
    LAUNCH_COUNTS["wgrad"] += 1
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
    plain = (all(kw.get(k) is None for k in ("res", "mask", "xaux", "oaux")) and not kw.get("accumulate", False)
             and kw.get("out_tf", TF_NONE) == TF_NONE and kw.get("alpha", 1.0) == 1.0)
    if (groups > 1 and Cg == 4 and K == 41 and stride == 4 and pad == 20 and dil == 1 and P == 1
            and M // groups in (4, 16) and plain and kw.get("in_tf", TF_NONE) == TF_NONE
            and kw.get("out_act", ACT_NONE) in (ACT_NONE, ACT_LEAKY)):
        # bf16 mode: the 16-channel groups on the bf16 matrix pipe (4 taps x 4 channels per MFMA step)
        fn = "vcv_grouped41_fwd_bf16" if (_COMPUTE[0] == "bf16" and M // groups == 16 and _GROUPED_BF16[0]) else "vcv_grouped41_fwd"
        check(getattr(lib(), fn)(ptr(x), ptr(w), ptr(bias), ptr(out), B, groups, M // groups, Tin, Tout,
                                 kw.get("out_act", ACT_NONE), kw.get("slope", 0.1), stream()), fn)
        return out
    if (C == 1 and groups == 1 and M <= 64 and K <= 16 and plain and kw.get("in_tf", TF_NONE) == TF_NONE
            and kw.get("out_act", ACT_NONE) in (ACT_NONE, ACT_LEAKY)):
        # one input channel: every position produces all M channels from K register taps (HBM write stream)
        check(lib().vcv_conv_c1_fwd(ptr(x), ptr(w), ptr(bias), ptr(out), B, M, Tin, Tout, P, K, stride, dil, pad,
                                    kw.get("out_act", ACT_NONE), kw.get("slope", 0.1), stream()), "vcv_conv_c1_fwd")
        return out
    if (M == 1 and groups == 1 and C >= 16 and plain
            and kw.get("in_tf", TF_NONE) in (TF_NONE, TF_LEAKY)
            and kw.get("out_act", ACT_NONE) in (ACT_NONE, ACT_TANH)):
        # one output channel: HBM-bound matrix-vector kernel instead of a 32-row MFMA tile
        check(lib().vcv_conv_m1_fwd(ptr(x), ptr(w), ptr(bias), ptr(out), B, C, Tin, Tout, P, K, stride, dil, pad,
                                    1 if kw.get("in_tf", TF_NONE) == TF_LEAKY else 0, kw.get("out_act", ACT_NONE),
                                    kw.get("slope", 0.1), stream()), "vcv_conv_m1_fwd")
        return out
    a = VcvConvArgs()
    a.x, a.w, a.y = ptr(x), ptr(w), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, groups, Cg, M // groups
    a.Tin, a.Tout, a.P, a.K = Tin, Tout, P, K
    a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q, a.a_mode = stride, dil, -pad, 1, 0, 1, Tout, 0
    _common(a, bias=bias, **kw)
    _launch_conv(a, wt=w)
    return out


def conv_dgrad(dy, w, x_shape, stride=1, pad=0, dil=1, groups=1, out=None, **kw):
    """Data gradient of conv_forward: dy [B,M,Tout(,P)] -> dx of shape x_shape."""
    B, M, Tout, P = _rows(dy)
    C, Tin = x_shape[1], x_shape[2]
    Cg, K = w.shape[1], w.shape[2]
    if out is None:
        out = torch.empty(tuple(x_shape), device=dy.device, dtype=torch.float32)
    if (groups > 1 and Cg == 4 and K == 41 and stride == 4 and pad == 20 and dil == 1 and P == 1
            and M // groups in (4, 16) and set(kw) <= {"in_tf", "xaux", "slope"}
            and kw.get("in_tf", TF_NONE) in (TF_NONE, TF_DLEAKY)):
        fn = "vcv_grouped41_dgrad_bf16" if (_COMPUTE[0] == "bf16" and M // groups == 16 and _GROUPED_BF16[0]) else "vcv_grouped41_dgrad"
        check(getattr(lib(), fn)(ptr(dy), ptr(kw.get("xaux")), ptr(w), ptr(out), B, groups, M // groups, Tin,
                                 Tout, kw.get("in_tf", TF_NONE), kw.get("slope", 0.1), stream()), fn)
        return out
    if (C == 1 and groups == 1 and M <= 64 and K <= 16 and kw.get("in_tf", TF_NONE) == TF_NONE
            and set(kw) <= {"in_tf", "xaux", "slope"}):
        check(lib().vcv_conv_c1_dgrad(ptr(dy), ptr(w), ptr(out), B, M, Tin, Tout, P, K, stride, dil, pad, stream()),
              "vcv_conv_c1_dgrad")
        return out
    if (M == 1 and groups == 1 and stride == 1 and C >= 16 and K <= 16 and kw.get("in_tf", TF_NONE) == TF_NONE
            and set(kw) <= {"in_tf", "xaux", "slope", "out_tf", "oaux"}
            and kw.get("out_tf", TF_NONE) in (TF_NONE, TF_DLEAKY)):
        # one OUTPUT channel (conv_post of the discriminators, 1024 -> 1; of the generator, 32 -> 1): the data gradient
        # is a one-input-channel convolution of dy with the flipped taps -- an HBM write stream, not a GEMM (was 35 us
        # on the generic kernel); the derivative of a leaky-ReLU on the conv's input rides in the launch's epilogue
        masked = kw.get("out_tf", TF_NONE) == TF_DLEAKY
        # (the [C, K] rows of w are read backwards by the kernel: no flipped copy, no launch of its own for it)
        check(lib().vcv_conv_c1_fwd_flip(ptr(dy), ptr(w), None, ptr(out), ptr(kw["oaux"]) if masked else None, B, C, Tout,
                                         Tin, P, K, 1, dil, (K - 1) * dil - pad, ACT_NONE, kw.get("slope", 0.1), 1, stream()),
              "vcv_conv_c1_fwd_flip")
        return out
    if stride == 1 and groups == 1 and M >= 32 and C >= 32:
        # stride-1 data gradient == forward conv with the flipped / transposed weights: the forward
        # staging path (row-major weight rows) is the faster one
        a = VcvConvArgs()
        a.x, a.y = ptr(dy), ptr(out)
        a.B, a.G, a.Cg, a.Mg = B, 1, M, C
        a.Tin, a.Tout, a.P, a.K = Tout, Tin, P, K
        a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q, a.a_mode = 1, dil, pad - (K - 1) * dil, 1, 0, 1, Tin, 0
        _common(a, **kw)
        a.w = ptr(w)
        if _USE_DMA[0] and ((_COMPUTE[0] == "bf16" and lib().vcv_conv_bf16_plan(ctypes.byref(a), 1, (ctypes.c_int64 * 3)()) == 0)
                            or (_COMPUTE[0] == "f32" and _USE_X3[0] and lib().vcv_conv_x3_plan(ctypes.byref(a), 1, (ctypes.c_int64 * 3)()) == 0)
                            or (_USE_PK[0] and lib().vcv_conv_pk_plan(ctypes.byref(a), 1, (ctypes.c_int64 * 3)()) == 0)
                            or lib().vcv_conv_dma_plan(ctypes.byref(a), 1, (ctypes.c_int64 * 3)()) == 0):
            _launch_conv(a, flip_w=w)
            return out
        wt = torch.empty((C, M, K), device=dy.device, dtype=torch.float32)
        check(lib().vcv_weight_flip_transpose(ptr(w), ptr(wt), M, C, K, stream()), "vcv_weight_flip_transpose")
        a.w = ptr(wt)
        check(lib().vcv_conv_gemm(ctypes.byref(a), stream()), "vcv_conv_gemm")
        return out
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
    _launch_conv(a, wt=w)
    return out


# Zero-initialised accumulators for weight gradients of INTERMEDIATE weights (weight-normed layers: the
# gradient is consumed by the weight-norm backward of the same pass and then dead).  Instead of one fill
# launch per layer, slices of one arena are handed out and the used part is re-zeroed once per backward pass
# (wgrad_arena_reset, called from the optimizer's zero_grad).
_ARENA = {"buf": None, "off": 0, "need": 0, "on": False}


def arena_swap(new):
    """Install `new` (a dict like _ARENA) as the weight-gradient arena and return the old one's state (light/graphed.py: a
    recorded batch has an arena of its own, so eager passes between replays cannot move or resize what it baked)."""
    old = dict(_ARENA)
    _ARENA.clear()
    _ARENA.update(new)
    return old


def wgrad_arena_reset():
    a = _ARENA
    if a["buf"] is not None and a["off"] > 0:
        a["buf"][:a["off"]].zero_()
    if a["need"] > (a["buf"].numel() if a["buf"] is not None else 0) and torch.cuda.is_available():
        if a["buf"] is not None:
            # a captured pass (light/graphed.py) may have slices of the old buffer baked into its kernels: retire it, never
            # free it (a freed buffer under a replayed graph was a memory fault in the eager step that grew the arena)
            a.setdefault("retired", []).append(a["buf"])
        a["buf"] = torch.zeros((int(a["need"] * 1.1) + 1024,), device=torch.device("cuda", torch.cuda.current_device()),
                               dtype=torch.float32)
    a["off"] = a["need"] = 0
    a["on"] = True


def _wgrad_zeros(shape, dev, arena):
    a = _ARENA
    n = 1
    for d in shape:
        n *= int(d)
    if arena and a["on"]:
        n64 = (n + 63) & ~63
        a["need"] += n64
        buf = a["buf"]
        if buf is not None and buf.device == dev and a["off"] + n64 <= buf.numel():
            o = a["off"]
            a["off"] = o + n64
            return buf[o:o + n].view(tuple(shape))
    return torch.zeros(tuple(shape), device=dev, dtype=torch.float32)


def conv_wgrad(dy, x, w_shape, stride=1, pad=0, dil=1, groups=1, out=None, a_tf=TF_NONE, aaux=None,
               b_tf=TF_NONE, baux=None, alpha=1.0, slope=0.1, arena=False, dbias=None):
    """Weight gradient of conv_forward (accumulates onto `out` when given, else onto zeros).  dbias [M]: the bias
    gradient sum(dy) is ADDED onto it -- inside the weight-gradient launch where the kernel supports it (the MFMA
    kernels collect the row sums of dy while staging it), by one extra streaming pass otherwise."""
    B, M, Tout, P = _rows(dy)
    _, C, Tin, _ = _rows(x)
    Cg, K = w_shape[1], w_shape[2]
    if out is None:
        out = _wgrad_zeros(w_shape, dy.device, arena)
    if dbias is not None and (_DETERMINISTIC[0] or a_tf != TF_NONE or groups != 1 or min(M, C) == 1):
        bias_grad(dy, aux=aaux, tf=a_tf, slope=slope, out=dbias)
        dbias = None
    if (groups > 1 and Cg == 4 and K == 41 and stride == 4 and pad == 20 and dil == 1 and P == 1
            and M // groups in (4, 16) and b_tf == TF_NONE and a_tf in (TF_NONE, TF_DLEAKY) and alpha == 1.0):
        fn = "vcv_grouped41_wgrad_bf16" if (_COMPUTE[0] == "bf16" and M // groups == 16 and _GROUPED_BF16[0]) else "vcv_grouped41_wgrad"
        check(getattr(lib(), fn)(ptr(dy), ptr(aaux), ptr(x), ptr(out), B, groups, M // groups, Tin, Tout, a_tf,
                                 slope, stream()), fn)
        return out
    if groups == 1 and min(M, C) == 1 and K <= 16:
        check(lib().vcv_thin_wgrad(ptr(dy), ptr(x), ptr(aaux), ptr(baux), ptr(out), B, M, C, Tout, Tin, P, K, stride,
                                   dil, -pad, a_tf, b_tf, slope, alpha, stream()), "vcv_thin_wgrad")
        return out
    a = VcvWgradArgs()
    a.a, a.b, a.aaux, a.baux, a.dw = ptr(dy), ptr(x), ptr(aaux), ptr(baux), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, groups, Cg, M // groups
    a.Ta, a.Tb, a.P, a.K = Tout, Tin, P, K
    a.s, a.dj, a.off = stride, dil, -pad
    a.a_tf, a.b_tf, a.transpose_out, a.alpha, a.slope = a_tf, b_tf, 0, alpha, slope
    a.dbias = ptr(dbias)
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
        out = torch.empty((B, M, Tout) if x.dim() == 3 else (B, M, Tout, P), device=x.device, dtype=torch.float32)
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
    _launch_conv(a, wt=w)
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
    _launch_conv(a, wt=w)
    return out


def convT_wgrad(dy, x, w_shape, stride=1, pad=0, out=None, a_tf=TF_NONE, aaux=None, b_tf=TF_NONE,
                baux=None, alpha=1.0, slope=0.1, arena=False):
    """dW[ci,co,k] of conv_transpose1d: `a` = x (un-shifted), `b` = dy (shifted)."""
    B, Cin, Tin, P = _rows(x)
    _, Cout, Tout, _ = _rows(dy)
    K = w_shape[2]
    if out is None:
        out = _wgrad_zeros(w_shape, dy.device, arena)
    a = VcvWgradArgs()
    a.a, a.b, a.aaux, a.baux, a.dw = ptr(x), ptr(dy), ptr(aaux), ptr(baux), ptr(out)
    a.B, a.G, a.Cg, a.Mg = B, 1, Cout, Cin
    a.Ta, a.Tb, a.P, a.K = Tin, Tout, P, K
    a.s, a.dj, a.off = stride, 1, -pad
    a.a_tf, a.b_tf, a.transpose_out, a.alpha, a.slope = a_tf, b_tf, 0, alpha, slope
    _launch_wgrad(a)
    return out


def bias_grad(dy, aux=None, tf=TF_NONE, slope=0.1, out=None):
    """Per-channel sum of tf(dy); with `out` the sums are ADDED onto it."""
    B, C = dy.shape[0], dy.shape[1]
    T = dy.numel() // (B * C)
    acc = 0 if out is None else 1
    if out is None:
        out = torch.empty((C,), device=dy.device, dtype=torch.float32)
    check(lib().vcv_bias_grad(ptr(dy), ptr(aux), ptr(out), B, C, T, tf, ctypes.c_float(slope), acc,
                              stream()), "vcv_bias_grad")
    return out


# ---------------------------------------------------------------------------------------------
# autograd
# ---------------------------------------------------------------------------------------------
_GRAD_B0 = [0]


class grad_batch_start:
    """Context: convolutions recorded inside only need data gradients for batch elements >= b0 (the
    leading b0 elements are inputs without gradient, e.g. the real waveforms stacked in front of the
    generated ones in the generator step).  Their backward then launches the data-gradient kernels
    on the trailing sub-batch only; the leading part of the returned gradient is unspecified."""

    def __init__(self, b0):
        self.b0 = int(b0)

    def __enter__(self):
        self.prev = _GRAD_B0[0]
        _GRAD_B0[0] = self.b0
        return self

    def __exit__(self, *exc):
        _GRAD_B0[0] = self.prev
        return False


class FmapTap:
    """A feature map recorded inside grad_batch_start(b0): `real` = the leading b0 batch elements (no gradient),
    `fake` = the trailing ones (gradient flows through fmap_tap's node)."""
    __slots__ = ("real", "fake")

    def __init__(self, real, fake):
        self.real, self.fake = real, fake


class _TapFn(torch.autograd.Function):
    """(x, x[b0:]) with one backward node: the gradient of the trailing slice is added IN PLACE onto the
    trailing part of the pass-through gradient (no zero-filled full-size temporary, no copy, no full-size add,
    which is what slicing after the fact costs).  The leading part of the returned gradient is unspecified, as
    grad_batch_start promises its consumers."""

    @staticmethod
    def forward(ctx, x, b0, producer=None):
        ctx.b0 = b0
        ctx.producer = producer  # the conv node that made x, when its backward can take the trailing gradient itself
        return x.view(x.shape), x[b0:]

    @staticmethod
    def backward(ctx, g_pass, g_tail):
        b0 = ctx.b0
        if g_pass is None and g_tail is None:
            return None, None, None
        if g_pass is None:
            g_pass = torch.empty((b0 + g_tail.shape[0],) + tuple(g_tail.shape[1:]), device=g_tail.device, dtype=g_tail.dtype)
            g_pass[b0:].copy_(g_tail)
            return g_pass, None, None
        if g_tail is not None:
            prod = ctx.producer
            if (_TAP_FUSE[0] and prod is not None and prod.tap_add is None and g_tail.dtype == torch.float32
                    and g_tail.is_contiguous() and g_pass.dtype == torch.float32):
                # x's producer is the next node of this backward pass: its activation-derivative pass reads g_pass anyway
                # and sums g_tail into it there (vcv_act_grad_add) -- no read-modify-write pass over g_pass here
                prod.tap_add = (g_tail, b0)
                return g_pass, None, None
            if not g_pass.is_contiguous():
                g_pass = g_pass.contiguous()
            g_pass[b0:].add_(g_tail)
        return g_pass, None, None


_TAP_FUSE = [tuning.flag("VCVITS_TAP_FUSE", True, "feature-map taps: the second gradient summed inside the producer's activation-derivative pass (A/B)")]


def fmap_tap(x):
    """Record a discriminator feature map: returns (x to continue with, the recorded map).  Inside
    grad_batch_start(b0) with gradients flowing, the record is a FmapTap whose `fake` half shares one backward
    node with the pass-through; otherwise it is x itself."""
    b0 = _GRAD_B0[0]
    if b0 > 0 and x.requires_grad and torch.is_grad_enabled() and b0 < x.shape[0]:
        fn = x.grad_fn
        producer = fn if (fn is not None and getattr(fn, "tap_ok", False)) else None
        xp, tail = _TapFn.apply(x, b0, producer)
        return xp, FmapTap(x.detach()[:b0], tail)
    return x, x


def _to_bt(t):
    """[B, C, T] -> [1, C, T, B] (batch as the innermost column; pure data movement)."""
    return t.permute(1, 2, 0).contiguous().unsqueeze(0)


def _from_bt(t):
    """[1, C, T, B] -> [B, C, T]."""
    return t[0].permute(2, 0, 1).contiguous()


class _ConvFn(torch.autograd.Function):
    """y = act(conv(in_act(x), w) + bias) + res      (act and res are mutually exclusive)."""

    @staticmethod
    def forward(ctx, x, w, bias, res, stride, pad, dil, groups, in_leaky, out_act, slope, transposed, link=None):
        ctx.link = link
        x, w = _f32c(x), _f32c(w)
        bias, res = _f32c(bias), _f32c(res)
        if out_act != ACT_NONE and res is not None:
            raise RuntimeError("conv: out_act and res cannot be combined")
        kw = dict(bias=bias, res=res, in_tf=TF_LEAKY if in_leaky else TF_NONE, out_act=out_act,
                  slope=slope)
        # Short sequences (the last layers of DiscriminatorS and its pooled scales: 5..64 frames): one batch element cannot fill
        # a GEMM tile, so the batch is folded into the kernel's column dimension -- x[b,c,t] is viewed as one
        # "image" [1,C,T,B] (P = B columns) and the tile's N runs over (t, b) pairs.
        ctx.bt = (not transposed and x.dim() == 3 and groups == 1 and res is None and x.shape[0] > 1
                  and x.shape[2] <= 64 and w.shape[0] >= 32 and w.shape[1] >= 32)
        if transposed and x.dim() == 3 and x.shape[2] <= 64 and x.shape[0] > 1 and res is None and stride > 1:
            # first generator stage (32 frames per utterance): a phase of the transposed conv has 32 columns per batch
            # element, below what the packed-weight kernels tile (the launch fell to the generic kernel at 14 TFLOP/s);
            # folded like the short convs, one phase has 32 x B columns
            y = _from_bt(convT_forward(_to_bt(x), w, stride=stride, pad=pad, **kw))
        elif transposed:
            y = convT_forward(x, w, stride=stride, pad=pad, **kw)
        elif ctx.bt:
            w3 = w.view(w.shape[0], w.shape[1], w.shape[2])
            x = _to_bt(x)
            y = _from_bt(conv_forward(x, w3, stride=stride, pad=pad, dil=dil, groups=groups, **kw))
        else:
            w3 = w.view(w.shape[0], w.shape[1], w.shape[2])
            y = conv_forward(x, w3, stride=stride, pad=pad, dil=dil, groups=groups, **kw)
        ctx.cfg = (stride, pad, dil, groups, in_leaky, out_act, slope, transposed)
        ctx.has_bias, ctx.has_res = bias is not None, res is not None
        ctx.w_sink, ctx.b_sink = _sink(w), _sink(bias)
        ctx.w_tmp = w.requires_grad and not w.is_leaf  # its gradient is an intermediate of this backward pass
        ctx.b0 = _GRAD_B0[0] if not (w.requires_grad or (bias is not None and bias.requires_grad)) else 0
        # (fmap_tap) this node's backward starts with a plain activation-derivative pass over its output gradient: a second
        # gradient of the output (the feature-matching loss's) can be summed inside that pass -- _TapFn leaves it in tap_add
        ctx.tap_ok = out_act != ACT_NONE and not (bias is not None and bias.requires_grad and not ctx.bt)
        ctx.tap_add = None
        ctx.save_for_backward(x, w, y if out_act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, pad, dil, groups, in_leaky, out_act, slope, transposed = ctx.cfg
        x, w, y = ctx.saved_tensors
        dy = _f32c(dy)
        # residual-gradient link (ResGradLink): the conv whose INPUT is another conv's residual adds that conv's
        # residual gradient inside its own data-gradient launch instead of leaving the sum to autograd
        link_dres = None
        if ctx.link is not None and ctx.link[1] == "dst":
            link_dres, ctx.link[0].dres = ctx.link[0].dres, None
        dtf = _ACT_TO_DTF[out_act]
        if ctx.tap_add is not None and (dtf == TF_NONE or (ctx.has_bias and ctx.needs_input_grad[2] and not ctx.bt)):
            tap, ctx.tap_add = ctx.tap_add, None  # (not the plain activation-derivative branch after all: add it here)
            dy[tap[1]:].add_(tap[0])
        dx = dw = db = dres = None
        db_done = None  # the bias gradient, once some launch has produced it
        w3 = w.view(w.shape[0], w.shape[1], w.shape[2])
        if dtf != TF_NONE:
            # apply the activation-derivative mask once; dgrad / wgrad / bias-grad then stream dye
            b0 = ctx.b0 if (0 < ctx.b0 < x.shape[0] and not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])) else 0
            dye = torch.empty_like(dy)
            if ctx.has_bias and ctx.needs_input_grad[2] and not ctx.bt:
                # the same pass collects the bias gradient (sum of the masked gradient per channel)
                db_done = ctx.b_sink[0] if ctx.b_sink is not None else torch.zeros((dy.shape[1],), device=dy.device,
                                                                                    dtype=torch.float32)
                check(lib().vcv_act_grad_bias(ptr(dy), ptr(y), ptr(dye), ptr(db_done), dy.shape[0], dy.shape[1],
                                              dy.numel() // (dy.shape[0] * dy.shape[1]), dtf, slope, stream()),
                      "vcv_act_grad_bias")
            else:
                tap, ctx.tap_add = ctx.tap_add, None
                tb0 = tap[1] if tap is not None else dy.shape[0]
                if b0 < tb0:
                    check(lib().vcv_act_grad(ptr(dy[b0:tb0]), ptr(y[b0:tb0]), ptr(dye[b0:tb0]), dtf, slope, dy[b0:tb0].numel(),
                                             stream()), "vcv_act_grad")
                if tap is not None:  # (the recorded half's second gradient, summed in the same pass)
                    lo = max(b0, tb0)
                    check(lib().vcv_act_grad_add(ptr(dy[lo:]), ptr(tap[0][lo - tb0:]), ptr(y[lo:]), ptr(dye[lo:]), dtf, slope,
                                                 dy[lo:].numel(), stream()), "vcv_act_grad_add")
            dy, y, dtf = dye, None, TF_NONE
        if ctx.bt:
            # x was saved in the folded layout; fold dy the same way, unfold dx
            dyt = _to_bt(dy)
            if ctx.needs_input_grad[0]:
                kw = dict(in_tf=TF_NONE, slope=slope)
                if in_leaky:
                    kw.update(out_tf=TF_DLEAKY, oaux=x)
                dx = _from_bt(conv_dgrad(dyt, w3, x.shape, stride=stride, pad=pad, dil=dil, groups=groups, **kw))
                if link_dres is not None:
                    dx.add_(link_dres)
            if ctx.needs_input_grad[1]:
                wout = ctx.w_sink[0].view(w3.shape) if ctx.w_sink is not None else None
                # rows of >= 64 frames fill the weight-gradient kernel's 64-position stages on their own: the
                # unfolded layout is faster there (the fold only pays for the forward / data-gradient tiles)
                wa, wb = (dy, _from_bt(x)) if dy.shape[2] >= 64 else (dyt, x)
                dw = conv_wgrad(wa, wb, w3.shape, stride=stride, pad=pad, dil=dil, groups=groups,
                                b_tf=TF_LEAKY if in_leaky else TF_NONE, slope=slope, out=wout,
                                arena=ctx.w_tmp).view(w.shape)
                dw = _sunk(ctx.w_sink, dw)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = _sunk(ctx.b_sink, bias_grad(dy, slope=slope, out=ctx.b_sink[0] if ctx.b_sink is not None else None))
            return dx, dw, db, None, None, None, None, None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            b0 = ctx.b0 if 0 < ctx.b0 < x.shape[0] else 0
            dys, ys, xs = (dy[b0:], (y[b0:] if y is not None else None), x[b0:]) if b0 else (dy, y, x)
            kw = dict(in_tf=dtf, xaux=ys, slope=slope)
            if in_leaky:
                kw.update(out_tf=TF_DLEAKY, oaux=xs)
            if link_dres is not None and not transposed:
                kw["res"] = link_dres[b0:] if b0 else link_dres
                link_dres = None
            dx = torch.empty_like(x)
            dxs = dx[b0:] if b0 else dx
            if transposed and xs.dim() == 3 and xs.shape[2] <= 64 and xs.shape[0] > 1 and kw.get("xaux") is None:
                # short input (first generator stage: 32 frames): fold the batch into the column dimension as the
                # forward convs of short sequences do, so the strided conv this gradient is has tiles to fill
                kwb = dict(kw)
                if kwb.get("oaux") is not None:
                    kwb["oaux"] = _to_bt(kwb["oaux"])
                dxb = convT_dgrad(_to_bt(dys), w3, (1, xs.shape[1], xs.shape[2], xs.shape[0]), stride=stride, pad=pad, **kwb)
                dxs.copy_(_from_bt(dxb))
            elif transposed:
                convT_dgrad(dys, w3, xs.shape, stride=stride, pad=pad, out=dxs, **kw)
            else:
                conv_dgrad(dys, w3, xs.shape, stride=stride, pad=pad, dil=dil, groups=groups, out=dxs, **kw)
        if ctx.needs_input_grad[1]:
            b_tf = TF_LEAKY if in_leaky else TF_NONE
            wout = ctx.w_sink[0].view(w3.shape) if ctx.w_sink is not None else None
            if transposed:
                dw = convT_wgrad(dy, x, w3.shape, stride=stride, pad=pad, a_tf=b_tf, b_tf=dtf,
                                 baux=y, slope=slope, out=wout, arena=ctx.w_tmp)
            else:
                if ctx.has_bias and ctx.needs_input_grad[2] and db_done is None:
                    # the weight-gradient launch collects sum(dy) while it stages dy
                    db_done = ctx.b_sink[0] if ctx.b_sink is not None else torch.zeros((dy.shape[1],), device=dy.device,
                                                                                        dtype=torch.float32)
                    dw = conv_wgrad(dy, x, w3.shape, stride=stride, pad=pad, dil=dil, groups=groups, a_tf=dtf, aaux=y,
                                    b_tf=b_tf, slope=slope, out=wout, arena=ctx.w_tmp, dbias=db_done)
                else:
                    dw = conv_wgrad(dy, x, w3.shape, stride=stride, pad=pad, dil=dil, groups=groups,
                                    a_tf=dtf, aaux=y, b_tf=b_tf, slope=slope, out=wout, arena=ctx.w_tmp)
            dw = _sunk(ctx.w_sink, dw.view(w.shape))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if db_done is not None:
                db = _sunk(ctx.b_sink, db_done)
            else:
                db = _sunk(ctx.b_sink, bias_grad(dy, aux=y, tf=dtf, slope=slope,
                                                 out=ctx.b_sink[0] if ctx.b_sink is not None else None))
        if link_dres is not None:  # not consumed by a fused launch above (no data gradient wanted / transposed)
            dx = link_dres if dx is None else dx.add_(link_dres)
        if ctx.has_res and ctx.needs_input_grad[3]:
            if ctx.link is not None and ctx.link[1] == "src":
                # handed to the linked conv's data gradient: that node consumes this conv's output, so whenever the
                # gradient of x is computed at all it runs later in this same backward pass and takes the hand-off
                # (a link object lives for one forward, so a hand-off nobody collects dies with the graph)
                if ctx.link[0].dres is not None:
                    raise RuntimeError("ResGradLink: a residual gradient of an earlier backward pass was never "
                                       "consumed (backward through the same graph twice?)")
                ctx.link[0].dres = dy
            else:
                dres = dy
        return dx, dw, db, dres, None, None, None, None, None, None, None, None, None


class _LinearT1Fn(torch.autograd.Function):
    """Pointwise conv on ONE frame: y[b, m, 0] = bias[m] + sum_c w[m, c, 0] x[b, c, 0] (the speaker-conditioning
    layers: a [M, C] matrix against <= 32 vectors -- matrix-vector kernels, not a GEMM tile)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x, w, bias = _f32c(x), _f32c(w), _f32c(bias)
        B, C, M = x.shape[0], x.shape[1], w.shape[0]
        y = torch.empty((B, M, 1), device=x.device, dtype=torch.float32)
        check(lib().vcv_linear_t1_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), B, C, M, stream()), "vcv_linear_t1_fwd")
        ctx.has_bias = bias is not None
        ctx.w_sink, ctx.b_sink = _sink(w), _sink(bias)
        ctx.w_tmp = w.requires_grad and not w.is_leaf
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _f32c(dy)
        B, C, M = x.shape[0], x.shape[1], w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib().vcv_linear_t1_dgrad(ptr(dy), ptr(w), ptr(dx), B, C, M, stream()), "vcv_linear_t1_dgrad")
        if ctx.needs_input_grad[1]:
            dw = ctx.w_sink[0].view(w.shape) if ctx.w_sink is not None else _wgrad_zeros(w.shape, dy.device, ctx.w_tmp)
            check(lib().vcv_linear_t1_wgrad(ptr(dy), ptr(x), ptr(dw), B, C, M, stream()), "vcv_linear_t1_wgrad")
            dw = _sunk(ctx.w_sink, dw)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _sunk(ctx.b_sink, bias_grad(dy, out=ctx.b_sink[0] if ctx.b_sink is not None else None))
        return dx, dw, db


class ResGradLink:
    """Shared by two conv1d calls of one residual pair y = c2(f(c1(x))) + x: pass link=(obj, "dst") to c1 (whose
    input is x) and link=(obj, "src") to c2 (whose `res` is the same x).  In backward c2 hands its residual gradient
    to c1, which adds it in its data-gradient kernel's epilogue; autograd then sees one gradient for x."""
    __slots__ = ("dres",)

    def __init__(self):
        self.dres = None


def conv1d(x, w, bias=None, stride=1, pad=0, dil=1, groups=1, in_leaky=False, out_act=ACT_NONE,
           slope=0.1, res=None, link=None):
    """Conv1d on [B,C,T] or the (k,1) Conv2d of the period discriminators on [B,C,H,P]."""
    if (x.dim() == 3 and x.shape[2] == 1 and w.dim() == 3 and w.shape[2] == 1 and groups == 1 and stride == 1 and pad == 0
            and not in_leaky and out_act == ACT_NONE and res is None and x.shape[0] <= 32 and w.shape[0] >= 32):
        return _LinearT1Fn.apply(x, w, bias)
    return _ConvFn.apply(x, w, bias, res, stride, pad, dil, groups, in_leaky, out_act, slope, False, link)


def conv_transpose1d(x, w, bias=None, stride=1, pad=0, in_leaky=False, out_act=ACT_NONE, slope=0.1):
    return _ConvFn.apply(x, w, bias, None, stride, pad, 1, 1, in_leaky, out_act, slope, True, None)
