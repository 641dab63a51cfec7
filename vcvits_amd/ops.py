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


_USE_DMA = [__import__("os").environ.get("VCVITS_CONV_DMA", "1") == "1"]
# fp32 launches try the channel-innermost packed kernel (vcv_conv_pk_*) before the LDS-DMA kernel
_USE_PK = [__import__("os").environ.get("VCVITS_CONV_PK", "1") == "1"]

# fp32 launches try the split-operand kernel first (vcv_conv_x3_*: fp32 operands as three exact bf16 terms each, nine -- or
# six -- bf16 MFMA products per fp32 product, fp32 accumulate: fp32 results at 1.8-2.7 x the fp32 MFMA peak)
_USE_X3 = [__import__("os").environ.get("VCVITS_CONV_X3", "1") == "1"]


# the weight gradient in the same arithmetic (vcv_wgrad_x3: wgrad_bf16.hip with three term planes and producer waves);
# the library takes the shapes where it beats the fp32 kernel (wgrad_dma.hip) and declines the rest
_USE_X3_WGRAD = [__import__("os").environ.get("VCVITS_WGRAD_X3", "1") == "1"]


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
_BF16_ACT = [__import__("os").environ.get("VCVITS_BF16_ACT", "1") == "1"]
# bf16 mode: DiscriminatorS's grouped k = 41 forward on the bf16 matrix pipe (VCVITS_GROUPED_BF16=0: fp32-input MFMA)
_GROUPED_BF16 = [__import__("os").environ.get("VCVITS_GROUPED_BF16", "1") == "1"]


def set_bf16_activations(on):
    _BF16_ACT[0] = bool(on)


# CAPTURING[0] is the _lib.Capture of the launch sequence being recorded into a HIP graph (vcvits_amd/light/graphed.py), None
# in eager execution.  While it is set: weights derived from parameters are made INSIDE the sequence (caches filled by eager
# passes are not consulted; entries made by this capture are, so the discriminators' weights are normalised and packed once
# per recorded batch as in the eager loop), device tables are filled once after the capture instead of by recorded copy
# nodes, and every tensor from outside the graph's pool that a launcher is handed is held by the graph (_lib.Capture).
CAPTURING = _lib.CAPTURE


_DBG_TABLE_NODES = __import__("os").environ.get("VCVITS_DBG_TABLE_NODES", "0") == "1"
_DBG_NO_LOCAL_CACHE = __import__("os").environ.get("VCVITS_DBG_NO_LOCAL_CACHE", "0") == "1"


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
    out = None if _DBG_TABLE_NODES else cap.table(tab, torch.int64, tab.shape)
    if out is None:  # (table arena full, or the A/B switch: round 4's form -- a recorded copy node re-reading the host array)
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


from ._lib import _GET_DEVICE  # noqa: E402

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


# Combine of the split weight-gradient reductions: True = per-workgroup slabs added in a fixed order (bit-reproducible),
# False (default: ~2 % faster per step) = fp32 atomics (order varies from run to run).  The bf16 kernel always uses slabs.
_DETERMINISTIC = [__import__("os").environ.get("VCVITS_DETERMINISTIC", "0") == "1"]


def set_deterministic(on):
    """Bit-reproducible gradients run to run: the MFMA weight-gradient kernels combine their split reductions through
    slabs added in a fixed order, bias gradients are summed by one workgroup per channel instead of inside the
    weight-gradient launch, and the library's other split reductions (thin / grouped / register-staged weight gradients,
    activation-derivative bias sums, the one-output-channel forward) run unsplit (vcv_set_deterministic).  Covers the
    GAN step of the vocoder workload (tests/test_determinism_gpu.py: two identical steps, gradients bit for bit); the
    full model's LayerNorm-parameter and relative-position-table gradients still meet in fp32 atomics."""
    _DETERMINISTIC[0] = bool(on)
    check(lib().vcv_set_deterministic(1 if on else 0), "vcv_set_deterministic")


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
        # on the generic kernel); the derivative of a leaky-ReLU on the conv's input is one more streaming pass in place
        wf = w.reshape(C, K).flip(-1).contiguous()
        masked = kw.get("out_tf", TF_NONE) == TF_DLEAKY
        raw = torch.empty_like(out) if masked else out
        check(lib().vcv_conv_c1_fwd(ptr(dy), ptr(wf), None, ptr(raw), B, C, Tout, Tin, P, K, 1, dil, (K - 1) * dil - pad,
                                    ACT_NONE, kw.get("slope", 0.1), stream()), "vcv_conv_c1_fwd")
        if masked:
            check(lib().vcv_act_grad(ptr(raw), ptr(kw["oaux"]), ptr(out), TF_DLEAKY, kw.get("slope", 0.1), out.numel(),
                                     stream()), "vcv_act_grad")
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


_CONVT_MERGED = [__import__("os").environ.get("VCVITS_CONVT_MERGED", "1") == "1"]  # (0: one launch phase per output residue)


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
_PAIR_FUSED = [__import__("os").environ.get("VCVITS_PAIR_FUSED", "1") == "1"]


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
    from ._lib import VcvResPairArgs
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


_TAP_FUSE = [__import__("os").environ.get("VCVITS_TAP_FUSE", "1") == "1"]  # (A/B switch)


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


class _SpectralNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, u, v, power_iteration, eps):
        w = _f32c(w)
        if not (u.is_contiguous() and v.is_contiguous() and u.dtype == torch.float32 and v.dtype == torch.float32):
            raise RuntimeError("spectral_norm: weight_u / weight_v must be contiguous fp32 buffers (updated in place)")
        R = w.shape[0]
        N = w.numel() // R
        if u.numel() != R or v.numel() != N:
            raise RuntimeError("spectral_norm: weight_u / weight_v do not match the weight's [%d, %d] matrix" % (R, N))
        w_sn = torch.empty_like(w)
        sigma = torch.empty((1,), device=w.device, dtype=torch.float32)
        work = torch.empty((R + N,), device=w.device, dtype=torch.float32)
        check(lib().vcv_spectral_norm_fwd(ptr(w), ptr(u), ptr(v), ptr(w_sn), ptr(sigma), ptr(work), R, N,
                                          1 if power_iteration else 0, eps, stream()), "vcv_spectral_norm_fwd")
        # the vectors sigma was formed from: the next training forward overwrites the buffers (torch clones them too)
        ctx.save_for_backward(w_sn, u.clone() if power_iteration else u, v.clone() if power_iteration else v, sigma)
        return w_sn

    @staticmethod
    def backward(ctx, dw_sn):
        w_sn, u, v, sigma = ctx.saved_tensors
        dw_sn = _f32c(dw_sn)
        R = w_sn.shape[0]
        N = w_sn.numel() // R
        dw = torch.empty_like(w_sn)
        work = torch.empty((256,), device=w_sn.device, dtype=torch.float32)
        check(lib().vcv_spectral_norm_bwd(ptr(dw_sn), ptr(w_sn), ptr(u), ptr(v), ptr(sigma), ptr(dw), ptr(work), R, N,
                                          stream()), "vcv_spectral_norm_bwd")
        return dw, None, None, None, None


def spectral_norm(w, u, v, power_iteration, eps=1e-12):
    """w / sigma, sigma = u . (W v) over the [out_channels, rest] matrix of w (torch.nn.utils.spectral_norm, dim 0, one
    power iteration; reference: discriminator.py:17,52 under use_spectral_norm=True).  With `power_iteration` (a training
    forward) the buffers u, v are advanced IN PLACE first, as torch's forward pre-hook does."""
    return _SpectralNormFn.apply(w, u, v, bool(power_iteration), float(eps))


_WN_TABLES = {}
# Cached results of _WeightNormManyFn per parameter set: {key: dict(versions, wbuf, norm, lo, hi, packs)}.  An
# entry is valid until one of its parameters changes: in place through torch (version counters) or through an
# optimizer's raw-pointer update (invalidate_weights).  The discriminators' weights are identical in the
# generator step and the discriminator step of a batch, and inference never changes them.
_WN_CACHE = {}
_WN_CACHE_ON = [__import__("os").environ.get("VCVITS_WEIGHT_CACHE", "1") == "1"]


# bumped by every raw write into parameter storage: weights handed to layers before it (modules._w_pre / _w_lazy) are
# stale afterwards even though no torch version counter moved
WEIGHT_EPOCH = [0]


def invalidate_weights(lo=None, hi=None):
    """Parameters stored in [lo, hi) (all parameters when None) were modified behind torch's back."""
    WEIGHT_EPOCH[0] += 1
    cap = CAPTURING[0]
    for e in list(_PARAM_REGIONS.values()) + (list(cap.__dict__.get("regions", {}).values()) if cap is not None else []):
        if lo is None or (lo < e["hi"] and e["lo"] < hi):
            e["dirty"] = True
    if lo is None:
        _WN_CACHE.clear()
        return
    for k in [k for k in _WN_CACHE if any(lo <= p < hi for p in k)]:
        del _WN_CACHE[k]


# Parameter regions: an optimizer that keeps its parameters in one flat buffer registers it (register_param_region).  Conv
# weights that are used as they are (no weight norm: the encoders' attention / FFN / projection layers) then get the same
# treatment as the weight-normed trees: their packed copies are cached until the region is written (invalidate_weights)
# and re-made in ONE batched launch at the first use afterwards, instead of one pack launch per layer and use (110 launches of
# ~8 us per bf16-mode step of the full model).
# bumped whenever parameter STORAGE may have moved (an optimizer re-seating parameters into a new flat buffer, a module
# replacing a layer): launch sequences recorded into HIP graphs bake parameter addresses and key on this counter
GRAPH_EPOCH = [0]

_PARAM_REGIONS = {}
_PARAM_REGIONS_ON = [__import__("os").environ.get("VCVITS_PARAM_REGIONS", "1") == "1"]


def register_param_region(flat):
    GRAPH_EPOCH[0] += 1
    lo = flat.data_ptr()
    hi = lo + 4 * flat.numel()
    _PARAM_REGIONS[lo] = dict(lo=lo, hi=hi, packs={}, key=("region", lo, hi), shapes=(int(flat.numel()),), wbuf=flat, dirty=True)


def unregister_param_region(flat):
    GRAPH_EPOCH[0] += 1
    _PARAM_REGIONS.pop(flat.data_ptr(), None)


def _stable_entry(w_ptr):
    """The cached weight-norm buffer (its cache entry) -- or the registered parameter region -- that contains address w_ptr,
    if any.  Inside a capture only entries made by that capture count (and in eager execution only eager ones): a recorded
    sequence must contain the launches that make the weights and packs it reads."""
    if w_ptr is None:
        return None
    cap = CAPTURING[0]
    capid = cap.id if cap is not None else None
    for e in _WN_CACHE.values():
        if e["lo"] <= w_ptr < e["hi"] and e.get("cap") == capid:
            return e
    if _PARAM_REGIONS_ON[0]:
        regions = _PARAM_REGIONS
        if cap is not None and _DBG_NO_LOCAL_CACHE:
            return None
        if cap is not None:
            regions = cap.__dict__.get("regions")
            if regions is None:  # the capture's own view of the regions: nothing packed yet
                regions = cap.regions = {lo: dict(lo=e["lo"], hi=e["hi"], packs={}, key=e["key"], shapes=e["shapes"],
                                                  wbuf=e["wbuf"], dirty=True) for lo, e in _PARAM_REGIONS.items()}
        for e in regions.values():
            if e["lo"] <= w_ptr < e["hi"]:
                if e["dirty"]:
                    e["dirty"] = False
                    e["packs"].clear()
                    _replay_packs(e["key"], e)
                return e
    return None


def _stable_packs(w_ptr):
    """The pack cache of the cached weight-norm buffer that contains address w_ptr, if any."""
    e = _stable_entry(w_ptr)
    return e["packs"] if e is not None else None


# Packed-weight jobs per parameter set: {wn key: {(offset of w in the buffer, pack words, layout signature, family): (launch
# arguments, flip)}} -- recorded when a launch had to pack (_launch_conv), replayed in ONE launch when the set is
# re-normalised (vcv_pack_many): 180-270 pack launches per step otherwise.
_PACK_JOBS = {}
_PACK_BATCH = [__import__("os").environ.get("VCVITS_PACK_BATCH", "1") == "1"]
_PACK_FILL = {"vcv_conv_x3_run": "vcv_conv_x3_pack_job", "vcv_conv_pk_run": "vcv_conv_pk_pack_job",
              "vcv_conv_bf16_run": "vcv_conv_bf16_pack_job"}


def _replay_packs(key, ent):
    """Make every recorded pack of parameter set `key` for its freshly normalised weights `ent` (one launch)."""
    rec = _PACK_JOBS.get(key)
    if not rec or not _PACK_BATCH[0]:
        return
    if rec["shapes"] != ent["shapes"]:
        # the same addresses now hold another module's parameters: its jobs would read outside the new buffer
        del _PACK_JOBS[key]
        return
    from ._lib import VcvPackJob
    L = lib()
    # only the families the current switches can launch (a job of another arithmetic would be packed for nothing)
    live = {"vcv_conv_pk_run"} if _USE_PK[0] else set()
    if _COMPUTE[0] == "bf16":
        live.add("vcv_conv_bf16_run")
    elif _USE_X3[0]:
        live.add("vcv_conv_x3_run")
    span = ent["hi"] - ent["lo"]
    todo = [(k, v) for k, v in rec["jobs"].items() if k[3] in _PACK_FILL and k[3] in live]
    if not todo:
        return
    arr = (VcvPackJob * len(todo))()
    total = sum(k[1] for k, _ in todo)
    dev = ent["wbuf"].device
    arena = torch.empty((total + 32 * len(todo),), device=dev, dtype=torch.float32)
    n = off = 0
    reg = []
    for (woff, words, sig, fam), job in todo:
        abytes, flip = job[0], job[1]
        wver = job[2] if len(job) > 2 else 0
        a = VcvConvArgs.from_buffer_copy(abytes)
        if woff < 0 or woff + 4 * a.Mg * a.Cg * a.K > span:
            continue
        a.w = ent["lo"] + woff
        if getattr(L, _PACK_FILL[fam])(ctypes.byref(a), flip, ctypes.byref(arr[n])) != 0:
            continue  # (the plan no longer takes this launch, e.g. a mode switch: it will pack lazily)
        view = arena[off:off + words]
        arr[n].w, arr[n].wp = a.w, view.data_ptr()
        reg.append(((a.w, words, sig) if wver == 0 else (a.w, words, sig, wver), view))
        off += (words + 31) & ~31  # 128-byte aligned slices
        n += 1
    if n == 0:
        return
    nwords = n * ctypes.sizeof(VcvPackJob) // 4 + 8
    cap = CAPTURING[0]
    table = None
    if cap is not None and not _DBG_TABLE_NODES:
        # recorded: the job table is finalised on the host by the call below and copied into `table` (cut from the capture's
        # table arena, outside the graph's pool) once after the capture; the packs themselves are re-made at every replay
        host = (ctypes.c_char * (4 * nwords))()
        table = cap.table(host, torch.float32, (nwords,))
    if table is not None:
        check(L.vcv_pack_many_prepared(arr, n, ptr(table), stream()), "vcv_pack_many_prepared")
        ctypes.memmove(host, arr, n * ctypes.sizeof(VcvPackJob))
    else:
        table = torch.empty((nwords,), device=dev, dtype=torch.float32)
        check(L.vcv_pack_many(arr, n, ptr(table), stream()), "vcv_pack_many")
        if cap is not None:
            cap.extend((arr, table))  # the recorded upload re-reads `arr` at every replay
    ent["pack_table"] = table  # (kept alive with the entry)
    for k, view in reg:
        ent["packs"][k] = view
    LAUNCH_COUNTS["pack_many"] = LAUNCH_COUNTS.get("pack_many", 0) + 1


class _WnHolder:
    """Result of one batched weight-norm forward launch: the buffer all effective weights live in, the row norms and the
    host copy of the launch table (one row per (v, g) pair: v, g, w offset, first row, rows, row length, ...)."""
    __slots__ = ("wbuf", "norm", "tab", "total", "rows")


def _wn_forward_all(vg, n):
    """Batched forward (no autograd), cached per parameter set until a parameter changes."""
    vs, gs = vg[:n], vg[n:]
    for t in vg:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("weight_norm_many: parameters must be contiguous fp32")
    dev = vs[0].device
    key = tuple(t.data_ptr() for t in vg)
    ent = _WN_TABLES.get(key)
    if ent is None:
        import numpy as np
        tab = np.zeros((n, 10), dtype=np.int64)
        woff = row0 = 0
        for i, (v, g) in enumerate(zip(vs, gs)):
            R = v.shape[0]
            C = v.numel() // R
            tab[i, :6] = (v.data_ptr(), g.data_ptr(), woff, row0, R, C)
            woff += R * C
            row0 += R
        ent = (tab, torch.from_numpy(tab).to(dev), woff, row0)
        if len(_WN_TABLES) > 256:
            _WN_TABLES.clear()
        _WN_TABLES[key] = ent
    tab, tab_dev, total, rows = ent
    versions = tuple(t._version for t in vg)
    cap = CAPTURING[0]
    hit = _WN_CACHE.get(key) if _WN_CACHE_ON[0] else None
    if hit is not None and (hit.get("cap") != (cap.id if cap is not None else None) or (cap is not None and _DBG_NO_LOCAL_CACHE)):
        hit = None  # (an eager pass's entry inside a capture, or a capture's entry in eager execution: not this sequence's)
    if hit is not None and hit["versions"] == versions and all(r() is t for r, t in zip(hit["refs"], vg)):
        wbuf, norm = hit["wbuf"], hit["norm"]  # (identity: a recycled address is not the same parameter)
    else:
        wbuf = torch.empty((total,), device=dev, dtype=torch.float32)
        norm = torch.empty((rows,), device=dev, dtype=torch.float32)
        check(lib().vcv_weight_norm_many_fwd(ptr(tab_dev), n, rows, ptr(wbuf), ptr(norm), stream()),
              "vcv_weight_norm_many_fwd")
        if _WN_CACHE_ON[0]:
            if len(_WN_CACHE) > 64:
                _WN_CACHE.clear()
            import weakref
            ent = dict(versions=versions, refs=tuple(weakref.ref(t) for t in vg), wbuf=wbuf, norm=norm,
                       lo=wbuf.data_ptr(), hi=wbuf.data_ptr() + 4 * total, packs={}, key=key,
                       shapes=tuple(tuple(t.shape) for t in vg), cap=cap.id if cap is not None else None)
            _WN_CACHE[key] = ent
            _replay_packs(key, ent)
    h = _WnHolder()
    h.wbuf, h.norm, h.tab, h.total, h.rows = wbuf, norm, tab, total, rows
    return h


class _WeightNormManyFn(torch.autograd.Function):
    """Autograd node of layers [i0, i1) of one batched weight-norm launch.  The forward launch covers the whole module
    tree (`holder`); the BACKWARD is one launch per node, so a tree split into several nodes (one per
    sub-discriminator / generator block) hands its parameter gradients to the optimizer -- and its gradient buckets to
    the all-reduce -- as soon as that part of the backward pass is done, not at the very end."""

    @staticmethod
    def forward(ctx, holder, i0, i1, *vg):
        n = i1 - i0
        vs, gs = vg[:n], vg[n:]
        tab = holder.tab
        ctx.holder, ctx.i0, ctx.i1 = holder, i0, i1
        ctx.shapes = [(v.shape, g.shape) for v, g in zip(vs, gs)]
        ctx.sinks = [(_sink(v), _sink(g)) for v, g in zip(vs, gs)]
        ctx.save_for_backward(*vg)  # keeps v / g alive; the table holds their addresses
        wbuf = holder.wbuf
        return tuple(wbuf[int(tab[i0 + i, 2]):int(tab[i0 + i, 2]) + vs[i].numel()].view(vs[i].shape) for i in range(n))

    @staticmethod
    def backward(ctx, *dws):
        holder, i0, i1 = ctx.holder, ctx.i0, ctx.i1
        n, dev = i1 - i0, holder.norm.device
        dws = [_f32c(d) for d in dws]
        tab = holder.tab[i0:i1].copy()
        first_row = int(tab[0, 3])
        rows = int(tab[-1, 3] + tab[-1, 4]) - first_row
        tab[:, 3] -= first_row  # the launch covers this node's rows only
        loose = [i for i in range(n) if ctx.sinks[i][0] is None or ctx.sinks[i][1] is None]
        dvbuf = dg = None
        if loose:
            dvbuf = torch.empty((sum(int(tab[i, 4] * tab[i, 5]) for i in loose),), device=dev, dtype=torch.float32)
            dg = torch.empty((sum(int(tab[i, 4]) for i in loose),), device=dev, dtype=torch.float32)
        dvs, dgs = [None] * n, [None] * n
        o = r0 = 0
        for i, d in enumerate(dws):
            R, C = int(tab[i, 4]), int(tab[i, 5])
            tab[i, 6] = d.data_ptr()
            sv, sg = ctx.sinks[i]
            if sv is not None and sg is not None:
                tab[i, 7], tab[i, 8], tab[i, 9] = sv[0].data_ptr(), sg[0].data_ptr(), 1
            else:
                vsh, gsh = ctx.shapes[i]
                dvs[i], dgs[i] = dvbuf[o:o + R * C].view(vsh), dg[r0:r0 + R].view(gsh)
                tab[i, 7], tab[i, 8], tab[i, 9] = dvs[i].data_ptr(), dgs[i].data_ptr(), 0
                o += R * C
                r0 += R
        tab_dev = _upload_table(tab, dev)
        check(lib().vcv_weight_norm_many_bwd(ptr(tab_dev), n, rows, ptr(holder.norm[first_row:first_row + rows]), stream()),
              "vcv_weight_norm_many_bwd")
        for sv, sg in ctx.sinks:
            if sv is not None and sg is not None:
                for e in (sv, sg):
                    if e[1] is not None:
                        e[1]()
        return (None, None, None) + tuple(dvs) + tuple(dgs)


def weight_norm_forward(vs, gs):
    """The batched forward launch alone (cached); autograd nodes are attached later with weight_norm_group."""
    return _wn_forward_all(tuple(vs) + tuple(gs), len(vs))


def weight_norm_group(holder, i0, i1, vs, gs):
    """Effective weights of layers [i0, i1) of a weight_norm_forward result, as ONE autograd node created NOW: a node
    created when its sub-block's forward starts sits right behind that block's conv nodes in autograd's (reverse
    creation order) schedule, so its backward -- and the parameter gradients it finalises -- run as soon as the block's
    backward is done, not after every other block's."""
    return _WeightNormManyFn.apply(holder, i0, i1, *vs, *gs)


def weight_norm_many(vs, gs, group_sizes=None):
    """[weight_norm(v, g) for v, g in zip(vs, gs)]: ONE forward launch; one autograd node (= one backward launch) per
    consecutive group of `group_sizes` layers (default: a single node)."""
    n = len(vs)
    holder = _wn_forward_all(tuple(vs) + tuple(gs), n)
    out = []
    i0 = 0
    for k in (group_sizes or [n]):
        out.extend(_WeightNormManyFn.apply(holder, i0, i0 + k, *vs[i0:i0 + k], *gs[i0:i0 + k]))
        i0 += k
    assert i0 == n
    return out


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
        if ctx.bt:
            B, C, T = ctx.xshape
            dx = _from_bt(conv_dgrad(_to_bt(dy), w, (1, C, T, B), in_tf=_lib.TF_DLOGCLAMP, xaux=yf, slope=ctx.clamp))
        else:
            dx = conv_dgrad(dy, w, ctx.xshape, in_tf=_lib.TF_DLOGCLAMP, xaux=yf, slope=ctx.clamp)
        return dx, None, None


def mel_log(spec, melmat, clamp=1e-5):
    return _MelLogFn.apply(spec, melmat, clamp)


# ---------------------------------------------------------------------------------------------
# AdamW on flat buffers
# ---------------------------------------------------------------------------------------------
def adamw_step(p, g, m, v, lr, betas, eps, weight_decay, step):
    check(lib().vcv_adamw(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps,
                          weight_decay, step, stream()), "vcv_adamw")


def adamw_step_dev(p, g, m, v, betas, eps, weight_decay, hyper, step_base):
    """The same step with lr and the step delta read from the device record `hyper` (int32[2]: fp32 bits of lr, delta):
    the form an optimizer step recorded into a HIP graph takes (light/graphed.py refreshes the record before each replay)."""
    check(lib().vcv_adamw_dev(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), betas[0], betas[1], eps, weight_decay,
                              ptr(hyper), int(step_base), stream()), "vcv_adamw_dev")


def set_hyper(hyper, lr, delta):
    import struct
    check(lib().vcv_set_words(ptr(hyper), 2, struct.unpack("<i", struct.pack("<f", float(lr)))[0], int(delta), 0, 0, stream()),
          "vcv_set_words")


# ---------------------------------------------------------------------------------------------
# WaveNet block glue
# ---------------------------------------------------------------------------------------------
class GateGradShare:
    """Shared by the `n` wn_gate calls of ONE WaveNet stack forward: every layer's conditioning gradient is its own slice
    of d(g) [B, 2H*L, 1], so the layers write their slices into one buffer and the layer whose backward runs last hands it
    to autograd -- instead of L full-size tensors that are zero outside one slice and L - 1 accumulation adds."""
    __slots__ = ("n", "left", "buf")

    def __init__(self, n):
        self.n, self.left, self.buf = int(n), int(n), None


class _WnGateFn(torch.autograd.Function):
    """acts = tanh(xin[:, :H] + g_l[:H]) * sigmoid(xin[:, H:] + g_l[H:]);  g: [B, 2H*L, 1] or None."""

    @staticmethod
    def forward(ctx, xin, g, goff, share=None):
        xin, g = _f32c(xin), _f32c(g)
        B, H2, T = xin.shape
        H = H2 // 2
        gstride = g.shape[1] if g is not None else 0
        acts = torch.empty((B, H, T), device=xin.device, dtype=torch.float32)
        check(lib().vcv_wn_gate_fwd(ptr(xin), ptr(g), gstride, goff, ptr(acts), B, H, T, stream()),
              "vcv_wn_gate_fwd")
        ctx.goff, ctx.share = goff, share
        ctx.save_for_backward(xin, g)
        return acts

    @staticmethod
    def backward(ctx, dacts):
        xin, g = ctx.saved_tensors
        dacts = _f32c(dacts)
        B, H2, T = xin.shape
        H = H2 // 2
        gstride = g.shape[1] if g is not None else 0
        dxin = torch.empty_like(xin)
        check(lib().vcv_wn_gate_bwd(ptr(xin), ptr(g), gstride, ctx.goff, ptr(dacts), ptr(dxin), B, H, T,
                                    stream()), "vcv_wn_gate_bwd")
        dg = None
        if g is not None and ctx.needs_input_grad[1]:
            sh = ctx.share
            if sh is None:
                dg = torch.zeros_like(g)
            else:
                if sh.buf is None:
                    sh.buf, sh.left = torch.zeros_like(g), sh.n
                dg = sh.buf
            check(lib().vcv_row_sum(ptr(dxin), ptr(dg), B * H2, T, H2, gstride, ctx.goff, stream()),
                  "vcv_row_sum")
            if sh is not None:
                sh.left -= 1
                if sh.left > 0:
                    dg = None  # (a later layer's backward of this stack returns the buffer)
                else:
                    sh.buf = None
        return dxin, dg, None, None


def wn_gate(xin, g, goff, share=None):
    return _WnGateFn.apply(xin, g, goff, share)


class _WnResSkipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out, rs, mask, last, link=None):
        ctx.link = link
        x, out, rs, mask = _f32c(x), _f32c(out), _f32c(rs), _f32c(mask)
        B, H, T = x.shape
        on = torch.empty_like(x)
        xn = x if last else torch.empty_like(x)
        check(lib().vcv_wn_res_skip_fwd(ptr(x), ptr(out), ptr(rs), ptr(mask), ptr(xn), ptr(on), B, H, T,
                                        1 if last else 0, stream()), "vcv_wn_res_skip_fwd")
        ctx.last, ctx.has_out = last, out is not None
        ctx.save_for_backward(mask)
        if last:
            return on
        return xn, on

    @staticmethod
    def backward(ctx, *grads):
        (mask,) = ctx.saved_tensors
        if ctx.last:
            don = _f32c(grads[0])
            return None, (don if ctx.has_out else None), don, None, None, None
        dxn, don = _f32c(grads[0]), _f32c(grads[1])
        ref = dxn if dxn is not None else don
        B, H, T = ref.shape
        drs = torch.empty((B, 2 * H, T), device=ref.device, dtype=torch.float32)
        dx = torch.empty_like(ref)
        check(lib().vcv_wn_res_skip_bwd(ptr(dxn), ptr(don), ptr(mask), ptr(drs), ptr(dx), B, H, T, stream()),
              "vcv_wn_res_skip_bwd")
        if ctx.link is not None:
            # x's other consumer is this layer's dilated conv, whose backward runs later in this pass (it is upstream of
            # `rs`): its data-gradient launch adds dx in its epilogue (ResGradLink), autograd sees one gradient for x
            if ctx.link.dres is not None:
                raise RuntimeError("ResGradLink: a residual gradient of an earlier backward pass was never consumed")
            ctx.link.dres, dx = dx, None
        return dx, (don if ctx.has_out else None), drs, None, None, None


def wn_res_skip(x, out, rs, mask, last, link=None):
    """(x_new, out_new) for a middle layer, out_new for the last one (modules.py:168-174).  link: a ResGradLink shared
    with the conv1d call that also consumes x (link=(obj, "dst")): x's residual-path gradient is handed to that conv."""
    return _WnResSkipFn.apply(x, out, rs, mask, last, link)


class _SplitSampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, stats, eps, mask):
        stats, eps, mask = _f32c(stats), _f32c(eps), _f32c(mask)
        B, C2, T = stats.shape
        C = C2 // 2
        m = torch.empty((B, C, T), device=stats.device, dtype=torch.float32)
        logs = torch.empty_like(m)
        z = torch.empty_like(m) if eps is not None else None
        check(lib().vcv_split_sample_fwd(ptr(stats), ptr(eps), ptr(mask), ptr(m), ptr(logs), ptr(z), B, C, T,
                                         stream()), "vcv_split_sample_fwd")
        ctx.save_for_backward(eps, logs, mask)
        ctx.has_eps = eps is not None
        if eps is None:
            return m, logs
        return z, m, logs

    @staticmethod
    def backward(ctx, *grads):
        eps, logs, mask = ctx.saved_tensors
        if ctx.has_eps:
            dz, dm, dlogs = (_f32c(g) for g in grads)
        else:
            dz = None
            dm, dlogs = (_f32c(g) for g in grads)
        B, C, T = logs.shape
        dstats = torch.empty((B, 2 * C, T), device=logs.device, dtype=torch.float32)
        check(lib().vcv_split_sample_bwd(ptr(dm), ptr(dlogs), ptr(dz), ptr(eps), ptr(logs), ptr(mask),
                                         ptr(dstats), B, C, T, stream()), "vcv_split_sample_bwd")
        return dstats, None, None


def split_stats(stats, mask):
    """m, logs = split(stats * mask)."""
    return _SplitSampleFn.apply(stats, None, mask)


def posterior_sample(stats, eps, mask):
    """z, m, logs with z = (m + eps*exp(logs)) * mask."""
    return _SplitSampleFn.apply(stats, eps, mask)


class _CouplingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, m, mask, reverse):
        x1, m, mask = _f32c(x1), _f32c(m), _f32c(mask)
        B, C, T = x1.shape
        y = torch.empty_like(x1)
        check(lib().vcv_coupling(ptr(x1), ptr(m), ptr(mask), ptr(y), B, C, T, 1 if reverse else 0, stream()),
              "vcv_coupling")
        ctx.reverse = reverse
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _f32c(dy)
        dmasked = mask_mul(dy, mask.view(mask.shape[0], 1, -1))
        if ctx.reverse:
            return dmasked, scale(dmasked, -1.0), None, None
        return dmasked, dy, None, None


def coupling(x1, m, mask, reverse=False):
    return _CouplingFn.apply(x1, m, mask, reverse)


def prior_sample(m_p, logs_p, noise, noise_scale=1.0):
    """z_p = m_p + noise * exp(logs_p) * noise_scale (synthesizer_svc.py:104; inference only, no autograd)."""
    m_p, logs_p, noise = _f32c(m_p.detach()), _f32c(logs_p.detach()), _f32c(noise.detach())
    z = torch.empty_like(m_p)
    check(lib().vcv_prior_sample(ptr(m_p), ptr(logs_p), ptr(noise), ptr(z), m_p.numel(), float(noise_scale), stream()),
          "vcv_prior_sample")
    return z


class _LayerNormCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, gamma, beta, eps):
        x, y, gamma, beta = _f32c(x), _f32c(y), _f32c(gamma), _f32c(beta)
        B, C, T = x.shape
        out = torch.empty_like(x)
        mean = torch.empty((B, T), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        check(lib().vcv_layernorm_c_fwd(ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(out), ptr(mean), ptr(rstd),
                                        B, C, T, eps, stream()), "vcv_layernorm_c_fwd")
        ctx.save_for_backward(x, y, gamma, mean, rstd)
        ctx.has_y = y is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, gamma, mean, rstd = ctx.saved_tensors
        dout = _f32c(dout)
        B, C, T = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
        # (the partial sums of the register-resident form go to a workspace of this call: nothing shared between streams / devices)
        nws = lib().vcv_layernorm_c_bwd_scratch(B, C, T)
        ws = torch.empty((nws,), device=x.device, dtype=torch.float32) if nws > 0 else None
        check(lib().vcv_layernorm_c_bwd_ws(ptr(x), ptr(y), ptr(gamma), ptr(mean), ptr(rstd), ptr(dout), ptr(dx),
                                           ptr(dgamma), ptr(dbeta), B, C, T, ptr(ws), nws, stream()), "vcv_layernorm_c_bwd_ws")
        return dx, (dx if ctx.has_y else None), dgamma, dbeta, None


def layernorm_c(x, y, gamma, beta, eps=1e-5):
    """LayerNorm over channels of (x + y) for [B,C,T] tensors (y may be None)."""
    return _LayerNormCFn.apply(x, y, gamma, beta, eps)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _f32c(x)
        y = torch.empty_like(x)
        check(lib().vcv_dropout(ptr(x), ptr(y), x.numel(), p, seed, stream()), "vcv_dropout")
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        dx = torch.empty_like(dy)
        check(lib().vcv_dropout(ptr(dy), ptr(dx), dy.numel(), ctx.p, ctx.seed, stream()), "vcv_dropout")
        return dx, None, None


_seed_state = [0x1234ABCD]


def next_seed():
    """Host-side seed stream for the counter-based dropout masks (reseeded by manual_seed)."""
    _seed_state[0] = (_seed_state[0] * 6364136223846793005 + 1442695040888963407) % (1 << 64)
    return _seed_state[0]


def manual_seed(seed):
    """Reseed the dropout stream (callers: VCVITS.configure_optimizers seeds it from torch.initial_seed() + rank so
    data-parallel ranks draw different masks; checkpoints carry get_seed_state())."""
    _seed_state[0] = (int(seed) * 2654435761 + 0x9E3779B97F4A7C15) % (1 << 64)


def get_seed_state():
    return int(_seed_state[0])


def set_seed_state(state):
    _seed_state[0] = int(state) % (1 << 64)


# test hook: when a list, every dropout draw of a step is appended as (kind, p, seed, shape) -- the masks are functions
# of (seed, flat index), so a checker can regenerate them (tests/test_dropout_step_gpu.py)
DROPOUT_TRACE = [None]


def dropout_mask(shape, p, seed, device):
    """The mask / (1 - p) tensor a dropout draw (p, seed) applies to a tensor of `shape` (flat-index hash, the same for
    ops.dropout and for the attention probabilities [B*H, T, T])."""
    return _DropoutFn.apply(torch.ones(tuple(shape), device=device, dtype=torch.float32), float(p), int(seed))


def dropout(x, p, training=True):
    if not training or p <= 0.0:
        return x
    seed = next_seed()
    if DROPOUT_TRACE[0] is not None:
        DROPOUT_TRACE[0].append(("drop", float(p), seed, tuple(x.shape)))
    return _DropoutFn.apply(x, float(p), seed)


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
_ATTN_FUSED = [__import__("os").environ.get("VCVITS_ATTN_FUSED", "1") == "1"]


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


# ---------------------------------------------------------------------------------------------
# KL loss, nearest interpolation, segment slicing
# ---------------------------------------------------------------------------------------------
class _KlFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_p, logs_q, m_p, logs_p, mask):
        z_p, logs_q, m_p, logs_p, mask = (_f32c(t) for t in (z_p, logs_q, m_p, logs_p, mask))
        B, C, T = z_p.shape
        out2 = torch.empty((2,), device=z_p.device, dtype=torch.float32)
        check(lib().vcv_kl_fwd(ptr(z_p), ptr(logs_q), ptr(m_p), ptr(logs_p), ptr(mask), ptr(out2), B, C, T,
                               stream()), "vcv_kl_fwd")
        ctx.save_for_backward(z_p, m_p, logs_p, mask, out2)
        return out2[0] / out2[1]

    @staticmethod
    def backward(ctx, gout):
        z_p, m_p, logs_p, mask, out2 = ctx.saved_tensors
        B, C, T = z_p.shape
        gout = _f32c(gout).reshape(1)
        den = out2[1:2]
        grads = [torch.empty_like(z_p) for _ in range(4)]
        check(lib().vcv_kl_bwd(ptr(z_p), ptr(m_p), ptr(logs_p), ptr(mask), ptr(gout), ptr(den), ptr(grads[0]),
                               ptr(grads[1]), ptr(grads[2]), ptr(grads[3]), B, C, T, stream()), "vcv_kl_bwd")
        return grads[0], grads[1], grads[2], grads[3], None


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    return _KlFn.apply(z_p, logs_q, m_p, logs_p, z_mask)


class _NearestFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tout):
        x = _f32c(x)
        B, C, Tin = x.shape
        y = torch.empty((B, C, tout), device=x.device, dtype=torch.float32)
        check(lib().vcv_nearest_fwd(ptr(x), ptr(y), B * C, Tin, tout, stream()), "vcv_nearest_fwd")
        ctx.tin = Tin
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        B, C, Tout = dy.shape
        dx = torch.empty((B, C, ctx.tin), device=dy.device, dtype=torch.float32)
        check(lib().vcv_nearest_bwd(ptr(dy), ptr(dx), B * C, ctx.tin, Tout, stream()), "vcv_nearest_bwd")
        return dx, None


def interpolate_nearest(x, size):
    return _NearestFn.apply(x, int(size))


class _EmbeddingTFn(torch.autograd.Function):
    """W[idx] laid out [B, C, T] (idx int64 [B, T]); the table gradient is one launch without atomics, sort or host
    read-back, added straight into the parameter's gradient sink when it has one."""

    @staticmethod
    def forward(ctx, idx, W):
        W = _f32c(W)
        idx = idx.to(torch.int64).contiguous()
        B, T = idx.shape
        rows, C = W.shape
        y = torch.empty((B, C, T), device=W.device, dtype=torch.float32)
        check(lib().vcv_embedding_t_fwd(ptr(idx), ptr(W), ptr(y), B, T, C, rows, stream()), "vcv_embedding_t_fwd")
        ctx.w_sink = _sink(W)
        ctx.shape = (B, T, C, rows)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, T, C, rows = ctx.shape
        if not ctx.needs_input_grad[1]:
            return None, None
        dy = _f32c(dy)
        sink = ctx.w_sink
        dW = sink[0].view(rows, C) if sink is not None else torch.empty((rows, C), device=dy.device, dtype=torch.float32)
        check(lib().vcv_embedding_t_bwd(ptr(idx), ptr(dy), ptr(dW), B, T, C, rows, 1 if sink is not None else 0, stream()),
              "vcv_embedding_t_bwd")
        return None, _sunk(sink, dW)


def embedding_t(idx, W):
    """F.embedding(idx, W).transpose(1, -1) for idx [B, T] -> [B, C, T] (content_encoder.py:58-60); idx [B] -> [B, C, 1]
    (emb_g(sid).unsqueeze(-1), synthesizer_svc.py:77)."""
    if idx.dim() == 1:
        idx = idx.view(-1, 1)
    return _EmbeddingTFn.apply(idx, W)


class _SliceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ids, mul, seg):
        x = _f32c(x)
        ids = ids.to(torch.int64).contiguous()
        B, C, T = x.shape
        y = torch.empty((B, C, seg), device=x.device, dtype=torch.float32)
        check(lib().vcv_slice_fwd(ptr(x), ptr(ids), mul, ptr(y), B, C, T, seg, stream()), "vcv_slice_fwd")
        ctx.shape, ctx.mul, ctx.seg = (B, C, T), mul, seg
        ctx.save_for_backward(ids)
        return y

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        dy = _f32c(dy)
        B, C, T = ctx.shape
        dx = torch.empty((B, C, T), device=dy.device, dtype=torch.float32)
        check(lib().vcv_slice_bwd(ptr(dy), ptr(ids), ctx.mul, ptr(dx), B, C, T, ctx.seg, stream()),
              "vcv_slice_bwd")
        return dx, None, None, None


def slice_segments(x, ids_str, segment_size=4, mul=1):
    """x[b, :, ids[b]*mul : ids[b]*mul + segment_size] (commons.py:48-54)."""
    return _SliceFn.apply(x, ids_str, int(mul), int(segment_size))
