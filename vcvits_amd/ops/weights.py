"""Weight side of the launches: weight / spectral norm (single and batched per module tree), the per-parameter-set cache of
normalised + packed weights, parameter regions of the flat optimizer buffer, and the batched operand packs.

Part of `vcvits_amd.ops` (the package re-exports every name: `from vcvits_amd import ops; ops.conv1d(...)`).  Everything here
runs on the GPU through libvcvits_hip.so; there is no CPU fallback."""
import ctypes

import torch

from .. import tuning
from .._lib import (VcvConvArgs, check, lib, ptr, stream)
from .core import (CAPTURING, LAUNCH_COUNTS, _COMPUTE, _USE_PK, _USE_X3, _f32c,
                   _sink, _upload_table)


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


_WN_CACHE_ON = [tuning.flag("VCVITS_WEIGHT_CACHE", True, "weight-normed weights and their packs cached until a parameter changes")]


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


_PARAM_REGIONS_ON = [tuning.flag("VCVITS_PARAM_REGIONS", True, "packs of plain (not weight-normed) conv weights cached per optimizer step")]


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


_PACK_BATCH = [tuning.flag("VCVITS_PACK_BATCH", True, "all packs of a module tree re-made in one launch after its weights changed")]


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
    from .._lib import VcvPackJob
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
    if cap is not None:
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
    if hit is not None and (hit.get("cap") != (cap.id if cap is not None else None)):
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
