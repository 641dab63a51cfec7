"""VITS building blocks: WaveNet gate / res-skip, posterior split + sample, coupling, prior sample, channel LayerNorm,
counter-based dropout and its seed stream, KL loss, nearest interpolation, transposed embedding, segment slicing.

Part of `vcvits_amd.ops` (the package re-exports every name: `from vcvits_amd import ops; ops.conv1d(...)`).  Everything here
runs on the GPU through libvcvits_hip.so; there is no CPU fallback."""
import torch

from .. import tuning
from .._lib import (check, lib, ptr, stream)
from .core import (_f32c, _sink, _sunk)
from .elementwise import (mask_mul, scale)


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


class _NearestRawFn(torch.autograd.Function):
    """_NearestFn with the index map of the batch's raw padded sizes (device int64 pair), zeros in the bucket padding."""

    @staticmethod
    def forward(ctx, x, tout, raw):
        x = _f32c(x)
        B, C, Tin = x.shape
        y = torch.empty((B, C, tout), device=x.device, dtype=torch.float32)
        check(lib().vcv_nearest_raw_fwd(ptr(x), ptr(y), B * C, Tin, tout, ptr(raw), stream()), "vcv_nearest_raw_fwd")
        ctx.tin = Tin
        ctx.save_for_backward(raw)
        return y

    @staticmethod
    def backward(ctx, dy):
        (raw,) = ctx.saved_tensors
        dy = _f32c(dy)
        B, C, Tout = dy.shape
        dx = torch.empty((B, C, ctx.tin), device=dy.device, dtype=torch.float32)
        check(lib().vcv_nearest_raw_bwd(ptr(dy), ptr(dx), B * C, ctx.tin, Tout, ptr(raw), stream()), "vcv_nearest_raw_bwd")
        return dx, None, None


def interpolate_nearest(x, size, raw_sizes=None):
    """F.interpolate(x, size=(size,), mode="nearest") along T.  raw_sizes (device int64 [2] = the batch's un-bucketed padded
    (Tin, Tout), data/collate.py: bucket_batch): the index map of THOSE sizes, zeros beyond Tout_raw."""
    if raw_sizes is not None:
        if raw_sizes.dtype != torch.int64 or raw_sizes.numel() != 2 or raw_sizes.device != x.device:
            raise RuntimeError("interpolate_nearest: raw_sizes must be an int64 [2] tensor on x's device")
        return _NearestRawFn.apply(x, int(size), raw_sizes.contiguous())
    return _NearestFn.apply(x, int(size))


# Out-of-range embedding indices: nn.Embedding (the reference's emb_pitch / emb_g) raises on one; the kernel zero-fills the
# column and COUNTS the position in a per-device int32 word, which check_indices() reads back (a device sync: called at check
# points -- validation, checkpoint save, epoch end -- or after every batch with VCVITS_CHECK_INDICES=1) and raises on.
_INDEX_ERR = {}
CHECK_INDICES_EVERY_BATCH = [tuning.flag("VCVITS_CHECK_INDICES", False, "read the out-of-range embedding index count back after EVERY batch (a device sync)")]


def _index_err_word(dev):
    w = _INDEX_ERR.get(dev)
    if w is None:
        w = _INDEX_ERR[dev] = torch.zeros(1, device=dev, dtype=torch.int32)  # (lives as long as the process: graphs bake it)
    return w


def index_errors(reset=True):
    """Number of embedding lookups with an index outside their table since the last reset (synchronises the device)."""
    n = 0
    for w in _INDEX_ERR.values():
        n += int(w.item())
        if reset:
            w.zero_()
    return n


def check_indices():
    n = index_errors()
    if n:
        raise IndexError("vcvits_amd: %d embedding lookup(s) used an index outside the table (pitch bin >= the pitch table's "
                         "rows, speaker id >= n_speakers?) -- nn.Embedding raises on these; the HIP kernel read them as zero "
                         "rows" % n)


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
        check(lib().vcv_embedding_t_fwd_checked(ptr(idx), ptr(W), ptr(y), B, T, C, rows, ptr(_index_err_word(W.device)), stream()),
              "vcv_embedding_t_fwd_checked")
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
