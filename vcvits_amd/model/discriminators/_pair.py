"""Shared (y, y_hat) driver of the multi-period / multi-scale discriminators.

The reference runs d(y) then d(y_hat) as two passes per discriminator
(multi_period_discriminator.py:22-28).  When the discriminator weights take gradients (D step)
both signals share every weight, so they are stacked into ONE batch of 2B: each conv, and each
weight-gradient reduction, is a single launch over both.  When the weights are frozen (G step,
Lightning-1.x toggle_optimizer semantics) the two signals are stacked as well, and the backward of
every conv is restricted to the generated half (`ops.grad_batch_start`), so no data-gradient work is
issued for the real branch."""
import torch

from ... import tuning


class _Halves(torch.autograd.Function):
    """(x[:b], x[b:]) of a stacked score tensor as one node.  Backward: ONE concatenation kernel, where the two torch
    slices it replaces each run zeros + copy_ (plus the add that sums them): five launches per sub-discriminator and pass,
    two of them device-to-device copy nodes once recorded."""

    @staticmethod
    def forward(ctx, x, b):
        ctx.b, ctx.shape = b, tuple(x.shape)
        return x[:b], x[b:]

    @staticmethod
    def backward(ctx, g0, g1):
        if g0 is None and g1 is None:
            return None, None
        b, shape = ctx.b, ctx.shape
        like = g0 if g0 is not None else g1
        if g0 is None:
            g0 = torch.zeros((b,) + shape[1:], device=like.device, dtype=like.dtype)
        if g1 is None:
            g1 = torch.zeros((shape[0] - b,) + shape[1:], device=like.device, dtype=like.dtype)
        return torch.cat([g0, g1], dim=0), None


def run_pair(disc, y, y_hat):
    if getattr(disc, "use_spectral_norm", False) and disc.training:
        # every training forward advances the layers' power-iteration vectors, and d(y) and d(y_hat) are two forwards in
        # the reference (multi_period_discriminator.py:22-28): the second divides by the sigma of the second iteration
        out_r, fmap_r = disc(y)
        out_g, fmap_g = disc(y_hat)
        return out_r, out_g, fmap_r, fmap_g
    wants_wgrad = torch.is_grad_enabled() and any(p.requires_grad for p in disc.parameters())
    B = y.shape[0]
    if wants_wgrad or not (y_hat.requires_grad and torch.is_grad_enabled()):
        out, fmap = disc(torch.cat([y, y_hat], dim=0))
        out_r, out_g = _Halves.apply(out, B) if out.requires_grad else (out[:B], out[B:])
        return out_r, out_g, [f[:B] for f in fmap], [f[B:] for f in fmap]
    # frozen weights, gradient wanted for y_hat only: still ONE stacked forward pass (wider GEMM tiles,
    # half the launches, weight norm computed once); the backward is told to skip the real half
    from ... import ops
    with ops.grad_batch_start(B):
        out, fmap = disc(torch.cat([y.detach(), y_hat], dim=0))
    taps = [isinstance(f, ops.FmapTap) for f in fmap]
    out_g = _Halves.apply(out, B)[1] if out.requires_grad else out[B:]
    return (out[:B].detach(), out_g, [f.real if t else f[:B].detach() for f, t in zip(fmap, taps)],
            [f.fake if t else f[B:] for f, t in zip(fmap, taps)])


_STREAMS = []
_N_STREAMS = tuning.integer("VCVITS_STREAMS", 1, "the independent sub-discriminator chains of MPD / MSD spread over N HIP streams (eager loop only; 1: one stream)")


def streams():
    return tuning.live_flag("VCVITS_STREAMS")


def join_streams():
    """Make the current stream wait for everything queued on the side streams.  Needed after a BACKWARD pass: the weight-,
    bias- and weight-norm-gradient kernels of a sub-discriminator run on its side stream and add into the optimizer's flat
    gradient buffer through gradient sinks -- autograd sees `None` for those parameters and therefore puts no stream
    dependency between the side stream and whatever reads the buffer next (the all-reduce, AdamW).  In the eager loop the
    omission is a latent race; recorded into a HIP graph it is a missing edge, and the replayed AdamW overtakes the
    gradients (tools/probes/streams_race_probe.py)."""
    n = streams()
    if n <= 1 or not _STREAMS:
        return
    cur = torch.cuda.current_stream()
    for s in _STREAMS[:n]:  # (the streams run_many deals the chains to; under a capture they are part of it since the forward)
        if s is not cur:
            cur.wait_stream(s)


class _Handoff(torch.autograd.Function):
    """Identity placed where a tensor crosses from one stream to another inside run_many.  Its BACKWARD marks the gradient's
    block as in use on both streams (record_stream).  Without it the block goes back to the free list of the stream that
    allocated it as soon as the consumer on the other stream has been QUEUED, and the allocating stream's next kernel may
    write it before that consumer has run.  The eager loop gets away with it on timing; found in a recorded batch, whose
    branches HIP runs in whatever order the edges allow: the replayed sub-discriminator read another loss's gradient in
    place of its own (tools/probes/streams_race_probe.py --small --trace-mpd1 follows the block;
    profiles/r6_multistream_replay_probe.txt item 1)."""

    @staticmethod
    def forward(ctx, x, other):
        ctx.other = other
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if g is not None and g.is_cuda:
            g.record_stream(ctx.other)
            g.record_stream(torch.cuda.current_stream())
        return g, None


def _hand(t, other):
    return _Handoff.apply(t, other) if (t.requires_grad and torch.is_grad_enabled()) else t


def run_many(discs, inputs):
    """Run independent discriminators on their (y, y_hat) pairs.  With VCVITS_STREAMS=N > 1 the chains
    are spread over N HIP streams so the short, low-occupancy layers (first / last convs, pooled scales)
    of one discriminator overlap the big GEMMs of another; autograd replays each chain's backward on
    the stream it ran on.  Eager loop only: light/graphed.py does not record a batch with the forks in it (a multi-branch
    HIP graph replayed wrong for some deals of the chains to the streams -- profiles/r6_multistream_replay_probe.txt)."""
    n = streams()
    if n <= 1 or not inputs[0][0].is_cuda:
        return [run_pair(d, y, y_hat) for d, (y, y_hat) in zip(discs, inputs)]
    while len(_STREAMS) < n:
        _STREAMS.append(torch.cuda.Stream())
    cur = torch.cuda.current_stream()
    # one node per chain ON THE CURRENT STREAM between y_hat and the chain: the chains' gradients of y_hat then meet in the
    # input buffer of y_hat's producer as the outputs of same-stream nodes, and autograd sums them on that stream -- not on
    # whichever side stream delivered first, with the partial sums changing streams in between
    mains = [_hand(y_hat, cur) for _, y_hat in inputs]
    ready = cur.record_event()
    outs, events = [], []
    for i, (d, (y, y_hat)) in enumerate(zip(discs, inputs)):
        s = _STREAMS[_stream_of(i, n)]
        s.wait_event(ready)
        with torch.cuda.stream(s):
            y.record_stream(s)
            y_hat.record_stream(s)
            o = run_pair(d, y, _hand(mains[i], cur))  # (gradient of y_hat: made on s, handed to the node on cur)
            # (gradients of the outputs and feature maps: made by the loss nodes on cur, consumed on s)
            o = (_hand(o[0], cur), _hand(o[1], cur), [_hand(f, cur) for f in o[2]], [_hand(f, cur) for f in o[3]])
            for t in [o[0], o[1]] + list(o[2]) + list(o[3]):
                t.record_stream(cur)
            events.append(s.record_event())
        outs.append(o)
    for e in events:
        cur.wait_event(e)
    return outs


def _stream_of(i, n):
    """Side stream of chain i (round robin; tools/probes/streams_race_probe.py tries other deals)."""
    return i % n
