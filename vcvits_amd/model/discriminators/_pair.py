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
        return out[:B], out[B:], [f[:B] for f in fmap], [f[B:] for f in fmap]
    # frozen weights, gradient wanted for y_hat only: still ONE stacked forward pass (wider GEMM tiles,
    # half the launches, weight norm computed once); the backward is told to skip the real half
    from ... import ops
    with ops.grad_batch_start(B):
        out, fmap = disc(torch.cat([y.detach(), y_hat], dim=0))
    taps = [isinstance(f, ops.FmapTap) for f in fmap]
    return (out[:B].detach(), out[B:], [f.real if t else f[:B].detach() for f, t in zip(fmap, taps)],
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


def run_many(discs, inputs):
    """Run independent discriminators on their (y, y_hat) pairs.  With VCVITS_STREAMS=N > 1 the chains
    are spread over N HIP streams so the short, low-occupancy layers (first / last convs, pooled scales)
    of one discriminator overlap the big GEMMs of another; autograd replays each chain's backward on
    the stream it ran on.  Eager loop only: a batch RECORDED with the forks replays 4 - 5 % faster (two streams: 253 vs 242
    utterances/s on the fp32 headline) but not reliably right -- light/graphed.py refuses to record it."""
    n = streams()
    if n <= 1 or not inputs[0][0].is_cuda:
        return [run_pair(d, y, y_hat) for d, (y, y_hat) in zip(discs, inputs)]
    while len(_STREAMS) < n:
        _STREAMS.append(torch.cuda.Stream())
    cur = torch.cuda.current_stream()
    ready = cur.record_event()
    outs, events = [], []
    for i, (d, (y, y_hat)) in enumerate(zip(discs, inputs)):
        s = _STREAMS[i % n]
        s.wait_event(ready)
        with torch.cuda.stream(s):
            y.record_stream(s)
            y_hat.record_stream(s)
            o = run_pair(d, y, y_hat)
            for t in [o[0], o[1]] + list(o[2]) + list(o[3]):
                t.record_stream(cur)
            events.append(s.record_event())
        outs.append(o)
    for e in events:
        cur.wait_event(e)
    return outs
