"""Race detector for the multi-stream discriminators inside a recorded batch: in deterministic mode (no atomics anywhere in
the GAN step) the replayed batches must equal the eager loop BIT FOR BIT, whatever the stream count.  GraphedBatch does not
record with VCVITS_STREAMS > 1; this probe makes it (profiles/r6_multistream_replay_probe.txt).
  VCVITS_STREAMS=2 [PROBE_MAP=001] python tools/probes/streams_race_probe.py [options]
    --full / --small / --periods23 / --seg4096 / --b2 / --b16 / --48k / --bf16 / --nondet / --one-batch   the workload
    PROBE_MAP=abc        chain i of a multi-discriminator on side stream digit i (default: round robin)
    PROBE_NOHAND=1       without the gradient hand-off nodes (model/discriminators/_pair.py: _Handoff)
    --trace-mpd1         every stage of one DiscriminatorP, forward and backward, eager batch beside first replayed batch
    --trace-hand         the gradient at every stream hand-off; --trace-gen: inputs / output gradients of every generator block
    --memhist            allocator history of the block one wrong value was read from (with --trace-mpd1)
    --dot PATH           hipGraphDebugDotPrint of every recorded graph (tools/probes/graph_dot_check.py reads them)"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from vcvits_amd import configs, ops, synthetic  # noqa: E402
from vcvits_amd.light import graphed  # noqa: E402
from vcvits_amd.light.vcvits import VCVITS, VocoderGAN  # noqa: E402


def _force_recording():
    """GraphedBatch refuses to record with VCVITS_STREAMS > 1 (the replays diverge: this probe shows it); look the other way
    while it asks."""
    orig = graphed.GraphedBatch.applicable

    def applicable(self):
        old = os.environ.get("VCVITS_STREAMS")
        os.environ["VCVITS_STREAMS"] = "1"
        try:
            return orig(self)
        finally:
            if old is None:
                os.environ.pop("VCVITS_STREAMS", None)
            else:
                os.environ["VCVITS_STREAMS"] = old
    graphed.GraphedBatch.applicable = applicable


def _variant_run_many():
    """PROBE_MAP=011: chain i of a multi-discriminator goes to the side stream digit i names (instead of round robin);
    PROBE_NOHAND=1: no gradient hand-off nodes at the stream boundaries."""
    from vcvits_amd.model.discriminators import _pair
    pmap = os.environ.get("PROBE_MAP")
    if pmap:
        _pair._stream_of = lambda i, n: int(pmap[i % len(pmap)])
    if os.environ.get("PROBE_NOHAND") == "1":
        _pair._hand = lambda t, other: t


TRACE = {"bufs": {}, "call": 0}


def _trace_mpd1(mod):
    """--trace-mpd1: every stage of net_period_d.discriminators[1] (input, effective weights, each layer's output) copied
    into buffers allocated BEFORE any recording (first eager batch), per call of the batch (0: generator pass, 1: discriminator
    pass) -- so the first replayed batch's intermediates can be laid beside the eager batch's."""
    from vcvits_amd.model.modules import prepare_weight_norm
    from vcvits_amd.model.discriminators.discriminator import ACT_LEAKY, LRELU_SLOPE
    d = mod.net_period_d.discriminators[1]

    def put(name, t):
        key = (TRACE["call"], name)
        buf = TRACE["bufs"].get(key)
        if buf is None:
            buf = TRACE["bufs"][key] = torch.zeros_like(t.detach(), memory_format=torch.contiguous_format)
        buf.copy_(t.detach())
        if ops.CAPTURING[0] is not None:
            TRACE.setdefault("ptr", {})[key] = (t.data_ptr(), t.numel() * t.element_size())

    def hook(name, t):
        if t.requires_grad:
            call = TRACE["call"]

            def h(g):
                old = TRACE["call"]
                TRACE["call"] = call
                put("d_" + name, g)
                TRACE["call"] = old
            t.register_hook(h)

    def forward(x):
        prepare_weight_norm(d)
        lazy = d.convs[0].__dict__.get("_w_lazy")
        if lazy is not None:
            put("wbuf", lazy[0].wbuf)
        put("in", x)
        fmap = []
        b, c, t = x.shape
        x = x.view(b, c, t // d.period, d.period)
        for i, l in enumerate(d.convs):
            x, rec = ops.fmap_tap(l(x, out_act=ACT_LEAKY, slope=LRELU_SLOPE))
            put("conv%d" % i, x)
            hook("conv%d" % i, x)
            fmap.append(rec)
        x, rec = ops.fmap_tap(d.conv_post(x))
        put("post", x)
        hook("post", x)
        fmap.append(rec)
        TRACE["call"] += 1
        return torch.flatten(x, 1, -1), fmap
    d.forward = forward
    from vcvits_amd.light import vcvits as V
    orig = V.discriminator_loss.__wrapped__ if hasattr(V.discriminator_loss, "__wrapped__") else V.discriminator_loss
    n = [0]

    def discriminator_loss(rs, gs):
        which = "mpd" if len(TRACE.get("seen", [])) % 2 == 0 else "msd"
        TRACE.setdefault("seen", []).append(1)
        TRACE["call"] = 9
        for i, (r, g) in enumerate(zip(rs, gs)):
            put("%s.main_sees_r%d" % (which, i), r)
            put("%s.main_sees_g%d" % (which, i), g)
        out = orig(rs, gs)
        put(which + ".r_terms", torch.stack(out[1]))
        put(which + ".g_terms", torch.stack(out[2]))
        put(which + ".loss", out[0])
        call = TRACE["call"]
        for nm, t in ((which + ".loss", out[0]),) + tuple(("%s.r%d" % (which, i), r) for i, r in enumerate(rs)):
            if t.requires_grad:
                def h(g, nm=nm):
                    old = TRACE["call"]
                    TRACE["call"] = 9
                    put("d_" + nm, g)
                    TRACE["call"] = old
                t.register_hook(h)
        for i, (r, g) in enumerate(zip(rs, gs)):
            put("%s.after_r%d" % (which, i), r)
        return out
    discriminator_loss.__wrapped__ = orig
    from vcvits_amd.ops import elementwise as E
    if not hasattr(E._LossTermsFn, "_orig_backward"):
        E._LossTermsFn._orig_backward = E._LossTermsFn.backward
        cnt = [0]

        def backward(ctx, gout):
            if ctx.mode == 1 and ctx.n_a == 3 and TRACE["call"] >= 2:  # the discriminator pass's MPD terms
                tag = "bw%d." % (len(TRACE.setdefault("bw", [])) % 2)
                TRACE["bw"].append(1)
                old = TRACE["call"]
                TRACE["call"] = 8
                put(tag + "gout", gout.contiguous())
                for i in range(3):
                    put(tag + "a%d_before" % i, ctx.saved_tensors[1 + i])
                res = E._LossTermsFn._orig_backward(ctx, gout)
                for i in range(3):
                    put(tag + "da%d" % i, res[4 + i])
                    put(tag + "a%d_after" % i, ctx.saved_tensors[1 + i])
                TRACE["call"] = old
                return res
            return E._LossTermsFn._orig_backward(ctx, gout)
        E._LossTermsFn.backward = staticmethod(backward)
    V.discriminator_loss = discriminator_loss


def _dump_graphs(path):
    """--dot PATH: every recorded graph written as a DOT file (hipGraphDebugDotPrint on the kept hipGraph_t) right after its
    capture ends."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
    G = torch.cuda.CUDAGraph
    n = [0]

    class Kept(G):
        def __new__(cls, *a, **k):
            return super().__new__(cls, True)

        def __init__(self, *a, **k):
            super().__init__(True)

        def capture_end(self):
            super().capture_end()
            rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(self.raw_cuda_graph()), ("%s.%d.dot" % (path, n[0])).encode(), 1)
            print("hipGraphDebugDotPrint ->", rc)
            n[0] += 1
            self.instantiate()
    torch.cuda.CUDAGraph = Kept


def _trace_handoffs():
    """--trace-hand: every gradient that passes a stream hand-off node copied out (in forward-call order per batch)."""
    from vcvits_amd.model.discriminators import _pair
    H = _pair._Handoff
    of, ob = H.forward, H.backward

    def forward(ctx, x, other):
        ctx.tag = TRACE.setdefault("hand", [0])[0]
        TRACE["hand"][0] += 1
        ctx.fstream = torch.cuda.current_stream() == other
        return of(ctx, x, other)

    def backward(ctx, g):
        key = (7, "hand%03d.%s.%s" % (ctx.tag, "main" if ctx.fstream else "side", "x".join(map(str, g.shape))))
        buf = TRACE["bufs"].get(key)
        if buf is None:
            buf = TRACE["bufs"][key] = torch.zeros_like(g, memory_format=torch.contiguous_format)
        buf.copy_(g)
        return ob(ctx, g)
    H.forward, H.backward = staticmethod(forward), staticmethod(backward)
    from vcvits_amd.model.discriminators.multi_period_discriminator import MultiPeriodDiscriminator as MPD
    mf = MPD.forward

    def mpd_forward(self, y, y_hat, g=None):
        if y_hat.requires_grad:
            def h(gr):
                key = (7, "y_hat_total")
                buf = TRACE["bufs"].get(key)
                if buf is None:
                    buf = TRACE["bufs"][key] = torch.zeros_like(gr, memory_format=torch.contiguous_format)
                buf.copy_(gr)
            y_hat.register_hook(h)
        return mf(self, y, y_hat, g)
    MPD.forward = mpd_forward


def _trace_generator(mod):
    """--trace-gen: per generator block, its input at forward time, the same tensor again when the block's backward starts,
    and the gradient arriving at its output."""
    def put(name, t):
        key = (6, name)
        buf = TRACE["bufs"].get(key)
        if buf is None:
            buf = TRACE["bufs"][key] = torch.zeros_like(t.detach(), memory_format=torch.contiguous_format)
        buf.copy_(t.detach())

    def attach(name, m):
        def fh(mod_, inp, out):
            if not (torch.is_grad_enabled() and out.requires_grad):
                return
            x = inp[0]
            put(name + ".in_fwd", x)

            def h(g):
                put(name + ".gout", g)
                put(name + ".in_bwd", x)
            out.register_hook(h)
        m.register_forward_hook(fh)
    g = mod.net_g if not hasattr(mod.net_g, "dec") else mod.net_g.dec
    attach("conv_post", g.conv_post)
    for i, m in enumerate(g.resblocks):
        attach("rb%02d" % i, m)
    for i, m in enumerate(g.ups):
        attach("ups%d" % i, m)
    attach("conv_pre", g.conv_pre)


def main():
    _force_recording()
    if "--trace-hand" in sys.argv:
        _trace_handoffs()
    _variant_run_many()
    if "--dot" in sys.argv:
        _dump_graphs(sys.argv[sys.argv.index("--dot") + 1])
    nb = 1 if "--one-batch" in sys.argv else 2
    full = "--full" in sys.argv
    dev = torch.device("cuda:0")
    ops.set_deterministic("--nondet" not in sys.argv)
    cfg = configs.base_48k() if "--48k" in sys.argv else configs.base()
    small = "--small" in sys.argv  # the reduced-width vocoder of tests/test_graphed_gpu.py::test_graphed_step_vocoder_workload
    if small:
        cfg["model"].update({"inter_channels": 16, "upsample_initial_channel": 32, "multi_period_discriminator_periods": [2, 3]})
        cfg["data"]["n_mel_channels"] = 40
        cfg["train"]["segment_size"] = 4096
    if "--periods23" in sys.argv:
        cfg["model"]["multi_period_discriminator_periods"] = [2, 3]
    if "--seg4096" in sys.argv:
        cfg["train"]["segment_size"] = 4096
    if "--bf16" in sys.argv:
        ops.set_compute_dtype("bf16")
    if full:
        cfg["model"]["p_dropout"] = 0.0
    torch.manual_seed(3)
    cls = VCVITS if full else VocoderGAN
    sd = copy.deepcopy(cls(**cfg).state_dict())
    B = 16 if "--b16" in sys.argv else (2 if small or "--b2" in sys.argv else 4)
    seg = {"segment_size": 4096} if small or "--seg4096" in sys.argv else {}
    m = cfg["model"]
    if full:
        batches = []
        for i in range(2):
            b = synthetic.full_batch(B, m["hubert_channels"], seed=40 + i, device=dev)
            g = torch.Generator().manual_seed(70 + i)
            b["noise"] = torch.randn(B, m["inter_channels"], 384, generator=g).to(dev)
            b["ids_slice"] = torch.tensor(([5, 100, 17, 200] * 4)[:B], device=dev)
            batches.append(b)
    else:
        batches = [synthetic.vocoder_batch(B, m["inter_channels"], seed=40 + i, device=dev, **seg)
                   for i in range(2)]
    res = {}
    if "--memhist" in sys.argv:
        torch.cuda.memory._record_memory_history(max_entries=2000000, stacks="python")
    for mode in (False, True):
        graphed.set_batch_enabled(mode)
        mod = cls(**cfg)
        mod.load_state_dict(sd)
        mod = mod.to(dev)
        mod.configure_optimizers()
        ls = []
        snap = None
        if "--trace-mpd1" in sys.argv:
            _trace_mpd1(mod)
        if "--trace-gen" in sys.argv:
            _trace_generator(mod)
        for i in range(10):
            TRACE["call"] = 0
            TRACE["seen"] = []
            TRACE["hand"] = [0]
            o = mod.fit_batch(batches[i % nb])
            if i == 2 and TRACE["bufs"]:
                torch.cuda.synchronize()
                TRACE[mode] = {k: v.clone() for k, v in TRACE["bufs"].items()}
            ls.append((o["g"].item(), o["d"].item()))
            if i == 2:  # the first replayed batch: the gradients it left in the flat buffers
                names = {id(p): n for n, p in mod.named_parameters()}
                snap = {}
                for tag, opt in (("G", mod.optim_g), ("D", mod.optim_d)):
                    for p, off in zip(opt.params, opt.offsets):
                        snap[tag + ":" + names[id(p)]] = opt.grad[off:off + p.numel()].clone()
        if mode and "--memhist" in sys.argv and "ptr" in TRACE:
            snap = torch.cuda.memory._snapshot()
            lo, n = TRACE["ptr"][(1, "post")]
            print("discriminator-pass output of MPD.1 while recording: 0x%x + %d" % (lo, n))
            for ev in snap["device_traces"][0]:
                a, sz = ev.get("addr", 0), ev.get("size", 0)
                if a < lo + n and lo < a + sz:
                    fr = [f for f in ev.get("frames", []) if "torch/" not in f["filename"] and "probe" not in f["filename"]][:4]
                    print("  %-16s 0x%x + %-8d stream %-14s %s" % (ev["action"], a, sz, ev.get("stream"), " <- ".join(
                        "%s:%d" % (os.path.basename(f["filename"]), f["line"]) for f in fr)))
        bg = mod.__dict__.get("_batch_graph")
        res[mode] = (ls, mod.optim_g.flat.clone(), mod.optim_d.flat.clone(), bg.replays if bg is not None else 0, snap)
        mod.optim_g.close()
        mod.optim_d.close()
    (l0, g0, d0, r0, s0), (l1, g1, d1, r1, s1) = res[False], res[True]
    if False in TRACE and True in TRACE:
        for k in sorted(TRACE[False]):
            a, b = TRACE[False][k], TRACE[True][k]
            if k[0] in (6, 7) and torch.equal(a, b):
                continue
            print("trace call %d %-6s eager |x| %.6e  replay |x| %.6e  max |diff| %.3e" % (k[0], k[1], float(a.norm()), float(b.norm()), float((a - b).abs().max())))
    if "--nondet" in sys.argv:  # atomics: compare to a tolerance
        bad = [(k, float((s0[k] - s1[k]).abs().max()), float(s0[k].abs().max())) for k in s0
               if float((s0[k] - s1[k]).norm()) > 1e-3 * float(s0[k].norm()) + 1e-12]
    else:
        bad = [(k, float((s0[k] - s1[k]).abs().max()), float(s0[k].abs().max())) for k in s0 if k in s1 and not torch.equal(s0[k], s1[k])]
    print("gradient tensors of the first replayed batch that differ from the eager batch's: %d of %d" % (len(bad), len(s0)))
    for k, e, sc in bad[:40]:
        print("   %-60s max |diff| %.3e  (scale %.3e)" % (k, e, sc))
    print("streams", os.environ.get("VCVITS_STREAMS", "1"), "full" if full else "vocoder", "replays", r1)
    same = all(a == b for a, b in zip(l0, l1))
    print("checksums of the final parameters (compare across stream counts): eager G %.10e D %.10e | replayed G %.10e D %.10e"
          % (float(g0.double().sum()), float(d0.double().sum()), float(g1.double().sum()), float(d1.double().sum())))
    print("losses bit-identical:", same, "| generator params max |diff|", float((g0 - g1).abs().max()), "| discriminator params max |diff|", float((d0 - d1).abs().max()))
    if not same:
        for i, (a, b) in enumerate(zip(l0, l1)):
            print(i, a, b)


if __name__ == "__main__":
    main()
