"""CPU, world_size 2, gloo: the data-parallel gradient path of FlatAdamW (bucket hooks fired from
backward, async all-reduce per contiguous bucket, averaging) and the rank-strided sharding of
synthetic utterances.  The HIP step itself needs a GPU; here the optimizer step is replaced by
an SGD stand-in on the averaged flat gradient, which is what DDP must deliver."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcvits_amd.light.optim import FlatAdamW
    torch.manual_seed(0)  # same init on both ranks
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 1))
    opt = FlatAdamW(net.parameters(), 1e-2, bucket_mb=0.0003)  # ~79 floats per bucket -> several buckets
    assert len(opt._buckets) >= 2
    opt.broadcast_parameters()
    # every rank sees a different shard of the "utterances"
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5, 8, generator=g)
    y = torch.randn(5, 1, generator=g)
    opt.zero_grad()
    loss = ((net(x) - y) ** 2).mean()
    loss.backward()  # hooks launch the bucket all-reduces
    opt.finish_grad_sync()
    out[rank] = (opt.grad.clone(), [p.grad.data_ptr() == opt.grad[o:o + p.numel()].data_ptr()
                                    for p, o in zip(opt.params, opt.offsets)])
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_global_batch_gradient():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    g0, views0 = out[0]
    g1, views1 = out[1]
    assert all(views0) and all(views1)  # .grad stayed a view of the flat buffer through backward
    assert torch.allclose(g0, g1, atol=0, rtol=0)  # both ranks hold the same averaged gradient
    # reference: gradient of the mean loss over the union of both shards
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 1))
    xs, ys = [], []
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        xs.append(torch.randn(5, 8, generator=g))
        ys.append(torch.randn(5, 1, generator=g))
    loss = ((net(torch.cat(xs)) - torch.cat(ys)) ** 2).mean()
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in reversed(list(net.parameters()))])
    assert torch.allclose(g0, ref, atol=1e-6, rtol=1e-5)


# ---- model-level path: gradient sinks (kernels add into the flat buffer and hand autograd None), frozen parameters,
# ---- buckets that never complete, per-parameter "touched" tracking -- on 2 ranks ---------------------------------------
class _SinkLinear(torch.autograd.Function):
    """y = x @ w.T + b the way the HIP conv ops treat parameters: when the optimizer registered a gradient sink the
    backward ADDS the parameter gradient into it and returns None (ops._sink / ops._sunk), else it returns the gradient."""

    @staticmethod
    def forward(ctx, x, w, b):
        from vcvits_amd import ops
        ctx.save_for_backward(x, w)
        ctx.w_sink, ctx.b_sink = ops._sink(w), ops._sink(b)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        from vcvits_amd import ops
        x, w = ctx.saved_tensors
        dx = dy @ w if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1]:
            dw = dy.t() @ x
            if ctx.w_sink is not None:
                ctx.w_sink[0].add_(dw)
            dw = ops._sunk(ctx.w_sink, dw)
        if ctx.needs_input_grad[2]:
            db = dy.sum(0)
            if ctx.b_sink is not None:
                ctx.b_sink[0].add_(db)
            db = ops._sunk(ctx.b_sink, db)
        return dx, dw, db


def _make_params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(16, 8), (16,), (12, 16), (12,), (4, 12), (4,), (5, 4), (5,)]  # last pair = an unused head
    return [torch.nn.Parameter(torch.randn(s, generator=g) * 0.3) for s in shapes]


def _forward(ps, x):
    h = torch.tanh(_SinkLinear.apply(x, ps[0], ps[1]))
    h = torch.tanh(_SinkLinear.apply(h, ps[2], ps[3]))
    return _SinkLinear.apply(h, ps[4], ps[5])  # ps[6], ps[7] never take part (cf. Generator.cond on the SVC path)


def _sink_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcvits_amd import ops
    from vcvits_amd.light.optim import FlatAdamW
    ps = _make_params(0)
    opt = FlatAdamW(ps, 1e-2, bucket_mb=0.0003)
    for p in opt.params:  # FlatAdamW registers sinks for GPU parameters; same mechanism on CPU tensors here
        ops.register_grad_sink(p, p.grad)
    assert len(opt._buckets) >= 3
    order = []
    orig = opt._launch_bucket

    def rec(b):
        order.append(b["lo"])
        return orig(b)
    opt._launch_bucket = rec
    g = torch.Generator().manual_seed(200 + rank)
    x, y = torch.randn(6, 8, generator=g), torch.randn(6, 4, generator=g)
    results = []
    for frozen in (False, True):  # second pass: the first layer is frozen (toggle_optimizer-style)
        ps[0].requires_grad_(not frozen)
        ps[1].requires_grad_(not frozen)
        opt.zero_grad()
        ((_forward(ps, x) - y) ** 2).mean().backward()
        opt.finish_grad_sync()  # buckets holding untouched parameters are reduced here, in the same order on every rank
        results.append((opt.grad.clone(), bytes(opt._touched)))
    ps[0].requires_grad_(True)
    ps[1].requires_grad_(True)
    out[rank] = (results, order)
    dist.barrier()
    ops.clear_grad_sinks()
    dist.destroy_process_group()


def test_sinks_frozen_and_unused_parameters_two_ranks():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sink_worker, args=(world, port, out), nprocs=world, join=True)
    (res0, order0), (res1, order1) = out[0], out[1]
    assert order0 == order1 and len(order0) >= 6  # identical collective sequence on both ranks, every bucket in every pass
    for (g0, t0), (g1, t1) in zip(res0, res1):
        assert torch.equal(g0, g1) and t0 == t1
    # reference: plain autograd on the union batch
    for pass_idx, frozen in enumerate((False, True)):
        ps = _make_params(0)
        xs, ys = [], []
        for rank in range(world):
            g = torch.Generator().manual_seed(200 + rank)
            xs.append(torch.randn(6, 8, generator=g))
            ys.append(torch.randn(6, 4, generator=g))
        x, y = torch.cat(xs), torch.cat(ys)
        h = torch.tanh(x @ ps[0].t() + ps[1])
        h = torch.tanh(h @ ps[2].t() + ps[3])
        ((h @ ps[4].t() + ps[5] - y) ** 2).mean().backward()
        ref = []
        for i, p in enumerate(reversed(ps)):  # FlatAdamW lays parameters out in reverse registration order
            idx = len(ps) - 1 - i
            dead = idx >= 6 or (frozen and idx < 2)
            ref.append(torch.zeros(p.numel()) if dead or p.grad is None else p.grad.reshape(-1))
        ref = torch.cat(ref)
        assert torch.allclose(res0[pass_idx][0], ref, atol=1e-6, rtol=1e-5), pass_idx
        touched = res0[pass_idx][1]
        # reversed order: [unused b, unused w, b3, w3, b2, w2, b1, w1]
        assert list(touched) == [0, 0, 1, 1, 1, 1, 0 if frozen else 1, 0 if frozen else 1]


def _uneven_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcvits_amd.light.optim import FlatAdamW
    torch.manual_seed(0)
    a, b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)  # `b` is a conditioning path only rank 0's batch exercises
    opt = FlatAdamW(list(a.parameters()) + list(b.parameters()), 1e-2, bucket_mb=0.0001)
    opt.broadcast_parameters()
    g = torch.Generator().manual_seed(300 + rank)
    x = torch.randn(3, 4, generator=g)
    opt.zero_grad()
    y = a(x) + (b(x) if rank == 0 else 0.0)
    y.pow(2).mean().backward()
    opt.finish_grad_sync()
    out[rank] = (bytes(opt._touched), opt.grad.clone(), [r for r in opt._update_ranges()])
    dist.barrier()
    dist.destroy_process_group()


def test_parameter_used_on_one_rank_is_updated_on_all():
    """a parameter one rank used and another did not received the averaged gradient on both, so both must
    step it -- the `touched` flags are OR-ed across the group before the update ranges are built."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_uneven_worker, args=(world, port, out), nprocs=world, join=True)
    (t0, g0, r0), (t1, g1, r1) = out[0], out[1]
    assert t0 == t1 and all(t0)          # every parameter counts as touched on BOTH ranks
    assert torch.equal(g0, g1)           # the same averaged gradient ...
    assert r0 == r1 and len(r0) == 1     # ... and the same single update range on both ranks


# ---- static-graph mode: the host-side used-parameter exchange stops once the ranks agreed for STATIC_AFTER steps ---------
def _static_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcvits_amd.light.optim import FlatAdamW
    FlatAdamW.CHECK_EVERY = 2  # (the periodic one-byte check of the frozen set: every second step here, 64 in production)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))
    extra = torch.nn.Linear(8, 1)  # used from step 5 on by rank 1 only: a change of the frozen set
    opt = FlatAdamW(list(net.parameters()) + list(extra.parameters()), 1e-2, bucket_mb=0.0003)
    opt.broadcast_parameters()
    g = torch.Generator().manual_seed(7 + rank)
    counts, err = [], None
    for step in range(6):
        x = torch.randn(4, 8, generator=g)
        opt.zero_grad()
        y = net(x)
        if step >= 5 and rank == 1:
            y = y + extra(x)
        y.pow(2).mean().backward()
        try:
            opt.finish_grad_sync()
        except RuntimeError as e:
            err = str(e)
            break
        counts.append(opt.flag_exchanges)
    out[rank] = (counts, err, opt._static_set is not None)
    dist.barrier()  # (both ranks raised at the same check: they are still in lock step)
    dist.destroy_process_group()


def test_flag_exchange_stops_after_agreement_and_a_later_change_raises_on_every_rank():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_static_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    from vcvits_amd.light.optim import FlatAdamW
    n = FlatAdamW.STATIC_AFTER
    c0, e0, f0 = out[0]
    c1, e1, f1 = out[1]
    # the per-step exchange ran in the first STATIC_AFTER steps only; afterwards one byte every CHECK_EVERY (= 2 here) steps.
    # Rank 1's set changes at step 5, which is a check step: BOTH ranks are told (rank 0 would otherwise sit in its next
    # collective until the watchdog fires), rank 1's message says it was the one
    assert c0 == c1 == [1, 2, 2, 3, 3] and f0 and f1, (c0, c1)
    assert e0 is not None and "static-graph" in e0 and "on this rank" not in e0, e0
    assert e1 is not None and "static-graph" in e1 and "on this rank" in e1, e1


# ---- a recording pass is not a step: ranks that capture at different times stay in step ----------------------
def _capture_skew_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcvits_amd import ops
    from vcvits_amd.light.optim import FlatAdamW
    FlatAdamW.CHECK_EVERY = 4
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))
    opt = FlatAdamW(net.parameters(), 1e-2, bucket_mb=0.0003)
    opt.broadcast_parameters()
    g = torch.Generator().manual_seed(11 + rank)
    record_at = {0: (3, 9), 1: (6,)}[rank]  # rank 0 records twice (an LRU re-capture), rank 1 once, at other steps
    trace = []

    def one_pass(recording):
        x = torch.randn(4, 8, generator=g)
        opt.zero_grad()
        net(x).pow(2).mean().backward()
        if recording:
            # what GraphedBatch._capture does around its recording pass: the pass runs with CAPTURING set (nothing executes),
            # host-side optimizer bookkeeping is snapshotted and restored, and the first replay counts the step
            snap = opt.static_state()
            ops.CAPTURING[0] = object()
            try:
                opt.finish_grad_sync()
            finally:
                ops.CAPTURING[0] = None
            opt._static_steps, opt._violation, opt.flag_exchanges = snap
            opt._synced = True
            opt.static_check()  # the replay that executes the recorded pass
        else:
            opt.finish_grad_sync()

    for step in range(12):
        one_pass(step in record_at and opt._static_set is not None)
        trace.append((opt._static_steps, opt.flag_exchanges))
    out[rank] = trace
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_that_record_at_different_steps_count_the_same_static_steps():
    """The periodic frozen-set check is a BLOCKING host collective: every rank must reach it at the same optimizer step.  A
    pass that is only recorded into a HIP graph used to count as a step of its own (finish_grad_sync) on top of the replay
    that executes it, so a rank that captured (or re-captured after an LRU eviction) ran one step ahead of its peers."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_capture_skew_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0] == out[1], (out[0], out[1])
    steps = [s for s, _ in out[0]]
    assert steps[-1] == 12 - 2 and steps == sorted(steps)  # frozen after STATIC_AFTER = 2 steps, one count per step since


def test_checkpoint_refused_while_a_violation_is_pending():
    """Between a rank's used-parameter set changing and the next periodic check the ranks may have stepped different
    parameter sets: FlatAdamW.state_dict refuses to produce a checkpoint in that window."""
    from vcvits_amd.light.optim import FlatAdamW
    net = torch.nn.Linear(4, 2)
    opt = FlatAdamW(net.parameters(), 1e-2)
    opt.state_dict()
    opt._violation = True
    with pytest.raises(RuntimeError, match="used-parameter set changed"):
        opt.state_dict()
