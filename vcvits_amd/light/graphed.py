"""HIP-graph replay of the training batch.

The reference's loop (vits/light/vcvits.py:54-183 under Lightning's automatic optimisation) issues, per batch, one generator
pass and one discriminator pass eagerly: here that is ~1,000-1,800 launcher calls through Python autograd per batch whose
shapes, addresses and order repeat from batch to batch, and every configuration was bound -- or within 20 % of being bound
-- by the host issuing them (bench.py: `host_issue_ms_per_step`).  So once a batch shape repeats the WHOLE batch

    zero_grad(G) -> training_step(batch, 0) -> backward -> [gradient all-reduces] -> AdamW(G)
 -> zero_grad(D) -> training_step(batch, 1) -> backward -> [gradient all-reduces] -> AdamW(D)

is recorded into ONE HIP graph (`GraphedBatch`) and replayed: one host call per batch.  `GraphedNoGrad` is the same
machinery for a no-grad callable (the discriminator step's generator pass of the eager loop).

What a recorded launch bakes is an address, and what makes the recorded sequence valid on replay is that everything behind
the addresses is still there and means the same.  That is enforced by construction (`_lib.Capture`), not by convention:

  * inputs are copied into static buffers; outputs / losses are the capture's own tensors (valid until the next replay);
  * every buffer the sequence allocates comes from the graph's private pool, and the sequence re-makes, at every replay,
    everything it derives from the parameters (weight norm, packed weights): caches filled by eager passes are not
    consulted while recording, entries made by the capture are (the discriminators' weights are normalised and packed
    once per recorded batch, as in the eager loop);
  * device tables (weight-norm records, pack jobs, loss items) are written ONCE, eagerly, right after the capture --
    their contents are addresses of the graph's own tensors -- and are held by the graph.  Round 4 recorded copy nodes that
    re-read pageable host memory at every replay instead;
  * every tensor from OUTSIDE the pool that any launcher was handed while recording (parameters, flat gradient / moment
    buffers, cached constant tables) is held by the graph, so its address cannot be recycled under it whatever the eager
    code does with its caches later (VCVITS_CHECK_PTRS=1 lists them with their call sites);
  * the weight-gradient arena of a recorded batch is its own (eager passes between replays cannot move or resize it) and
    is left all-zero by the recorded sequence itself;
  * per-step scalars are not baked: the optimizer's learning rate and step count (bias corrections) are read from an
    8-byte device record refreshed before each replay (vcv_adamw_dev), dropout kernels add a device-side counter bumped
    per replay to their baked host seed (vcv_set_seed_offset_ptr) -- fresh masks every step, the backward kernels of a
    replay regenerate the masks of their own forward;
  * host-side effects of a pass that later code reads are re-applied after each replay: per-parameter step counts, the
    "received a gradient" flags, `module.logged`, derived-weight invalidation;
  * ONE stream: `VCVITS.fit_batch` runs the eager batches, the capture and the replays on the module's own (non-null)
    stream, and the optimizers' gradient hooks are registered under it, so the AccumulateGrad nodes of the parameters live
    on the stream the capture records: the recorded graph is a linear chain.  Round 4 warmed up on the legacy null stream
    and captured on a side stream: every autograd accumulation (LayerNorm / embedding / relative-position parameters: the
    ones without a gradient sink) then forked the graph onto the null stream -- not a capturable stream -- and replays of
    the full model raced (wrong losses from the third replay on, a GPU memory fault once eager batches were interleaved);
  * data parallel (round 6): the batch is recorded as THREE graphs sharing one pool -- [G pass] [AdamW(G) + D pass]
    [AdamW(D)] -- and the bucket all-reduces are issued eagerly between their replays, in the blocking form, on the same
    stream, in the order the eager loop issues them (light/optim.py).  No RCCL kernel is inside a graph, the chain stays
    linear, a rank that replays and a rank that still runs eagerly issue the same collectives in the same order, and the
    host's work per batch is three replays + ~20 collective calls instead of ~1,500 launcher calls.  What this form gives up
    is the overlap of a bucket's all-reduce with the rest of the backward pass (the eager loop's side stream): across eight
    ranks that is the all-reduce time of ~0.5 GB of gradients over xGMI per batch.  Recording starts only after the
    used-parameter set was frozen (FlatAdamW static mode: no host-side flag exchange left in the step);
  * the key of a graph holds the batch shapes, the arithmetic switches and the parameter-storage epoch
    (ops.GRAPH_EPOCH: a rebuilt optimizer, a replaced layer) -- a graph recorded for other storage is never replayed;
    entries are LRU-bounded by count (`MAX_ENTRIES`) AND by the memory their pools hold (`GRAPH_MEM_FRACTION` of the
    device / `VCVITS_GRAPH_BYTES`); an evicted graph releases its pool; a capture that runs out of memory releases every
    recorded graph of the object and is retried when the shape has repeated again (`MAX_OOM_RETRIES` times);
  * the per-launch profiler and the dropout trace of the tests force the eager pass (their events / lists are not
    capturable); any failure while recording disables the graph for that object (eager from then on).
"""
import gc
import os
import sys
from collections import OrderedDict

import torch

from .. import _lib, ops, tuning
from .._lib import lib
from ..model.discriminators._pair import join_streams

ENABLED = [tuning.flag("VCVITS_GRAPHS", True, "HIP-graph replay of repeated launch sequences at all")]
# the whole batch (both optimizer passes and their AdamW steps) as one graph: on by default (VCVITS_BATCH_GRAPHS=0 keeps
# the eager loop with the graphed no-grad generator pass)
BATCH_ENABLED = [tuning.flag("VCVITS_BATCH_GRAPHS", True, "the whole training batch (both passes + AdamW) as one recorded sequence")]
# record batches whose gradient all-reduces span real ranks (in three segments: the collectives stay OUTSIDE the graphs)
DDP_GRAPHS = [tuning.flag("VCVITS_DDP_GRAPHS", True, "record batches whose gradient all-reduces span real ranks")]
MAX_ENTRIES = tuning.integer("VCVITS_GRAPH_ENTRIES", 12, "recorded graphs kept per object (distinct batch shapes), LRU")
# ... and the memory they may hold together: every recorded batch owns a private pool with the whole activation footprint of
# its shape (plus a table arena and a weight-gradient arena), next to the eager working set.  Fraction of the device's
# memory (default 0.4: 115 GB of the MI355X's 288) or VCVITS_GRAPH_BYTES in bytes; least-recently-used graphs go first.
GRAPH_MEM_FRACTION = tuning.number("VCVITS_GRAPH_MEM_FRACTION", 0.4, "share of the device memory the recorded graphs' pools may hold together")
GRAPH_BYTES = tuning.integer("VCVITS_GRAPH_BYTES", 0, "... as an absolute byte budget (0: the fraction)")
MAX_OOM_RETRIES = 3  # captures that ran out of memory before the object gives up recording (other failures: at once)
MAX_COUNTED = 256  # distinct shapes whose repeat counts are remembered

# Capture with the thread-local error mode: under the default ("global") any other thread's event query during the capture is
# an error that kills the process -- and in a data-parallel run torch's RCCL watchdog thread polls the events of the gradient
# all-reduces it is still retiring at about that time.  Only the capturing thread's own calls are policed.
CAPTURE_ERROR_MODE = tuning.text("VCVITS_CAPTURE_ERROR_MODE", "thread_local", "stream-capture error mode of the recordings")


def set_enabled(on):
    ENABLED[0] = bool(on)


def set_batch_enabled(on):
    BATCH_ENABLED[0] = bool(on)


set_step_enabled = set_batch_enabled  # (round-4 name)


def _no_gc_during_capture():
    """Collect garbage NOW and keep the cyclic collector off until the capture is over (returns whether it was on).  A
    collection that starts inside a capture can reach an earlier module's HIP graph (modules sit in reference cycles, so
    their graphs die in the collector, not at `del`), and destroying a graph while a stream is capturing is an error raised
    in a destructor: the process aborts (seen one full-suite run in eight: many modules with graphs in one process)."""
    gc.collect()
    was_on = gc.isenabled()
    gc.disable()
    return was_on


def _shape_key(batch):
    return tuple((k, tuple(v.shape), str(v.dtype), str(v.device)) for k, v in sorted(batch.items()))


class _Recorder:
    """Shared by GraphedNoGrad / GraphedBatch: repeat counting, the LRU of recorded graphs and the capture itself."""

    what = "sequence"

    def __init__(self, warmup):
        self.warmup = int(warmup)
        self.entries, self.counts = OrderedDict(), OrderedDict()
        self.failed = False
        self.replays = self.captures = 0
        self.ooms = 0        # captures that ran out of memory (each one releases every recorded graph and retries later)
        self.evictions = 0   # graphs released for the entry count or the byte budget

    def _lookup(self, key):
        """The entry of `key` (moved to the young end), or None after counting one more sighting; True when the key has now
        been seen more than `warmup` times and should be recorded."""
        ent = self.entries.get(key)
        if ent is not None:
            self.entries.move_to_end(key)
            return ent, False
        n = self.counts[key] = self.counts.get(key, 0) + 1
        self.counts.move_to_end(key)
        while len(self.counts) > MAX_COUNTED:
            self.counts.popitem(last=False)
        return None, n > self.warmup

    @staticmethod
    def _budget(dev):
        if GRAPH_BYTES > 0:
            return GRAPH_BYTES
        try:
            return int(GRAPH_MEM_FRACTION * torch.cuda.get_device_properties(dev).total_memory)
        except Exception:  # noqa: BLE001
            return 1 << 62

    def held_bytes(self):
        return sum(e.get("bytes", 0) for e in self.entries.values())

    def _store(self, key, ent):
        self.entries[key] = ent
        self.counts.pop(key, None)
        budget = self._budget(ent.get("device"))
        # the newest entry always stays (a single shape larger than the budget still replays; it just never has company)
        while len(self.entries) > 1 and (len(self.entries) > max(1, MAX_ENTRIES) or self.held_bytes() > budget):
            _k, old = self.entries.popitem(last=False)
            self._release(old)
            self.evictions += 1

    @staticmethod
    def _release(ent):
        g = ent.pop("graph", None)
        ent.clear()
        del g  # (the CUDAGraph object owns the private pool: destroying it releases the pool's memory)

    def drop(self):
        """Forget every recorded graph (parameter storage moved, module rebuilt)."""
        while self.entries:
            _k, old = self.entries.popitem(last=False)
            self._release(old)
        self.counts.clear()

    def _record(self, dev, body):
        """Run `body(cap)` under a stream capture; returns (graph, cap, seed, result) or None after a failure.  `body` may be
        a LIST of callables: each is recorded into a graph of its own, all sharing one private pool (torch: graphs that share
        a pool must replay in the order they were captured -- GraphedBatch.run does); returns ([graphs], cap, seed, [results])."""
        L = lib()
        torch.cuda.synchronize()
        gc_was_on = _no_gc_during_capture()
        torch.cuda.empty_cache()  # (torch's capture entry does the same: done first so the segment snapshot below is final)
        cap = _lib.Capture(dev)
        seed = torch.zeros(1, dtype=torch.int64, device=dev)
        reserved0 = torch.cuda.memory_reserved(dev)  # (after empty_cache: what the capture reserves on top is its pool)
        graph = torch.cuda.CUDAGraph()
        ops.CAPTURING[0] = cap
        L.vcv_set_seed_offset_ptr(seed.data_ptr())
        # record ON the stream the caller is already working on when that is a real stream (VCVITS.fit_batch runs everything
        # on the module's own stream): autograd's AccumulateGrad nodes run on the stream they were CREATED on, and a node
        # created on another stream forks the recorded graph -- onto the legacy null stream when the eager warm-up ran there,
        # which a capture must never touch (torch warns "may break CUDA graph capture"; on this stack the capture went
        # through and the replays raced: round 4's memory fault).  On the null stream torch's side capture stream is used.
        cur = torch.cuda.current_stream(dev)
        cap_stream = cur if cur.cuda_stream != 0 else None
        try:
            if isinstance(body, (list, tuple)):
                graphs, outs = [], []
                for k, seg in enumerate(body):
                    g = graph if k == 0 else torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=None if k == 0 else graphs[0].pool(), stream=cap_stream,
                                          capture_error_mode=CAPTURE_ERROR_MODE):
                        outs.append(seg(cap))
                    graphs.append(g)
                graph, out = graphs, outs
            else:
                with torch.cuda.graph(graph, stream=cap_stream, capture_error_mode=CAPTURE_ERROR_MODE):
                    out = body(cap)
            torch.cuda.synchronize()
            cap.flush()  # device tables of the recorded launches: written once, before the first replay
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001 -- whatever the capture tripped over: stay eager
            oom = isinstance(e, torch.cuda.OutOfMemoryError) or "out of memory" in str(e).lower()
            if oom and self.ooms < MAX_OOM_RETRIES:
                # transient by nature (the eager working set, other graphs' pools): give back everything this object holds
                # and let the shape be recorded again once it has repeated `warmup` more times -- do not latch `failed`
                self.ooms += 1
                graph = cap = None
                self.drop()
                torch.cuda.empty_cache()
                sys.stderr.write("vcvits_amd: HIP-graph capture of the %s ran out of memory (%d of %d tries); recorded graphs "
                                 "released, running eagerly for now\n" % (self.what, self.ooms, MAX_OOM_RETRIES))
                return None
            self.failed = True
            sys.stderr.write("vcvits_amd: HIP-graph capture of the %s failed (%s: %s); running eagerly\n"
                             % (self.what, type(e).__name__, str(e)[:300]))
            return None
        finally:
            L.vcv_set_seed_offset_ptr(None)
            ops.CAPTURING[0] = None
            # derived weights made while recording hold no data yet (a capture records, it does not run) and belong to the graph
            ops.invalidate_weights()
            if gc_was_on:
                gc.enable()
        self.captures += 1
        self.last_pool_bytes = max(0, torch.cuda.memory_reserved(dev) - reserved0)
        if cap.log:
            sys.stderr.write("vcvits_amd[capture %d]: %s recorded; %d external tensors held, %d device tables\n"
                             % (cap.id, self.what, len(cap.external), len(cap.keep)))
        return graph, cap, seed, out


class GraphedNoGrad(_Recorder):
    """fn(batch) under torch.no_grad(), replayed from a HIP graph once `batch`'s shapes have been seen `warmup` + 1 times."""

    what = "no-grad generator pass"

    def __init__(self, fn, warmup=1):
        super().__init__(warmup)
        self.fn = fn

    def __call__(self, batch, extra=()):
        if (not ENABLED[0] or self.failed or not torch.cuda.is_available() or torch.is_grad_enabled()
                or lib().vcv_prof_active() or ops.DROPOUT_TRACE[0] is not None or ops.CAPTURING[0] is not None):
            return self.fn(batch)  # (inside another capture -- GraphedBatch -- the pass is part of that graph)
        key = _shape_key(batch) + tuple(extra) + (ops.GRAPH_EPOCH[0],)
        ent, record = self._lookup(key)
        if ent is None:
            if not record:
                # eager warm-up: plans, device tables and allocator pools exist before the capture (one pass: the capture --
                # tens of milliseconds -- then falls into a run's warm-up steps, not into its steady state)
                return self.fn(batch)
            ent = self._capture(key, batch)
            if ent is None:
                return self.fn(batch)
        for k, v in batch.items():
            ent["inputs"][k].copy_(v)
        ent["seed"].add_(1)
        ent["graph"].replay()
        self.replays += 1
        return ent["outputs"]

    def _capture(self, key, batch):
        dev = next(iter(batch.values())).device
        static = {k: v.clone() for k, v in batch.items()}
        rec = self._record(dev, lambda cap: self.fn(static))
        if rec is None:
            return None
        graph, cap, seed, out = rec
        ent = {"graph": graph, "inputs": static, "outputs": out, "seed": seed, "cap": cap, "device": dev,
               "bytes": self.last_pool_bytes}
        self._store(key, ent)
        return ent


class GraphedBatch(_Recorder):
    """One batch of the training loop -- both optimizer passes and their AdamW steps -- as ONE HIP graph (module docstring).

    `run(batch, extra)` returns {"g": loss, "d": loss} (static tensors, valid until the next replay) or None: the caller runs
    the batch eagerly (shapes not seen often enough yet, profiler / dropout trace active, graphs switched off, capture
    failed).  Eager batches may be interleaved with replays freely: a replay depends on nothing an eager batch can move."""

    what = "training batch"

    def __init__(self, module, warmup=2):
        super().__init__(warmup)
        self.module = module

    def applicable(self):
        m = self.module
        og, od = m.optim_g, m.optim_d
        if not (ENABLED[0] and BATCH_ENABLED[0] and not self.failed and torch.cuda.is_available() and torch.is_grad_enabled()
                and m.training and og is not None and od is not None and og.grad.is_cuda
                and not lib().vcv_prof_active() and ops.DROPOUT_TRACE[0] is None and ops.CAPTURING[0] is None):
            return False
        from ..model.discriminators._pair import streams
        if streams() > 1:
            # the sub-discriminators spread over several HIP streams are for the EAGER loop only.  Recorded with the forks the
            # batch replays 4 - 5 % faster -- and, for some deals of the chains to the streams, wrong (3 of the 8 deals of three
            # chains to two streams; round robin is not one of them).  Three causes were found and removed: autograd's root
            # gradient read stale (`root_one` below), gradient blocks recycled by the stream that allocated them before the
            # other stream had read them (_pair._Handoff), device-to-device copy nodes (_pair._Halves).  What is left is not
            # in the recording: the dumped graph of a failing deal has every edge it needs (reachability over all node
            # pairs: nothing is unordered against the generator's backward pass, where the first wrong value appears), the
            # allocator history shows no overlap, and the SAME graph replays bit-exactly on one graph queue
            # (DEBUG_HIP_FORCE_GRAPH_QUEUES=1) or two hardware queues (GPU_MAX_HW_QUEUES=2) -- which also removes the gain.
            # profiles/r6_multistream_replay_probe.txt; tools/probes/streams_race_probe.py reproduces every step.
            return False
        for o in (og, od):
            if getattr(o, "_ddp", False):
                # data parallel: only once the used-parameter set is frozen (no host-side flag exchange left in the step) ...
                if o._static_set is None:
                    return False
                # ... and, across REAL ranks, unless switched off (VCVITS_DDP_GRAPHS=0).  No RCCL kernel is recorded: the
                # collectives run eagerly between the replays of three segments
                if o.world > 1 and not DDP_GRAPHS[0]:
                    return False
        return True

    def run(self, batch, extra=()):
        if not self.applicable():
            return None
        m = self.module
        key = _shape_key(batch) + tuple(extra) + (ops.GRAPH_EPOCH[0], id(m.optim_g), id(m.optim_d))
        ent, record = self._lookup(key)
        if ent is None:
            if not record:
                return None
            ent = self._capture(key, batch)
            if ent is None:
                return None
        # the per-step scalars of the two recorded AdamW steps
        deltas = []
        for opt, idx, steps in ((m.optim_g, ent["idx_g"], ent["steps_g"]), (m.optim_d, ent["idx_d"], ent["steps_d"])):
            ps = opt._pstep
            # (over ALL touched parameters, not only the first of each recorded run: eager batches in between may have
            # stepped part of a run, and the run's later parameters would then get the wrong bias correction silently)
            ds = {ps[i] - s0 for i, s0 in zip(idx, steps)}
            if len(ds) > 1:
                # eager batches in between stepped a different parameter set: the recorded runs' relative step counts no
                # longer hold -- forget this graph (it is recorded again when the shape repeats)
                self._release(self.entries.pop(key))
                return None
            deltas.append(ds.pop() if ds else 0)
        for opt, d in zip((m.optim_g, m.optim_d), deltas):
            glr = opt.param_groups[0]["lr"]
            if glr != opt.lr:  # (an external scheduler wrote the group's rate: FlatAdamW.step does the same)
                opt.lr = float(glr)
            if opt.hyper is not None:
                ops.set_hyper(opt.hyper, opt.lr, d)
        for k, v in batch.items():
            ent["inputs"][k].copy_(v)
        ent["seed"].add_(1)
        if ent.get("segments"):
            # [G pass] -> all-reduce G's buckets -> [AdamW(G) + D pass] -> all-reduce D's buckets -> [AdamW(D)]: three replays
            # and the bucket collectives in between, everything on this one stream in order (blocking form: no fork, no join)
            g0, g1, g2 = ent["graph"]
            g0.replay()
            m.optim_g.allreduce_buckets_now()
            g1.replay()
            m.optim_d.allreduce_buckets_now()
            g2.replay()
        else:
            ent["graph"].replay()
        # host-side effects of the two passes
        for opt, touched, idx in ((m.optim_g, ent["touched_g"], ent["idx_g"]), (m.optim_d, ent["touched_d"], ent["idx_d"])):
            ps = opt._pstep
            for i in idx:
                ps[i] += 1
            opt.step_count += 1
            opt._touched[:] = touched
            opt._synced = True
            opt.static_check()  # (data parallel: the periodic "did any rank's used-parameter set change" byte)
        ops.invalidate_weights()  # the recorded AdamW steps wrote the parameters
        m.logged = ent["logged"]
        self.replays += 1
        return ent["losses"]

    def _capture(self, key, batch):
        m = self.module
        og, od = m.optim_g, m.optim_d
        dev = og.grad.device
        static = {k: v.clone() for k, v in batch.items()}
        # the recorded batch's own weight-gradient arena: as large as the eager one has grown to, zero now and left zero by
        # the recorded sequence (every reset zeroes what the pass before it used; the last reset is recorded explicitly)
        eager_buf = ops._ARENA.get("buf")
        n_arena = max(int(eager_buf.numel()) if eager_buf is not None else 0, 1 << 20)
        arena = {"buf": torch.zeros((n_arena,), device=dev, dtype=torch.float32), "off": 0, "need": 0, "on": True}
        for o in (og, od):  # the {lr, step delta} records of the two recorded AdamW steps: allocated (and kept) outside the pool
            if o.hyper is None:
                o.hyper = torch.zeros(2, device=dev, dtype=torch.int32)
        state = {}
        # the root gradient of both backward passes: ONE tensor outside the graph's pool.  autograd's own `ones_like(loss)` is
        # a pool allocation, and with the sub-discriminators on several streams (VCVITS_STREAMS > 1) the replayed branch that
        # read it last saw the block's previous contents instead (a generator-pass loss term: all of MPD's gradients came out
        # 5.13 x too large from the first replay on, deterministically -- tools/probes/streams_race_probe.py)
        root_one = torch.ones((), device=dev, dtype=torch.float32)

        def body(cap):
            losses = {}
            old_arena = ops.arena_swap(arena)
            try:
                for idx, opt in ((0, og), (1, od)):
                    m._toggle(idx)
                    opt.zero_grad()
                    loss = m.training_step(static, 0, idx)
                    loss.backward(root_one)
                    join_streams()  # (VCVITS_STREAMS > 1: side-stream gradient kernels before the all-reduces / AdamW)
                    opt.step()  # (finish_grad_sync inside: the all-reduce joins are part of the graph)
                    losses["g" if idx == 0 else "d"] = loss.detach()
                    state["touched_%s" % ("g" if idx == 0 else "d")] = bytes(opt._touched)
                    state["ranges_%s" % ("g" if idx == 0 else "d")] = list(opt.captured_ranges or [])
                ops.wgrad_arena_reset()
                if ops._ARENA.get("buf") is not arena["buf"]:
                    cap.append(ops._ARENA.get("buf"))  # (grown while recording: allocated in the graph's pool, zeroed by it)
            finally:
                ops.arena_swap(old_arena)
                for p in og.params:
                    p.requires_grad_(True)
                for p in od.params:
                    p.requires_grad_(True)
            return losses

        segments = bool(getattr(og, "_ddp", False) or getattr(od, "_ddp", False))

        def seg_bodies():
            """The same batch as three recordings: the gradient all-reduces (not recorded: FlatAdamW._launch_bucket returns
            at once while recording) run eagerly between their replays."""
            losses = {}
            old_arena = [None]

            def seg_g(cap):
                old_arena[0] = ops.arena_swap(arena)
                m._toggle(0)
                og.zero_grad()
                loss = m.training_step(static, 0, 0)
                loss.backward(root_one)
                join_streams()
                losses["g"] = loss.detach()

            def seg_d(cap):
                og.step()
                state["touched_g"], state["ranges_g"] = bytes(og._touched), list(og.captured_ranges or [])
                m._toggle(1)
                od.zero_grad()
                loss = m.training_step(static, 0, 1)
                loss.backward(root_one)
                join_streams()
                losses["d"] = loss.detach()

            def seg_end(cap):
                try:
                    od.step()
                    state["touched_d"], state["ranges_d"] = bytes(od._touched), list(od.captured_ranges or [])
                    ops.wgrad_arena_reset()
                    if ops._ARENA.get("buf") is not arena["buf"]:
                        cap.append(ops._ARENA.get("buf"))
                finally:
                    restore()
                return losses

            def restore():
                if old_arena[0] is not None:
                    ops.arena_swap(old_arena[0])
                    old_arena[0] = None
                for p in og.params:
                    p.requires_grad_(True)
                for p in od.params:
                    p.requires_grad_(True)
            return [seg_g, seg_d, seg_end], restore

        # the recorded passes advance host-side optimizer state as an executed pass would; the first replay (right after
        # the capture) is that pass's execution
        # (the data-parallel bookkeeping too: static-step count, pending violation, exchange counter -- a rank that records
        # must stay in step with a rank that does not, or they reach the periodic blocking check at different steps)
        snap = [(o, list(o._pstep), o.step_count, o.static_state()) for o in (og, od)]
        if segments:
            bodies, restore = seg_bodies()
            try:
                rec = self._record(dev, bodies)
            finally:
                restore()  # (a failure between two segments must not leave the recording arena / the toggles behind)
            if rec is not None:
                rec = (rec[0], rec[1], rec[2], rec[3][-1])
        else:
            rec = self._record(dev, body)
        for o, ps, sc, st in snap:
            o._pstep, o.step_count = ps, sc
            o._static_steps, o._violation, o.flag_exchanges = st
        if rec is None:
            return None
        graph, cap, seed, losses = rec
        cap.append(arena["buf"])
        ent = {"graph": graph, "inputs": static, "losses": losses, "seed": seed, "cap": cap, "logged": dict(m.logged), "root": root_one,
               "device": dev, "bytes": self.last_pool_bytes + 4 * int(arena["buf"].numel()), "segments": segments,
               "touched_g": state["touched_g"], "touched_d": state["touched_d"],
               "ranges_g": state["ranges_g"], "ranges_d": state["ranges_d"],
               "idx_g": [i for i, t in enumerate(state["touched_g"]) if t],
               "idx_d": [i for i, t in enumerate(state["touched_d"]) if t]}
        # the step count EVERY touched parameter had when the batch was recorded (the recorded AdamW launches bake `st` per
        # run of parameters; a replay is valid only if all of them have advanced by the same amount since)
        ent["steps_g"] = [og._pstep[i] for i in ent["idx_g"]]
        ent["steps_d"] = [od._pstep[i] for i in ent["idx_d"]]
        self._store(key, ent)
        return ent
