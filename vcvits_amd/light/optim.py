"""Flat-buffer AdamW + exponential LR + data-parallel gradient averaging.

MI355X-first replacement for what the reference delegates to Lightning
(vits/light/vcvits.py:247-263: two torch.optim.AdamW + ExponentialLR; train.py:99-100:
strategy="ddp"): all parameters of one optimizer live in ONE contiguous fp32 buffer (and so do
their gradients and both moments), so the optimizer step is a single streaming kernel launch
and the gradient all-reduce is a handful of large RCCL collectives over contiguous bucket views
-- launched from post-accumulate-grad hooks as soon as a bucket is complete, i.e. overlapped
with the rest of the backward pass.
"""
import os

import torch
import torch.distributed as dist

from .. import ops, tuning

# gloo side channels for the host-side "received a gradient" flag exchange, one per set of ranks, shared by every optimizer
# over that set (optim_g and optim_d of one model; a rebuilt optimizer) and destroyed by shutdown_flag_groups()
_FLAG_GROUPS = {}


def _flag_group(process_group):
    ranks = tuple(dist.get_process_group_ranks(process_group if process_group is not None else dist.group.WORLD))
    g = _FLAG_GROUPS.get(ranks)
    if g is None:
        g = _FLAG_GROUPS[ranks] = dist.new_group(ranks=list(ranks), backend="gloo")
    return g


def shutdown_flag_groups():
    """Destroy the cached gloo side channels (before dist.destroy_process_group())."""
    for g in _FLAG_GROUPS.values():
        try:
            dist.destroy_process_group(g)
        except Exception:
            pass
    _FLAG_GROUPS.clear()


# A RECORDED batch (light/graphed.py) is three graphs -- [G pass] [AdamW(G) + D pass] [AdamW(D)] -- and the bucket all-reduces are
# issued EAGERLY between their replays, blocking form, on the same stream: nothing of RCCL is inside a graph, the chain of
# replays and collectives is linear, and the host issues three replays and ~20 collectives per batch.  (Round 5's one-graph
# forms -- the collectives recorded in the blocking form, or on torch's communication stream with forks and joins -- measured
# level with this one on a forced one-rank group and crash inside RCCL's capture path once the sub-discriminators run on
# several streams; they are gone.)
FORCE_DDP = tuning.flag("VCVITS_FORCE_DDP", False, "keep the bucket hooks / collectives in a ONE-rank group (exercises the RCCL path on a 1-GPU box)")
DDP_STATIC = tuning.flag("VCVITS_DDP_STATIC", True, "freeze the used-parameter set after two steps of cross-rank agreement (no per-step host exchange)")
GRAD_SINK = tuning.flag("VCVITS_GRAD_SINK", True, "kernels add parameter gradients straight into the optimizer's flat buffer")

class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay 0.01 by default) over a flat buffer.
    Parameters are re-pointed at views of the buffer; `.grad` of every parameter is a permanent
    view of the flat gradient buffer (autograd accumulates into it in place).

    A `torch.optim.Optimizer` (one parameter group), so what consumes the reference's `configure_optimizers`
    (vits/light/vcvits.py:247-263 returns two torch.optim.AdamW to Lightning, which type-checks them) accepts it:
    `param_groups[0]["lr"]` is the live learning rate (an external scheduler may write it), `step(closure)` runs the
    closure first as Lightning's automatic optimisation does, `zero_grad()` keeps the permanent gradient views."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, bucket_mb=32,
                 process_group=None):
        params = [p for p in params]
        if not params:
            raise ValueError("FlatAdamW: empty parameter list")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        dev = params[0].device
        # reverse registration order ~ the order gradients become ready in backward, so a bucket
        # (a contiguous range of the buffer) completes early and its all-reduce overlaps the rest
        self.params = list(reversed(params))
        n = sum(p.numel() for p in self.params)
        self.numel = n
        self.flat = torch.empty(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                view = self.flat[off:off + k].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[off:off + k].view(p.shape)
                self.offsets.append(off)
                off += k
        self.lr, self.base_lr = lr, lr
        self.betas, self.eps, self.weight_decay = tuple(betas), eps, weight_decay
        self.step_count = 0
        self.hyper = None  # device record {lr, step delta} of optimizer steps recorded into a HIP graph (step())
        self.captured_ranges = None
        # (the reference reads optim_g.param_groups[0]['lr'], vcvits.py:122: torch's own group dict, kept in step with self.lr)
        # ---- data parallel ----
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self._works = []
        self._buckets = []
        self._bucket_order, self._order_now = None, []
        self._synced = False
        # VCVITS_FORCE_DDP=1 keeps the bucket hooks / collectives active in a 1-rank group (used to exercise
        # the RCCL path on a single-GPU box)
        self._ddp = self.world > 1 or (tuning.live_flag("VCVITS_FORCE_DDP") and dist.is_initialized())
        self._flag_pg = None
        # Static-graph mode (default; torch DDP's `static_graph` contract): once STATIC_AFTER consecutive steps had every
        # rank report the SAME set of parameters with a gradient, the set is frozen and the per-step flag exchange -- a
        # blocking host collective, i.e. a cross-rank host barrier twice per batch -- is skipped; a later step whose local
        # set differs raises (the other ranks no longer take part in an exchange).  VCVITS_DDP_STATIC=0 exchanges every step.
        self._static_ok = tuning.live_flag("VCVITS_DDP_STATIC")
        self._static_set = None
        self._static_steps = 0
        self._violation = False
        self._agree_steps = 0
        self.flag_exchanges = 0  # (observable: tests / bench)
        if self._ddp:
            self._make_buckets(int(bucket_mb * 1024 * 1024 / 4))
            # the per-parameter "received a gradient" flags are OR-ed across ranks on the HOST (finish_grad_sync): over
            # RCCL that would be a device round trip -- a host sync in the middle of the step (measured: -5 % on one rank)
            # -- so GPU groups get a gloo side channel (over the SAME ranks, shared between optimizers) for this small
            # reduction; CPU (gloo) groups use their own
            self._flag_pg = _flag_group(process_group) if dist.get_backend(process_group) == "nccl" else process_group
        # One post-accumulate hook per parameter: marks the parameter as "received a gradient in this pass"
        # (torch.optim.AdamW skips parameters whose .grad is None -- no decay, no moment update, no step count)
        # and, under data parallelism, counts its bucket down.  The handles are kept so a rebuilt optimizer can
        # detach the old one (close()).
        self._touched = bytearray(len(self.params))
        self._pstep = [0] * len(self.params)
        self._hook_handles = [p.register_post_accumulate_grad_hook(self._make_hook(i))
                              for i, p in enumerate(self.params)]
        # gradient sinks: the conv / weight-norm backward kernels add parameter gradients straight into the
        # flat buffer (ops.register_grad_sink) instead of handing autograd a temporary to accumulate.  The
        # backward then returns None for the parameter; autograd still evaluates its AccumulateGrad node (no
        # kernel) and fires the post-accumulate hook above, after the producing kernel was enqueued.
        if dev.type == "cuda" and tuning.live_flag("VCVITS_GRAD_SINK"):
            for p in self.params:
                ops.register_grad_sink(p, p.grad)
        if dev.type == "cuda":
            ops.register_param_region(self.flat)  # packed copies of plain conv weights: cached per optimizer step, batched

    # -- gradient buckets ------------------------------------------------------------------------
    def _make_buckets(self, cap):
        start, count, bucket_of = 0, 0, []
        cur_first = 0
        for i, p in enumerate(self.params):
            count += p.numel()
            bucket_of.append(len(self._buckets))
            last = i == len(self.params) - 1
            if count >= cap or last:
                end = self.offsets[i] + p.numel()
                self._buckets.append({"lo": start, "hi": end, "n": i - cur_first + 1, "ready": 0, "i": len(self._buckets)})
                start, count, cur_first = end, 0, i + 1
        self._bucket_of = bucket_of

    def _make_hook(self, i):
        touched = self._touched
        if not self._ddp:
            def hook(_p):
                touched[i] = 1
            return hook

        def hook(_p):
            touched[i] = 1
            b = self._buckets[self._bucket_of[i]]
            b["ready"] += 1
            if b["ready"] == b["n"]:
                self._launch_bucket(b)
        return hook

    def close(self):
        """Detach this optimizer from its parameters: removes the hooks and the gradient sinks (call before
        building a replacement over the same parameters, or the old buckets keep launching all-reduces)."""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        for p in self.params:
            ops.unregister_grad_sink(p)
        ops.unregister_param_region(self.flat)

    def bucket_views(self):
        """The bucket all-reduces of one backward pass as (view of the flat gradient buffer) in the order the last EAGER pass
        issued them (hook order, then the leftovers of finish_grad_sync): what a segmented recorded batch issues between
        its replays, so that a rank that replays and a rank that still runs eagerly line their collectives up."""
        order = self._bucket_order if self._bucket_order and sorted(self._bucket_order) == list(range(len(self._buckets))) \
            else list(range(len(self._buckets)))
        return [self.grad[self._buckets[i]["lo"]:self._buckets[i]["hi"]] for i in order]

    def allreduce_buckets_now(self):
        """Average every bucket across the group, blocking form on the current stream (segmented replay)."""
        if not self._ddp:
            return
        avg = dist.get_backend(self.pg) == "nccl"  # (gloo has no AVG: sum, then scale -- as _launch_bucket does)
        for view in self.bucket_views():
            dist.all_reduce(view, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.pg)
            if not avg:
                view.div_(self.world)

    def _launch_bucket(self, b):
        view = self.grad[b["lo"]:b["hi"]]
        backend = dist.get_backend(self.pg)
        if ops.CAPTURING[0] is not None:
            return  # (recording: the all-reduces run eagerly between the replays of the three segments -- GraphedBatch.run)
        self._order_now.append(b["i"])
        # a bucket may hold gradients of sub-discriminators that ran on DIFFERENT side streams (VCVITS_STREAMS > 1), and the
        # collective orders itself after the CURRENT stream only: make that one wait for the others first
        from ..model.discriminators._pair import join_streams
        join_streams()
        if backend == "nccl":
            work = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
            self._works.append((work, None))
        else:
            work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works.append((work, view))

    def finish_grad_sync(self):
        """Wait for the bucket all-reduces of this backward pass (buckets whose parameters got no
        gradient in this pass are reduced here so every rank stays in step)."""
        if not self._ddp or self._synced:
            return
        self._synced = True  # once per backward pass (zero_grad re-arms it)
        for b in self._buckets:
            if b["ready"] != b["n"]:
                self._launch_bucket(b)
            b["ready"] = 0
        for work, view in self._works:
            work.wait()
            if view is not None:
                view.div_(self.world)
        self._works = []
        if ops.CAPTURING[0] is None:  # (a recording pass issues no collectives: it must not erase the eager order)
            self._bucket_order, self._order_now = self._order_now, []
        # A parameter one rank used and another did not (a batch-dependent conditioning path) still received the
        # averaged gradient everywhere: every rank must apply the same update, so the "received a gradient" flags are
        # OR-ed across the group (torch DDP reduces its used-parameter bitmap for the same reason).
        if self._ddp:  # (also in a forced 1-rank group, so that the single-GPU tests run this collective)
            if self._static_set is not None:
                # frozen set: no per-step exchange.  A rank whose local set changes must not raise on its own -- its peers
                # have no exchange left in the step and would sit in the next collective until the RCCL watchdog fires --
                # so the violation is remembered and surfaced on EVERY rank at the next periodic check (static_check).
                if bytes(self._touched) != self._static_set:
                    self._violation = True
                # (a pass that is only being RECORDED into a HIP graph is not a step yet: the first replay executes it and
                # counts it -- counting here too made a rank that captured drift one step ahead of a rank that did not, and
                # ranks that reach the blocking check at different steps deadlock against each other's collectives)
                if ops.CAPTURING[0] is None:
                    self.static_check()
                return
            n = len(self._touched)
            # one MAX all-reduce of [flags, 1 - flags]: the first half is the OR over ranks, the second half NOT(AND) --
            # OR == AND means every rank saw the same set
            local = list(self._touched)
            flags = torch.tensor(local + [1 - f for f in local], dtype=torch.uint8)  # host tensor, gloo: no device sync
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self._flag_pg)
            self.flag_exchanges += 1
            out = flags.tolist()
            self._touched[:] = bytes(out[:n])
            same = all(o == 1 - a for o, a in zip(out[:n], out[n:]))
            self._agree_steps = self._agree_steps + 1 if same else 0
            if self._static_ok and self._agree_steps >= self.STATIC_AFTER:
                self._static_set = bytes(self._touched)

    STATIC_AFTER = 2  # steps of cross-rank agreement before the used-parameter set is frozen
    # frozen set: every this many steps the ranks exchange one "my set changed" byte (host side, gloo: ~50 us).  Until the
    # check a rank whose set changed steps a different parameter set than its peers, so the window is kept short and a
    # checkpoint taken inside it is refused (state_dict); graph replays re-apply the RECORDED set and cannot see a change.
    CHECK_EVERY = 8

    def static_check(self, steps=1):
        """Count `steps` optimizer steps under the frozen used-parameter set; every CHECK_EVERY of them all ranks exchange one
        byte over the host-side channel and ALL raise if any rank saw its set change (light/graphed.py calls this for the
        steps a graph replay stands for)."""
        if not self._ddp or self._static_set is None:
            return
        before = self._static_steps // self.CHECK_EVERY
        self._static_steps += int(steps)
        if self._static_steps // self.CHECK_EVERY == before:
            return
        flag = torch.tensor([1 if self._violation else 0], dtype=torch.uint8)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self._flag_pg)
        self.flag_exchanges += 1
        if int(flag.item()):
            raise RuntimeError("FlatAdamW: on at least one rank the set of parameters that received a gradient changed after "
                               "it was frozen (static-graph mode)%s; set VCVITS_DDP_STATIC=0 for models whose "
                               "used-parameter set varies from step to step" % (" -- on this rank" if self._violation else ""))

    # -- optimizer ---------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        self.grad.zero_()
        self._synced = False
        self._touched[:] = bytes(len(self._touched))  # in place: the hooks hold this object
        if self.grad.is_cuda:
            ops.wgrad_arena_reset()  # temporaries of the previous backward pass are dead by now
        for b in self._buckets:
            b["ready"] = 0

    def _update_ranges(self):
        """[(lo, hi, step)] of the flat buffer to update in this step: maximal runs of parameters that received
        a gradient in this pass and share a step count (torch.optim.AdamW keeps a step per parameter and skips
        parameters without a gradient, e.g. a conditioning layer the forward never fed).  One run -- one launch
        -- when every parameter took part."""
        ranges = []
        for i, p in enumerate(self.params):
            if not self._touched[i]:
                continue
            self._pstep[i] += 1
            lo, hi, st = self.offsets[i], self.offsets[i] + p.numel(), self._pstep[i]
            if ranges and ranges[-1][1] == lo and ranges[-1][2] == st:
                ranges[-1] = (ranges[-1][0], hi, st, ranges[-1][3])
            else:
                ranges.append((lo, hi, st, i))  # (i: the run's first parameter -- its step count stands for the run's)
        return ranges

    def step(self, closure=None):
        loss = None
        if closure is not None:  # (Lightning's automatic optimisation: zero_grad + training_step + backward live in the closure)
            with torch.enable_grad():
                loss = closure()
        glr = self.param_groups[0]["lr"]
        if glr != self.lr:  # an external scheduler (torch.optim.lr_scheduler / Lightning) wrote the group's rate
            self.lr = float(glr)
        self.finish_grad_sync()
        self.step_count += 1
        if not self.flat.is_cuda:
            raise RuntimeError("FlatAdamW.step: parameters are not on the GPU (no CPU fallback)")
        ranges = self._update_ranges()
        if ops.CAPTURING[0] is not None:
            # recorded into a HIP graph (light/graphed.py): the learning rate and the step count come from a device record
            # the host refreshes before each replay; `captured_ranges` lets the replay work out the step delta
            if self.hyper is None:
                self.hyper = torch.zeros(2, device=self.flat.device, dtype=torch.int32)
            for lo, hi, st, _i in ranges:
                ops.adamw_step_dev(self.flat[lo:hi], self.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                                   self.betas, self.eps, self.weight_decay, self.hyper, st)
            self.captured_ranges = [(i, st) for _lo, _hi, st, i in ranges]
        else:
            for lo, hi, st, _i in ranges:
                ops.adamw_step(self.flat[lo:hi], self.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr,
                               self.betas, self.eps, self.weight_decay, st)
        # the kernel wrote the parameters through raw pointers: drop weights derived from them
        ops.invalidate_weights(self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.numel)
        return loss

    def set_lr(self, lr):
        self.lr = float(lr)
        self.param_groups[0]["lr"] = self.lr

    def set_epoch(self, epoch, gamma, reference_stack=None):
        """Closed form of ExponentialLR after `epoch` epoch-end steps from the base rate, in the same mode as the
        ExponentialLR class below (reference_stack: the first epoch-end step of a fresh run does not decay)."""
        if reference_stack is None:
            reference_stack = ExponentialLR.DEFAULT_REFERENCE_STACK
        e = max(epoch - 1, 0) if reference_stack else epoch
        self.set_lr(self.base_lr * (gamma ** e))

    def static_state(self):
        """(static steps counted, violation pending, flag exchanges): what a graph capture snapshots and restores around
        its recording pass (light/graphed.py) and what the data-parallel tests compare across ranks."""
        return self._static_steps, self._violation, self.flag_exchanges

    def state_dict(self):
        if self._violation:
            raise RuntimeError("FlatAdamW.state_dict: this rank's used-parameter set changed after it was frozen and the ranks "
                               "have not compared notes yet (next periodic check): its parameters may have stepped differently "
                               "from its peers', so a checkpoint now could be inconsistent across ranks")
        return {"step": self.step_count, "lr": self.lr, "exp_avg": self.exp_avg.clone(),
                "exp_avg_sq": self.exp_avg_sq.clone(), "param_steps": list(self._pstep)}

    def load_state_dict(self, sd):
        # (a restored run re-learns its used-parameter set: the frozen one belonged to the run that saved)
        self._static_set, self._agree_steps, self._static_steps, self._violation = None, 0, 0, False
        self.step_count = int(sd["step"])
        self.set_lr(sd["lr"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        ps = sd.get("param_steps")
        self._pstep = list(ps) if ps is not None and len(ps) == len(self.params) else [self.step_count] * len(self.params)

    def write_flat(self, fn):
        """Run fn(self.flat) -- a raw write into the parameter buffer -- and drop everything derived from the old
        values (cached weight-norm results / packed weights validate on torch version counters, which a write
        through the flat view does not bump for the per-parameter views)."""
        fn(self.flat)
        if self.flat.is_cuda:
            ops.invalidate_weights(self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.numel)

    def broadcast_parameters(self, src=0):
        if self._ddp:
            self.write_flat(lambda flat: dist.broadcast(flat, src=src, group=self.pg))


class ExponentialLR:
    """torch.optim.lr_scheduler.ExponentialLR over a FlatAdamW, as the reference builds it
    (vits/light/vcvits.py:258-261: one scheduler per optimizer, `last_epoch` re-seated to current_epoch - 1,
    stepped by Lightning at every epoch end).  Chainable form: step() multiplies the optimizer's CURRENT rate by
    gamma, so after a resume the rate continues from whatever the restored optimizer state holds (and from the
    base rate when that state was dropped -- exactly what the reference does)."""

    # VCVITS_LR_REFERENCE_STACK=0 / hparams train.lr_reference_stack=False select the torch >= 2.2 behaviour
    DEFAULT_REFERENCE_STACK = tuning.flag("VCVITS_LR_REFERENCE_STACK", True, "ExponentialLR as in the reference's pinned torch 2.0 (first epoch-end step does not decay)")

    def __init__(self, optimizer, gamma, last_epoch=-1, reference_stack=None):
        """reference_stack=True (default) follows the scheduler of the stack the reference was written for: its
        requirements pin lightning==2.0.0 (requirements.txt) and torch==2.0.0 (requirements_torch.txt:1),
        whose `ExponentialLR.get_lr` returns the rate unchanged when `last_epoch == 0`: after the reference re-seats
        `last_epoch = current_epoch - 1 = -1` on a fresh run (vcvits.py:258-261), the first epoch-end step therefore does
        NOT decay, and the rate after e >= 1 epochs is base * gamma^(e - 1).  False: torch >= 2.2 (`_is_initial`; the torch
        installed HERE is 2.10): every step decays, base * gamma^e.  None: the process default (environment
        VCVITS_LR_REFERENCE_STACK, or `train.lr_reference_stack` in the hparams via VCVITS.configure_optimizers)."""
        self.optimizer, self.gamma = optimizer, float(gamma)
        if reference_stack is None:
            reference_stack = ExponentialLR.DEFAULT_REFERENCE_STACK
        self.reference_stack = bool(reference_stack)
        self.last_epoch = last_epoch + 1  # torch's constructor performs the initial step: rate unchanged
        self._last_lr = [optimizer.lr]

    def step(self):
        self.last_epoch += 1
        if not (self.reference_stack and self.last_epoch == 0):
            self.optimizer.set_lr(self.optimizer.lr * self.gamma)
        self._last_lr = [self.optimizer.lr]

    def get_last_lr(self):
        return list(self._last_lr)

    def state_dict(self):
        return {"gamma": self.gamma, "last_epoch": self.last_epoch, "_last_lr": list(self._last_lr)}

    def load_state_dict(self, sd):
        self.gamma = float(sd.get("gamma", self.gamma))
        self.last_epoch = int(sd["last_epoch"])
        self._last_lr = list(sd.get("_last_lr", self._last_lr))
