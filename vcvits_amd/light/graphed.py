"""HIP-graph replay of a no-grad pass.

The discriminator step of the reference re-runs the whole generator forward under `torch.no_grad()` and keeps only
`y_hat.detach()` (vits/light/vcvits.py:119,153).  That pass is ~300 launcher calls whose shapes, addresses and order do
not change from step to step, and at bf16 speeds the step is bound by the host issuing them (bench.py:
`host_issue_ms_per_step`), so it is captured once into a HIP graph (torch.cuda.CUDAGraph: the library's launches go to the
capturing stream like torch's own) and replayed: one host call instead of ~300.

What makes the captured sequence valid on replay:
  * inputs are copied into static buffers, outputs are the capture's own tensors (valid until the next replay);
  * every buffer the sequence allocates comes from the graph's private pool, so the addresses baked into device tables
    (weight-norm records, packed-weight jobs) stay the same; host arrays a captured copy reads from are kept alive;
  * derived-weight caches are bypassed while capturing (ops.CAPTURING): the graph always re-normalises and re-packs the
    weights it uses, whatever the cache state was at capture time;
  * dropout: the host seed is baked into the kernel arguments, so the kernels add a device-side counter that is bumped
    before each replay (vcv_set_seed_offset_ptr) -- fresh masks every step; torch's own generators are graph-safe;
  * the per-launch profiler must be idle (its events are not capturable): while it records, the pass runs eagerly.
Any failure while capturing disables the graph for that callable (eager from then on)."""
import os

import gc

import torch

from .. import ops
from .._lib import lib

ENABLED = [os.environ.get("VCVITS_GRAPHS", "1") == "1"]
# the forward + backward pass of each optimizer index as a graph too (GraphedStep below): EXPERIMENTAL, off unless
# VCVITS_STEP_GRAPHS=1 / set_step_enabled(True) -- see the class docstring for what was measured and what is unresolved
STEP_ENABLED = [os.environ.get("VCVITS_STEP_GRAPHS", "0") == "1"]


def set_enabled(on):
    ENABLED[0] = bool(on)


class GraphedNoGrad:
    def __init__(self, fn, warmup=1):
        self.fn, self.warmup = fn, int(warmup)
        self.entries, self.counts = {}, {}
        self.failed = False
        self.replays = 0

    def _key(self, batch, extra):
        return tuple((k, tuple(v.shape), str(v.dtype), str(v.device)) for k, v in sorted(batch.items())) + tuple(extra)

    def __call__(self, batch, extra=()):
        if (not ENABLED[0] or self.failed or not torch.cuda.is_available() or torch.is_grad_enabled()
                or lib().vcv_prof_active() or ops.DROPOUT_TRACE[0] is not None or ops.CAPTURING[0] is not None):
            return self.fn(batch)  # (inside another capture -- GraphedStep -- the pass is part of that graph)
        key = self._key(batch, extra)
        ent = self.entries.get(key)
        if ent is None:
            n = self.counts[key] = self.counts.get(key, 0) + 1
            if n <= self.warmup:
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
        L = lib()
        dev = next(iter(batch.values())).device
        static = {k: v.clone() for k, v in batch.items()}
        seed = torch.zeros(1, dtype=torch.int64, device=dev)
        keep = []
        graph = torch.cuda.CUDAGraph()
        ops.CAPTURING[0] = keep
        L.vcv_set_seed_offset_ptr(seed.data_ptr())
        gc_was_on = _no_gc_during_capture()
        try:
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, capture_error_mode=CAPTURE_ERROR_MODE):
                out = self.fn(static)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001 -- whatever the capture tripped over: stay eager
            self.failed = True
            import sys
            sys.stderr.write("vcvits_amd: HIP-graph capture of the no-grad generator pass failed (%s: %s); running eagerly\n"
                             % (type(e).__name__, str(e)[:200]))
            ops.invalidate_weights()
            return None
        finally:
            L.vcv_set_seed_offset_ptr(None)
            ops.CAPTURING[0] = None
            if gc_was_on:
                gc.enable()
        # derived weights created while capturing hold no data yet (a capture records, it does not run)
        ops.invalidate_weights()
        ent = self.entries[key] = {"graph": graph, "inputs": static, "outputs": out, "seed": seed, "keep": keep}
        return ent



def _no_gc_during_capture():
    """Collect garbage NOW and keep the cyclic collector off until the capture is over (returns whether it was on).  A
    collection that starts inside a capture can reach an earlier module's HIP graph (modules sit in reference cycles, so
    their graphs die in the collector, not at `del`), and destroying a graph while a stream is capturing is an error raised
    in a destructor: the process aborts (seen one full-suite run in eight: many modules with graphs in one process)."""
    gc.collect()
    was_on = gc.isenabled()
    gc.disable()
    return was_on


# Capture with the thread-local error mode: under the default ("global") any other thread's event query during the capture is
# an error that kills the process -- and in a data-parallel run torch's RCCL watchdog thread polls the events of the gradient
# all-reduces it is still retiring at about that time.  Only the capturing thread's own calls are policed.
CAPTURE_ERROR_MODE = __import__("os").environ.get("VCVITS_CAPTURE_ERROR_MODE", "thread_local")


def set_step_enabled(on):
    STEP_ENABLED[0] = bool(on)


class GraphedStep:
    """`zero_grad -> training_step(batch, idx) -> backward` of one optimizer index as ONE HIP graph (the optimizer's own
    step stays eager: two launches whose scalar arguments -- learning rate, bias corrections -- change per step).

    The reference's loop issues this pass eagerly (vits/light/vcvits.py:55-183 under Lightning's automatic optimisation);
    here it is ~450 (generator index) / ~500 (discriminator index) launcher calls through autograd per step whose shapes,
    addresses and order repeat, and every bf16-mode configuration is bound by the host issuing them (and the fp32 one is
    within 20 %: a loaded host turns it host-bound).  Same validity rules as GraphedNoGrad above, plus:
      * the backward pass runs inside the capture (torch's autograd engine keeps the capturing stream current for its
        worker thread; the library's launches take the stream from torch per call);
      * host-side effects of the pass that the optimizer reads are re-applied after each replay: the "received a gradient"
        flags the post-accumulate hooks set (FlatAdamW._touched) -- hooks do not fire in a replay;
      * the weight-gradient arena (ops._ARENA) is sized by the eager warm-up passes, so the capture neither grows nor moves
        it; its re-zeroing is part of the captured zero_grad;
      * dropout in a backward kernel (attention, vcv_dropout's mask regeneration) reads the same device-side seed offset as
        its forward: one bump per replay covers both;
      * single-process only: with a process group the gradient hooks launch collectives -- that pass stays eager.
    Any failure while capturing disables the graph for that index (eager from then on).

    State at the end of round 4 (why it is off by default): tests/test_graphed_gpu.py -- losses step for step and the
    parameters after nine steps equal the eager loop's at reduced width (full model, f32 and bf16 mode; the benchmark's
    vocoder module), and a captured backward regenerates the dropout mask of its own replay.  bench.py with it on: host issue
    time per step 57 -> 25 ms (fp32 vocoder workload), 65 -> 46 ms (48k full model, one rank), but throughput -1 % on the
    GPU-bound workloads (235.4-239.5 vs 236-241.8 utterances/s: eager steps normalise and pack the discriminators' weights
    once per batch and reuse them in the second optimizer index; each graph has to make its own) and +1 % on the 48k one.
    UNRESOLVED: the base-width full model at B = 32 takes a GPU memory fault when eager passes (the profiler's sampled
    steps) are interleaved with replays; without interleaving it runs (372 utterances/s).  Until that is understood the eager
    loop is the product path."""

    def __init__(self, module, warmup=2):
        self.module, self.warmup = module, int(warmup)
        self.entries, self.counts = {}, {}
        self.failed = False
        self.replays = 0

    def applicable(self, opt):
        return (ENABLED[0] and STEP_ENABLED[0] and not self.failed and torch.cuda.is_available() and torch.is_grad_enabled()
                and self.module.training and not getattr(opt, "_ddp", False) and opt.grad.is_cuda
                and not lib().vcv_prof_active() and ops.DROPOUT_TRACE[0] is None and ops.CAPTURING[0] is None)

    def run(self, idx, opt, batch, extra=()):
        """The loss of the pass (a static tensor, valid until the next replay), or None: the caller runs the pass eagerly."""
        if not self.applicable(opt):
            return None
        key = (idx,) + tuple((k, tuple(v.shape), str(v.dtype), str(v.device)) for k, v in sorted(batch.items())) + tuple(extra)
        ent = self.entries.get(key)
        if ent is None:
            n = self.counts[key] = self.counts.get(key, 0) + 1
            if n <= self.warmup:
                return None
            ent = self._capture(key, idx, opt, batch)
            if ent is None:
                return None
        for k, v in batch.items():
            ent["inputs"][k].copy_(v)
        ent["seed"].add_(1)
        ent["graph"].replay()
        opt._synced = False
        opt._touched[:] = ent["touched"]
        self.module.logged = ent["logged"]
        self.replays += 1
        return ent["loss"]

    def _capture(self, key, idx, opt, batch):
        L = lib()
        dev = opt.grad.device
        static = {k: v.clone() for k, v in batch.items()}
        seed = torch.zeros(1, dtype=torch.int64, device=dev)
        keep = []
        graph = torch.cuda.CUDAGraph()
        ops.CAPTURING[0] = keep
        L.vcv_set_seed_offset_ptr(seed.data_ptr())
        gc_was_on = _no_gc_during_capture()
        try:
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, capture_error_mode=CAPTURE_ERROR_MODE):
                opt.zero_grad()
                loss = self.module.training_step(static, 0, idx)
                loss.backward()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001 -- whatever the capture tripped over: stay eager
            self.failed = True
            import sys
            sys.stderr.write("vcvits_amd: HIP-graph capture of the optimizer-%d pass failed (%s: %s); running eagerly\n"
                             % (idx, type(e).__name__, str(e)[:300]))
            ops.invalidate_weights()
            return None
        finally:
            L.vcv_set_seed_offset_ptr(None)
            ops.CAPTURING[0] = None
            if gc_was_on:
                gc.enable()
        ops.invalidate_weights()  # derived weights created while capturing hold no data (a capture records, it does not run)
        ent = self.entries[key] = {"graph": graph, "inputs": static, "loss": loss.detach(), "seed": seed, "keep": keep,
                                   "touched": bytes(opt._touched), "logged": dict(self.module.logged)}
        return ent
