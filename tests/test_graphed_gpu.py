"""GPU: the discriminator step's no-grad generator pass replayed from a HIP graph (vcvits_amd/light/graphed.py; the
reference runs that pass eagerly under torch.no_grad(): vits/light/vcvits.py:119,153).  The replayed pass must be the
eager pass: same losses step for step from identical state; fresh dropout masks on every replay."""
import copy
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

# The whole-batch graphs (graphed.GraphedBatch: both optimizer passes and their AdamW steps as one HIP graph) are the default
# path of fit_batch.  These tests still run in a child pytest process: a GPU fault in a graph replay kills the process, and
# it must not take the session (and every test after it) down; the parent test fails with the child's output instead.
IN_CHILD = os.environ.get("VCVITS_STEP_GRAPH_TESTS_CHILD") == "1"
step_graph = pytest.mark.skipif(not IN_CHILD, reason="runs inside test_step_graph_tests_in_child_process")


def test_step_graph_tests_in_child_process(gpu):
    if IN_CHILD:
        pytest.skip("this is the child")
    env = dict(os.environ, VCVITS_STEP_GRAPH_TESTS_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "graphed_step or captured_backward or graphed_batch"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    assert "8 passed" in out, out[-1500:]


def _cfg(p_dropout):
    from vcvits_amd import configs
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 32, "hidden_channels": 32, "filter_channels": 64, "n_heads": 2,
                         "upsample_initial_channel": 64, "hubert_channels": 48, "gin_channels": 16, "p_dropout": p_dropout,
                         "multi_period_discriminator_periods": [2, 3]})
    cfg["data"]["n_mel_channels"] = 40
    return cfg


def _batches(cfg, n, gpu, with_draws):
    from vcvits_amd import synthetic
    m = cfg["model"]
    out = []
    for i in range(n):
        b = synthetic.full_batch(4, m["hubert_channels"], seed=50 + i)
        if with_draws:
            g = torch.Generator().manual_seed(90 + i)
            b["noise"] = torch.randn(4, m["inter_channels"], 384, generator=g)
            b["ids_slice"] = torch.tensor([5, 100, 17, 200])
        out.append({k: v.to(gpu) for k, v in b.items()})
    return out


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_generator_pass_equals_eager(gpu, dtype):
    from vcvits_amd import ops
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VCVITS
    cfg = _cfg(0.0)
    torch.manual_seed(11)
    ref = VCVITS(**cfg)
    sd = copy.deepcopy(ref.state_dict())
    batches = _batches(cfg, 2, gpu, with_draws=True)
    losses = {}
    ops.set_compute_dtype(dtype)
    try:
        graphed.set_batch_enabled(False)  # the eager loop: its discriminator step's generator pass is the graph under test
        for mode in (False, True):
            graphed.set_enabled(mode)
            torch.manual_seed(12)
            mod = VCVITS(**cfg)
            mod.load_state_dict(sd)
            mod = mod.to(gpu)
            mod.configure_optimizers()
            ls = []
            for i in range(8):
                out = mod.fit_batch(batches[i % 2])
                ls.append((float(out["g"]), float(out["d"])))
            losses[mode] = ls
            if mode:
                g = mod.__dict__["_g_graph"]
                assert not g.failed and g.replays >= 4, (g.failed, g.replays)
            mod.optim_g.close()
            mod.optim_d.close()
    finally:
        graphed.set_enabled(True)
        graphed.set_batch_enabled(True)
        ops.set_compute_dtype("f32")
    tol = 2e-5 if dtype == "f32" else 2e-3  # (weight-gradient atomics / split orders move the parameters in the last bits)
    for (g0, d0), (g1, d1) in zip(losses[False], losses[True]):
        assert abs(g0 - g1) <= tol * abs(g0) and abs(d0 - d1) <= tol * abs(d0), (losses[False], losses[True])


def test_graph_replays_draw_fresh_dropout_masks(gpu):
    """A captured sequence bakes the host seed into its kernel arguments; the device-side seed offset bumped before every
    replay must give each replay its own masks (forward dropout and the attention kernel's dropout)."""
    from vcvits_amd import ops
    from vcvits_amd.light import graphed

    def fn(b):
        y = ops.dropout(b["x"], 0.5, True)
        o, _ = ops.rel_attention(b["q"], b["q"], b["q"], b["ek"], b["ek"], b["mask"], 2, 4, pdrop=0.5, training=True, want_attn=False)
        return y, o

    g = graphed.GraphedNoGrad(fn, warmup=2)
    gen = torch.Generator().manual_seed(3)
    batch = {"x": torch.randn(4, 32, 200, generator=gen), "q": torch.randn(2, 64, 120, generator=gen),
             "ek": torch.randn(9, 32, generator=gen) * 0.1, "mask": torch.ones(2, 120)}
    batch = {k: v.to(gpu) for k, v in batch.items()}
    graphed.set_enabled(True)
    outs = []
    with torch.no_grad():
        for _ in range(7):
            y, o = g(batch)
            outs.append((y.clone(), o.clone()))
    assert not g.failed and g.replays >= 4, (g.failed, g.replays)
    for y, o in outs:
        assert bool(torch.isfinite(o).all())
        kept = (y != 0).float().mean().item()
        assert 0.4 < kept < 0.6, kept
    for i in range(len(outs)):
        for j in range(i + 1, len(outs)):
            assert not torch.equal(outs[i][0], outs[j][0]) and not torch.equal(outs[i][1], outs[j][1]), (i, j)


@step_graph
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_step_equals_eager(gpu, dtype):
    """The whole batch (both optimizer passes and their AdamW steps) replayed from ONE HIP graph (graphed.GraphedBatch)
    against the eager loop: same losses step for step from identical state, the recorded optimizer steps move the
    parameters as the eager ones do."""
    from vcvits_amd import ops
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VCVITS
    cfg = _cfg(0.0)
    torch.manual_seed(11)
    ref = VCVITS(**cfg)
    sd = copy.deepcopy(ref.state_dict())
    batches = _batches(cfg, 2, gpu, with_draws=True)
    losses, params = {}, {}
    ops.set_compute_dtype(dtype)
    try:
        for mode in (False, True):
            graphed.set_step_enabled(mode)
            torch.manual_seed(12)
            mod = VCVITS(**cfg)
            mod.load_state_dict(sd)
            mod = mod.to(gpu)
            mod.configure_optimizers()
            ls = []
            for i in range(9):
                out = mod.fit_batch(batches[i % 2])
                ls.append((float(out["g"]), float(out["d"])))
            losses[mode] = ls
            params[mode] = (mod.optim_g.flat.clone(), mod.optim_d.flat.clone())
            if mode:
                sg = mod.__dict__["_batch_graph"]
                assert not sg.failed and sg.replays >= 7 and sg.captures == 1, (sg.failed, sg.replays, sg.captures)
                assert mod.optim_g.step_count == 9 and mod.optim_d.step_count == 9
            else:
                assert "_batch_graph" not in mod.__dict__ or mod.__dict__["_batch_graph"].replays == 0
            mod.optim_g.close()
            mod.optim_d.close()
    finally:
        graphed.set_step_enabled(True)
        ops.set_compute_dtype("f32")
    tol = 2e-5 if dtype == "f32" else 2e-3  # (weight-gradient atomics / split orders move the parameters in the last bits)
    for (g0, d0), (g1, d1) in zip(losses[False], losses[True]):
        assert abs(g0 - g1) <= tol * abs(g0) and abs(d0 - d1) <= tol * abs(d0), (losses[False], losses[True])
    # parameters: an Adam step moves an element by about the learning rate whatever the gradient's size, so a gradient that
    # differs in its last bits (atomics, split orders) around zero can move it the other way: a few learning rates per element
    lr = float(cfg["train"]["learning_rate"])
    for a, b in zip(params[False], params[True]):
        assert float((a - b).abs().max()) <= 2.5 * lr * 9 + 1e-4 * float(a.abs().max()), float((a - b).abs().max())
    assert losses[True][0] != losses[True][-1]  # the parameters moved


@step_graph
def test_graphed_step_vocoder_workload(gpu):
    """The benchmark's module (VocoderGAN, BASELINE configs[1]) at reduced width: graph steps equal eager steps."""
    from vcvits_amd import configs, synthetic
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VocoderGAN
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 16, "upsample_initial_channel": 32, "multi_period_discriminator_periods": [2, 3]})
    cfg["data"]["n_mel_channels"] = 40
    cfg["train"]["segment_size"] = 4096
    torch.manual_seed(5)
    sd = copy.deepcopy(VocoderGAN(**cfg).state_dict())
    batch = {k: v.to(gpu) for k, v in synthetic.vocoder_batch(2, 16, segment_size=4096, seed=3).items()}
    res = {}
    try:
        for mode in (False, True):
            graphed.set_step_enabled(mode)
            mod = VocoderGAN(**cfg)
            mod.load_state_dict(sd)
            mod = mod.to(gpu)
            mod.configure_optimizers()
            ls = []
            for _ in range(7):
                out = mod.fit_batch(batch)
                ls.append((float(out["g"]), float(out["d"])))
            res[mode] = ls
            if mode:
                sg = mod.__dict__["_batch_graph"]
                assert not sg.failed and sg.replays >= 5, (sg.failed, sg.replays)
            mod.optim_g.close()
            mod.optim_d.close()
    finally:
        graphed.set_step_enabled(True)
    for (g0, d0), (g1, d1) in zip(res[False], res[True]):
        assert abs(g0 - g1) <= 2e-5 * abs(g0) and abs(d0 - d1) <= 2e-5 * abs(d0), (res[False], res[True])


@step_graph
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_batch_interleaved_with_eager_batches(gpu, dtype):
    """Replays and eager batches interleaved (what bench.py's sampled profiler steps do, and what variable-length data
    does): the mixed loop's losses equal the all-eager loop's step for step.  Round 4's step graphs took a GPU memory
    fault in exactly this pattern at full width."""
    from vcvits_amd import ops
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VCVITS
    cfg = _cfg(0.0)
    torch.manual_seed(11)
    sd = copy.deepcopy(VCVITS(**cfg).state_dict())
    batches = _batches(cfg, 2, gpu, with_draws=True)
    odd = {k: (v[:3].contiguous() if v.dim() > 0 and v.shape[0] == 4 else v) for k, v in batches[0].items()}  # another shape
    losses = {}
    ops.set_compute_dtype(dtype)
    try:
        for mode in ("eager", "mixed"):
            torch.manual_seed(12)
            mod = VCVITS(**cfg)
            mod.load_state_dict(sd)
            mod = mod.to(gpu)
            mod.configure_optimizers()
            ls = []
            for i in range(14):
                b = odd if i in (6, 11) else batches[i % 2]
                graphed.set_step_enabled(mode == "mixed" and i % 4 != 1)  # every fourth batch of the mixed loop is eager
                out = mod.fit_batch(b)
                ls.append((float(out["g"]), float(out["d"])))
            losses[mode] = ls
            if mode == "mixed":
                sg = mod.__dict__["_batch_graph"]
                assert not sg.failed and sg.replays >= 5, (sg.failed, sg.replays)
            assert mod.optim_g.step_count == 14 and mod.optim_d.step_count == 14
            mod.optim_g.close()
            mod.optim_d.close()
    finally:
        graphed.set_step_enabled(True)
        ops.set_compute_dtype("f32")
    tol = 5e-5 if dtype == "f32" else 4e-3
    for (g0, d0), (g1, d1) in zip(losses["eager"], losses["mixed"]):
        assert abs(g0 - g1) <= tol * abs(g0) and abs(d0 - d1) <= tol * abs(d0), (losses["eager"], losses["mixed"])


@step_graph
def test_graphed_batch_follows_a_learning_rate_change_and_a_rebuilt_optimizer(gpu):
    """The recorded AdamW steps read the learning rate from device memory (a scheduler step between replays takes effect)
    and a rebuilt optimizer (new flat buffers) never replays a graph recorded for the old storage."""
    from vcvits_amd import configs, synthetic
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VocoderGAN
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 16, "upsample_initial_channel": 32, "multi_period_discriminator_periods": [2, 3]})
    cfg["data"]["n_mel_channels"] = 40
    cfg["train"]["segment_size"] = 4096
    cfg["train"]["lr_decay"] = 0.5
    torch.manual_seed(5)
    sd = copy.deepcopy(VocoderGAN(**cfg).state_dict())
    batch = {k: v.to(gpu) for k, v in synthetic.vocoder_batch(2, 16, segment_size=4096, seed=3).items()}
    res = {}
    try:
        for mode in (False, True):
            graphed.set_step_enabled(mode)
            mod = VocoderGAN(**cfg)
            mod.load_state_dict(sd)
            mod = mod.to(gpu)
            mod.configure_optimizers()
            ls = []
            for i in range(12):
                if i in (5, 6):
                    mod.on_epoch_end()  # ExponentialLR: the second call halves the rate
                if i == 9:
                    st = (mod.optim_g.state_dict(), mod.optim_d.state_dict())
                    mod.configure_optimizers()  # new flat buffers; moments restored, parameters carried over
                    mod.optim_g.load_state_dict(st[0])
                    mod.optim_d.load_state_dict(st[1])
                out = mod.fit_batch(batch)
                ls.append((float(out["g"]), float(out["d"])))
            res[mode] = (ls, mod.optim_g.flat.clone(), mod.optim_g.lr)
            if mode:
                sg = mod.__dict__["_batch_graph"]
                assert not sg.failed and sg.replays >= 1, (sg.failed, sg.replays)
            mod.optim_g.close()
            mod.optim_d.close()
    finally:
        graphed.set_step_enabled(True)
    assert res[False][2] == res[True][2] and res[True][2] < float(cfg["train"]["learning_rate"])
    for (g0, d0), (g1, d1) in zip(res[False][0], res[True][0]):
        assert abs(g0 - g1) <= 5e-5 * abs(g0) and abs(d0 - d1) <= 5e-5 * abs(d0), (res[False][0], res[True][0])
    lr = float(cfg["train"]["learning_rate"])
    assert float((res[False][1] - res[True][1]).abs().max()) <= 2.5 * lr * 12 + 1e-4 * float(res[False][1].abs().max())


@step_graph
def test_graphed_batch_records_the_bucket_all_reduces(gpu):
    """Data parallel: the recorded batch is three graphs with the bucket all-reduces issued eagerly between their replays, and
    the replayed batches follow the eager loop -- on the one-GPU box with a forced ONE-rank RCCL group (VCVITS_FORCE_DDP=1: the
    hooks, the bucket bookkeeping, RCCL's one-rank kernels)."""
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    from vcvits_amd import configs, synthetic
    from vcvits_amd.light import graphed
    from vcvits_amd.light.optim import shutdown_flag_groups
    from vcvits_amd.light.vcvits import VocoderGAN
    import socket
    with socket.socket() as sk:  # a free port (a fixed one collided with a lingering listener once)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update({"VCVITS_FORCE_DDP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cfg = configs.base()
        cfg["model"].update({"inter_channels": 16, "upsample_initial_channel": 32, "multi_period_discriminator_periods": [2, 3]})
        cfg["data"]["n_mel_channels"] = 40
        cfg["train"]["segment_size"] = 4096
        torch.manual_seed(5)
        sd = copy.deepcopy(VocoderGAN(**cfg).state_dict())
        batch = {k: v.to(gpu) for k, v in synthetic.vocoder_batch(2, 16, segment_size=4096, seed=3).items()}
        res = {}
        # eager, then the recorded batch: three graphs, the bucket all-reduces issued eagerly between their replays
        for mode in (False, "segments"):
            graphed.set_step_enabled(bool(mode))
            mod = VocoderGAN(**cfg)
            mod.load_state_dict(sd)
            mod = mod.to(gpu)
            mod.configure_optimizers()
            assert mod.optim_g._ddp and mod.optim_d._ddp and len(mod.optim_d._buckets) >= 1
            ls = []
            for _ in range(10):
                out = mod.fit_batch(batch)
                ls.append((float(out["g"]), float(out["d"])))
            res[mode] = ls
            if mode:
                sg = mod.__dict__["_batch_graph"]
                # (recording waits for the frozen used-parameter set: two steps of agreement, then three sightings)
                assert not sg.failed and sg.replays >= 4, (sg.failed, sg.replays)
                ent = next(iter(sg.entries.values()))
                assert ent["segments"] and len(ent["graph"]) == 3
                # the eager passes before the recording left the order their hooks issued the buckets in
                assert sorted(mod.optim_d._bucket_order) == list(range(len(mod.optim_d._buckets)))
            mod.optim_g.close()
            mod.optim_d.close()
        for mode in ("segments",):
            for (g0, d0), (g1, d1) in zip(res[False], res[mode]):
                assert abs(g0 - g1) <= 5e-5 * abs(g0) and abs(d0 - d1) <= 5e-5 * abs(d0), (mode, res[False], res[mode])
    finally:
        graphed.set_step_enabled(True)
        os.environ.pop("VCVITS_FORCE_DDP", None)
        shutdown_flag_groups()
        dist.destroy_process_group()


@step_graph
def test_captured_backward_regenerates_the_replays_dropout_mask(gpu):
    """Inside a captured pass the attention backward (and vcv_dropout's) must regenerate the mask of ITS replay: the
    device-side seed offset read by the forward.  A replay with offset k equals the eager pass run with host seed S + k."""
    from vcvits_amd import ops
    from vcvits_amd._lib import lib
    gen = torch.Generator().manual_seed(4)
    q = torch.randn(2, 64, 120, generator=gen).to(gpu).requires_grad_(True)
    x = torch.randn(4, 32, 200, generator=gen).to(gpu).requires_grad_(True)
    ek = (torch.randn(1, 9, 32, generator=gen) * 0.1).to(gpu).requires_grad_(True)
    mask = torch.ones(2, 120, device=gpu)
    gy = torch.randn(2, 64, 120, generator=gen).to(gpu)

    def run():
        o, _ = ops.rel_attention(q, q, q, ek, ek, mask, 2, 4, pdrop=0.3, training=True, want_attn=False)
        y = ops.dropout(x, 0.3, True)
        gq, gek, gx = torch.autograd.grad([o, y], [q, ek, x], [gy, torch.ones_like(y)])
        return o, y, gq, gek, gx

    seeds = []
    real_next = ops.next_seed
    L = lib()
    off = torch.zeros(1, dtype=torch.int64, device=gpu)
    try:
        def recording():
            s = real_next()
            seeds.append(s)
            return s
        ops.replace("next_seed", recording)
        run()  # warm-up (plans, allocator)
        seeds.clear()
        graph = torch.cuda.CUDAGraph()
        L.vcv_set_seed_offset_ptr(off.data_ptr())
        torch.cuda.synchronize()
        from vcvits_amd.light import graphed
        gc_was_on = graphed._no_gc_during_capture()  # (no collection of an earlier test's graphs inside this capture)
        try:
            with torch.cuda.graph(graph):
                outs = run()
        finally:
            if gc_was_on:
                import gc
                gc.enable()
        L.vcv_set_seed_offset_ptr(None)
        baked = list(seeds)
        assert len(baked) == 2, baked  # attention, dropout
        for k in (1, 5):
            off.fill_(k)
            graph.replay()
            torch.cuda.synchronize()
            got = [t.clone() for t in outs]
            it = iter((s + k) % (1 << 64) for s in baked)
            ops.replace("next_seed", lambda: next(it))
            want = run()
            for a, b in zip(got, want):
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), float((a - b).abs().max())
            assert float((got[1] == 0).float().mean()) > 0.2
    finally:
        ops.replace("next_seed", real_next)
        L.vcv_set_seed_offset_ptr(None)

