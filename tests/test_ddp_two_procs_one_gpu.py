"""GPU, TWO PROCESSES on the ONE GPU of the test box, gloo between them (RCCL refuses two ranks on one device): the
data-parallel training loop across real processes -- used-parameter agreement, the frozen set, and the batch recorded as
three HIP graphs with the bucket all-reduces issued between their replays (light/graphed.py: GraphedBatch, `segments`) --
run the way a multi-GPU job runs it, minus the transport.  What it pins: both ranks record at the same batch and replay
from then on, every rank holds bit-identical parameters after every phase (eager, recording, replays), and the recorded
run follows the eager data-parallel run of the same two processes."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, recorded, path):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "GLOO_SOCKET_IFNAME": "lo"})
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vcvits_amd import configs, ops, synthetic
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VocoderGAN
    ops.set_deterministic(True)  # (no atomics: the two runs compared below differ by the arithmetic of nothing)
    graphed.set_batch_enabled(bool(recorded))
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 16, "upsample_initial_channel": 32, "multi_period_discriminator_periods": [2, 3]})
    cfg["data"]["n_mel_channels"] = 40
    cfg["train"]["segment_size"] = 4096
    torch.manual_seed(5)
    module = VocoderGAN(**cfg).to(dev)
    module.configure_optimizers()
    assert module.optim_g._ddp and module.optim_g.world == world
    batches = [{k: v.to(dev) for k, v in synthetic.vocoder_batch(2, 16, segment_size=4096, seed=10 * i + rank).items()}
               for i in range(3)]
    losses, sums = [], []
    for i in range(14):
        out = module.fit_batch(batches[i % 3])
        losses.append((float(out["g"]), float(out["d"])))
        sums.append((float(module.optim_g.flat.double().sum()), float(module.optim_d.flat.double().sum())))
    bg = module.__dict__.get("_batch_graph")
    ent = next(iter(bg.entries.values())) if (bg is not None and bg.entries) else None
    res = dict(losses=losses, sums=sums, replays=bg.replays if bg is not None else 0,
               captures=bg.captures if bg is not None else 0, failed=bool(bg.failed) if bg is not None else None,
               segments=bool(ent["segments"]) if ent is not None else None,
               static=module.optim_g._static_set is not None and module.optim_d._static_set is not None,
               flag_exchanges=(module.optim_g.flag_exchanges, module.optim_d.flag_exchanges),
               g=module.optim_g.flat.cpu(), d=module.optim_d.flat.cpu())
    torch.save(res, path)
    dist.barrier()
    from vcvits_amd.light.optim import shutdown_flag_groups
    shutdown_flag_groups()
    dist.destroy_process_group()


def _run(recorded, tmp_path):
    port = _free_port()
    procs, paths = [], []
    for rank in range(2):
        path = str(tmp_path / ("r%d_%d.pt" % (int(recorded), rank)))
        paths.append(path)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), str(rank), "2", str(port), str(int(recorded)), path],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung (ranks out of step with their collectives?)")
        outs.append(o)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(p, weights_only=False) for p in paths]


def test_two_processes_record_the_same_batch_and_stay_bit_identical(gpu, tmp_path):
    eager = _run(False, tmp_path)
    rec = _run(True, tmp_path)
    for run in (eager, rec):
        a, b = run
        # one set of parameters, whatever the phase: the ranks average the same buckets and apply the same update
        assert a["sums"] == b["sums"], (a["sums"], b["sums"])
        assert torch.equal(a["g"], b["g"]) and torch.equal(a["d"], b["d"])
        assert a["static"] and b["static"] and a["flag_exchanges"] == b["flag_exchanges"]
    a, b = rec
    assert a["failed"] is False and b["failed"] is False
    assert a["captures"] == b["captures"] == 1 and a["segments"] and b["segments"], (a["captures"], b["captures"], a["segments"])
    assert a["replays"] == b["replays"] and a["replays"] >= 3, (a["replays"], b["replays"])  # (same batch recorded on both)
    assert eager[0]["replays"] == 0
    # the recorded run against the eager data-parallel run (deterministic mode: the same arithmetic in the same order)
    for r in (0, 1):
        for (g0, d0), (g1, d1) in zip(eager[r]["losses"], rec[r]["losses"]):
            assert abs(g0 - g1) <= 2e-5 * abs(g0) and abs(d0 - d1) <= 2e-5 * abs(d0), (eager[r]["losses"], rec[r]["losses"])
    lr = 2e-4
    for k in ("g", "d"):
        diff = float((eager[0][k] - rec[0][k]).abs().max())
        assert diff <= 2.5 * lr * 14 + 1e-4 * float(eager[0][k].abs().max()), diff


if __name__ == "__main__":
    _worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
