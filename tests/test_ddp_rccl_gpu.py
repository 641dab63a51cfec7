"""GPU, 2 ranks over RCCL (skipped where fewer than two GPUs are visible: the round-end test box has one): the averaged
gradients of a data-parallel VocoderGAN batch -- two utterances per rank -- against the single-rank step on the four
utterances together (reference: train.py:99-100, Lightning's strategy="ddp" averages per-rank mean losses), and the
static-graph flag exchange (FlatAdamW: the used-parameter exchange stops after two steps of cross-rank agreement)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg():
    from vcvits_amd import configs
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 16, "upsample_initial_channel": 32, "multi_period_discriminator_periods": [2, 3]})
    cfg["data"]["n_mel_channels"] = 40
    cfg["train"]["segment_size"] = 4096
    return cfg


def _grads(module, batch):
    names = {id(p): n for n, p in module.named_parameters()}
    got = {}

    def probe(idx, opt):
        for p in opt.params:
            got[names[id(p)]] = p.grad.detach().cpu().clone()

    out = module.fit_batch(batch, after_backward=probe)
    return got, (float(out["g"]), float(out["d"]))


def _rank(rank, world, port, state, out):
    import torch.distributed as dist
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "GLOO_SOCKET_IFNAME": "lo",
                       "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from vcvits_amd import synthetic
    from vcvits_amd.light.vcvits import VocoderGAN
    module = VocoderGAN(**_cfg())
    module.load_state_dict(state)
    module = module.to(dev)
    module.configure_optimizers()
    full = synthetic.vocoder_batch(4, 16, segment_size=4096, seed=3)
    mine = {k: v[2 * rank:2 * rank + 2].contiguous().to(dev) for k, v in full.items()}
    grads, losses = _grads(module, mine)
    # three more steps: the used-parameter flags agree on both ranks, so the exchange stops after STATIC_AFTER steps
    for _ in range(3):
        module.fit_batch(mine)
    torch.cuda.synchronize()
    out[rank] = (grads, losses, module.optim_g.flag_exchanges, module.optim_d.flag_exchanges,
                 module.optim_g._static_set is not None)
    dist.barrier()
    from vcvits_amd.light.optim import shutdown_flag_groups
    shutdown_flag_groups()
    dist.destroy_process_group()


def test_two_rank_rccl_gradients_match_the_global_batch(gpu):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI); the 1-GPU box covers the same path with VCVITS_FORCE_DDP=1 "
                    "(tests/test_ddp_overlap_gpu.py) and gloo (tests/test_ddp_gloo.py)")
    import torch.multiprocessing as mp
    from vcvits_amd import synthetic
    from vcvits_amd.light.optim import FlatAdamW
    from vcvits_amd.light.vcvits import VocoderGAN
    torch.manual_seed(0)
    ref = VocoderGAN(**_cfg())
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    ref = ref.to(gpu)
    ref.configure_optimizers()
    full = synthetic.vocoder_batch(4, 16, segment_size=4096, seed=3)
    want, _ = _grads(ref, {k: v.to(gpu) for k, v in full.items()})
    ref.optim_g.close()
    ref.optim_d.close()
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.spawn(_rank, args=(2, _free_port(), state, out), nprocs=2, join=True)
    for rank in (0, 1):
        grads, losses, ex_g, ex_d, frozen = out[rank]
        assert ex_g == FlatAdamW.STATIC_AFTER and ex_d == FlatAdamW.STATIC_AFTER and frozen, (ex_g, ex_d, frozen)
        num = den = 0.0
        for k, w in want.items():
            num += (grads[k].double() - w.double()).pow(2).sum().item()
            den += w.double().pow(2).sum().item()
        # (per-rank batches of two against one batch of four: other tiles, other summation orders, a few leaky-ReLU kinks)
        assert (num / den) ** 0.5 <= 2e-3, (rank, (num / den) ** 0.5)
    g0, g1 = out[0][0], out[1][0]
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k  # both ranks hold the same averaged gradient
