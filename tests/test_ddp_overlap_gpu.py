"""GPU: the data-parallel gradient buckets of the DISCRIMINATOR pass launch their all-reduce progressively during the
backward pass (1-rank RCCL group, VCVITS_FORCE_DDP=1): with the weight-norm backward split per sub-discriminator, a
sub-discriminator's parameter gradients are final -- and its bucket goes out -- as soon as its own backward is done.
Reference behaviour: DDP's bucketed overlap, /root/reference/train.py:99-100."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_discriminator_buckets_overlap_backward(gpu, monkeypatch):
    import torch.distributed as dist
    from vcvits_amd import configs, ops, synthetic
    from vcvits_amd.light.vcvits import VocoderGAN
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    monkeypatch.setenv("VCVITS_FORCE_DDP", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29641")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(0)
        cfg = configs.base()  # full-width discriminators: 9 + 5 sub-discriminators, ~100 M parameters
        cfg["model"].update({"inter_channels": 32, "upsample_initial_channel": 64})
        module = VocoderGAN(**cfg).to(gpu)
        module.configure_optimizers()
        opt = module.optim_d
        assert opt._ddp and len(opt._buckets) >= 8
        events = []   # (GEMM launches issued so far, bucket bytes) at every bucket launch
        orig = opt._launch_bucket

        def counting(b):
            events.append((sum(ops.LAUNCH_COUNTS.values()), 4 * (b["hi"] - b["lo"])))
            return orig(b)
        opt._launch_bucket = counting
        batch = synthetic.vocoder_batch(2, 32, seed=3, device=gpu)
        module.fit_batch(batch)             # warm-up (caches, arena)
        events.clear()
        # discriminator pass only, instrumented
        module._toggle(1)
        opt.zero_grad()
        loss = module.training_step(batch, 0, 1)
        start = sum(ops.LAUNCH_COUNTS.values())
        loss.backward()
        end = sum(ops.LAUNCH_COUNTS.values())
        opt.finish_grad_sync()
        total = sum(b for _, b in events)
        assert total == 4 * opt.numel
        span = end - start
        early = sum(b for at, b in events if at - start <= 0.75 * span)
        # at least half of the gradient bytes are on the wire before the last quarter of the backward GEMM launches
        assert early >= 0.5 * total, (early / total, [(round((at - start) / span, 2), b >> 20) for at, b in events])
        # and the launches are spread out: not everything in one burst
        assert len({at for at, _ in events}) >= 6
    finally:
        dist.destroy_process_group()
