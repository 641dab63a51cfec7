"""GPU: one whole reference batch (generator step + discriminator step, AdamW included) on the
HIP path against the CPU oracle trainer, at reduced widths so the oracle finishes in seconds.
Checks the two losses and the updated parameters of both optimizers."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def small_cfg():
    from vcvits_amd import configs
    c = configs.base()
    c["model"].update({"inter_channels": 16, "hidden_channels": 16, "filter_channels": 32, "n_heads": 2,
                       "n_layers": 2, "upsample_initial_channel": 32, "hubert_channels": 24, "gin_channels": 8,
                       "p_dropout": 0.0, "multi_period_discriminator_periods": [2, 3]})
    c["data"].update({"n_mel_channels": 40, "hubert_channels": 24, "n_speakers": 8})
    c["train"]["segment_size"] = 4096
    return c


def _run(module, trainer, batch, gpu, tol_loss=2e-4, tol_grad=3e-4):
    """One batch on both sides; compares the two losses and EVERY parameter gradient of both
    passes.  (Parameters after the step are not compared element-wise: Adam's first steps move
    each weight by lr*sign(g), which turns fp32 rounding noise on analytically-zero gradients --
    e.g. the key bias of a softmax attention -- into +-lr differences.  AdamW itself is checked in
    tests/test_elementwise_gpu.py and through the second batch's losses here.)"""
    lc = trainer.batch(batch)
    names = {id(p): n for n, p in module.named_parameters()}
    grads = {}

    def probe(idx, opt):
        for p in opt.params:
            grads[names[id(p)]] = p.grad.detach().cpu().clone()

    out = module.fit_batch({k: v.to(gpu) for k, v in batch.items()}, after_backward=probe)
    for a, b, n in zip((out["g"], out["d"]), lc, ("loss_g", "loss_d")):
        assert abs(float(a) - float(b)) <= tol_loss * abs(float(b)) + 1e-5, (n, float(a), float(b))
    ref = dict(trainer.grads_g)
    ref.update(trainer.grads_d)
    assert set(ref) == set(grads), set(ref) ^ set(grads)
    gscale = max(float(v.abs().max()) for v in ref.values())
    for k, b in ref.items():
        err = (grads[k].double() - b.double()).abs().max().item()
        bound = tol_grad * b.abs().max().item() + 1e-6 * gscale
        assert err <= bound, (k, err, bound)


def test_vocoder_gan_batch(gpu):
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import synthetic
    from vcvits_amd.light.vcvits import VocoderGAN
    torch.manual_seed(0)
    cfg = small_cfg()
    module = VocoderGAN(**cfg)
    trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, [2, 3], vocoder_only=True)
    module = module.to(gpu)
    module.configure_optimizers()
    _run(module, trainer, synthetic.vocoder_batch(2, 16, segment_size=4096, seed=3), gpu)
    # second batch: losses now depend on both AdamW updates of the first one
    _run(module, trainer, synthetic.vocoder_batch(2, 16, segment_size=4096, seed=4), gpu, tol_loss=1e-3,
         tol_grad=2e-2)


def test_vocoder_gan_batch_resblock2(gpu):
    """model.resblock = "2" (vits/model/modules.py:225-247: ResBlock2, two dilated convs per block) through a whole batch:
    both losses and every parameter gradient of the G and D passes against the oracle trainer."""
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import synthetic
    from vcvits_amd.light.vcvits import VocoderGAN
    from vcvits_amd.model.modules import ResBlock2
    torch.manual_seed(1)
    cfg = small_cfg()
    cfg["model"].update({"resblock": "2", "resblock_dilation_sizes": [[1, 3], [2, 6], [3, 12]]})
    module = VocoderGAN(**cfg)
    assert all(isinstance(b, ResBlock2) for b in module.net_g.resblocks)
    assert "net_g.resblocks.11.convs.1.weight_g" in module.state_dict()
    trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, [2, 3], vocoder_only=True)
    module = module.to(gpu)
    module.configure_optimizers()
    _run(module, trainer, synthetic.vocoder_batch(2, 16, segment_size=4096, seed=6), gpu)


def test_vocoder_gan_batch_spectral_norm(gpu):
    """use_spectral_norm=True through a whole batch: the discriminators' power-iteration vectors advance with each of the
    four forwards a batch runs them through (d(y), d(y_hat) in the generator step and again in the discriminator step),
    losses and every gradient against the oracle trainer; the vectors after the batch are the oracle's."""
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import synthetic
    from vcvits_amd.light.vcvits import VocoderGAN
    torch.manual_seed(0)
    cfg = small_cfg()
    cfg["model"]["use_spectral_norm"] = True
    module = VocoderGAN(**cfg)
    assert "net_period_d.discriminators.1.convs.0.weight_orig" in module.state_dict()
    trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, [2, 3], vocoder_only=True)
    module = module.to(gpu)
    module.configure_optimizers()
    _run(module, trainer, synthetic.vocoder_batch(2, 16, segment_size=4096, seed=3), gpu)
    sd = module.state_dict()
    n = 0
    for k, v in trainer.sd.items():
        if k.endswith(".weight_u") or (k.endswith(".weight_v") and k[:-1] + "orig" in trainer.sd):
            assert (sd[k].cpu() - v).abs().max().item() <= 1e-4, k
            n += 1
    assert n == 2 * ((7 + 2 * 6) + 7)  # MPD: DiscS (7 convs) + two DiscP (6 each); MSD: its first DiscS only
    _run(module, trainer, synthetic.vocoder_batch(2, 16, segment_size=4096, seed=4), gpu, tol_loss=1e-3,
         tol_grad=2e-2)


def test_vocoder_gan_batch_filter_length_1024(gpu):
    """A config with another STFT size (filter_length 1024, hop 256, win 1024: the mel loss of the generator step runs the
    generic STFT kernels forward and backward, the decoder upsamples by 256): losses and every gradient vs the oracle."""
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import synthetic
    from vcvits_amd.light.vcvits import VocoderGAN
    torch.manual_seed(2)
    cfg = small_cfg()
    cfg["data"].update({"filter_length": 1024, "hop_length": 256, "win_length": 1024})
    cfg["model"].update({"upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4]})
    module = VocoderGAN(**cfg)
    trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, [2, 3], vocoder_only=True)
    module = module.to(gpu)
    module.configure_optimizers()
    _run(module, trainer, synthetic.vocoder_batch(2, 16, segment_size=4096, hop=256, seed=5), gpu)


def test_full_vcvits_batch(gpu):
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import synthetic
    from vcvits_amd.light.vcvits import VCVITS
    torch.manual_seed(1)
    cfg = small_cfg()
    module = VCVITS(**cfg)
    # the flow's `post` convs are zero-initialised (modules.py:314-315): randomise them or the
    # coupling layers are identity and the test is vacuous
    with torch.no_grad():
        for n, p in module.named_parameters():
            if ".post." in n:
                p.normal_(0.0, 0.05)
    trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, [2, 3], vocoder_only=False)
    module = module.to(gpu)
    module.configure_optimizers()
    batch = synthetic.full_batch(2, 24, t_y=40, t_x=22, seed=5)
    batch["y_wav_lengths"][1] = 30 * 512
    batch["y_wav_values"][1, :, 30 * 512:] = 0
    gen = torch.Generator().manual_seed(9)
    batch["noise"] = torch.randn(2, 16, 40, generator=gen)
    batch["ids_slice"] = torch.tensor([3, 11])
    batch["sid"] = batch["sid"] % 8
    _run(module, trainer, batch, gpu)


def test_vocoder_gan_batch_without_grad_sinks(gpu, monkeypatch):
    """The autograd-accumulation path (VCVITS_GRAD_SINK=0: kernels hand temporaries to autograd instead of
    adding into the optimizer's flat gradient buffer) gives the same gradients."""
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import ops, synthetic
    from vcvits_amd.light.vcvits import VocoderGAN
    monkeypatch.setenv("VCVITS_GRAD_SINK", "0")
    ops.clear_grad_sinks()
    torch.manual_seed(0)
    cfg = small_cfg()
    module = VocoderGAN(**cfg)
    trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, [2, 3], vocoder_only=True)
    module = module.to(gpu)
    module.configure_optimizers()
    assert not ops._GRAD_SINKS
    _run(module, trainer, synthetic.vocoder_batch(2, 16, segment_size=4096, seed=3), gpu)


def test_vocoder_gan_batch_ddp_buckets_single_rank(gpu, monkeypatch):
    """Bucketed all-reduce machinery on (a 1-rank RCCL group): every bucket must see all of its parameters
    reported ready -- by autograd hooks or by the gradient sinks' notify -- exactly once per backward pass, and
    the averaged gradients equal the plain ones."""
    import torch.distributed as dist
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import synthetic
    from vcvits_amd.light.vcvits import VocoderGAN
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    monkeypatch.setenv("VCVITS_FORCE_DDP", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29631")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        torch.manual_seed(0)
        cfg = small_cfg()
        module = VocoderGAN(**cfg)
        trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, [2, 3], vocoder_only=True)
        module = module.to(gpu)
        opts = module.configure_optimizers()
        assert opts and all(o._ddp for o in (module.optim_g, module.optim_d))
        counts = []
        for o in (module.optim_g, module.optim_d):
            orig = o._launch_bucket

            def wrapped(b, _orig=orig, _o=o):
                counts.append((id(_o), b["lo"], b["ready"], b["n"]))
                return _orig(b)
            o._launch_bucket = wrapped
        _run(module, trainer, synthetic.vocoder_batch(2, 16, segment_size=4096, seed=3), gpu)
        # ready never exceeds the bucket's parameter count (a parameter reported twice would launch the
        # all-reduce before its last contribution), most buckets complete from the backward pass itself, and
        # no bucket is reduced twice
        assert counts and all(r <= n for _, _, r, n in counts), counts
        assert sum(r == n for _, _, r, n in counts) >= len(counts) - 2, counts
        assert len(set((i, lo) for i, lo, _, _ in counts)) == len(counts), "a bucket was reduced twice"
    finally:
        dist.destroy_process_group()
