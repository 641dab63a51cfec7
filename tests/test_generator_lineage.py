"""The decoder's conv_pre / conv_post in both published forms.  The reference pulls its generator from torch.hub
(synthesizer_svc.py:59, "vtuber-plan/hifi-gan:v0.3.1" -- absent offline): the VITS lineage keeps both
layers plain and conv_post bias-free, the original HiFi-GAN lineage weight-norms both and keeps the bias.  Either
state_dict must load into the product Generator and run on the HIP path like the oracle's restatement."""
import numpy as np
import pytest
import torch

from golden_util import fill_state_dict, keys_shapes_of

ARGS = (16, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 4, 2], 64, [16, 16, 4, 4])
HIFIGAN = dict(conv_pre_weight_norm=True, conv_post_weight_norm=True, conv_post_bias=True)


def test_state_dict_of_either_lineage_loads():
    from vcvits_amd.model.generator import Generator
    plain, hub = Generator(*ARGS), Generator(*ARGS, **HIFIGAN)
    kp, kh = [k for k, _ in keys_shapes_of(plain)], [k for k, _ in keys_shapes_of(hub)]
    assert "conv_post.bias" not in kp and "conv_pre.weight" in kp
    assert {"conv_pre.weight_g", "conv_pre.weight_v", "conv_post.weight_g", "conv_post.weight_v", "conv_post.bias"} <= set(kh)
    sd = fill_state_dict(keys_shapes_of(hub), 7)
    wrapper = torch.nn.Module()
    wrapper.dec = Generator(*ARGS)  # built in the VITS form, as SynthesizerSVC builds it
    wrapper.load_state_dict({"dec." + k: v for k, v in sd.items()})  # strict
    assert [k for k, _ in keys_shapes_of(wrapper.dec)] == kh
    assert torch.equal(wrapper.dec.conv_post.bias, sd["conv_post.bias"])
    wrapper.load_state_dict({"dec." + k: v for k, v in plain.state_dict().items()})  # and back
    assert [k for k, _ in keys_shapes_of(wrapper.dec)] == kp
    # a layer that already has the right form keeps its parameter objects (optimizers hold them)
    before = wrapper.dec.conv_pre.weight
    wrapper.load_state_dict({"dec." + k: v for k, v in plain.state_dict().items()})
    assert wrapper.dec.conv_pre.weight is before


@pytest.mark.gpu
@pytest.mark.parametrize("bf16_infer", [False, True])
def test_hifigan_lineage_vs_oracle(gpu, bf16_infer):
    from oracle import vits_oracle as O
    from vcvits_amd import ops
    from vcvits_amd.model.generator import Generator
    # (the 16-bit-activation kernels take the configs' channel counts, not the reduced ones: 48k widths there)
    args = (128, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 4, 2], 512, [16, 16, 4, 4]) if bf16_infer else ARGS
    gen = Generator(*args)
    sd = fill_state_dict(keys_shapes_of(Generator(*args, **HIFIGAN)), 13)
    gen.load_state_dict(sd)
    gen = gen.to(gpu)
    rng = np.random.default_rng(6)
    T = 128 if bf16_infer else 9
    x = torch.from_numpy(rng.standard_normal((2, args[0], T)).astype(np.float32))
    sdo = {"g." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xc = x.clone().requires_grad_(True)
    yc = O.generator_forward(sdo, "g", xc)
    if bf16_infer:  # the 16-bit-activation inference pass takes the same layers (bias of conv_post in its last launch)
        ops.set_compute_dtype("bf16")
        try:
            count = lambda: ops.LAUNCH_COUNTS["bf16io"] + 2 * ops.LAUNCH_COUNTS.get("pair_fused", 0)  # (a fused pair = two convs)
            before = count()
            with torch.no_grad():
                yg = gen(x.to(gpu))
            assert count() - before == 4 + 72
        finally:
            ops.set_compute_dtype("f32")
        sig = yc.detach().pow(2).mean().sqrt().item()
        err = (yg.cpu().double() - yc.detach().double()).pow(2).mean().sqrt().item()
        assert sig >= 0.1 and err <= 1e-3, (sig, err)  # north_star's bf16 bound on the waveform
        return
    r = torch.from_numpy(rng.standard_normal((2, 1, T * 512)).astype(np.float32))
    (yc * r).sum().backward()
    xg = x.clone().to(gpu).requires_grad_(True)
    yg = gen(xg)
    (yg * r.to(gpu)).sum().backward()

    def close(name, a, b, tol=1e-4):
        a, b = a.detach().double().cpu(), b.detach().double()
        assert (a - b).abs().max().item() <= tol * b.abs().max().item() + 2e-6, name

    close("y", yg, yc)
    close("dx", xg.grad, xc.grad)
    for n, p in gen.named_parameters():
        close("d" + n, p.grad, sdo["g." + n].grad, tol=2e-4)
