"""GPU: the batched weight pack (conv_pack.hip: every packed-weight buffer of a module tree in one launch) writes, bit for
bit, the buffers the per-launch pack kernels of conv_pk.hip / conv_x3.hip write -- all three element kinds (split fp32,
fp32, bf16) and all three weight views (forward, flipped stride-1 data gradient, phased ConvTranspose / strided data
gradient) -- and a training step that replays its packs gives the same losses and gradients as one that packs lazily."""
import copy
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(ops, x, w, y, *, mode, stride=1, pad=0, dil=1):
    from vcvits_amd._lib import VcvConvArgs, ptr
    a = VcvConvArgs()
    B, C, Tin = x.shape
    a.x, a.w, a.y = ptr(x), ptr(w), ptr(y)
    if mode == "fwd":
        M, _, K = w.shape
        Tout = y.shape[2]
        a.B, a.G, a.Cg, a.Mg = B, 1, C, M
        a.Tin, a.Tout, a.P, a.K = Tin, Tout, 1, K
        a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q, a.a_mode = stride, dil, -pad, 1, 0, 1, Tout, 0
    elif mode == "dgrad":  # x = dy [B, M, Tout], w [M, C, K] viewed flipped, y = dx [B, C, Tin]
        M, Cc, K = w.shape
        a.B, a.G, a.Cg, a.Mg = B, 1, M, Cc
        a.Tin, a.Tout, a.P, a.K = Tin, y.shape[2], 1, K
        a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q, a.a_mode = 1, dil, pad - (K - 1) * dil, 1, 0, 1, y.shape[2], 0
    else:  # convT forward: w [Cin, Cout, K], phased
        Cin, M, K = w.shape
        Tout = y.shape[2]
        a.B, a.G, a.Cg, a.Mg = B, 1, C, M
        a.Tin, a.Tout, a.P, a.K = Tin, Tout, 1, K
        a.a_mode = 1
        a.s, a.dj, a.off, a.os, a.oo, a.phases = 1, -1, 0, stride, -pad, stride
        a.Q = (Tout - 1 + pad) // stride + 1
    a.alpha, a.slope = 1.0, 0.1
    return a


CASES = [  # mode, B, C, M, T, K, stride, pad, dil
    ("fwd", 2, 128, 128, 512, 11, 1, 5, 1),
    ("fwd", 2, 512, 1024, 300, 5, 3, 2, 1),
    ("fwd", 2, 32, 32, 2048, 7, 1, 3, 1),
    ("fwd", 2, 130, 100, 333, 5, 1, 2, 1),
    ("dgrad", 2, 256, 256, 256, 7, 1, 9, 3),
    ("convT", 2, 128, 64, 512, 4, 2, 1, 1),
    ("convT", 2, 256, 128, 128, 16, 8, 4, 1),
]


@pytest.mark.parametrize("family", ["x3", "pk", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-C%d-M%d-T%d-K%d-s%d" % (c[0], c[2], c[3], c[4], c[5], c[6]))
def test_batched_pack_equals_per_launch_pack(gpu, family, case):
    from vcvits_amd import _lib, ops
    from vcvits_amd._lib import VcvPackJob, ptr, stream
    L = _lib.lib()
    mode, B, C, M, T, K, s, pad, dil = case
    rng = np.random.default_rng(K * 1000 + C)
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32)).to(gpu)
    if mode == "fwd":
        x, w = t(B, C, T), t(M, C, K)
        y = torch.empty(B, M, ops.conv_out_len(T, K, s, pad, dil), device=gpu)
    elif mode == "dgrad":
        x, w = t(B, M, T), t(M, C, K)  # dy, w
        y = torch.empty(B, C, T + (K - 1) * dil - 2 * pad, device=gpu)
    else:
        x, w = t(B, C, T), t(C, M, K)
        y = torch.empty(B, M, ops.convT_out_len(T, K, s, pad), device=gpu)
    a = _args(ops, x, w, y, mode=mode, stride=s, pad=pad, dil=dil)
    flip = 1 if mode == "dgrad" else 0
    L.vcv_conv_x3_set_all(1)
    try:
        plan_fn, run_fn, job_fn = {"x3": (L.vcv_conv_x3_plan, L.vcv_conv_x3_run, L.vcv_conv_x3_pack_job),
                                   "pk": (L.vcv_conv_pk_plan, L.vcv_conv_pk_run, L.vcv_conv_pk_pack_job),
                                   "bf16": (L.vcv_conv_bf16_plan, L.vcv_conv_bf16_run, L.vcv_conv_bf16_pack_job)}[family]
        plan = (ctypes.c_int64 * 3)()
        if plan_fn(ctypes.byref(a), flip, plan) != 0:
            pytest.skip("this family does not take the launch")
        lazy = torch.full((plan[0],), float("nan"), device=gpu)
        scratch = torch.empty((max(int(plan[1]), 1),), device=gpu)
        _lib.check(run_fn(ctypes.byref(a), ptr(lazy), ptr(scratch), flip, 0, stream()), "run")
        y1 = y.clone()
        jobs = (VcvPackJob * 1)()
        assert job_fn(ctypes.byref(a), flip, ctypes.byref(jobs[0])) == 0
        batched = torch.full((plan[0],), float("nan"), device=gpu)
        jobs[0].w, jobs[0].wp = w.data_ptr(), batched.data_ptr()
        table = torch.empty((64,), device=gpu)
        _lib.check(L.vcv_pack_many(jobs, 1, ptr(table), stream()), "vcv_pack_many")
        torch.cuda.synchronize()
        assert torch.equal(lazy.view(torch.int32), batched.view(torch.int32)), "packed buffers differ"
        # and a launch that trusts the batched pack (pack_valid = 1) gives the same output
        y.fill_(float("nan"))
        _lib.check(run_fn(ctypes.byref(a), ptr(batched), ptr(scratch), flip, 1, stream()), "run")
        assert torch.equal(y, y1)
    finally:
        L.vcv_conv_x3_set_all(0)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_step_with_replayed_packs_equals_lazy_packs(gpu, dtype):
    from vcvits_amd import configs, ops, synthetic
    from vcvits_amd.light.vcvits import VocoderGAN
    torch.manual_seed(0)
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 64, "upsample_initial_channel": 128, "multi_period_discriminator_periods": [2, 5]})
    state = copy.deepcopy(VocoderGAN(**cfg).state_dict())
    batches = [synthetic.vocoder_batch(2, 64, seed=3 + i) for i in range(3)]
    res = {}
    ops.set_compute_dtype(dtype)
    try:
        for batch_packs in (True, False):
            ops._PACK_BATCH[0] = batch_packs
            ops._PACK_JOBS.clear()
            ops.invalidate_weights()
            m = VocoderGAN(**cfg)
            m.load_state_dict(copy.deepcopy(state))
            m = m.to(gpu)
            m.configure_optimizers()
            before = ops.LAUNCH_COUNTS.get("pack_many", 0)
            outs = [m.fit_batch({k: v.to(gpu) for k, v in b.items()}) for b in batches]
            torch.cuda.synchronize()
            res[batch_packs] = ([float(o["g"]) for o in outs] + [float(o["d"]) for o in outs], m.optim_g.flat.clone(), m.optim_d.flat.clone())
            n = ops.LAUNCH_COUNTS.get("pack_many", 0) - before
            assert (n >= 4) if batch_packs else (n == 0), n  # replays from the second batch on
            m.optim_g.close()
            m.optim_d.close()
    finally:
        ops._PACK_BATCH[0] = True
        ops.set_compute_dtype("f32")
    la, ga, da = res[True]
    lb, gb, db = res[False]
    # the pack buffers are bit-equal (test above); the steps are not bit-equal run to run (fp32 atomics in bias sums and thin
    # weight gradients), and AdamW turns gradient noise on near-zero gradients into updates of +-lr: three steps at
    # lr = 2e-4 bound the parameter difference by 1.2e-3
    assert la == pytest.approx(lb, rel=1e-3)
    assert float((ga - gb).abs().max()) <= 1.2e-3 and float((ga - gb).abs().mean()) <= 2e-5
    assert float((da - db).abs().max()) <= 1.2e-3 and float((da - db).abs().mean()) <= 2e-5
