"""HiFi-GAN generator -- the decoder the reference pulls from torch.hub
("vtuber-plan/hifi-gan:v0.3.1", synthesizer_svc.py:59).  Not in the reference tree: built from the
canonical VITS/HiFi-GAN definition with the in-tree ResBlock1
(modules.py:186-222), ctor signature of synthesizer_tts.py:71-78, hyper-parameters of
configs/base.json:55-63.  Parity with the hub weights is unpinned."""
import torch
from torch import nn

from .. import ops
from .._lib import ACT_TANH
from . import modules
from .modules import Conv, ConvT, LRELU_SLOPE


class Generator(nn.Module):
    """Defaults = the canonical VITS decoder (plain conv_pre, plain bias-free conv_post).  The original HiFi-GAN lineage
    (jik876) weight-norms conv_pre / conv_post and keeps conv_post's bias; which of the two the hub model
    "vtuber-plan/hifi-gan:v0.3.1" follows cannot be checked offline, so both load: the three
    `conv_*` switches select the form, and `load_state_dict` switches the two layers to the form its keys show
    (`conv_pre.weight_g`, `conv_post.weight_g`, `conv_post.bias`) before copying -- load weights before creating
    optimizers, as the reference does (synthesizer_svc.py:59 builds the decoder with its weights)."""

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0, conv_pre_weight_norm=False,
                 conv_post_weight_norm=False, conv_post_bias=False):
        super().__init__()
        self.num_kernels = len(resblock_kernel_sizes)
        self.num_upsamples = len(upsample_rates)
        self.conv_pre = Conv(initial_channel, upsample_initial_channel, 7, padding=3, weight_norm=conv_pre_weight_norm)
        block = modules.ResBlock1 if str(resblock) == "1" else modules.ResBlock2
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            self.ups.append(ConvT(upsample_initial_channel // (2 ** i), upsample_initial_channel // (2 ** (i + 1)),
                                  k, stride=u, padding=(k - u) // 2, weight_norm=True))
        self.resblocks = nn.ModuleList()
        ch = upsample_initial_channel
        for i in range(len(self.ups)):
            ch = upsample_initial_channel // (2 ** (i + 1))
            for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes):
                self.resblocks.append(block(ch, k, tuple(d)))
        self.conv_post = Conv(ch, 1, 7, padding=3, bias=conv_post_bias, weight_norm=conv_post_weight_norm)
        if gin_channels != 0:
            self.cond = Conv(gin_channels, upsample_initial_channel, 1)
        for m in self.ups:
            m.weight_v.data.normal_(0.0, 0.01)
        self._register_load_state_dict_pre_hook(self._match_lineage, with_module=True)

    @staticmethod
    def _match_lineage(self, state_dict, prefix, *_):
        """Before a state_dict is copied in: give conv_pre / conv_post the form (weight norm, bias) its keys show."""
        for name in ("conv_pre", "conv_post"):
            k = prefix + name
            if k + ".weight_g" not in state_dict and k + ".weight" not in state_dict:
                continue  # (a partial state_dict without this layer: nothing to go by)
            old = getattr(self, name)
            wn = k + ".weight_g" in state_dict
            bias = k + ".bias" in state_dict
            if wn == old.is_wn and bias == (old.bias is not None):
                continue
            w = old.weight_v if old.is_wn else old.weight
            new = Conv(w.shape[1], w.shape[0], w.shape[2], padding=old.padding, bias=bias, weight_norm=wn)
            setattr(self, name, new.to(device=w.device, dtype=w.dtype))
            for key in ("_wn_mods", "_wn_mods_skip", "_wn_mods_groups", "_wn_mods_skip_groups"):
                self.__dict__.pop(key, None)  # prepare_weight_norm's cached layer list

    def _forward_bf16_activations(self, x, g):
        """Inference in bf16 mode: every conv <-> conv tensor between conv_pre and conv_post lives in 16 bits in HBM (the
        reference's autocast does the same to every conv output: train.py:104-106).  The residual stream (stage inputs,
        ResBlock inputs / outputs, the accumulated stage mean) is stored as fp16 -- re-rounded at every residual add, it
        needs the 11 significand bits to stay below the operand rounding; the tensor between the two convs of a pair is
        stored as the bf16 matrix-core operand itself (after the leaky-ReLU its only consumer applies).  The mean over a
        stage's three ResBlocks is accumulated by their last convs (post_scale 1/3) and conv_post reads the stream.
        Arithmetic: bf16 MFMA operands, fp32 accumulate and epilogue, as in the fp32-activation path."""
        stream = torch.float16
        x = self.conv_pre(x)
        if g is not None:
            x = x + self.cond(g)
        x = ops.cast_x16(x, stream)
        nk = self.num_kernels
        for i in range(self.num_upsamples):
            up = self.ups[i]
            x = ops.convT_forward_x16(x, up.effective_weight(), up.bias, stride=up.stride, pad=up.padding, in_leaky=True,
                                      slope=LRELU_SLOPE, out_dtype=stream)
            acc = None
            for j in range(nk):
                acc = self.resblocks[i * nk + j].forward_x16(x, acc, 1.0 / nk)
            x = acc
        cp = self.conv_post
        return ops.conv_m1_x16(x, cp.effective_weight(), cp.bias, pad=cp.padding, in_leaky=True, slope=0.01, out_act=ACT_TANH)

    def forward(self, x, g=None):
        modules.prepare_weight_norm(self)
        if (ops.bf16_activations() and not torch.is_grad_enabled() and x.is_cuda and x.shape[-1] % 2 == 0
                and x.shape[-1] >= 96 and isinstance(self.resblocks[0], modules.ResBlock1)
                and (self.conv_post.weight_v if self.conv_post.is_wn else self.conv_post.weight).shape[2] in (3, 5, 7)):
            return self._forward_bf16_activations(x, g)
        x = self.conv_pre(x)
        if g is not None:
            x = x + self.cond(g)
        for i in range(self.num_upsamples):
            x = self.ups[i](x, in_leaky=True, slope=LRELU_SLOPE)
            r = [self.resblocks[i * self.num_kernels + j](x) for j in range(self.num_kernels)]
            if len(r) == 3:  # (both reference configs: configs/base.json:57)
                x = ops.avg3(r[0], r[1], r[2])
            else:  # any other count: sum (autograd's own accumulation adds) and one scaling pass
                acc = r[0]
                for t in r[1:]:
                    acc = acc + t
                x = ops.scale_grad(acc, 1.0 / len(r))
        # F.leaky_relu default slope 0.01 -> conv_post -> tanh, all in one launch
        return self.conv_post(x, in_leaky=True, slope=0.01, out_act=ACT_TANH)
