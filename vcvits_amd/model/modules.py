"""Building blocks with the reference's names and state_dict keys (vits/model/modules.py), whose
forward/backward run on the HIP kernels behind ops/."""
import torch
from torch import nn

from .. import ops, tuning
from .._lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_TANH

LRELU_SLOPE = 0.1  # vits/model/modules.py:16
_WN_LINK = tuning.flag("VCVITS_WN_LINK", True, "WN: the residual gradient handed to the next layer's data-gradient launch (A/B)")


def _default_conv_init(weight, bias):
    """PyTorch's Conv default init (kaiming_uniform(a=sqrt(5)) + uniform bias), as the reference's
    nn.Conv1d/Conv2d/ConvTranspose1d constructors do."""
    nn.init.kaiming_uniform_(weight, a=5 ** 0.5)
    if bias is not None:
        fan_in = weight[0].numel() if weight.dim() > 1 else weight.numel()
        bound = 1.0 / fan_in ** 0.5 if fan_in > 0 else 0.0
        nn.init.uniform_(bias, -bound, bound)


def prepare_weight_norm(root, skip=()):
    """Compute the effective weights of every weight-normed Conv / ConvT under `root` (minus the sub-modules
    in `skip`) in one launch and hand them to the layers: each consumes its weight once, in its next forward;
    a layer whose parameters were re-assigned in between recomputes its own.  Call at the top of a module's
    forward.  A parent that prepares for its children marks them with `_wn_parent_prepared` so their own
    call is a no-op for that forward."""
    if root.__dict__.pop("_wn_parent_prepared", False):
        return
    key = "_wn_mods" if not skip else "_wn_mods_skip"
    mods = root.__dict__.get(key)
    if mods is None:
        excluded = set()
        for sm in skip:
            excluded.update(id(m) for m in sm.modules())
        named = [(n, m) for n, m in root.named_modules() if isinstance(m, (Conv, ConvT)) and m.is_wn and id(m) not in excluded]
        mods = [m for _, m in named]
        root.__dict__[key] = mods
        # backward granularity: consecutive layers of one sub-block ("discriminators.3", "resblocks.7", "ups.1") share
        # an autograd node, so their parameter gradients -- and the data-parallel buckets they fill -- are final as
        # soon as THAT block's backward is done (ops._WeightNormManyFn)
        sizes, last = [], None
        for n, _ in named:
            parts = n.split(".")
            k = ".".join(parts[:2]) if parts[0] in ("discriminators", "resblocks", "ups") else ""
            if k == last:
                sizes[-1] += 1
            else:
                sizes.append(1)
                last = k
        root.__dict__[key + "_groups"] = sizes
    if len(mods) < 2 or not mods[0].weight_v.is_cuda:
        return
    # one forward launch for the whole tree now; the autograd node of each group is created when the group's first
    # layer runs (_take_prepared), i.e. at that sub-block's place in the backward schedule
    holder = ops.weight_norm_forward([m.weight_v for m in mods], [m.weight_g for m in mods])
    i0 = 0
    for k in root.__dict__[key + "_groups"]:
        group = mods[i0:i0 + k]
        lazy = (holder, i0, i0 + k, group, [(m.weight_v._version, m.weight_g._version, m.weight_v.data_ptr()) for m in group],
                ops.WEIGHT_EPOCH[0])
        for m in group:
            m.__dict__.pop("_w_pre", None)
            m._w_lazy = lazy
        i0 += k


def _materialise_group(lazy):
    holder, i0, i1, group, stamps, epoch = lazy
    fresh = epoch == ops.WEIGHT_EPOCH[0]  # no raw write into parameter storage since the forward launch
    ws = ops.weight_norm_group(holder, i0, i1, [m.weight_v for m in group], [m.weight_g for m in group]) if fresh else None
    for i, m in enumerate(group):
        if m.__dict__.get("_w_lazy") is lazy:
            del m.__dict__["_w_lazy"]
            if fresh:
                m._w_pre = (ws[i],) + stamps[i]  # validated against the parameter versions of the forward launch


def _take_prepared(m):
    lazy = m.__dict__.get("_w_lazy")
    if lazy is not None:
        _materialise_group(lazy)
    pre = m.__dict__.pop("_w_pre", None)
    if pre is not None and pre[1] == m.weight_v._version and pre[2] == m.weight_g._version \
            and pre[3] == m.weight_v.data_ptr():
        return pre[0]
    return None


class Conv(nn.Module):
    """Conv1d ([M, C/g, K]) or period Conv2d with a (K,1) kernel ([M, C/g, K, 1]).  With
    ``weight_norm=True`` the parameters are the old-style ``weight_g`` / ``weight_v`` pair of
    torch.nn.utils.weight_norm (same state_dict keys as the reference)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, weight_norm=False, two_d=False, spectral_norm=False):
        super().__init__()
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups
        shape = (out_channels, in_channels // groups, kernel_size) + ((1,) if two_d else ())
        w = torch.empty(shape)
        b = torch.empty(out_channels) if bias else None
        _default_conv_init(w, b)
        self.is_wn = weight_norm and not spectral_norm
        self.is_sn = spectral_norm
        if spectral_norm:
            # torch.nn.utils.spectral_norm (old hook form, what discriminator.py:17,52 apply): parameters (bias,
            # weight_orig), buffers weight_u [out] and weight_v [in/groups * k] = normalised N(0, 1) draws
            self.bias = nn.Parameter(b) if bias else None
            self.weight_orig = nn.Parameter(w)
            self.register_buffer("weight_u", nn.functional.normalize(torch.randn(shape[0]), dim=0, eps=self.SN_EPS))
            self.register_buffer("weight_v", nn.functional.normalize(torch.randn(w[0].numel()), dim=0, eps=self.SN_EPS))
            return
        # registration order follows torch: (weight, bias) for a plain conv, (bias, weight_g, weight_v)
        # after torch.nn.utils.weight_norm
        if weight_norm:
            self.bias = nn.Parameter(b) if bias else None
            gshape = (out_channels,) + (1,) * (len(shape) - 1)
            self.weight_g = nn.Parameter(w.reshape(out_channels, -1).norm(dim=1).reshape(gshape))
            self.weight_v = nn.Parameter(w)
        else:
            self.weight = nn.Parameter(w)
            self.bias = nn.Parameter(b) if bias else None

    SN_EPS = 1e-12  # torch.nn.utils.spectral_norm's default

    def effective_weight(self):
        if self.is_sn:
            # one power iteration per training forward, none in eval mode (the hook's do_power_iteration = module.training)
            return ops.spectral_norm(self.weight_orig, self.weight_u, self.weight_v, self.training, self.SN_EPS)
        if not self.is_wn:
            return self.weight
        w = _take_prepared(self)
        return w if w is not None else ops.weight_norm(self.weight_v, self.weight_g)

    def forward(self, x, in_leaky=False, out_act=ACT_NONE, slope=LRELU_SLOPE, res=None, weight=None, link=None):
        w = self.effective_weight() if weight is None else weight
        return ops.conv1d(x, w, self.bias, stride=self.stride, pad=self.padding, dil=self.dilation,
                          groups=self.groups, in_leaky=in_leaky, out_act=out_act, slope=slope, res=res, link=link)


class ConvT(nn.Module):
    """ConvTranspose1d ([Cin, Cout, K]); weight norm is over dim 0 = Cin as torch does."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, weight_norm=True):
        super().__init__()
        self.stride, self.padding = stride, padding
        w = torch.empty(in_channels, out_channels, kernel_size)
        b = torch.empty(out_channels)
        _default_conv_init(w, b)
        self.bias = nn.Parameter(b)
        self.is_wn = weight_norm
        if weight_norm:
            self.weight_g = nn.Parameter(w.reshape(in_channels, -1).norm(dim=1).reshape(in_channels, 1, 1))
            self.weight_v = nn.Parameter(w)
        else:
            self.weight = nn.Parameter(w)

    def effective_weight(self):
        if not self.is_wn:
            return self.weight
        w = _take_prepared(self)
        return w if w is not None else ops.weight_norm(self.weight_v, self.weight_g)

    def forward(self, x, in_leaky=False, slope=LRELU_SLOPE):
        return ops.conv_transpose1d(x, self.effective_weight(), self.bias, stride=self.stride,
                                    pad=self.padding, in_leaky=in_leaky, slope=slope)


class LayerNorm(nn.Module):
    """LayerNorm over the channel dim of [B,C,T] (vits/model/modules.py:19-31); ``residual`` fuses
    the ``x + y`` of the post-LN transformer."""

    def __init__(self, channels, eps=1e-5):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))

    def forward(self, x, residual=None):
        return ops.layernorm_c(x, residual, self.gamma, self.beta, self.eps)


class WN(nn.Module):
    """WaveNet-style gated stack (vits/model/modules.py:109-175)."""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1
        self.hidden_channels, self.kernel_size = hidden_channels, kernel_size
        self.dilation_rate, self.n_layers = dilation_rate, n_layers
        self.gin_channels, self.p_dropout = gin_channels, p_dropout
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        if gin_channels != 0:
            self.cond_layer = Conv(gin_channels, 2 * hidden_channels * n_layers, 1, weight_norm=True)
        for i in range(n_layers):
            dilation = dilation_rate ** i
            padding = int((kernel_size * dilation - dilation) / 2)
            self.in_layers.append(Conv(hidden_channels, 2 * hidden_channels, kernel_size, dilation=dilation,
                                       padding=padding, weight_norm=True))
            rs_ch = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(Conv(hidden_channels, rs_ch, 1, weight_norm=True))

    def forward(self, x, x_mask, g=None, **kwargs):
        prepare_weight_norm(self, skip=() if g is not None or not hasattr(self, "cond_layer") else (self.cond_layer,))
        mask2 = x_mask.reshape(x_mask.shape[0], x_mask.shape[-1])
        if g is not None:
            g = self.cond_layer(g)  # [B, 2H*L, 1]
        output = None
        H = self.hidden_channels
        # the layers' conditioning gradients are disjoint slices of d(g): one shared buffer (ops.GateGradShare)
        share = ops.GateGradShare(self.n_layers) if (g is not None and g.requires_grad and torch.is_grad_enabled()) else None
        for i in range(self.n_layers):
            # x feeds this layer's conv and its residual add: the two gradients are summed in the conv's data-gradient launch
            link = ops.ResGradLink() if (_WN_LINK and i < self.n_layers - 1 and x.requires_grad and torch.is_grad_enabled()) else None
            x_in = self.in_layers[i](x, link=(link, "dst") if link else None)
            acts = ops.wn_gate(x_in, g, i * 2 * H, share)
            acts = ops.dropout(acts, self.p_dropout, self.training)
            rs = self.res_skip_layers[i](acts)
            if i < self.n_layers - 1:
                x, output = ops.wn_res_skip(x, output, rs, mask2, False, link=link)
            else:
                output = ops.wn_res_skip(x, output, rs, mask2, True)
        return ops.mask_mul(output, mask2)


class ResBlock1(nn.Module):
    """vits/model/modules.py:186-222; leaky-ReLU inputs and the residual add are fused into the
    convs."""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        pad = lambda d: int((kernel_size * d - d) / 2)
        self.convs1 = nn.ModuleList([Conv(channels, channels, kernel_size, dilation=d, padding=pad(d),
                                          weight_norm=True) for d in dilation])
        self.convs2 = nn.ModuleList([Conv(channels, channels, kernel_size, dilation=1, padding=pad(1),
                                          weight_norm=True) for _ in dilation])

    def forward(self, x, x_mask=None):
        if x_mask is not None:
            raise NotImplementedError("ResBlock1 with x_mask is never used on the VC path")
        for c1, c2 in zip(self.convs1, self.convs2):
            link = ops.ResGradLink() if x.requires_grad else None  # x's two gradients are summed in c1's dgrad launch
            xt = c1(x, in_leaky=True, slope=LRELU_SLOPE, link=(link, "dst") if link else None)
            x = c2(xt, in_leaky=True, slope=LRELU_SLOPE, res=x, link=(link, "src") if link else None)
        return x

    def forward_x16(self, x, acc, scale):
        """Inference over 16-bit activations (ops.conv_forward_x16): x is the residual stream (fp16 [B, C, T]); the block's
        output times `scale` is ADDED onto `acc` (created when None) by the last conv's epilogue -- the stage mean of the
        generator without a pass over three stored outputs.  c1 stores leaky(xt) in bf16 (its only consumer is c2's fused
        input leaky-ReLU followed by the rounding to the bf16 operand), so c2 stages its operand without any conversion
        and the result is the fp32-activation path's, rounding for rounding."""
        n = len(self.convs1)
        for i, (c1, c2) in enumerate(zip(self.convs1, self.convs2)):
            w1, w2 = c1.effective_weight(), c2.effective_weight()
            if (c1.bias is not None and c2.bias is not None and c2.dilation == 1 and c1.padding == (w1.shape[2] - 1) * c1.dilation // 2
                    and c2.padding == (w2.shape[2] - 1) // 2 and (acc is None or acc.dtype == x.dtype)
                    and ops.resblock_pair_supported(x, w1, w2, c1.dilation)):
                # the 32- / 64-channel stages: the pair as ONE launch, the intermediate never leaves the CU (resblock_pair.hip)
                if i < n - 1:
                    x = ops.resblock_pair_x16(x, w1, c1.bias, w2, c2.bias, c1.dilation, slope=LRELU_SLOPE)
                elif acc is None:
                    acc = ops.resblock_pair_x16(x, w1, c1.bias, w2, c2.bias, c1.dilation, slope=LRELU_SLOPE, post_scale=scale)
                else:
                    acc = ops.resblock_pair_x16(x, w1, c1.bias, w2, c2.bias, c1.dilation, slope=LRELU_SLOPE, out=acc,
                                                accumulate=True, post_scale=scale)
                continue
            xt = ops.conv_forward_x16(x, w1, c1.bias, pad=c1.padding, dil=c1.dilation, in_leaky=True,
                                      out_act=ACT_LEAKY, slope=LRELU_SLOPE, out_dtype=torch.bfloat16)
            if i < n - 1:
                x = ops.conv_forward_x16(xt, w2, c2.bias, pad=c2.padding, dil=c2.dilation, res=x,
                                         out_dtype=x.dtype)
            else:
                acc = ops.conv_forward_x16(xt, w2, c2.bias, pad=c2.padding, dil=c2.dilation, res=x,
                                           out=acc, accumulate=acc is not None, post_scale=scale, out_dtype=x.dtype)
        return acc


class ResBlock2(nn.Module):
    """vits/model/modules.py:225-247"""

    def __init__(self, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        pad = lambda d: int((kernel_size * d - d) / 2)
        self.convs = nn.ModuleList([Conv(channels, channels, kernel_size, dilation=d, padding=pad(d),
                                         weight_norm=True) for d in dilation])

    def forward(self, x, x_mask=None):
        if x_mask is not None:
            raise NotImplementedError("ResBlock2 with x_mask is never used on the VC path")
        for c in self.convs:
            x = c(x, in_leaky=True, slope=LRELU_SLOPE, res=x)
        return x


class Flip(nn.Module):
    """vits/model/modules.py:261-268 (pure data movement)."""

    def forward(self, x, *args, reverse=False, **kwargs):
        x = torch.flip(x, [1])
        if not reverse:
            return x, torch.zeros(x.size(0), dtype=x.dtype, device=x.device)
        return x


class ResidualCouplingLayer(nn.Module):
    """vits/model/modules.py:289-336."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0,
                 gin_channels=0, mean_only=False):
        assert channels % 2 == 0, "channels should be divisible by 2"
        super().__init__()
        if not mean_only:
            raise NotImplementedError("the VC path only uses mean_only=True (flow.py:27)")
        self.channels, self.hidden_channels = channels, hidden_channels
        self.half_channels = channels // 2
        self.mean_only = mean_only
        self.pre = Conv(self.half_channels, hidden_channels, 1)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=p_dropout,
                      gin_channels=gin_channels)
        self.post = Conv(hidden_channels, self.half_channels, 1)
        self.post.weight.data.zero_()
        self.post.bias.data.zero_()

    def forward(self, x, x_mask, g=None, reverse=False):
        mask2 = x_mask.reshape(x_mask.shape[0], x_mask.shape[-1])
        x0 = x[:, :self.half_channels].contiguous()
        x1 = x[:, self.half_channels:].contiguous()
        h = ops.mask_mul(self.pre(x0), mask2)
        h = self.enc(h, x_mask, g=g)
        m = ops.mask_mul(self.post(h), mask2)
        x1 = ops.coupling(x1, m, mask2, reverse)
        x = torch.cat([x0, x1], 1)
        if not reverse:
            return x, torch.zeros(x.size(0), dtype=x.dtype, device=x.device)  # logs == 0 (mean_only)
        return x
