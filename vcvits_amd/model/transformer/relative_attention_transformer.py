"""Relative-position transformer encoder (vits/model/transformer/relative_attention_transformer.py:
TransformerEncoder :13-47, MultiHeadAttention :103-251 with window_size=4 / heads_share,
FFN :265-311 non-causal).  QK^T and P.V run on the fp32 MFMA GEMM kernel; the banded relative
logits/values, the -1e4 mask fill, softmax and dropout are one fused kernel each way."""
import math

import torch
from torch import nn

from ... import ops
from ..._lib import ACT_RELU
from ..modules import Conv, LayerNorm


class MultiHeadAttention(nn.Module):
    def __init__(self, channels, out_channels, n_heads, p_dropout=0., window_size=None, heads_share=True,
                 block_length=None, proximal_bias=False, proximal_init=False):
        super().__init__()
        assert channels % n_heads == 0
        if window_size is None or not heads_share or block_length is not None or proximal_bias:
            raise NotImplementedError("only the configuration the VC path instantiates is built "
                                      "(window_size set, heads_share, no block_length/proximal_bias)")
        self.channels, self.out_channels, self.n_heads = channels, out_channels, n_heads
        self.p_dropout, self.window_size = p_dropout, window_size
        self.attn = None
        # the reference keeps the attention probabilities of the last call in `self.attn` (:138); set False to skip the
        # [B, H, T, T] write when nothing reads them (TransformerEncoder does so for its own layers)
        self.store_attn = True
        self.k_channels = channels // n_heads
        self.conv_q = Conv(channels, channels, 1)
        self.conv_k = Conv(channels, channels, 1)
        self.conv_v = Conv(channels, channels, 1)
        self.conv_o = Conv(channels, out_channels, 1)
        rel_stddev = self.k_channels ** -0.5
        self.emb_rel_k = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * rel_stddev)
        self.emb_rel_v = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * rel_stddev)
        nn.init.xavier_uniform_(self.conv_q.weight)
        nn.init.xavier_uniform_(self.conv_k.weight)
        nn.init.xavier_uniform_(self.conv_v.weight)
        if proximal_init:
            with torch.no_grad():
                self.conv_k.weight.copy_(self.conv_q.weight)
                self.conv_k.bias.copy_(self.conv_q.bias)

    def forward(self, x, c, attn_mask=None, x_mask=None):
        """Self-attention only (x is c).  The reference's attn_mask [B,1,T,T] is always the outer
        product of x_mask with itself (:36); pass x_mask [B,1,T] (or let it be recovered from the
        diagonal of attn_mask)."""
        if x is not c:
            raise NotImplementedError("relative attention is only available for self-attention (:160)")
        B, _, T = x.shape
        if x_mask is None:
            if attn_mask is None:
                x_mask = torch.ones(B, 1, T, device=x.device, dtype=x.dtype)
            else:
                x_mask = torch.diagonal(attn_mask[:, 0], dim1=-2, dim2=-1).unsqueeze(1)
        mask2 = x_mask.reshape(B, T).contiguous()
        q = self.conv_q(x)
        k = self.conv_k(x)
        v = self.conv_v(x)
        out, self.attn = ops.rel_attention(q, k, v, self.emb_rel_k, self.emb_rel_v, mask2, self.n_heads,
                                           self.window_size, self.p_dropout, self.training, want_attn=self.store_attn)
        return self.conv_o(out)


class FFN(nn.Module):
    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0., activation=None,
                 causal=False):
        super().__init__()
        if causal or activation == "gelu":
            raise NotImplementedError("the VC path uses the non-causal ReLU FFN only")
        if kernel_size % 2 != 1:
            raise NotImplementedError("same padding is fused as symmetric zero padding (odd kernels)")
        self.in_channels, self.out_channels, self.filter_channels = in_channels, out_channels, filter_channels
        self.kernel_size, self.p_dropout = kernel_size, p_dropout
        pad = (kernel_size - 1) // 2
        self.conv_1 = Conv(in_channels, filter_channels, kernel_size, padding=pad)
        self.conv_2 = Conv(filter_channels, out_channels, kernel_size, padding=pad)

    def forward(self, x, x_mask):
        mask2 = x_mask.reshape(x_mask.shape[0], -1)
        h = self.conv_1(ops.mask_mul(x, mask2), out_act=ACT_RELU)
        h = ops.dropout(h, self.p_dropout, self.training)
        h = self.conv_2(ops.mask_mul(h, mask2))
        return ops.mask_mul(h, mask2)


class TransformerEncoder(nn.Module):
    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0.,
                 window_size=4, **kwargs):
        super().__init__()
        self.hidden_channels, self.filter_channels, self.n_heads = hidden_channels, filter_channels, n_heads
        self.n_layers, self.kernel_size, self.p_dropout, self.window_size = n_layers, kernel_size, p_dropout, window_size
        self.attn_layers = nn.ModuleList()
        self.norm_layers_1 = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.norm_layers_2 = nn.ModuleList()
        for _ in range(n_layers):
            self.attn_layers.append(MultiHeadAttention(hidden_channels, hidden_channels, n_heads,
                                                       p_dropout=p_dropout, window_size=window_size))
            self.attn_layers[-1].store_attn = False  # nothing reads the encoder layers' `attn` (set True to inspect it)
            self.norm_layers_1.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(FFN(hidden_channels, hidden_channels, filter_channels, kernel_size,
                                       p_dropout=p_dropout))
            self.norm_layers_2.append(LayerNorm(hidden_channels))

    def forward(self, x, x_mask):
        mask2 = x_mask.reshape(x_mask.shape[0], -1)
        x = ops.mask_mul(x, mask2)
        for i in range(self.n_layers):
            y = self.attn_layers[i](x, x, x_mask=x_mask)
            y = ops.dropout(y, self.p_dropout, self.training)
            x = self.norm_layers_1[i](x, residual=y)
            y = self.ffn_layers[i](x, x_mask)
            y = ops.dropout(y, self.p_dropout, self.training)
            x = self.norm_layers_2[i](x, residual=y)
        return ops.mask_mul(x, mask2)
