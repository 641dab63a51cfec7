"""Host-side helpers with the reference's names (vits/commons.py); the tensor work goes through
the HIP kernels in ops/."""
import torch

from . import ops


def init_weights(m, mean=0.0, std=0.01):
    """vits/commons.py:8-11"""
    if m.__class__.__name__.find("Conv") != -1 and hasattr(m, "weight") and isinstance(m.weight, torch.Tensor):
        m.weight.data.normal_(mean, std)


def get_padding(kernel_size, dilation=1):
    """vits/commons.py:14-15"""
    return int((kernel_size * dilation - dilation) / 2)


def sequence_mask(length, max_length=None):
    """vits/commons.py:120-124 (index arithmetic on tiny integer tensors: torch glue)."""
    if max_length is None:
        max_length = int(length.max())
    ar = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return ar.unsqueeze(0) < length.unsqueeze(1)


def slice_segments(x, ids_str, segment_size=4):
    """vits/commons.py:48-54 as one gather kernel (the reference loops over the batch in Python)."""
    return ops.slice_segments(x, ids_str, segment_size, 1)


def rand_slice_segments(x, x_lengths=None, segment_size=4):
    """vits/commons.py:57-64"""
    b, d, t = x.size()
    if x_lengths is None:
        x_lengths = torch.full((b,), t, device=x.device, dtype=torch.long)
    ids_str_max = x_lengths - segment_size + 1
    ids_str = (torch.rand([b], device=x_lengths.device) * ids_str_max).to(dtype=torch.long)
    return slice_segments(x, ids_str, segment_size), ids_str


def fused_add_tanh_sigmoid_multiply(input_a, input_b, n_channels):
    """vits/commons.py:99-106: tanh((a+b)[:, :H]) * sigmoid((a+b)[:, H:]) with b [B, 2H, 1]."""
    return ops.wn_gate(input_a, input_b, 0)


def clip_grad_value_(parameters, clip_value, norm_type=2):
    """vits/commons.py:145-160 -- the reference calls it with clip_value=None purely to LOG the
    gradient norm (one .item() per tensor).  Kept for API parity; one reduction, one sync."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return 0.0
    total = torch.stack([g.detach().norm(float(norm_type)) ** norm_type for g in grads]).sum()
    if clip_value is not None:
        for g in grads:
            g.clamp_(min=-float(clip_value), max=float(clip_value))
    return float(total) ** (1.0 / norm_type)
