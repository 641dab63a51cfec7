"""vits/model/flow.py:7-37"""
import torch
from torch import nn

from .. import ops, tuning
from . import modules

# Mixed-precision policy of the inference path in bf16 mode: the flow is under 1 % of the decode's FLOPs (5.9 of 770 GFLOP per
# 10 s at 48 kHz) but its output error passes through the whole decoder, so a no-grad pass runs it in the fp32
# arithmetic (split-operand kernels) and leaves bf16 to the HiFi-GAN decoder, where the time is.  VCVITS_INFER_FLOW_F32=0:
# the flow in bf16 too.
_INFER_F32 = [tuning.flag("VCVITS_INFER_FLOW_F32", True, "bf16 mode: the reverse flow of infer() keeps fp32 arithmetic")]


class ResidualCouplingBlock(nn.Module):
    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows=4,
                 gin_channels=0):
        super().__init__()
        self.channels, self.hidden_channels, self.kernel_size = channels, hidden_channels, kernel_size
        self.dilation_rate, self.n_layers, self.n_flows, self.gin_channels = dilation_rate, n_layers, n_flows, gin_channels
        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(modules.ResidualCouplingLayer(channels, hidden_channels, kernel_size, dilation_rate,
                                                            n_layers, gin_channels=gin_channels, mean_only=True))
            self.flows.append(modules.Flip())

    def forward(self, x, x_mask, g=None, reverse=False):
        if _INFER_F32[0] and not torch.is_grad_enabled() and ops.compute_dtype() == "bf16":
            ops.set_compute_dtype("f32")
            try:
                return self._run(x, x_mask, g, reverse)
            finally:
                ops.set_compute_dtype("bf16")
        return self._run(x, x_mask, g, reverse)

    def _run(self, x, x_mask, g, reverse):
        if not reverse:
            for flow in self.flows:
                x, _ = flow(x, x_mask, g=g, reverse=reverse)
        else:
            for flow in reversed(self.flows):
                x = flow(x, x_mask, g=g, reverse=reverse)
        return x
