"""Flat-buffer AdamW launches (host-scalar and device-record forms).

Part of `vcvits_amd.ops` (the package re-exports every name: `from vcvits_amd import ops; ops.conv1d(...)`).  Everything here
runs on the GPU through libvcvits_hip.so; there is no CPU fallback."""

from .._lib import (check, lib, ptr, stream)


# ---------------------------------------------------------------------------------------------
# AdamW on flat buffers
# ---------------------------------------------------------------------------------------------
def adamw_step(p, g, m, v, lr, betas, eps, weight_decay, step):
    check(lib().vcv_adamw(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps,
                          weight_decay, step, stream()), "vcv_adamw")


def adamw_step_dev(p, g, m, v, betas, eps, weight_decay, hyper, step_base):
    """The same step with lr and the step delta read from the device record `hyper` (int32[2]: fp32 bits of lr, delta):
    the form an optimizer step recorded into a HIP graph takes (light/graphed.py refreshes the record before each replay)."""
    check(lib().vcv_adamw_dev(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), betas[0], betas[1], eps, weight_decay,
                              ptr(hyper), int(step_base), stream()), "vcv_adamw_dev")


def set_hyper(hyper, lr, delta):
    import struct
    check(lib().vcv_set_words(ptr(hyper), 2, struct.unpack("<i", struct.pack("<f", float(lr)))[0], int(delta), 0, 0, stream()),
          "vcv_set_words")
