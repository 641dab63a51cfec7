"""Launch wrappers + autograd Functions over the C ABI of libvcvits_hip.so, by kernel family:

  core         switches, launch counters, capture hook, table uploads, gradient sinks
  weights      weight / spectral norm, the normalised + packed weight cache, parameter regions, batched packs
  conv         conv / transposed conv launches (forward, data, weight and bias gradients), feature-map taps, autograd Functions
  x16          16-bit activations (inference decoder), the fused ResBlock pair
  elementwise  scale / mask / pad / pool streaming kernels, the batched loss terms
  stft         STFT / iSTFT / log-mel
  adamw        flat-buffer AdamW
  blocks       WaveNet gate / res-skip, posterior / prior sampling, coupling, LayerNorm, dropout, KL, embedding, slicing
  attention    relative-position attention

Everything runs on the GPU through hand-written HIP kernels; torch is used for buffer allocation, the current stream and
the autograd graph only.  There is no CPU fallback: calling any op with CPU tensors raises.

`from vcvits_amd import ops` sees every name of every family module (`ops.conv1d`, `ops.LAUNCH_COUNTS`, `ops._USE_X3`, ...):
the switches are one-element lists / dicts shared by reference, so `ops._USE_X3[0] = False` acts on the family modules too.
To REPLACE a function (a probe or a test double) use `ops.replace(name, fn)`: the family modules call each other through
their own globals, so an assignment on the package alone would not reach them.
"""
from . import core, weights, conv, x16, elementwise, stft, adamw, blocks, attention  # noqa: F401  (dependency order)

FAMILIES = (core, weights, conv, x16, elementwise, stft, adamw, blocks, attention)
for _m in FAMILIES:
    globals().update({_k: _v for _k, _v in vars(_m).items() if not _k.startswith("__")})
del _m


def replace(name, fn):
    """Bind `name` to `fn` here and in every family module that defines or imports it; returns the previous object."""
    prev = globals()[name]
    for m in FAMILIES:
        if name in vars(m):
            setattr(m, name, fn)
    globals()[name] = fn
    return prev
