"""Host-side data path of the voice-conversion trainer: batch schema + collate,
cache-key layout of the pre-processed tensors, pitch binning and the inference length-scale plumbing.
Pure CPU host logic, as in the reference (vits/data/*); it feeds the HIP hot path and launches no kernels."""
from .audio import coarse_f0, infer_length_scale  # noqa: F401
from .collate import VoiceConversionMultiSpeakerCollate  # noqa: F401
from .sampler import DistributedUtteranceSampler, rank_indices  # noqa: F401
