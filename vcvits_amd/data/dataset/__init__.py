from .vc_ms import VoiceConversionMultiSpeakerDataset, cache_paths, hash_string  # noqa: F401
