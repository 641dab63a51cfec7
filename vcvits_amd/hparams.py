"""Attribute/mapping view of a nested config dict (interface of vits/hparams.py:3-32 of the
reference, so ``VCVITS(**hparams)`` and ``hparams.train.segment_size`` both work)."""


class HParams:
    def __init__(self, **kwargs):
        for key, value in kwargs.items():
            self[key] = HParams(**value) if isinstance(value, dict) else value

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def values(self):
        return self.__dict__.values()

    def __len__(self):
        return len(self.__dict__)

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.__dict__

    def __repr__(self):
        return repr(self.__dict__)

    def get(self, key, default=None):
        return self.__dict__.get(key, default)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, HParams) else v) for k, v in self.__dict__.items()}
