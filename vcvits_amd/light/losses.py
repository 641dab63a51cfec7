"""`vits.light.losses` import path of the reference (vits/light/losses.py)."""
from ..losses import discriminator_loss, feature_loss, generator_loss, kl_loss  # noqa: F401
