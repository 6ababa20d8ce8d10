"""Learning-rate schedule of the reference's training recipe (``routeformer/optimizers``)."""
from .lr_scheduler import LinearWarmupCosineAnnealingLR

__all__ = ["LinearWarmupCosineAnnealingLR"]
