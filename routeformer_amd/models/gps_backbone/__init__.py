"""GPS backbones.  Only the Informer (the backbone Routeformer is defined with) is in scope; the
ablation backbones of the reference plug into the same slot:
``gps_backbone(configs=GPSBackboneConfig)``."""
from .config import GPSBackboneConfig
from .informer import Informer

__all__ = ["GPSBackboneConfig", "Informer"]
