"""GPS backbones: the Informer (the backbone Routeformer is defined with) and the vanilla Transformer
(SURVEY 8(f) #4).  The other ablation backbones of the reference plug into the same slot:
``gps_backbone(configs=GPSBackboneConfig)``."""
from .config import GPSBackboneConfig
from .informer import Informer
from .transformer import Transformer

__all__ = ["GPSBackboneConfig", "Informer", "Transformer"]
