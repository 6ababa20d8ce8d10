"""Video-backbone plugin slot.  In scope: the conv encoder (HRNet-16 trunk, the reference's
``InverseForm`` backbone).  The timm backbones (SwinV2/DINOv2/SAM) need third-party weights and are
out of scope; any ``VideoBackboneModule`` subclass plugs into ``Routeformer``."""
from .config import InverseFormBackboneConfig, VideoBackboneConfig, VideoBackboneModule
from .hrnet16 import HRNet16Backbone
from .token_cache import TokenCache

InverseForm = HRNet16Backbone  # name used by the reference's experiment driver

__all__ = ["VideoBackboneConfig", "VideoBackboneModule", "InverseFormBackboneConfig", "HRNet16Backbone",
           "InverseForm", "TokenCache"]
