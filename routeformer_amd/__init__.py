"""routeformer_amd -- the Routeformer multimodal forward/backward hot path, MI355X-native.

Drop-in surface: ``Routeformer``, ``routeformer_amd.models.RouteformerConfig``,
``routeformer_amd.models.gps_backbone.{GPSBackboneConfig, Informer}``,
``routeformer_amd.models.video_backbone.{VideoBackboneConfig, VideoBackboneModule, InverseForm}``,
``routeformer_amd.losses.future_discounted_mse.FutureDiscountedLoss``, ``routeformer_amd.score.{ade, fde}``.
All dense arithmetic runs in ``csrc/librf_hip.so`` (hand-written HIP for gfx950); there is no CPU or
eager-PyTorch fallback -- operations raise if the library is missing or tensors are not on the GPU.
"""
from routeformer_amd.models import Routeformer, RouteformerConfig

__all__ = ["Routeformer", "RouteformerConfig"]
__version__ = "0.1.0"
