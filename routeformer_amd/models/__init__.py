"""``from routeformer_amd.models import Routeformer, RouteformerConfig`` (mirrors ``routeformer.models``)."""
from .config import RouteformerConfig
from .routeformer import Routeformer

__all__ = ["Routeformer", "RouteformerConfig"]
