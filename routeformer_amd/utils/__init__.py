"""Small tensor helpers of the hot path (device-agnostic torch plumbing; API of
``routeformer/utils/vector.py`` and ``routeformer/utils/filter.py``)."""
from .config import BaseConfig
from .tensor import estimate_angle, estimate_angle_and_norm, median_downsampler, rotate

__all__ = ["BaseConfig", "rotate", "estimate_angle", "estimate_angle_and_norm", "median_downsampler"]
