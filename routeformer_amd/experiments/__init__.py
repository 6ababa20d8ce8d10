"""Baselines of the reference's ``experiments/`` tree that are built from the hot-path blocks."""
from .multimodal_transformer import MultiModalTransformer

__all__ = ["MultiModalTransformer"]
