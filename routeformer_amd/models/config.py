"""Routeformer config (API of ``routeformer/models/config.py:10-107``)."""
from dataclasses import dataclass, field
from typing import Literal

from routeformer_amd.models.gps_backbone.config import GPSBackboneConfig
from routeformer_amd.models.video_backbone.config import VideoBackboneConfig
from routeformer_amd.utils.config import BaseConfig


@dataclass
class RouteformerConfig(BaseConfig):
    gps_backbone_config: GPSBackboneConfig
    video_backbone_config: VideoBackboneConfig = None
    output_attention: bool = False
    with_video: bool = None  # None -> "a video backbone config was given"
    with_gaze: bool = False
    with_scene: bool = True
    discount_factor: dict = field(default_factory=lambda: {0: 0.9})  # epoch -> gamma
    decoder_mode: Literal["vanilla", "recursive", "smart"] = "vanilla"
    rotate_motion: bool = False
    loss_function: Literal["mse", "mae", "smooth_l1"] = "smooth_l1"
    epsilon: float = None
    visual_epsilon: float = None
    autoregressive: bool = False
    autoregressive_step_size: int = 1
    dense_prediction: bool = False
    dense_loss_ratio: float = 0.25
    video_fps: int = 1
    gaze_fps: int = 1
    encoder_hidden_size: int = 64
    encoder_heads: int = 8
    encoder_layers: int = 2
    encoder_d_ff: int = 64
    cross_modal_decoder_heads: int = 8
    cross_modal_decoder_layers: int = 1
    normalize_motion: bool = False
    motion_mean: float = 0.0
    motion_std: float = 1.0
    motion_noise: float = 0.0
    view_dropout: float = 0.0
    gaze_dropout: float = 0.0
    feature_dropout: float = 0.0
    image_embedding_size: int = 128
    # training-harness knobs carried on the config (unused by the model itself)
    lr: float = 5e-4
    wd: float = 0
    optimizer: str = "Adam"
    batch_size: int = 32
    min_pci: float = 0.0
    step_size: int = 1
    epochs: int = 100
    output_fps: int = 5
    gopro_scaling_factor: float = 1.0
    front_scaling_factor: float = 1.0
    num_workers: int = 0
    use_cache: bool = False
    cache_dir: str = None
    _only_motion: bool = False

    def __post_init__(self, **_):
        assert self.output_fps % self.video_fps == 0, "Video FPS must be a divisor of the output FPS"
        assert self.output_fps % self.gaze_fps == 0, "Gaze FPS must be a divisor of the output FPS"
        if self.with_video is None:
            self.with_video = self.video_backbone_config is not None
        if self.with_gaze:
            assert self.with_video, "Gaze backbone requires video backbone to be used"
        g = self.gps_backbone_config
        for name in ("output_attention", "with_video", "with_gaze", "dense_prediction",
                     "image_embedding_size", "encoder_hidden_size", "output_fps",
                     "dense_loss_ratio", "discount_factor"):
            setattr(g, name, getattr(self, name))
        g.smart_decoder = self.decoder_mode == "smart"
