"""GPS-backbone config (field-for-field the API of ``routeformer/models/gps_backbone/config.py:8-75``)."""
from dataclasses import dataclass, field

from routeformer_amd.utils.config import BaseConfig


@dataclass
class GPSBackboneConfig(BaseConfig):
    seq_len: int
    label_len: int
    pred_len: int
    embed: str = "timeF"
    freq: str = "m"
    d_model: int = 128
    n_heads: int = 8
    e_layers: int = 2
    d_layers: int = 1
    d_ff: int = 512
    moving_avg: int = 25
    factor: int = 1
    distil: bool = True
    dropout: float = 0.1
    activation: str = "gelu"
    individual: bool = False
    # pushed down by RouteformerConfig.__post_init__
    output_attention: bool = field(init=False)
    with_video: bool = field(init=False)
    with_gaze: bool = field(init=False)
    dense_prediction: bool = field(init=False)
    encoder_hidden_size: int = field(init=False)
    image_embedding_size: int = field(init=False)
    output_fps: int = field(init=False)
    dense_loss_ratio: float = field(init=False)
    discount_factor: dict = field(init=False)
    smart_decoder: bool = field(init=False)
    _enc_in: int = None
    _c_out: int = None

    @property
    def enc_in(self) -> int:
        """motion(2) + angle, speed, acceleration (+ fused visual tokens when video is on)."""
        if self._enc_in is not None:
            return self._enc_in
        return 5 + (self.encoder_hidden_size if self.with_video else 0)

    @property
    def dec_in(self) -> int:
        return self.enc_in

    @property
    def c_out(self) -> int:
        if self._c_out is not None:
            return self._c_out
        return self.enc_in - 3 if self.dense_prediction else 2
