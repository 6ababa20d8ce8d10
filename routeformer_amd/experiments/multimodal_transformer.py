"""The "simple multi-modal transformer" baseline on the HIP kernels -- interface, forward semantics and
state_dict keys of ``experiments/multimodal_transformer/multimodal_transformer.py:35-122``:

    frames of left / right / front video --(frozen conv encoder + frame_encoder, EVERY frame)--> (B,T,E) each
    motion (gps differences) --Linear(2,h)-->, gaze (median-downsampled) --Linear(2,h)-->
    cat -> vanilla ``Transformer`` GPS backbone (enc_in = 5 h, c_out = 2) -> cumsum from the last position.

Needs ``image_embedding_size == encoder_hidden_size`` (the reference's ``enc_in = 5 * hidden``).  The video
backbone is the plugin slot of the hot path (``VideoBackboneModule``); the frame encoder is called once per
stream, in the reference's order (left, right, front), so the host-RNG draws line up call for call."""
from typing import Optional, Type

import torch
import torch.nn.functional as F
from torch import nn

from routeformer_amd import kernels as K
from routeformer_amd.models.blocks import PerceiveEncoder
from routeformer_amd.models.config import RouteformerConfig
from routeformer_amd.models.gps_backbone import Transformer
from routeformer_amd.utils import median_downsampler


class MultiModalTransformer(nn.Module):
    def __init__(self, configs: RouteformerConfig, video_backbone: Optional[Type[nn.Module]] = None):
        super().__init__()
        self.configs = configs
        self.video_backbone = video_backbone(configs=configs.video_backbone_config)
        self.frame_encoder = PerceiveEncoder(
            in_channels=self.video_backbone.output_feature_shape[0], out_len=1,
            out_channels=configs.image_embedding_size, n_heads=configs.encoder_heads, layers=configs.encoder_layers,
            dropout=configs.feature_dropout, d_ff=configs.encoder_d_ff)
        self.motion_linear = nn.Linear(2, configs.encoder_hidden_size)
        self.gaze_linear = nn.Linear(2, configs.encoder_hidden_size)
        gcfg = configs.gps_backbone_config.copy()
        gcfg._enc_in = configs.encoder_hidden_size * 5
        gcfg._c_out = 2
        self.transformer = Transformer(configs=gcfg)

    def forward(self, batch, eval=False):  # noqa: A002  (signature of the reference)
        gps = batch["gps"].to(torch.float32)
        motions = F.pad(gps[:, 1:, :] - gps[:, :-1, :], (0, 0, 1, 0))
        motion_feats = K.linear(motions, self.motion_linear.weight, self.motion_linear.bias)
        left = batch["left_video"]
        right = batch.get("right_video", left)
        left_feats = self._forward_single_video(left)
        right_feats = self._forward_single_video(right)
        gaze_video_feats = self._forward_single_video(batch["front_video"])
        gazes = median_downsampler(batch["gaze"].to(torch.float32), self.configs.gps_backbone_config.seq_len)
        gaze_feats = K.linear(gazes, self.gaze_linear.weight, self.gaze_linear.bias)
        feats = torch.cat([motion_feats, left_feats, right_feats, gaze_video_feats, gaze_feats], dim=2)
        output = self.transformer(feats)
        return gps[:, -1:, :] + torch.cumsum(output, dim=1)

    def _forward_single_video(self, video):
        B, T = video.shape[:2]
        if hasattr(self.video_backbone, "encode_tokens"):  # native trunk: tokens (B*T, 65, C) incl. the -1 row
            tokens = self.video_backbone.encode_tokens([video], torch.arange(T))
        else:
            fmap = self.video_backbone(video.flatten(0, 1)).to(torch.float32)
            t = fmap.permute(0, 2, 3, 1).reshape(fmap.shape[0], -1, fmap.shape[1])
            tokens = torch.cat([t, -torch.ones_like(t)[:, :1, :]], dim=1)
        return self.frame_encoder(tokens).view(B, -1, self.configs.image_embedding_size)
