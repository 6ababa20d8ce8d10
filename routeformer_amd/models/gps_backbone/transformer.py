"""Vanilla Transformer GPS backbone on the HIP kernels (interface and state_dict keys of
``routeformer/models/gps_backbone/Transformer.py:12-141``): dense softmax attention everywhere (causal in
the decoder's self-attention), decoder input = history followed by ``pred_len`` zero rows, no distilling
and no host randomness.  Plugs into the same ``gps_backbone(configs=GPSBackboneConfig)`` slot."""
import torch
from torch import nn

from routeformer_amd.models.blocks import AttentionLayer, DataEmbedding, Decoder, DecoderLayer, Encoder, EncoderLayer

from .config import GPSBackboneConfig


class Transformer(nn.Module):
    def __init__(self, configs: GPSBackboneConfig):
        super().__init__()
        c = configs
        if c.embed != "timeF":
            raise NotImplementedError("only the timeF embedding used by Routeformer is implemented")
        self.pred_len = c.pred_len
        self.output_attention = bool(c.output_attention)
        self.enc_embedding = DataEmbedding(c.enc_in, c.d_model, c.dropout)
        self.dec_embedding = DataEmbedding(c.dec_in, c.d_model, c.dropout)

        def attn(kind):
            return AttentionLayer(kind, c.d_model, c.n_heads, c.factor, attn_dropout=c.dropout)

        self.encoder = Encoder(
            [EncoderLayer(attn("full"), c.d_model, c.d_ff, c.dropout, c.activation) for _ in range(c.e_layers)],
            norm_layer=nn.LayerNorm(c.d_model))
        self.decoder = Decoder(
            [DecoderLayer(attn("full_masked"), attn("full"), c.d_model, c.d_ff, c.dropout, c.activation)
             for _ in range(c.d_layers)],
            norm_layer=nn.LayerNorm(c.d_model),
            projection=nn.Linear(c.d_model, c.c_out, bias=True))
        if self.output_attention:  # the ENCODER's maps (Transformer.py:45,138)
            self.encoder.set_output_attention()

    def forward(self, x):
        B, L, C = x.shape
        x_dec = torch.cat([x, torch.zeros(B, self.pred_len, C, device=x.device, dtype=torch.float32)], dim=1)
        memory = self.encoder(self.enc_embedding(x))
        out = self.decoder(self.dec_embedding(x_dec), memory)[:, -self.pred_len:, :]
        return (out, self.encoder.attentions) if self.output_attention else out
