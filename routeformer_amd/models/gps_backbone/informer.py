"""Informer GPS backbone on the HIP kernels (interface of ``routeformer/models/gps_backbone/
Informer.py:18-167``): ``Informer(configs: GPSBackboneConfig)``, ``forward(x (B,L,enc_in)) ->
(B,pred_len,c_out)``, mutable ``.pred_len`` (the autoregressive eval loop patches it,
``routeformer.py:171-172,191``)."""
import torch
from torch import nn

from routeformer_amd import kernels as K

from routeformer_amd.models.blocks import (AttentionLayer, DataEmbedding, Decoder, DecoderLayer,
                                           DistilConv, Encoder, EncoderLayer)

from .config import GPSBackboneConfig


class Informer(nn.Module):
    def __init__(self, configs: GPSBackboneConfig):
        super().__init__()
        c = configs
        if c.embed != "timeF":
            raise NotImplementedError("only the timeF embedding used by Routeformer is implemented")
        self.pred_len = c.pred_len
        self.output_attention = bool(c.output_attention)
        self.smart_decoder = c.smart_decoder
        self.enc_embedding = DataEmbedding(c.enc_in, c.d_model, c.dropout)
        self.dec_embedding = DataEmbedding(c.dec_in, c.d_model, c.dropout)

        def attn(kind):
            return AttentionLayer(kind, c.d_model, c.n_heads, c.factor, gps_variant=True)

        self.encoder = Encoder(
            [EncoderLayer(attn("prob"), c.d_model, c.d_ff, c.dropout, c.activation) for _ in range(c.e_layers)],
            [DistilConv(c.d_model) for _ in range(c.e_layers - 1)] if c.distil else None,
            norm_layer=nn.LayerNorm(c.d_model))
        self.decoder = Decoder(
            [DecoderLayer(attn("prob_masked"), attn("prob"), c.d_model, c.d_ff, c.dropout, c.activation)
             for _ in range(c.d_layers)],
            norm_layer=nn.LayerNorm(c.d_model),
            projection=nn.Linear(c.d_model, c.c_out, bias=True))
        if self.output_attention:  # the ENCODER's maps (Informer.py:53,164); decoder attentions are never returned
            self.encoder.set_output_attention()

    def forward(self, x):
        # decoder input: the history followed by its last row repeated ("smart") or by zeros; one launch, and the two
        # gradients of x (encoder path, decoder path) meet in one backward launch
        x_dec, x = K.smart_tail(x, self.pred_len, self.smart_decoder)
        fork = None
        if (K.OVERLAP and (K.OVERLAP_MASK & 8) and x.is_cuda and not K.on_side_stream() and len(self.decoder.layers) > 0):
            # the decoder's embedding and the self-attention block of its first layer do not depend on the encoder:
            # they run on a side stream that forks HERE (before the encoder), although the host issues them after the
            # encoder -- the reference's draw order (encoder layers, then decoder) is untouched
            fork = K.fork_side_stream("decoder")
        memory = self.encoder(self.enc_embedding(x))
        if fork is not None:
            with torch.cuda.stream(fork):
                first = self.decoder.layers[0].self_block(self.dec_embedding(x_dec))
            torch.cuda.current_stream().wait_stream(fork)
            out = self.decoder(None, memory, first=first)
        else:
            out = self.decoder(self.dec_embedding(x_dec), memory)
        out = out[:, -self.pred_len:, :]
        return (out, self.encoder.attentions) if self.output_attention else out
