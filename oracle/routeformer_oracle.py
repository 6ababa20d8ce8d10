"""CPU ORACLE for the Routeformer hot path -- TEST INFRASTRUCTURE ONLY.

A from-scratch, functional, pure-torch (CPU, fp32) restatement of the reference algorithm
(`routeformer/models/routeformer.py` forward and everything below it), driven by a plain
``state_dict``.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product (``routeformer_amd``) never does and has no CPU fallback.

Pinned: every function here is checked against golden vectors produced by running the reference
itself in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``, see
``tests/test_oracle_golden.py``).  The reference has no tests / known-answer vectors of its own
(SURVEY.md section 4), so those fixtures are the pin.

Each function cites the reference file:line it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
TAPS: Optional[dict] = None  # debug: name -> list of intermediate activations (tools/debug_taps.py)


def _tap(name: str, x):
    if TAPS is not None:
        TAPS.setdefault(name, []).append(x.detach().clone())
    return x


# ---------------------------------------------------------------------------------------------
# host-RNG index source (ProbSparse key sampling)
# ---------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------
# Arithmetic model of the reference's OWN training precision on a GPU (measurement aid, bench.py ade_vs_cpu_ref)
# ---------------------------------------------------------------------------------------------
# experiments/full_comparison.py:48 sets torch.set_float32_matmul_precision("medium"): fp32 matmuls (nn.Linear, matmul /
# bmm / einsum: every projection and all three attention products) may round their operands to bfloat16; PyTorch's cuDNN
# default (torch.backends.cudnn.allow_tf32 = True) lets the convolutions (token embeddings, the Conv1d(k=1) feed-forward
# pairs, the distilling convolution, the conv trunk) run on TF32 operands (10 mantissa bits).  Accumulation stays fp32.
# ARITH = None: exact fp32 (the oracle proper; every parity test).  ARITH = "medium": operands rounded as above, so that
# the rate at which the REFERENCE's precision mode flips ProbSparse selections can be put next to the product's bf16 mode.
ARITH: Optional[str] = None


def _mm(t):
    """Operand of a matmul-class op under ARITH."""
    return t.to(torch.bfloat16).to(torch.float32) if ARITH == "medium" else t


def _cv(t):
    """Operand of a convolution under ARITH: TF32 = fp32 with the mantissa rounded to 10 bits (round to nearest even)."""
    if ARITH != "medium":
        return t
    i = t.contiguous().view(torch.int32)
    i = (i + 0xFFF + ((i >> 13) & 1)) & ~0x1FFF
    return i.view(torch.float32)


class IndexSource:
    """Supplies ``index_sample`` tensors in call order.

    ``replay=None``  -> draw ``torch.randint(L_K, (L_Q, k))`` from the global CPU generator, the
    exact call of cross_modal_transformer.py:95 / layers/SelfAttentionFamily.py:94, so the same
    ``torch.manual_seed`` reproduces the reference's draws.
    ``replay=[...]`` -> pop recorded tensors (fixtures)."""

    def __init__(self, replay: Optional[Sequence[torch.Tensor]] = None):
        self.replay = list(replay) if replay is not None else None
        self.log: List[torch.Tensor] = []
        # top-u selections made by each ProbSparse call, (B,H,u) ascending, in call order; tests use
        # them to teacher-force the HIP kernels (the selection is discontinuous in the inputs) and to
        # count selection flips.  ``margins`` = gap between the last selected and the first rejected
        # sparsity measure per (b,h), relative to the largest sampled |q.k| it was computed from (M is a
        # difference of such dot products, so their rounding error is what can flip a selection).
        self.tops: List[torch.Tensor] = []
        self.margins: List[torch.Tensor] = []
        # measurement aid: ``forced`` = selections imposed on the calls in order (teacher forcing; the call's own selection
        # still goes to ``tops``), so that an ARITH = "medium" run can be counted for flips on the fp32 run's trajectory
        self.forced: Optional[List[torch.Tensor]] = None

    def randint(self, high: int, size) -> torch.Tensor:
        if self.replay is not None:
            t = self.replay.pop(0).long()
            assert tuple(t.shape) == tuple(size) and int(t.max()) < high, (t.shape, size, high)
        else:
            t = torch.randint(high, size)
        self.log.append(t)
        return t


class DropoutSource:
    """``nn.Dropout`` masks (train mode, p > 0), one per dropout call in the reference's call order.

    ``replay=[keep-masks]`` -> pop recorded masks (fixtures recorded from the reference run, or masks materialised
    from the product's Philox generator); else draw ``bernoulli(1 - p)`` from a PRIVATE generator, so the global CPU
    generator -- the ProbSparse key samples -- is consumed exactly as on a GPU run of the reference, where dropout
    draws from the device generator.  ``log`` keeps every mask used.
    A mask is stored in the layout of the tensor the reference drops (the FFN's hidden activation is dropped in its
    (B, d_ff, L) Conv1d layout, cross_modal_transformer.py:298)."""

    def __init__(self, replay: Optional[Sequence[torch.Tensor]] = None, seed: int = 0):
        self.replay = list(replay) if replay is not None else None
        self.gen = torch.Generator().manual_seed(seed)
        self.log: List[torch.Tensor] = []

    def keep(self, shape, p: float) -> torch.Tensor:
        if self.replay is not None:
            m = self.replay.pop(0).to(torch.bool)
            assert tuple(m.shape) == tuple(shape), (tuple(m.shape), tuple(shape))
        else:
            m = torch.empty(tuple(shape)).bernoulli_(1.0 - p, generator=self.gen).to(torch.bool)
        self.log.append(m)
        return m


def _drop(x, p: float, src: Optional[DropoutSource]):
    """F.dropout(x, p, training=True): x * keep / (1 - p).  p == 0 (or eval: callers pass 0) is the identity."""
    if p <= 0.0:
        return x
    assert src is not None, "dropout > 0 in train mode needs a DropoutSource"
    return x * (src.keep(x.shape, p).to(x.dtype) / (1.0 - p))


# ---------------------------------------------------------------------------------------------
# attention (cross_modal_transformer.py:36-198, layers/SelfAttentionFamily.py:35-194)
# ---------------------------------------------------------------------------------------------
def prob_sizes(L_Q: int, L_K: int, factor: int):
    """(sample_k, n_top) -- cross_modal_transformer.py:149-153."""
    U_part = factor * int(math.ceil(math.log(L_K)))
    u = factor * int(math.ceil(math.log(L_Q)))
    return (U_part if U_part < L_K else L_K), (u if u < L_Q else L_Q)


def full_attention(q, k, v, scale=None, masked: bool = False, dropout: float = 0.0,
                   drop: Optional[DropoutSource] = None):
    """dropout(softmax(scale * Q K^T)) V -- cross_modal_transformer.py:51-69; ``masked`` = the TriangularCausalMask
    branch of the GPS copy (layers/SelfAttentionFamily.py:9-17,53-57).  (B,L,H,E) in/out."""
    E = q.shape[-1]
    scale = scale or 1.0 / math.sqrt(E)
    s = torch.einsum("blhe,bshe->bhls", _mm(q), _mm(k))
    if masked:
        L = q.shape[1]
        s = s.masked_fill(torch.triu(torch.ones(L, s.shape[-1], dtype=torch.bool), diagonal=1), float("-inf"))
    a = _drop(torch.softmax(scale * s, dim=-1), dropout, drop)  # (:63) on the (B,H,L,S) probabilities
    return torch.einsum("bhls,bshd->blhd", _mm(a), _mm(v)).contiguous()


def prob_attention(q, k, v, index_sample, factor: int, masked: bool, scale=None,
                   gps_variant: bool = False, return_top: bool = False, trace: Optional[IndexSource] = None):
    """Informer ProbSparse attention -- cross_modal_transformer.py:88-166 (A.3 of SURVEY).

    q,k,v: (B, L, H, D).  Returns (B, L_Q, H, D) for the cross-modal variant, or the
    UN-transposed (B, H, L_Q, D) tensor for the GPS variant (SelfAttentionFamily.py:165)."""
    B, L_Q, H, D = q.shape
    L_K = k.shape[1]
    Q, K, V = (t.transpose(1, 2) for t in (q, k, v))  # (B,H,L,D)
    sample_k, n_top = prob_sizes(L_Q, L_K, factor)
    assert tuple(index_sample.shape) == (L_Q, sample_k)
    # sampled scores Q[q] . K[index_sample[q, j]]  (:94-97)
    K_s = K[:, :, index_sample, :]  # (B,H,L_Q,k,D)
    qk_s = torch.einsum("bhqd,bhqjd->bhqj", _mm(Q), _mm(K_s))  # (a matmul in the reference: :96)
    M = qk_s.max(-1).values - qk_s.sum(-1) / L_K  # sparsity measure (:100)
    top = M.topk(n_top, sorted=False).indices  # (B,H,u)  (:101)
    if trace is not None and trace.forced is not None:
        own, top = top, trace.forced.pop(0).long()
        trace.tops.append(own.sort(dim=-1).values)
    elif trace is not None:
        trace.tops.append(top.sort(dim=-1).values)
        srt = M.detach().sort(dim=-1, descending=True).values
        gap = (srt[..., n_top - 1] - srt[..., n_top]) if n_top < L_Q else torch.full(srt.shape[:-1], float("inf"))
        trace.margins.append(gap / qk_s.detach().abs().amax(dim=(-1, -2)).clamp_min(1e-12))
    Q_red = torch.gather(Q, 2, top.unsqueeze(-1).expand(-1, -1, -1, D))
    scores = torch.matmul(_mm(Q_red), _mm(K).transpose(-2, -1)) * (scale or 1.0 / math.sqrt(D))  # (:107,158-160)
    if masked:  # ProbMask: key s visible to selected query i iff s <= top[i]  (:22-29)
        assert L_Q == L_K
        key_pos = torch.arange(L_K).view(1, 1, 1, L_K)
        scores = scores.masked_fill(key_pos > top.unsqueeze(-1), float("-inf"))
        ctx = V.cumsum(dim=-2)  # (:117-119)
    else:
        ctx = V.mean(dim=-2, keepdim=True).expand(B, H, L_Q, V.shape[-1]).clone()  # (:113-116)
    attn = torch.softmax(scores, dim=-1)
    upd = torch.matmul(_mm(attn), _mm(V))
    ctx = ctx.scatter(2, top.unsqueeze(-1).expand(-1, -1, -1, V.shape[-1]), upd)  # (:131-133)
    out = ctx.contiguous() if gps_variant else ctx.transpose(1, 2).contiguous()
    return (out, top) if return_top else out


def _linear(sd: SD, p: str, x):
    return F.linear(_mm(x), _mm(sd[p + ".weight"]), sd.get(p + ".bias"))


def attention_layer(sd: SD, p: str, xq, xk, xv, n_heads: int, kind: str, idx: IndexSource,
                    factor: int = 5, gps_variant: bool = False, mix: bool = False, dropout: float = 0.0,
                    drop: Optional[DropoutSource] = None):
    """AttentionLayer -- cross_modal_transformer.py:169-198 / SelfAttentionFamily.py:168-194.
    kind in {"prob", "prob_masked", "full", "full_masked"}."""
    B, L, _ = xq.shape
    S = xk.shape[1]
    q = _linear(sd, p + ".query_projection", xq).view(B, L, n_heads, -1)
    k = _linear(sd, p + ".key_projection", xk).view(B, S, n_heads, -1)
    v = _linear(sd, p + ".value_projection", xv).view(B, S, n_heads, -1)
    if kind in ("full", "full_masked"):
        out = full_attention(q, k, v, masked=kind == "full_masked", dropout=dropout, drop=drop)
    else:  # ProbAttention defines a Dropout it never applies (cross_modal_transformer.py:86)
        sample_k, _ = prob_sizes(L, S, factor)
        index_sample = idx.randint(S, (L, sample_k))
        out = prob_attention(q, k, v, index_sample, factor, kind == "prob_masked",
                             gps_variant=gps_variant, trace=idx)
    if mix and not gps_variant:
        out = out.transpose(2, 1).contiguous()
    # GPS variant: (B,H,L,D) memory reinterpreted as (B,L,H*D) -- the "head scramble" (:192)
    out = out.view(B, L, -1)
    return _linear(sd, p + ".out_projection", out)


def _act(name: str):
    return F.relu if name == "relu" else F.gelu


def _ln(sd: SD, p: str, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _ffn(sd: SD, p: str, x, activation: str, dropout: float = 0.0, drop: Optional[DropoutSource] = None):
    """dropout(Conv1d(k=1)(dropout(act(Conv1d(k=1)(x))))) on (B,L,C) -- cross_modal_transformer.py:298-299: the
    first dropout acts on the (B, d_ff, L) Conv1d layout, the second after the transpose back."""
    w1, w2 = sd[p + ".conv1.weight"].squeeze(-1), sd[p + ".conv2.weight"].squeeze(-1)
    y = _act(activation)(F.linear(_cv(x), _cv(w1), sd[p + ".conv1.bias"]))  # (Conv1d(k=1) in the reference: a convolution)
    y = _drop(y.transpose(-1, 1), dropout, drop).transpose(-1, 1)
    return _drop(F.linear(_cv(y), _cv(w2), sd[p + ".conv2.bias"]), dropout, drop)


def encoder_layer(sd: SD, p: str, x, n_heads, idx, factor, activation, gps_variant, kind: str = "prob",
                  dropout: float = 0.0, drop: Optional[DropoutSource] = None):
    """Post-LN encoder block -- cross_modal_transformer.py:288-301 (dropout sites :295,298,299)."""
    x = x + _drop(attention_layer(sd, p + ".attention", x, x, x, n_heads, kind, idx, factor,
                                  gps_variant, dropout=dropout, drop=drop), dropout, drop)
    x = _ln(sd, p + ".norm1", x)
    return _tap(p, _ln(sd, p + ".norm2", x + _ffn(sd, p, x, activation, dropout, drop)))


def decoder_layer(sd: SD, p: str, x, cross, n_heads, idx, factor, activation, gps_variant,
                  cross_kind: str, mix: bool = False, self_kind: str = "prob_masked", dropout: float = 0.0,
                  drop: Optional[DropoutSource] = None):
    """Decoder block -- cross_modal_transformer.py:223-233 / TransformerEncoderDecoder.py:106-115."""
    x = x + _drop(attention_layer(sd, p + ".self_attention", x, x, x, n_heads, self_kind, idx, factor,
                                  gps_variant, mix=mix, dropout=dropout, drop=drop), dropout, drop)
    x = _ln(sd, p + ".norm1", x)
    x = x + _drop(attention_layer(sd, p + ".cross_attention", x, cross, cross, n_heads, cross_kind, idx,
                                  factor, gps_variant, dropout=dropout, drop=drop), dropout, drop)
    x = _ln(sd, p + ".norm2", x)
    return _tap(p, _ln(sd, p + ".norm3", x + _ffn(sd, p, x, activation, dropout, drop)))


def _count(sd: SD, prefix: str) -> int:
    """Number of consecutive integer children ``prefix.N.`` present in the state dict."""
    n = 0
    while any(k.startswith(f"{prefix}.{n}.") for k in sd):
        n += 1
    return n


def circular_conv3(x, weight, bias=None, padding: int = 1):
    """Conv1d(k=3, padding_mode='circular') on (B,L,C) -> (B,L+2*padding-2,C_out).
    cross_modal_transformer.py:352-369 (padding 1, bias) / layers/Embedding.py:28-46 (no bias) /
    TransformerEncoderDecoder.py:12-18 (padding 2: output length L+2)."""
    L = x.shape[1]
    pos = torch.arange(-padding, L + padding) % L  # circularly padded source positions
    xp = x[:, pos, :]  # (B, L+2p, C)
    taps = [xp[:, t: t + L + 2 * padding - 2, :] for t in range(3)]
    y = sum(torch.matmul(_cv(taps[t]), _cv(weight[:, :, t]).t()) for t in range(3))
    return y if bias is None else y + bias


def perceive_encoder(sd: SD, p: str, x, n_heads: int, out_len: int, idx: IndexSource,
                     factor: int = 5, activation: str = "gelu", dropout: float = 0.0,
                     drop: Optional[DropoutSource] = None):
    """PerceiveEncoder.forward -- cross_modal_transformer.py:425-433."""
    L = x.shape[1]
    h = circular_conv3(x, sd[p + ".value_embedding.tokenConv.weight"],
                       sd[p + ".value_embedding.tokenConv.bias"])
    h = h + sd[p + ".position_embedding.pe"][:, :L]
    for i in range(_count(sd, p + ".encoder.attn_layers")):
        h = encoder_layer(sd, f"{p}.encoder.attn_layers.{i}", h, n_heads, idx, factor, activation,
                          gps_variant=False, dropout=dropout, drop=drop)
    h = _ln(sd, p + ".encoder.norm", h)
    return _tap(p, _linear(sd, p + ".projection", h)[:, -out_len:, :])


def perceive_decoder(sd: SD, p: str, x_enc, x_dec, n_heads: int, out_len: int, idx: IndexSource,
                     factor: int = 5, activation: str = "gelu", mix: bool = False, dropout: float = 0.0,
                     drop: Optional[DropoutSource] = None):
    """PerceiveDecoder.forward -- cross_modal_transformer.py:498-503 (cross attention = FullAttention)."""
    L = x_dec.shape[1]
    h = circular_conv3(x_dec, sd[p + ".value_embedding.tokenConv.weight"],
                       sd[p + ".value_embedding.tokenConv.bias"])
    h = h + sd[p + ".position_embedding.pe"][:, :L]
    for i in range(_count(sd, p + ".decoder.layers")):
        h = decoder_layer(sd, f"{p}.decoder.layers.{i}", h, x_enc, n_heads, idx, factor, activation,
                          gps_variant=False, cross_kind="full", mix=mix, dropout=dropout, drop=drop)
    h = _ln(sd, p + ".decoder.norm", h)
    return _tap(p, _linear(sd, p + ".projection", h)[:, -out_len:, :])


# ---------------------------------------------------------------------------------------------
# Informer GPS backbone (gps_backbone/Informer.py:105-167 + layers/*)
# ---------------------------------------------------------------------------------------------
def data_embedding(sd: SD, p: str, x, dropout: float = 0.0, drop: Optional[DropoutSource] = None):
    """DataEmbedding (timeF): dropout(circular conv (no bias) + Linear(1->d)(position as float) + PE).
    layers/Embedding.py:111-126; x_mark = arange(L) (Informer.py:119-123,152-156)."""
    B, L, _ = x.shape
    mark = torch.arange(L, dtype=torch.float32).view(1, L, 1).expand(B, L, 1)
    return _drop(circular_conv3(x, sd[p + ".value_embedding.tokenConv.weight"])
                 + F.linear(_mm(mark), _mm(sd[p + ".temporal_embedding.embed.weight"]))
                 + sd[p + ".position_embedding.pe"][:, :L], dropout, drop)


def distil_conv(sd: SD, p: str, x, training: bool, bn_state: Optional[dict] = None):
    """Conv1d(k3, circular pad 2) -> BatchNorm1d -> ELU -> MaxPool1d(3,2,1).
    layers/TransformerEncoderDecoder.py:9-29.  Train mode uses batch statistics (biased variance)
    and reports the running-stat update (momentum 0.1, unbiased variance) into ``bn_state``."""
    y = circular_conv3(x, sd[p + ".downConv.weight"], sd[p + ".downConv.bias"], padding=2)
    if training:
        mean = y.mean(dim=(0, 1))
        var = y.var(dim=(0, 1), unbiased=False)
        if bn_state is not None:
            n = y.shape[0] * y.shape[1]
            bn_state[p + ".norm.running_mean"] = 0.9 * sd[p + ".norm.running_mean"] + 0.1 * mean.detach()
            bn_state[p + ".norm.running_var"] = (0.9 * sd[p + ".norm.running_var"]
                                                 + 0.1 * var.detach() * n / (n - 1))
    else:
        mean, var = sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"]
    y = (y - mean) / torch.sqrt(var + 1e-5) * sd[p + ".norm.weight"] + sd[p + ".norm.bias"]
    y = F.elu(y)
    return _tap(p, F.max_pool1d(y.transpose(1, 2), kernel_size=3, stride=2, padding=1).transpose(1, 2))


def informer(sd: SD, p: str, x, *, pred_len: int, n_heads: int, factor: int, activation: str,
             smart_decoder: bool, training: bool, idx: IndexSource,
             bn_state: Optional[dict] = None, dropout: float = 0.0, drop: Optional[DropoutSource] = None):
    """Informer.forward -- gps_backbone/Informer.py:105-167 (``dropout``: the active probability, 0 in eval)."""
    dk = dict(dropout=dropout, drop=drop)
    tail = x[:, -1:, :].repeat(1, pred_len, 1) if smart_decoder else \
        torch.zeros(x.shape[0], pred_len, x.shape[-1])
    x_dec = torch.cat([x, tail], dim=1)
    h = data_embedding(sd, p + ".enc_embedding", x, **dk)
    n_attn = _count(sd, p + ".encoder.attn_layers")
    n_conv = _count(sd, p + ".encoder.conv_layers")
    if n_conv:  # distilling: zip(attn, conv) then one last attn layer (TransformerEncoderDecoder.py:66-71)
        for i in range(n_conv):
            h = encoder_layer(sd, f"{p}.encoder.attn_layers.{i}", h, n_heads, idx, factor,
                              activation, gps_variant=True, **dk)
            h = distil_conv(sd, f"{p}.encoder.conv_layers.{i}", h, training, bn_state)
        h = encoder_layer(sd, f"{p}.encoder.attn_layers.{n_attn - 1}", h, n_heads, idx, factor,
                          activation, gps_variant=True, **dk)
    else:
        for i in range(n_attn):
            h = encoder_layer(sd, f"{p}.encoder.attn_layers.{i}", h, n_heads, idx, factor,
                              activation, gps_variant=True, **dk)
    enc = _ln(sd, p + ".encoder.norm", h)
    d = data_embedding(sd, p + ".dec_embedding", x_dec, **dk)
    for i in range(_count(sd, p + ".decoder.layers")):
        d = decoder_layer(sd, f"{p}.decoder.layers.{i}", d, enc, n_heads, idx, factor, activation,
                          gps_variant=True, cross_kind="prob", **dk)
    d = _ln(sd, p + ".decoder.norm", d)
    return _tap(p, _linear(sd, p + ".decoder.projection", d)[:, -pred_len:, :])


def transformer_gps(sd: SD, p: str, x, *, pred_len: int, n_heads: int, activation: str, dropout: float = 0.0,
                    drop: Optional[DropoutSource] = None):
    """The vanilla ``Transformer`` GPS backbone -- gps_backbone/Transformer.py:98-141:
    FullAttention encoder, causal FullAttention + cross FullAttention decoder, decoder input = history
    followed by ``pred_len`` zero rows, no distilling, no host RNG."""
    dk = dict(dropout=dropout, drop=drop)
    x_dec = torch.cat([x, torch.zeros(x.shape[0], pred_len, x.shape[-1])], dim=1)
    h = data_embedding(sd, p + ".enc_embedding", x, **dk)
    for i in range(_count(sd, p + ".encoder.attn_layers")):
        h = encoder_layer(sd, f"{p}.encoder.attn_layers.{i}", h, n_heads, None, 0, activation, False, kind="full", **dk)
    enc = _ln(sd, p + ".encoder.norm", h)
    d = data_embedding(sd, p + ".dec_embedding", x_dec, **dk)
    for i in range(_count(sd, p + ".decoder.layers")):
        d = decoder_layer(sd, f"{p}.decoder.layers.{i}", d, enc, n_heads, None, 0, activation, False,
                          cross_kind="full", self_kind="full_masked", **dk)
    d = _ln(sd, p + ".decoder.norm", d)
    return _linear(sd, p + ".decoder.projection", d)[:, -pred_len:, :]


def multimodal_transformer(sd: SD, cfg, batch, idx: IndexSource):
    """The "simple multi-modal transformer" baseline --
    experiments/multimodal_transformer/multimodal_transformer.py:67-122: every frame of the three streams through
    the frozen conv encoder and the frame encoder (calls in the order left, right, front), Linear(2,h) on motion
    and on median-downsampled gaze, cat, vanilla Transformer backbone, cumsum from the last position."""
    gps = batch["gps"].to(torch.float32)
    motions = F.pad(gps[:, 1:] - gps[:, :-1], (0, 0, 1, 0))
    g = cfg.gps_backbone_config

    def single(video):
        B = video.shape[0]
        fmap = hrnet16_features(sd, "video_backbone._Backbone", video.flatten(0, 1).to(torch.float32))
        t = fmap.permute(0, 2, 3, 1).reshape(fmap.shape[0], -1, fmap.shape[1])
        t = torch.cat([t, -torch.ones_like(t)[:, :1, :]], dim=1)
        return perceive_encoder(sd, "frame_encoder", t, cfg.encoder_heads, 1, idx).view(B, -1, cfg.image_embedding_size)

    left = batch["left_video"]
    feats = [_linear(sd, "motion_linear", motions), single(left), single(batch.get("right_video", left)),
             single(batch["front_video"]),
             _linear(sd, "gaze_linear", median_downsampler(batch["gaze"].to(torch.float32), g.seq_len))]
    out = transformer_gps(sd, "transformer", torch.cat(feats, dim=2), pred_len=g.pred_len, n_heads=g.n_heads,
                          activation=g.activation)
    return gps[:, -1:, :] + torch.cumsum(out, dim=1)


def warmup_cosine_lr(base_lr: float, epochs: int, warmup_epochs: int, max_epochs: int,
                     warmup_start_lr: float = 0.0, eta_min: float = 0.0):
    """Learning rate in force during epochs 0 .. epochs-1 under LinearWarmupCosineAnnealingLR stepped once
    per epoch (optimizers/lr_scheduler.py:62-113, the CHAINABLE form Lightning drives; driver:
    full_comparison.py:702-709 with warmup_epochs=2, max_epochs=200).  Python floats, same expression
    order as the reference so the values are equal bit for bit."""
    lrs, lr = [], base_lr
    for e in range(epochs):
        if e == warmup_epochs:
            lr = base_lr
        elif e == 0:
            lr = warmup_start_lr
        elif e < warmup_epochs:
            lr = lr + (base_lr - warmup_start_lr) / (warmup_epochs - 1)
        elif (e - 1 - max_epochs) % (2 * (max_epochs - warmup_epochs)) == 0:
            lr = lr + (base_lr - eta_min) * (1 - math.cos(math.pi / (max_epochs - warmup_epochs))) / 2
        else:
            lr = ((1 + math.cos(math.pi * (e - warmup_epochs) / (max_epochs - warmup_epochs)))
                  / (1 + math.cos(math.pi * (e - warmup_epochs - 1) / (max_epochs - warmup_epochs)))
                  * (lr - eta_min) + eta_min)
        lrs.append(lr)
    return lrs


# ---------------------------------------------------------------------------------------------
# frozen conv encoder: HRNet-16 trunk (inverse_form_layers/hrnetv2.py:282-500, config.py:177-206)
# ---------------------------------------------------------------------------------------------
def _conv_bn(sd: SD, conv: str, bn: Optional[str], x, stride=1, relu=False):
    w = sd[conv + ".weight"]
    y = F.conv2d(_cv(x), _cv(w), None, stride=stride, padding=(w.shape[-1] - 1) // 2 if w.shape[-1] == 3 else 0)
    if bn is not None:  # eval-mode BatchNorm2d (frozen, InverseForm.py:69-71)
        y = F.batch_norm(y, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"],
                         sd[bn + ".bias"], False, 0.1, 1e-5)
    return F.relu(y) if relu else y


def _basic_block(sd, p, x):  # hrnetv2.py:45-61
    y = _conv_bn(sd, p + ".conv1", p + ".bn1", x, relu=True)
    y = _conv_bn(sd, p + ".conv2", p + ".bn2", y)
    return F.relu(y + x)


def _bottleneck(sd, p, x):  # hrnetv2.py:79-99
    y = _conv_bn(sd, p + ".conv1", p + ".bn1", x, relu=True)
    y = _conv_bn(sd, p + ".conv2", p + ".bn2", y, relu=True)
    y = _conv_bn(sd, p + ".conv3", p + ".bn3", y)
    res = _conv_bn(sd, p + ".downsample.0", p + ".downsample.1", x) \
        if (p + ".downsample.0.weight") in sd else x
    return F.relu(y + res)


def _up(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


def _hr_module(sd, p, xs):  # HighResolutionModule.forward, hrnetv2.py:250-277
    nb = len(xs)
    xs = list(xs)
    for b in range(nb):
        for k in range(_count(sd, f"{p}.branches.{b}")):
            xs[b] = _basic_block(sd, f"{p}.branches.{b}.{k}", xs[b])
    outs = []
    for i in range(nb):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:  # 1x1 conv + BN, bilinear up to branch i's resolution
                t = _up(_conv_bn(sd, f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", xs[j]),
                        xs[i].shape[-2:])
            else:  # chain of stride-2 3x3 convs; ReLU on all but the last
                t = xs[j]
                for k in range(i - j):
                    t = _conv_bn(sd, f"{p}.fuse_layers.{i}.{j}.{k}.0", f"{p}.fuse_layers.{i}.{j}.{k}.1",
                                 t, stride=2, relu=(k != i - j - 1))
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hrnet16_features(sd: SD, p: str, images):
    """(N,3,H,W) -> (N,240,8,8): HRNet16 trunk output[-1] + AdaptiveAvgPool2d((8,8)).
    hrnetv2.py:430-500 (only the final concat is used, InverseForm.py:66-67)."""
    x = images.to(torch.float32)
    x = _conv_bn(sd, p + ".conv0", None, x, stride=2)  # DOWN_CONV 2x2 s2 (config.py:268-269)
    x = _conv_bn(sd, p + ".conv1", p + ".bn1", x, stride=2, relu=True)
    x = _conv_bn(sd, p + ".conv2", p + ".bn2", x, stride=2, relu=True)
    for k in range(_count(sd, p + ".layer1")):
        x = _bottleneck(sd, f"{p}.layer1.{k}", x)
    xs = [_conv_bn(sd, p + ".transition1.0.0", p + ".transition1.0.1", x, relu=True),
          _conv_bn(sd, p + ".transition1.1.0.0", p + ".transition1.1.0.1", x, stride=2, relu=True)]
    for stage, nb in (("stage2", 2), ("stage3", 3), ("stage4", 4)):
        if len(xs) < nb:  # new lowest-resolution branch from the previous lowest (transitionN)
            t = f"{p}.transition{nb - 1}.{nb - 1}.0"
            xs.append(_conv_bn(sd, t + ".0", t + ".1", xs[-1], stride=2, relu=True))
        for m in range(_count(sd, f"{p}.{stage}")):
            xs = _hr_module(sd, f"{p}.{stage}.{m}", xs)
    size = xs[0].shape[-2:]
    feats = torch.cat([xs[0]] + [_up(t, size) for t in xs[1:]], dim=1)
    return F.adaptive_avg_pool2d(feats, (8, 8))


# ---------------------------------------------------------------------------------------------
# helpers (utils/vector.py, utils/filter.py), losses and scores
# ---------------------------------------------------------------------------------------------
def angle_and_norm(v):  # utils/vector.py:85-111
    v = v.float()
    return torch.atan2(v[..., 1], v[..., 0]).unsqueeze(-1), torch.linalg.vector_norm(v, dim=-1, keepdim=True)


def rotate(v, angle):
    """R(angle) . v with R=[[c,-s],[s,c]], fp32 -- utils/vector.py:6-54.  v (B,L,2), angle (B,1,1)."""
    c, s = torch.cos(angle.float()).reshape(-1, 1), torch.sin(angle.float()).reshape(-1, 1)
    x, y = v[..., 0].float(), v[..., 1].float()
    return torch.stack([c * x - s * y, s * x + c * y], dim=-1).to(v.dtype)


def median_downsampler(x, target: int):
    """Lower median over consecutive windows of T//target samples -- utils/filter.py:5-43."""
    B, T, C = x.shape
    if target >= T:
        raise ValueError("Target length must be less than the current time steps.")
    w = T // target
    win = x[:, : w * target].reshape(B, target, w, C)
    return win.sort(dim=2).values[:, :, (w - 1) // 2, :]


def future_discounted_loss(pred, true, discount: float, kind: str = "smooth_l1",
                           epsilon: Optional[float] = 1.0):
    """FutureDiscountedLoss.forward -- losses/future_discounted_mse.py:56-95."""
    T = pred.shape[1]
    f = torch.pow(torch.tensor(float(discount)), torch.arange(T)).view(1, T, *([1] * (pred.dim() - 2)))
    if kind == "smooth_l1":  # epsilon is evaluated but unused on this path (:85-93)
        return (F.smooth_l1_loss(pred, true, reduction="none") * f).mean()
    err = pred - true
    err = torch.where(err.abs() < epsilon, torch.zeros_like(err), err)
    return ((err.abs() if kind == "mae" else err.pow(2)) * f).mean()


def discount_for_epoch(table, epoch: int, current: Optional[float] = None) -> float:
    """current_discount_factor logic (:47-50,71-74): starts at table[0], switches when the epoch
    is a key.  For a stateless oracle: the value of the largest key <= epoch."""
    if isinstance(table, float):
        return table
    keys = sorted(k for k in table if k <= epoch)
    return table[keys[-1]] if keys else table[0]


def ade(pred, true):  # score/error.py:29
    return torch.linalg.vector_norm(pred - true, dim=-1).mean()


def fde(pred, true):  # score/error.py:51 -- indexes dim 0 (last BATCH element), Frobenius over (T,2)
    return torch.linalg.vector_norm(pred[-1] - true[-1])


# ---------------------------------------------------------------------------------------------
# Routeformer (models/routeformer.py)
# ---------------------------------------------------------------------------------------------
class OracleRouteformer:
    """Functional Routeformer over a state dict.  ``cfg`` is any object exposing the
    RouteformerConfig fields (reference's or this repo's)."""

    def __init__(self, cfg, sd: SD, *, training: bool = False, idx: Optional[IndexSource] = None,
                 drop: Optional[DropoutSource] = None):
        self.cfg, self.sd, self.training = cfg, sd, training
        self.idx = idx or IndexSource()
        # nn.Dropout follows the MODULE's train/eval state (so it is also active in the target-side feature pass of
        # a train step, preprocess_batch(target, training=False) -- full_comparison.py:482), not the `training` argument
        self.drop = drop or DropoutSource()
        self.p_feat = float(getattr(cfg, "feature_dropout", 0.0)) if training else 0.0
        self.p_gps = float(getattr(cfg.gps_backbone_config, "dropout", 0.0)) if training else 0.0
        self.bn_state: dict = {}
        g = cfg.gps_backbone_config
        self.gps_kw = dict(n_heads=g.n_heads, factor=g.factor, activation=g.activation,
                           smart_decoder=(cfg.decoder_mode == "smart"))
        self.pred_len = g.pred_len
        self.seq_len = g.seq_len

    # -- visual path -------------------------------------------------------------------------
    def _frame_indices(self, T: int, fps: int):
        rel = self.cfg.output_fps // fps
        return torch.flip(torch.arange(T - 1, 0, -rel), dims=[0])  # never frame 0 (:418-419)

    def _single_video(self, frames, drop: bool, training: bool):
        """(N,3,H,W) -> (N,E) -- routeformer.py:463-491."""
        E = self.cfg.image_embedding_size
        if drop and training:
            return torch.zeros(frames.shape[0], E)
        f = hrnet16_features(self.sd, "video_backbone._Backbone", frames)
        tok = f.permute(0, 2, 3, 1).reshape(f.shape[0], -1, f.shape[1])
        tok = torch.cat([tok, -torch.ones_like(tok[:, :1])], dim=1)
        out = perceive_encoder(self.sd, "frame_encoder", tok, self.cfg.encoder_heads, 1, self.idx,
                               dropout=self.p_feat, drop=self.drop)
        return out.reshape(frames.shape[0], E)

    def _timeline(self, feats, B, T, indices):
        full = torch.zeros(B, T, feats.shape[-1])
        full[:, indices] = feats.view(B, -1, feats.shape[-1])
        return full

    def _scene(self, batch, training: bool):
        """routeformer.py:397-461: right stream first, then left."""
        left = batch["left_video"]
        right = batch.get("right_video", left)
        drop_left, drop_right = False, "right_video" not in batch
        if self.cfg.view_dropout > 0.0 and training:
            drop_one = bool(torch.rand(1) < self.cfg.view_dropout)
            drop_left = drop_one and bool(torch.rand(1) < 0.5)
            drop_right = (drop_one and not drop_left) or "right_video" not in batch
        B, T = left.shape[:2]
        ind = self._frame_indices(T, self.cfg.video_fps)
        rf = self._single_video(right[:, ind].flatten(0, 1), drop_right, training)
        lf = self._single_video(left[:, ind].flatten(0, 1), drop_left, training)
        return self._timeline(lf, B, T, ind), self._timeline(rf, B, T, ind)

    def _gaze_video(self, batch, training: bool):
        v = batch["front_video"]
        B, T = v.shape[:2]
        ind = self._frame_indices(T, self.cfg.gaze_fps)
        f = self._single_video(v[:, ind].flatten(0, 1), False, training)
        return self._timeline(f, B, T, ind)

    def preprocess_batch(self, batch, training: Optional[bool] = None):
        """routeformer.py:254-348."""
        cfg, sd = self.cfg, self.sd
        if training is None:
            training = self.training
        gps = batch["gps"].to(torch.float32)
        if cfg.motion_noise > 0.0 and self.training:
            gps = gps + torch.randn_like(gps) * cfg.motion_noise
        mv = gps[:, 1:] - gps[:, :-1]
        if cfg.normalize_motion:
            mv = (mv - cfg.motion_mean) / cfg.motion_std
        motion = F.pad(mv, (0, 0, 1, 0))
        if not cfg.with_video:
            return motion, []
        feats = []
        if cfg.with_scene:
            feats.extend(self._scene(batch, training))
        if cfg.with_gaze:
            drop_gaze = bool(torch.rand(1) < cfg.gaze_dropout) if (cfg.gaze_dropout > 0.0 and training) else False
            if drop_gaze:
                fv = batch["front_video"]
                g = torch.zeros(fv.shape[0], fv.shape[1], cfg.image_embedding_size)
            else:
                gv = self._gaze_video(batch, training)
                gp = median_downsampler(batch["gaze"].to(torch.float32), self.seq_len)
                gp = perceive_encoder(sd, "gaze_encoder", gp, cfg.encoder_heads, self.seq_len, self.idx,
                                      dropout=self.p_feat, drop=self.drop)
                g = perceive_decoder(sd, "gaze_video_decoder", gv, gp, cfg.cross_modal_decoder_heads,
                                     self.seq_len, self.idx, mix=False, dropout=self.p_feat,
                                     drop=self.drop)[:, : gv.shape[1]]
            feats.append(g)
        if cfg.with_scene:
            feats[0] = feats[0] + sd["left_video_embedding"]
            feats[1] = feats[1] + sd["right_video_embedding"]
        if cfg.with_gaze:
            feats[-1] = feats[-1] + sd["gaze_video_embedding"]
        seq = torch.cat(feats + [torch.zeros_like(feats[-1]) + sd["video_output_embedding"]], dim=1)
        vis = perceive_encoder(sd, "video_encoder", seq, cfg.encoder_heads, self.seq_len, self.idx,
                               dropout=self.p_feat, drop=self.drop)
        return motion, vis

    # -- GPS path ----------------------------------------------------------------------------
    def _forward(self, motion, vis, pred_len):
        """routeformer.py:204-252."""
        cfg = self.cfg
        angle, norm = angle_and_norm(motion)
        origin = angle[:, -1:] if cfg.rotate_motion else angle[:, :1]
        nangle = (angle - origin) / torch.pi
        accel = F.pad(norm[:, 1:] - norm[:, :-1], (0, 0, 1, 0))
        if cfg.rotate_motion:
            motion = rotate(motion, -origin)
        parts = [torch.cat([motion, nangle, norm, accel], dim=-1)]
        if cfg.with_video:
            parts.append(vis)
        if cfg._only_motion:
            parts[-1] = torch.zeros_like(parts[-1])
        x = torch.cat(parts, dim=-1)
        if getattr(self, "gps_kind", "informer") == "transformer":
            out = transformer_gps(self.sd, "gps_backbone", x, pred_len=pred_len, n_heads=self.gps_kw["n_heads"],
                                  activation=self.gps_kw["activation"], dropout=self.p_gps, drop=self.drop)
        else:
            out = informer(self.sd, "gps_backbone", x, pred_len=pred_len, training=self.training,
                           idx=self.idx, bn_state=self.bn_state, dropout=self.p_gps, drop=self.drop, **self.gps_kw)
        if cfg.decoder_mode == "recursive":
            out = out + (x[:, -1:, :] if cfg.dense_prediction else x[:, -1:, :2])
        if cfg.rotate_motion:
            out = torch.cat([rotate(out[..., :2], origin), out[..., 2:]], dim=-1)
        return out

    def _post(self, last_gps, out):
        """routeformer.py:350-395."""
        cfg = self.cfg
        mv = out[..., :2]
        if cfg.normalize_motion:
            mv = mv * cfg.motion_std + cfg.motion_mean
        pos = (last_gps + torch.cumsum(mv, dim=1)).to(last_gps.dtype)
        rest = out[..., 2:]
        vis = None
        if cfg.with_video and cfg.dense_prediction:
            vis, rest = rest[..., : cfg.image_embedding_size], rest[..., cfg.image_embedding_size:]
        assert rest.shape[-1] == 0
        return mv, pos, vis

    def forward(self, batch):
        """routeformer.py:124-202 (incl. the eval-time autoregressive loop)."""
        cfg = self.cfg
        motion, vis = self.preprocess_batch(batch)
        last = batch["gps"][:, -1:, :]
        if self.training or not cfg.autoregressive:
            _, pos, fvis = self._post(last, self._forward(motion, vis, self.pred_len))
        else:
            outs, done, step = [], 0, cfg.autoregressive_step_size
            while done < self.pred_len:
                mv, p, fvis = self._post(last, self._forward(motion, vis, step))
                outs.append((p, fvis))
                motion = torch.cat([motion[:, step:], mv], dim=1)
                last = p[:, -1:, :]
                vis = torch.cat([vis[:, step:], fvis], dim=1)
                done += step
            pos = torch.cat([o[0] for o in outs], dim=1)[:, : self.pred_len]
            if cfg.with_video:
                fvis = torch.cat([o[1] for o in outs], dim=1)[:, : self.pred_len]
        return (pos, fvis) if cfg.dense_prediction else pos

    # -- the evaluation protocol (experiments/full_comparison.py:654-679) ------------------------
    def eval_step(self, item, epoch: int = 0, passes: int = 5):
        """Mean trajectory over ``passes`` forwards (each consuming fresh key samples from ``self.idx``),
        then per-sample discounted loss / ADE / FDE.  -> (losses (B,), ades (B,), fdes (B,), mean (B,P,2))."""
        gamma = discount_for_epoch(self.cfg.discount_factor, epoch)
        runs = []
        for _ in range(passes):
            out = self.forward(item["train"])
            runs.append(out[0] if self.cfg.dense_prediction else out)
        mean = torch.stack(runs).mean(dim=0)
        tgt = item["target"]["gps"]
        rows = [(future_discounted_loss(mean[i:i + 1], tgt[i:i + 1], gamma), ade(mean[i:i + 1], tgt[i:i + 1]),
                 fde(mean[i:i + 1], tgt[i:i + 1])) for i in range(mean.shape[0])]
        return (torch.stack([r[0] for r in rows]), torch.stack([r[1] for r in rows]),
                torch.stack([r[2] for r in rows]), mean)

    # -- the train-step recipe (experiments/full_comparison.py:476-521) -------------------------
    def train_step(self, item, epoch: int = 0):
        cfg = self.cfg
        gamma = discount_for_epoch(cfg.discount_factor, epoch)
        target_gps = item["target"]["gps"].to(torch.float32)
        res = {}
        if cfg.dense_prediction:
            pos, fvis = self.forward(item["train"])
            _, tvis = self.preprocess_batch(item["target"], training=False)
            tvis = tvis[:, : fvis.shape[1]].detach()
            traj = future_discounted_loss(pos, target_gps, gamma)
            dense = future_discounted_loss(fvis, tvis, gamma)
            w = (cfg.dense_loss_ratio * traj / torch.clamp(dense, min=1e-6)).detach() if epoch >= 10 else 0
            loss = traj + w * dense
            res.update(dense_loss=dense, future_vis=fvis, target_vis=tvis)
        else:
            pos = self.forward(item["train"])
            traj = future_discounted_loss(pos, target_gps, gamma)
            loss = traj
        res.update(loss=loss, traj_loss=traj, future_gps=pos, ade=ade(pos, target_gps),
                   fde=fde(pos, target_gps))
        return res
