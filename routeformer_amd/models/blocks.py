"""Transformer building blocks of the Routeformer hot path, executing on librf_hip.so.

These modules are *parameter containers with the reference's state_dict layout* (so reference
checkpoints load) whose ``forward`` hands raw weights to the HIP kernels through
``routeformer_amd.kernels``.  The torch ``nn.Linear`` / ``nn.Conv1d`` / ``nn.LayerNorm`` children are
never *called*: they only own parameters (and give the reference's default initialisation).

Interface mirrored (file:line in /root/reference):
  PerceiveEncoder / PerceiveDecoder ............ routeformer/models/cross_modal_transformer.py:372-503
  AttentionLayer (both variants) ............... cross_modal_transformer.py:169-198,
                                                 gps_backbone/layers/SelfAttentionFamily.py:168-194
  EncoderLayer / DecoderLayer (post-LN) ........ cross_modal_transformer.py:201-301
  distilling ConvLayer ......................... gps_backbone/layers/TransformerEncoderDecoder.py:9-29
  DataEmbedding (timeF) ........................ gps_backbone/layers/Embedding.py:111-126
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from routeformer_amd import kernels as K


# ---------------------------------------------------------------------------------------------
# host RNG for the ProbSparse key samples (cross_modal_transformer.py:95): drawn on the CPU from the
# global generator in call order, exactly like the reference, then shipped to the device.
# ---------------------------------------------------------------------------------------------
class IndexSampler:
    """Process-wide source of the HOST random draws of the hot path: the ``index_sample`` tensors of ProbSparse
    attention and the view / gaze dropout decisions.

    ``draw_host`` performs the reference's own call ``torch.randint(L_K, (L_Q, sample_k))`` on the host, so
    a ``torch.manual_seed`` reproduces the reference's samples; ``bernoulli(p)`` is its ``torch.rand(1) < p``
    (routeformer.py:301,406-407), drawn from the same generator at the same point of the sequence.  ``replay``
    injects recorded samples (tests); ``log`` keeps what was drawn (draw-order tests).

    Static mode (HIP-graph replay, ``engine.GraphedTrainEngine``): the sequence of draws of one step is fixed by
    the shapes and by the dropout decisions made so far, so the engine records one PLAN (list of draw operations)
    per decision variant up front.  Before every replay ``refill_static`` walks the plans on the host -- same
    calls, same order as the reference; a decision narrows the candidate variants -- into one pinned buffer,
    ships it with a single async copy and returns the variant, whose graph the engine then replays; inside a
    capture ``draw`` / ``bernoulli`` hand out views of the static device buffer / the variant's decisions in
    call order (no host work and no copies inside the captured region)."""

    def __init__(self):
        self.replay: Optional[list] = None
        self.log: Optional[list] = None
        self.plan: Optional[list] = None      # recording: [("int", L_K, L_Q, k) | ("bern", p, outcome)] of one step
        self.forcing: Optional[list] = None   # plan recording: decisions to impose (exhausted -> False), no host draw
        self._static = None                   # (device buffer, pinned buffer, [plan per variant])
        self._variant = 0
        self._cursor = 0

    def draw_host(self, L_K: int, L_Q: int, sample_k: int) -> torch.Tensor:
        """One host draw (int64, CPU) -- consumes the global CPU generator exactly like the reference."""
        if self.replay is not None:
            t = self.replay.pop(0)
            assert tuple(t.shape) == (L_Q, sample_k), (tuple(t.shape), (L_Q, sample_k))
        else:
            t = torch.randint(L_K, (L_Q, sample_k))
        if self.log is not None:
            self.log.append(t.clone())
        return t

    def draw(self, L_K: int, L_Q: int, sample_k: int, device) -> torch.Tensor:
        """(L_Q, sample_k) int32 on ``device``."""
        if self._static is not None:
            dev_buf, _, plans = self._static
            op = plans[self._variant][self._cursor]
            assert op[:4] == ("int", L_K, L_Q, sample_k), "draw sequence changed since the graph was planned"
            self._cursor += 1
            return dev_buf[op[4]:op[4] + L_Q * sample_k].view(L_Q, sample_k)
        if self.plan is not None:
            self.plan.append(("int", L_K, L_Q, sample_k))
        return self.draw_host(L_K, L_Q, sample_k).to(torch.int32).to(device, non_blocking=True)

    def bernoulli(self, p: float) -> bool:
        """The reference's host decision ``bool(torch.rand(1) < p)`` (view / gaze dropout)."""
        if self._static is not None:
            op = self._static[2][self._variant][self._cursor]
            assert op[0] == "bern" and op[1] == p, "draw sequence changed since the graph was planned"
            self._cursor += 1
            return op[2]
        if self.forcing is not None:
            out = self.forcing.pop(0) if self.forcing else False
        else:
            out = bool(torch.rand(1) < p)
        if self.plan is not None:
            self.plan.append(("bern", p, out))
        return out

    # -- static mode ------------------------------------------------------------------------------
    def make_static(self, plans, device):
        """``plans``: one op list per decision variant (a single list of ("int", ...) ops = no decisions)."""
        if plans and not isinstance(plans[0], list):
            plans = [plans]
        plans = [[("int",) + tuple(op) if op[0] not in ("int", "bern") else tuple(op) for op in plan] for plan in plans]
        laid, size = [], 0
        for plan in plans:
            off, ops = 0, []
            for op in plan:
                if op[0] == "int":
                    ops.append(op[:4] + (off,))
                    off += op[2] * op[3]
                else:
                    ops.append(op)
            laid.append(ops)
            size = max(size, off)
        host = torch.zeros(max(size, 1), dtype=torch.int32)
        self._static = (torch.zeros(max(size, 1), dtype=torch.int32, device=device),
                        host.pin_memory() if torch.cuda.is_available() else host, laid)
        self._variant, self._cursor = 0, 0

    @property
    def n_variants(self) -> int:
        return len(self._static[2]) if self._static is not None else 0

    def decisions(self, variant: int):
        return tuple(op[2] for op in self._static[2][variant] if op[0] == "bern")

    def refill_static(self) -> int:
        """Host side of one step: the reference's draws in order (a decision narrows the candidate variants), then
        ONE async H2D copy.  Returns the variant the decisions selected."""
        dev_buf, pinned, plans = self._static
        ev = self.__dict__.get("_copied")
        if ev is not None:
            ev.synchronize()  # the previous step's copy must have left the pinned buffer before it is rewritten
        cand, pos = list(range(len(plans))), 0
        while pos < len(plans[cand[0]]):
            op = plans[cand[0]][pos]
            if op[0] == "bern":
                out = bool(torch.rand(1) < op[1])
                cand = [v for v in cand if plans[v][pos][0] == "bern" and plans[v][pos][2] == out]
                if not cand:
                    raise RuntimeError("host dropout decisions selected a step variant that was not planned")
            else:
                _, lk, lq, k, off = op
                pinned[off:off + lq * k].copy_(self.draw_host(lk, lq, k).view(-1))
            pos += 1
        assert len(cand) == 1, "two planned variants with the same decisions"
        dev_buf.copy_(pinned, non_blocking=True)
        if dev_buf.is_cuda:
            self._copied = torch.cuda.Event()
            self._copied.record()
        self._variant, self._cursor = cand[0], 0
        return cand[0]

    def select_static(self, variant: int):
        """Capture / eager pass of one variant on the static buffers."""
        self._variant, self._cursor = variant, 0

    def rewind_static(self):
        self._cursor = 0

    def drop_static(self):
        self._static, self._cursor, self._variant = None, 0, 0


SAMPLER = IndexSampler()


def _tail(h, n: int):
    """h[:, -n:, :] -- the tensor itself when that is all of it (the gaze encoder / decoder keep every position): a
    full-range slice is still an autograd node whose backward is a zero-fill and a copy."""
    return h if n >= h.shape[1] else h[:, -n:, :]


def _dropout(x, p: float, training: bool):
    """nn.Dropout(p): device-side Philox mask, regenerated (not stored) in backward (kernels.dropout)."""
    return K.dropout(x, p, training)


class PositionalEmbedding(nn.Module):
    """Fixed sin/cos table kept as the buffer ``pe`` (1, max_len, d)."""

    def __init__(self, d_model: int, max_len: int = 5000):
        super().__init__()
        pos = torch.arange(0, max_len).float().unsqueeze(1)
        freq = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()
        pe = torch.zeros(max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * freq)
        pe[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, L: int):
        return self.pe[:, :L]


class TokenEmbedding(nn.Module):
    """Circular Conv1d(k=3) token embedding; bias only in the cross-modal flavour."""

    def __init__(self, c_in: int, d_model: int, bias: bool):
        super().__init__()
        self.tokenConv = nn.Conv1d(c_in, d_model, kernel_size=3, padding=1, padding_mode="circular",
                                   bias=bias)
        nn.init.kaiming_normal_(self.tokenConv.weight, mode="fan_in", nonlinearity="leaky_relu")

    def forward(self, x, residual=None):
        return K.circular_conv3(x, self.tokenConv.weight, self.tokenConv.bias, pad=1, residual=residual)


class _TimeFeature(nn.Module):
    def __init__(self, d_model: int):
        super().__init__()
        self.embed = nn.Linear(1, d_model, bias=False)


class DataEmbedding(nn.Module):
    """token conv (no bias) + Linear(1->d)(position index) + PE, then dropout."""

    def __init__(self, c_in: int, d_model: int, dropout: float):
        super().__init__()
        self.value_embedding = TokenEmbedding(c_in, d_model, bias=False)
        self.position_embedding = PositionalEmbedding(d_model)
        self.temporal_embedding = _TimeFeature(d_model)
        self.p = dropout

    def forward(self, x):
        L = x.shape[1]
        # rank-1 time feature (the position index l as the one "timeF" feature) + positional table: an (L, d) table built
        # in one launch, added to every sequence in the token convolution's epilogue
        offset = K.time_table(self.temporal_embedding.embed.weight, self.position_embedding.pe, L)
        return _dropout(self.value_embedding(x, residual=offset), self.p, self.training)


class AttentionLayer(nn.Module):
    """Q/K/V/O projections around the attention kernel.

    kind: "prob" | "prob_masked" | "full" | "full_masked" (causal softmax attention, the
    TriangularCausalMask branch of layers/SelfAttentionFamily.py:53-57).  ``gps_variant`` reproduces the Informer copy whose core
    returns (B,H,L,D) and is then *viewed* as (B,L,H*D) without a transpose."""

    def __init__(self, kind: str, d_model: int, n_heads: int, factor: int = 5, gps_variant: bool = False,
                 mix: bool = False, attn_dropout: float = 0.0):
        super().__init__()
        d_head = d_model // n_heads
        self.query_projection = nn.Linear(d_model, d_head * n_heads)
        self.key_projection = nn.Linear(d_model, d_head * n_heads)
        self.value_projection = nn.Linear(d_model, d_head * n_heads)
        self.out_projection = nn.Linear(d_head * n_heads, d_model)
        self.n_heads, self.kind, self.factor = n_heads, kind, factor
        self.gps_variant, self.mix, self.attn_dropout = gps_variant, mix, attn_dropout

    def packing_groups(self):
        """Parameters the training engine should lay out back to back (in this order) in its flat buffers."""
        q, k, v = self.query_projection, self.key_projection, self.value_projection
        return [[q.weight, k.weight, v.weight], [q.bias, k.bias, v.bias]]

    def forward(self, x, memory=None, idx=None, idx_group: int = 0, norm=None):
        """``norm`` = the LayerNorm that follows ``x + attention(x)`` in the layer: the out-projection, the
        residual add and the norm then run as one launch and the NORMALISED tensor is returned.
        Self-attention when ``memory is None`` (one packed QKV GEMM), else queries from ``x`` and
        keys/values from ``memory`` (packed KV GEMM).  ``idx`` (G,L,k) int32 on the device = pre-drawn
        key samples, one table per ``idx_group`` consecutive sequences (stream-batched encoders)."""
        B, L, _ = x.shape
        H = self.n_heads
        qp, kp, vp = self.query_projection, self.key_projection, self.value_projection
        HE = qp.weight.shape[0]
        E = HE // H
        pk = self.__dict__.get("_packed") if K.SINK.active else None
        # with a norm to follow, the projection that consumes x also hands back an alias of x for the skip
        # connection, so both gradients of x meet inside that projection's dX GEMM (no autograd add launch)
        fork = norm is not None and torch.is_grad_enabled() and x.requires_grad
        skip = x
        if memory is None:
            S = L
            if pk is not None:  # [Wq;Wk;Wv] are adjacent in the engine's flat buffers: no cat, one dW GEMM
                a = K.linear_packed(x.reshape(B * L, -1), pk["w"], pk["b"], pk["gw"], pk["gb"], fork=fork)
            else:
                w = torch.cat([qp.weight, kp.weight, vp.weight], dim=0)
                b = torch.cat([qp.bias, kp.bias, vp.bias], dim=0)
                a = K.linear(x.reshape(B * L, -1), w, b, fork=fork)
            if fork:
                a, skip = a[0], a[1].view(x.shape)
            bm = a
            offs = (0, HE, 2 * HE)
        else:
            S = memory.shape[1]
            a = K.linear(x.reshape(B * L, -1), qp.weight, qp.bias, fork=fork)
            if fork:
                a, skip = a[0], a[1].view(x.shape)
            if pk is not None:
                bm = K.linear_packed(memory.reshape(B * S, -1), pk["w"][HE:], pk["b"][HE:], pk["gw"][HE:], pk["gb"][HE:])
            else:
                w = torch.cat([kp.weight, vp.weight], dim=0)
                b = torch.cat([kp.bias, vp.bias], dim=0)
                bm = K.linear(memory.reshape(B * S, -1), w, b)
            offs = (0, 0, HE)
        dims = (B, H, L, S, E)
        layout = 1 if self.gps_variant else 0
        # nn.Dropout on the softmax probabilities (FullAttention only, cross_modal_transformer.py:63): in-kernel
        drop_p = self.attn_dropout if self.training else 0.0
        want_map = self.__dict__.get("output_attention", False)
        if want_map:  # the dense map is rebuilt from the kernel's own selection (recorded for this one call)
            if drop_p > 0.0:
                raise NotImplementedError("output_attention with attention dropout in train mode (the dropped map)")
            rec_outer, K.TOPS.record = K.TOPS.record, []
        if self.kind == "full":
            ctx = K.attention(a, bm, offs, dims, 0, out_layout=layout, drop_p=drop_p)
        elif self.kind == "full_masked":
            # causal softmax attention = the masked ProbSparse kernel with EVERY query row active (imposed
            # selection 0..L-1: the sampling stage is skipped, no lazy rows remain)
            every = torch.arange(L, device=x.device, dtype=torch.int32).expand(B, H, L).contiguous()
            ctx = K.attention(a, bm, offs, dims, 2, n_top=L, out_layout=layout, forced_top=every, drop_p=drop_p)
        else:
            sample_k, n_top = K.prob_sizes(L, S, self.factor)
            if idx is None:
                idx = SAMPLER.draw(S, L, sample_k, x.device)
            ctx = K.attention(a, bm, offs, dims, 2 if self.kind == "prob_masked" else 1, index_sample=idx,
                              n_top=n_top, out_layout=layout, idx_group=idx_group)
        if want_map:
            mine, K.TOPS.record = K.TOPS.record, rec_outer
            if rec_outer is not None:
                rec_outer.extend(mine)
            mode = {"full": 0, "full_masked": 2, "prob": 1, "prob_masked": 2}[self.kind]
            self.__dict__["attention_map"] = K.attention_map(a, bm, offs, dims, mode, mine[-1] if mine else None)
        if self.mix and not self.gps_variant:
            ctx = ctx.transpose(2, 1).contiguous()
        ctx = ctx.view(B, L, HE)  # GPS variant: (B,H,L,D) memory reinterpreted -- the head scramble
        if norm is not None:
            # (the context goes to the out-projection and nowhere else)
            return K.linear_add_layer_norm(ctx, self.out_projection.weight, self.out_projection.bias, skip, norm.weight,
                                           norm.bias, norm.eps, sole_consumer=True)
        return K.linear(ctx, self.out_projection.weight, self.out_projection.bias)


class EncoderLayer(nn.Module):
    def __init__(self, attention: AttentionLayer, d_model: int, d_ff: Optional[int], dropout: float,
                 activation: str):
        super().__init__()
        d_ff = d_ff or 4 * d_model
        self.attention = attention
        self.conv1 = nn.Conv1d(d_model, d_ff, kernel_size=1)
        self.conv2 = nn.Conv1d(d_ff, d_model, kernel_size=1)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.p = dropout
        self.act = "relu" if activation == "relu" else "gelu"

    def _ffn_norm(self, x, norm, unfold: bool = False):
        """LayerNorm(x + dropout(conv2(dropout(act(conv1 x))))) -- cross_modal_transformer.py:297-301.
        ``unfold``: -> (output, is it the im2col image of the distilling convolution that follows)."""
        if self.p > 0.0 and self.training:  # dropout sites between the products: the unfused FFN + fused add-norm
            y, skip = K.ffn(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias, self.act,
                            fork=True, drop_p=self.p)
            y = K.add_layer_norm(skip, y, norm.weight, norm.bias, norm.eps)
            return (y, False) if unfold else y
        # (``x`` -- the output of norm1 / norm2 -- goes to this FFN + skip and nowhere else: its gradient may travel as slabs)
        return K.ffn_add_layer_norm(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias, self.act,
                                    norm.weight, norm.bias, norm.eps, sole_consumer=True, **({"unfold": True} if unfold else {}))

    def forward(self, x, idx=None, idx_group: int = 0, unfold: bool = False):
        """``unfold`` (the Informer encoder's distilling layers, Encoder._forward): see ``_ffn_norm``."""
        if self.p > 0.0 and self.training:
            x = K.add_layer_norm(x, _dropout(self.attention(x, idx=idx, idx_group=idx_group), self.p, self.training),
                                 self.norm1.weight, self.norm1.bias)
        else:
            x = self.attention(x, idx=idx, idx_group=idx_group, norm=self.norm1)
        return self._ffn_norm(x, self.norm2, unfold)


class DecoderLayer(nn.Module):
    def __init__(self, self_attention: AttentionLayer, cross_attention: AttentionLayer, d_model: int,
                 d_ff: Optional[int], dropout: float, activation: str):
        super().__init__()
        d_ff = d_ff or 4 * d_model
        self.self_attention = self_attention
        self.cross_attention = cross_attention
        self.conv1 = nn.Conv1d(d_model, d_ff, kernel_size=1)
        self.conv2 = nn.Conv1d(d_ff, d_model, kernel_size=1)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.p = dropout
        self.act = "relu" if activation == "relu" else "gelu"

    _ffn_norm = EncoderLayer._ffn_norm

    def self_block(self, x):
        """x + self-attention -> norm1: the part of the layer that does not read the encoder memory."""
        if self.p > 0.0 and self.training:
            return K.add_layer_norm(x, _dropout(self.self_attention(x), self.p, self.training), self.norm1.weight,
                                    self.norm1.bias)
        return self.self_attention(x, norm=self.norm1)

    def forward(self, x, memory, after_self=None):
        """``after_self``: the output of ``self_block`` if the caller ran it already (on another stream)."""
        x = after_self if after_self is not None else self.self_block(x)
        if self.p > 0.0 and self.training:
            x = K.add_layer_norm(x, _dropout(self.cross_attention(x, memory), self.p, self.training),
                                 self.norm2.weight, self.norm2.bias)
        else:
            x = self.cross_attention(x, memory, norm=self.norm2)
        return self._ffn_norm(x, self.norm3)


class DistilConv(nn.Module):
    """Conv1d(k3, circular pad 2) -> BatchNorm1d -> ELU -> MaxPool1d(3,2,1): halves the sequence."""

    def __init__(self, c: int):
        super().__init__()
        self.downConv = nn.Conv1d(c, c, kernel_size=3, padding=2, padding_mode="circular")
        self.norm = nn.BatchNorm1d(c)

    def forward(self, x, unfolded: bool = False):
        """``unfolded``: ``x`` is already the (B, L + 2, 3 C) im2col image (written by the norm in front, kernels.
        ffn_add_layer_norm(unfold=True))."""
        n = self.norm
        if unfolded:
            # train mode, the one-launch BatchNorm tail applies (all rows of a 32-channel slab in LDS): it sums the split-K slabs
            # of this product itself -- z is handed over unwritten (kernels.LAZY) and goes nowhere else
            Bz, Lz = x.shape[0], x.shape[1]
            lazy = bool(K.LAZY_BN_FWD and self.training and x.is_cuda and Bz * Lz * 32 * 4 <= 96 * 1024
                        and n.num_batches_tracked is not None)
            z = K.circular_conv3_unfolded(x, self.downConv.weight, self.downConv.bias, lazy=lazy)
        else:
            z = K.circular_conv3(x, self.downConv.weight, self.downConv.bias, pad=2)
        return K.bn_elu_pool(z, n.weight, n.bias, n.running_mean, n.running_var, n.num_batches_tracked,
                             training=self.training, momentum=n.momentum, eps=n.eps)


class FusedStack:
    """Host side of the fused per-sequence encoder stack (csrc/seqlayer.hip): the layers' weights in bf16 MFMA
    fragment order + the fp32 vectors, one blob per layer, re-packed whenever the parameters have changed
    (``refresh``: parameter versions and ``kernels.WEIGHTS_EPOCH``; the training engine re-packs at the head of
    every step, inside its captured graph)."""

    def __init__(self, layers):
        self.layers = list(layers)
        self.wpack, self.stride, self._key = None, 0, None
        self.wpack_bwd, self.stride_bwd = None, 0  # transposed fragments for the fused backward (training only)

    def _params(self):
        for lay in self.layers:
            a = lay.attention
            yield from (a.query_projection.weight, a.key_projection.weight, a.value_projection.weight,
                        a.out_projection.weight, lay.conv1.weight, lay.conv2.weight, lay.norm1.weight, lay.norm2.weight)

    def refresh(self, force: bool = False, collect=None):
        """``collect``: a list that takes the pack entries instead of launching them (kernels.PackPlan: the engine packs
        every stack of the model with one launch per step)."""
        dev = self.layers[0].conv1.weight.device
        key = (K.WEIGHTS_EPOCH, str(dev), tuple(p._version for p in self._params()), tuple(p.data_ptr() for p in self._params()))
        if not force and key == self._key and self.wpack is not None:
            return
        F_ = self.layers[0].conv1.weight.shape[0]
        stride = K.seqstack_pack_bytes(F_)
        if self.wpack is None or self.wpack.device != dev or self.stride != stride:
            self.wpack = torch.empty(len(self.layers) * stride, dtype=torch.uint8, device=dev)
            self.stride = stride
        descs = []
        with torch.no_grad():
            for lay in self.layers:
                a = lay.attention
                d = dict(wo=a.out_projection.weight, bo=a.out_projection.bias, w1=lay.conv1.weight.view(F_, -1),
                         b1=lay.conv1.bias, w2=lay.conv2.weight.view(-1, F_), b2=lay.conv2.bias, g1=lay.norm1.weight,
                         be1=lay.norm1.bias, g2=lay.norm2.weight, be2=lay.norm2.bias)
                pk = a.__dict__.get("_packed")
                if pk is not None:
                    d.update(wqkv=pk["w"], bqkv=pk["b"])
                else:
                    d.update(wq=a.query_projection.weight, wk=a.key_projection.weight, wv=a.value_projection.weight,
                             bq=a.query_projection.bias, bk=a.key_projection.bias, bv=a.value_projection.bias)
                descs.append({k: v.detach() for k, v in d.items()})
            K.seqstack_pack(descs, self.wpack, stride, collect)
            if all("wqkv" in d for d in descs) and K.SEQSTACK_BWD:  # packed projections exist: a training engine owns the model
                sb = K.seqstack_bwd_pack_bytes(F_)
                if self.wpack_bwd is None or self.wpack_bwd.device != dev or self.stride_bwd != sb:
                    self.wpack_bwd = torch.empty(len(self.layers) * sb, dtype=torch.uint8, device=dev)
                    self.stride_bwd = sb
                K.seqstack_bwd_pack(descs, self.wpack_bwd, sb, collect)
            else:
                self.wpack_bwd = None
        self._key = key


class Encoder(nn.Module):
    def __init__(self, attn_layers, conv_layers=None, norm_layer=None):
        super().__init__()
        self.attn_layers = nn.ModuleList(attn_layers)
        self.conv_layers = nn.ModuleList(conv_layers) if conv_layers is not None else None
        self.norm = norm_layer

    # -- fused per-sequence stack (one launch for all layers) -----------------------------------------
    def fused_stack(self) -> Optional[FusedStack]:
        """The FusedStack of this encoder if its architecture is the one the fused kernel implements (ProbSparse
        cross-modal layers, d_model 128, 8 heads, no distilling), else None.  Shape / mode checks happen per call."""
        st = self.__dict__.get("_fused")
        if st is None:
            ok = self.conv_layers is None and len(self.attn_layers) <= 8 and all(
                isinstance(lay, EncoderLayer) and lay.attention.kind == "prob" and not lay.attention.gps_variant
                and not lay.attention.mix and lay.attention.n_heads == 8 and lay.conv1.weight.shape[1] == 128
                and lay.attention.query_projection.weight.shape == (128, 128) and lay.conv1.bias is not None
                and lay.act == self.attn_layers[0].act and lay.conv1.weight.shape == self.attn_layers[0].conv1.weight.shape
                and lay.attention.factor == self.attn_layers[0].attention.factor
                for lay in self.attn_layers)
            st = self.__dict__["_fused"] = FusedStack(self.attn_layers) if ok else False
        return st or None

    def _fused_forward(self, x, idx_list, idx_group):
        st = self.fused_stack()
        if st is None or not x.is_cuda or x.dtype != torch.float32:
            return None
        lay0 = self.attn_layers[0]
        B, L, D = x.shape
        drop_p = lay0.p if self.training else 0.0
        if drop_p > 0.0 and (K.RNG.forced is not None or any(lay.p != lay0.p for lay in self.attn_layers)):
            return None  # injected masks (parity tests) go through the layer-by-layer path
        sample_k, n_top = K.prob_sizes(L, L, lay0.attention.factor)
        tiled = False
        if not K.seqstack_supported(L, D, 8, lay0.conv1.weight.shape[0], sample_k, n_top):
            # longer sequences (the fusion encoder's L = 160 / 320): attention launch + row-tile launch per layer
            tiled = K.tiled_stack_supported(L, D, 8, lay0.conv1.weight.shape[0])
            if not tiled:
                return None
        need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in st._params()))
        if need_grad:  # backward = layer-by-layer kernels writing parameter gradients through the engine's sinks
            if not (K.SINK.active and x.requires_grad and all("_packed" in lay.attention.__dict__ for lay in self.attn_layers)
                    and all(K._slot(p) is not None for p in st._params())):
                return None
        if idx_list is None:  # same host draws, same order as the layer-by-layer path
            idx_list = [SAMPLER.draw(L, L, sample_k, x.device).unsqueeze(0) for _ in self.attn_layers]
            idx_group = B
        strides = {(t.stride(0) if t.shape[0] > 1 else L * sample_k) for t in idx_list}
        if len(strides) != 1:
            idx_list = [t.contiguous() for t in idx_list]
        if not K.SINK.active:
            st.refresh()  # (the training engine re-packs at the head of every step instead)
        elif st.wpack is None:
            st.refresh(force=True)
        if tiled:
            return K._TiledStack.apply(x, st, idx_list, idx_group or B, need_grad, drop_p)
        return K._SeqStack.apply(x, st, idx_list, idx_group or B, need_grad, drop_p)

    def set_output_attention(self, on: bool = True):
        """``output_attention=True`` of the reference encoders: after a forward, ``self.attentions`` holds the dense
        (B, H, L, L) map of every layer (kernels.attention_map); the layers then run one by one (no fused stack)."""
        self.__dict__["output_attention"] = bool(on)
        for lay in self.attn_layers:
            lay.attention.__dict__["output_attention"] = bool(on)
        return self

    def forward(self, x, idx_list=None, idx_group: int = 0, tail: Optional[int] = None):
        """``tail``: the caller consumes only the last ``tail`` positions -- they are cut out BEFORE the final LayerNorm (a
        row-wise op: same values; the camera-token encoder keeps 1 position of 65, so the norm and its backward run on 192 rows
        instead of 12 480)."""
        x = self._forward(x, idx_list, idx_group, tail)
        if self.__dict__.get("output_attention", False):
            self.__dict__["attentions"] = [lay.attention.__dict__.pop("attention_map") for lay in self.attn_layers]
        return x

    def _forward(self, x, idx_list=None, idx_group: int = 0, tail: Optional[int] = None):
        if self.conv_layers is not None:
            for attn, conv in zip(self.attn_layers, self.conv_layers):
                if isinstance(attn, EncoderLayer) and isinstance(conv, DistilConv) and not attn.attention.__dict__.get("output_attention"):
                    x = conv(*attn(x, unfold=True))  # the layer's last norm writes the convolution's im2col image itself
                else:
                    x = conv(attn(x))
            x = self.attn_layers[-1](x)
        else:
            y = None if self.__dict__.get("output_attention", False) else self._fused_forward(x, idx_list, idx_group)
            if y is not None:
                x = y
            else:
                for i, attn in enumerate(self.attn_layers):
                    x = attn(x, None if idx_list is None else idx_list[i], idx_group)
        if tail is not None:
            x = _tail(x, tail)
        if self.norm is not None:
            x = K.add_layer_norm(x, None, self.norm.weight, self.norm.bias)
        return x


class Decoder(nn.Module):
    def __init__(self, layers, norm_layer=None, projection=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self.norm = norm_layer
        self.projection = projection

    def _chain_forward(self, x, memory):
        """The layers as attention launches + row-local chains (csrc/rowchain.hip: everything between two attention
        launches in one launch; 5 launches per layer instead of 13) -- d_model 64 decoders in bf16 mode without dropout,
        i.e. the gaze-video PerceiveDecoder.  None: not applicable, take the layer-by-layer path."""
        if not (x.is_cuda and x.dtype == torch.float32 and len(self.layers) > 0):
            return None
        l0 = self.layers[0]
        D = x.shape[-1]
        F_ = l0.conv1.weight.shape[0]
        for lay in self.layers:
            sa, ca = lay.self_attention, lay.cross_attention
            if not (isinstance(lay, DecoderLayer) and sa.kind == "prob_masked" and ca.kind == "full" and not sa.gps_variant
                    and not ca.gps_variant and not ca.mix and sa.n_heads == ca.n_heads
                    and sa.query_projection.weight.shape == (D, D) and ca.query_projection.weight.shape == (D, D)
                    and lay.conv1.weight.shape[0] == F_ and lay.act == l0.act and lay.conv1.bias is not None
                    and not sa.__dict__.get("output_attention") and not ca.__dict__.get("output_attention")):
                return None
            if self.training and (lay.p > 0.0 or ca.attn_dropout > 0.0) and (K.RNG.forced is not None or lay.p != l0.p):
                return None  # injected masks (parity tests) go through the layer-by-layer path
        if not (K.rowchain_supported(D, F_, D) and K.rowchain_supported(D, F_, 3 * D)):
            return None
        need_grad = torch.is_grad_enabled() and (x.requires_grad or memory.requires_grad
                                                 or any(p.requires_grad for p in self.layers.parameters()))
        if need_grad:  # parameter gradients leave through the engine's sinks only
            if not (K.SINK.active and not K.DETERMINISTIC and x.requires_grad
                    and all("_packed" in a.__dict__ for lay in self.layers for a in (lay.self_attention, lay.cross_attention))
                    and all(K._slot(p) is not None for p in self.layers.parameters())):
                return None
        B, L, _ = x.shape
        S = memory.shape[1]
        H = l0.self_attention.n_heads
        E = D // H
        mem2 = memory.reshape(B * S, -1)
        qkv = None
        for i, lay in enumerate(self.layers):
            sa, ca = lay.self_attention, lay.cross_attention
            pks, pkc = sa.__dict__.get("_packed"), ca.__dict__.get("_packed")
            if qkv is None:  # first layer: the packed q | k | v projection as a GEMM (later ones come out of the chain)
                if pks is not None:
                    qkv = K.linear_packed(x.reshape(B * L, D), pks["w"], pks["b"], pks["gw"], pks["gb"])
                else:
                    qkv = K.linear(x.reshape(B * L, D),
                                   torch.cat([sa.query_projection.weight, sa.key_projection.weight, sa.value_projection.weight]),
                                   torch.cat([sa.query_projection.bias, sa.key_projection.bias, sa.value_projection.bias]))
            qkv = qkv.reshape(B * L, 3 * D)
            sample_k, n_top = K.prob_sizes(L, L, sa.factor)
            idx = SAMPLER.draw(L, L, sample_k, x.device)
            # mix = the (B, H, L, E) context VIEWED as (B, L, H E) (cross_modal_transformer.py:203-205): the kernel's
            # un-transposed output layout, no transpose copy
            ctx1 = K.attention(qkv, qkv, (0, D, 2 * D), (B, H, L, L, E), 2, index_sample=idx, n_top=n_top,
                               out_layout=1 if sa.mix else 0).view(B, L, D)
            drop_p = lay.p if self.training else 0.0  # (sites in the layer-by-layer order: self out, cross probabilities,
            #                                                  cross out, hidden activation, conv2 out)
            x1, q2 = K.rowchain(ctx1, x, (sa.out_projection.weight, sa.out_projection.bias), (lay.norm1.weight, lay.norm1.bias),
                                None, (ca.query_projection.weight, ca.query_projection.bias,
                                       K._slot(ca.query_projection.weight), K._slot(ca.query_projection.bias)),
                                lay.act, lay.norm1.eps, drop_p)
            if pkc is not None:
                kv = K.linear_packed(mem2, pkc["w"][D:], pkc["b"][D:], pkc["gw"][D:], pkc["gb"][D:])
            else:
                kv = K.linear(mem2, torch.cat([ca.key_projection.weight, ca.value_projection.weight]),
                              torch.cat([ca.key_projection.bias, ca.value_projection.bias]))
            ctx2 = K.attention(q2.reshape(B * L, D), kv, (0, 0, D), (B, H, L, S, E), 0,
                               drop_p=ca.attn_dropout if self.training else 0.0).view(B, L, D)
            nxt = self.layers[i + 1].self_attention if i + 1 < len(self.layers) else None
            proj = None
            if nxt is not None:
                pkn = nxt.__dict__.get("_packed")
                if pkn is not None:
                    proj = (pkn["w"], pkn["b"], pkn["gw"], pkn["gb"])
                else:
                    proj = (torch.cat([nxt.query_projection.weight, nxt.key_projection.weight, nxt.value_projection.weight]),
                            torch.cat([nxt.query_projection.bias, nxt.key_projection.bias, nxt.value_projection.bias]), None, None)
            ffn = (lay.conv1.weight, lay.conv1.bias, lay.conv2.weight, lay.conv2.bias, lay.norm3.weight, lay.norm3.bias)
            x, qkv = K.rowchain(ctx2, x1, (ca.out_projection.weight, ca.out_projection.bias), (lay.norm2.weight, lay.norm2.bias),
                                ffn, proj, lay.act, lay.norm2.eps, drop_p)
        return x

    def forward(self, x, memory, first=None):
        """``first``: the first layer's ``self_block`` output, computed by the caller (then ``x`` is not used)."""
        y = self._chain_forward(x, memory) if first is None else None
        if y is not None:
            x = y
        else:
            for i, layer in enumerate(self.layers):
                x = layer(x, memory, after_self=first if i == 0 else None)
        if self.norm is not None:
            x = K.add_layer_norm(x, None, self.norm.weight, self.norm.bias)
        if self.projection is not None:
            x = K.linear(x, self.projection.weight, self.projection.bias)
        return x


class PerceiveEncoder(nn.Module):
    """ProbSparse encoder used for the frame, gaze and fusion encoders (internal width 128)."""

    def __init__(self, in_channels, out_channels, out_len, factor=5, d_model=128, n_heads=8, layers=3,
                 d_ff=None, dropout=0.1, activation="gelu", output_attention=False):
        super().__init__()
        self.output_attention = bool(output_attention)
        self.pred_len = out_len
        d_ff = d_ff if d_ff is not None else 4 * d_model
        self.value_embedding = TokenEmbedding(in_channels, d_model, bias=True)
        self.position_embedding = PositionalEmbedding(d_model)
        self.encoder = Encoder(
            [EncoderLayer(AttentionLayer("prob", d_model, n_heads, factor), d_model, d_ff, dropout, activation)
             for _ in range(layers)],
            None, norm_layer=nn.LayerNorm(d_model))
        self.projection = nn.Linear(d_model, out_channels, bias=True)
        if self.output_attention:  # (cross_modal_transformer.py:430-433: returns (output, attentions))
            self.encoder.set_output_attention()

    def predraw(self, L: int, device):
        """The draws one forward over length-L sequences makes, in layer order ((L,k) int32 on device)."""
        out = []
        for layer in self.encoder.attn_layers:
            sample_k, _ = K.prob_sizes(L, L, layer.attention.factor)
            out.append(SAMPLER.draw(L, L, sample_k, device))
        return out

    def forward(self, x_enc, idx_list=None, idx_group: int = 0):
        """``idx_list``: per layer a (G,L,k) int32 device tensor of pre-drawn key samples (several
        reference calls batched into one: rows [g*idx_group, (g+1)*idx_group) use table g)."""
        h = self.value_embedding(x_enc, residual=self.position_embedding(x_enc.shape[1])[0])
        if K.STEP_HOOKS and not K.on_side_stream() and torch.is_grad_enabled():
            hook = K.STEP_HOOKS.pop("after_frame_embedding", None)  # (engine: let the streaming optimizer update loose now)
            if hook is not None:
                hook()
        h = self.encoder(h, idx_list, idx_group, tail=self.pred_len)  # only the last pred_len tokens are consumed
        y = K.linear(h, self.projection.weight, self.projection.bias)
        return (y, self.encoder.attentions) if self.output_attention else y


class PerceiveDecoder(nn.Module):
    """Masked-ProbSparse self attention + full cross attention (gaze tokens query FoV features)."""

    def __init__(self, query_channels, value_channels, out_channels, out_len, factor=5, n_heads=8, layers=2,
                 d_ff=None, dropout=0.1, activation="gelu", mix=True):
        super().__init__()
        self.pred_len = out_len
        d_model = value_channels
        d_ff = d_ff if d_ff is not None else 4 * d_model
        self.value_embedding = TokenEmbedding(query_channels, d_model, bias=True)
        self.position_embedding = PositionalEmbedding(d_model)
        self.decoder = Decoder(
            [DecoderLayer(AttentionLayer("prob_masked", d_model, n_heads, factor, mix=mix),
                          AttentionLayer("full", d_model, n_heads, factor, attn_dropout=dropout),
                          d_model, d_ff, dropout, activation)
             for _ in range(layers)],
            norm_layer=nn.LayerNorm(d_model))
        self.projection = nn.Linear(d_model, out_channels, bias=True)

    def forward(self, x_enc, x_dec):
        h = self.value_embedding(x_dec, residual=self.position_embedding(x_dec.shape[1])[0])
        h = self.decoder(h, x_enc)
        return K.linear(_tail(h, self.pred_len), self.projection.weight, self.projection.bias)
