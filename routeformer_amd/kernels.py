"""Autograd glue: ``torch.autograd.Function``s whose forward AND backward are librf_hip.so calls.

Host-side responsibilities only: allocate outputs/workspaces with torch's caching allocator, pass
raw pointers + the current HIP stream over the C ABI, wire gradients.  No arithmetic happens here
and nothing falls back to eager PyTorch: a missing library or a CPU tensor raises.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _hip
from ._hip import check, ptr

ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2, "elu": 3}

_PRECISION = 0  # 0 = exact fp32 MFMA, 1 = bf16 MFMA inputs / fp32 accumulate


def set_precision(name: str):
    """``"f32"`` (parity mode) or ``"bf16"`` (matrix-core inputs rounded to bf16)."""
    global _PRECISION
    _PRECISION = {"f32": 0, "fp32": 0, "bf16": 1}[name]


def get_precision() -> str:
    return "bf16" if _PRECISION else "f32"


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Profiler:
    """HIP-event timing of kernel classes on the launch stream (bench.py's live roofline numbers).
    Off by default: a single attribute test per launch."""

    def __init__(self):
        self.on = False
        self.events = []

    def enable(self):
        self.on, self.events = True, []

    def disable(self):
        self.on = False

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, tag, start, flops, nbytes):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.events.append((tag, start, e, flops, nbytes))

    def summary(self):
        """tag -> {launches, total_ms, flops, bytes} (algorithmic flops / bytes summed over launches)."""
        if not self.events:
            return {}
        torch.cuda.synchronize()
        out = {}
        for tag, s, e, fl, by in self.events:
            d = out.setdefault(tag, {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["total_ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += by
        return out


PROFILE = _Profiler()


def _req(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _hip.HipLibraryError(
            f"{what}: routeformer_amd runs on the GPU only (got a {t.device} tensor); there is no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: expected float32, got {t.dtype}")


def _splits(tiles: int, depth: int) -> int:
    """Split-K factor for skinny-output GEMMs (weight gradients): aim for >= 512 workgroups while
    keeping >= 128 of reduction depth per slice."""
    return max(1, min(64, depth // 128, -(-512 // max(tiles, 1))))


def gemm(A, lda_m, lda_k, B, ldb_k, ldb_n, C, ldc, M, N, K, *, bias=None, residual=None, ldr=0,
         res_rows=0, res_before_act=0, act=0, preact=None, ldp=0, dact_src=None, ldd=0, dact=0,
         splitk=1):
    ws = None
    if splitk > 1:
        ws = torch.empty(splitk * M * N, device=C.device, dtype=torch.float32)
    ev = PROFILE.begin() if PROFILE.on else None
    check(_hip.lib().rf_gemm(ptr(A), lda_m, lda_k, ptr(B), ldb_k, ldb_n, ptr(C), ldc, M, N, K,
                             ptr(bias), ptr(residual), ldr, res_rows, res_before_act, act,
                             ptr(preact), ldp, ptr(dact_src), ldd, dact, _PRECISION, splitk, ptr(ws),
                             _stream()), "rf_gemm")
    if ev is not None:
        PROFILE.end("gemm", ev, 2.0 * M * N * K, 4.0 * (M * K + K * N + M * N))


def colsum(X2d: torch.Tensor) -> torch.Tensor:
    M, N = X2d.shape
    out = torch.empty(N, device=X2d.device, dtype=torch.float32)
    parts = _hip.lib().rf_colsum_parts(M, N)
    ws = torch.empty(parts * N, device=X2d.device, dtype=torch.float32)
    check(_hip.lib().rf_colsum(ptr(X2d), X2d.stride(0), M, N, ptr(out), ptr(ws), _stream()), "rf_colsum")
    return out


def _weight_grad(dy2: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """dW[N,K] = dY[M,N]^T X[M,K]  (split-K over the row dimension M)."""
    M, N = dy2.shape
    K = x2.shape[1]
    dw = torch.empty(N, K, device=dy2.device, dtype=torch.float32)
    tiles = -(-N // 64) * -(-K // 64)
    gemm(dy2, 1, dy2.stride(0), x2, x2.stride(0), 1, dw, K, N, K, M, splitk=_splits(tiles, M))
    return dw


def _input_grad(dy2: torch.Tensor, w: torch.Tensor, **epi) -> torch.Tensor:
    """dX[M,K] = dY[M,N] W[N,K]."""
    M, N = dy2.shape
    K = w.shape[1]
    dx = torch.empty(M, K, device=dy2.device, dtype=torch.float32)
    gemm(dy2, dy2.stride(0), 1, w, w.stride(0), 1, dx, K, M, K, N, **epi)
    return dx


class _Linear(torch.autograd.Function):
    """y = x W^T + b (+ residual broadcast over row blocks)."""

    @staticmethod
    def forward(ctx, x, w, b):
        _req(x, "linear.x"); _req(w, "linear.w")
        K = x.shape[-1]
        N = w.shape[0]
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        w = w.contiguous()
        M = x2.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        gemm(x2, x2.stride(0), 1, w, 1, K, y, N, M, N, K, bias=b)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = b is not None
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.stride(1) != 1 or dy2.stride(0) != dy2.shape[1]:
            dy2 = dy2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dy2, w).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dw = _weight_grad(dy2, x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy2)
        return dx, dw, db


def linear(x, w, b=None):
    return _Linear.apply(x, w, b)


class _FFN(torch.autograd.Function):
    """y = act(x W1^T + b1) W2^T + b2  -- the Conv1d(k=1) pair of every encoder/decoder layer.
    The activation and its derivative ride in GEMM epilogues."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act: str):
        _req(x, "ffn.x")
        D, F = w1.shape[1], w1.shape[0]
        x2 = x.reshape(-1, D)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        w1, w2 = w1.contiguous(), w2.contiguous()
        M = x2.shape[0]
        h = torch.empty(M, F, device=x.device, dtype=torch.float32)
        z = torch.empty_like(h) if act == "gelu" else None
        gemm(x2, x2.stride(0), 1, w1, 1, D, h, F, M, F, D, bias=b1, act=ACT[act], preact=z, ldp=F)
        y = torch.empty(M, D, device=x.device, dtype=torch.float32)
        gemm(h, F, 1, w2, 1, F, y, D, M, D, F, bias=b2)
        ctx.save_for_backward(x2, w1, w2, h, z if z is not None else h)
        ctx.act = act
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, h, zsrc = ctx.saved_tensors
        D = w1.shape[1]
        dy2 = dy.reshape(-1, D)
        if dy2.stride(1) != 1 or dy2.stride(0) != D:
            dy2 = dy2.contiguous()
        # dZ = (dY W2) * act'(Z)   (relu: mask from H > 0; gelu: from the saved pre-activation)
        dz = _input_grad(dy2, w2, dact_src=zsrc, ldd=zsrc.stride(0), dact=ACT[ctx.act])
        dw2 = _weight_grad(dy2, h)
        db2 = colsum(dy2)
        dw1 = _weight_grad(dz, x2)
        db1 = colsum(dz)
        dx = _input_grad(dz, w1).view(ctx.xshape) if ctx.needs_input_grad[0] else None
        return dx, dw1, db1, dw2, db2, None


def ffn(x, w1, b1, w2, b2, act: str):
    return _FFN.apply(x, w1, b1, w2, b2, act)


class _AddLayerNorm(torch.autograd.Function):
    """y = LayerNorm(x + residual) (eps 1e-5); residual optional."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps):
        _req(x, "layernorm.x")
        cols = x.shape[-1]
        x2 = x.reshape(-1, cols).contiguous()
        r2 = residual.reshape(-1, cols).contiguous() if residual is not None else None
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        xhat = torch.empty_like(x2)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        ev = PROFILE.begin() if PROFILE.on else None
        check(_hip.lib().rf_layernorm_fwd(ptr(x2), ptr(r2), ptr(gamma), ptr(beta), ptr(y), ptr(xhat),
                                          ptr(rstd), rows, cols, eps, _stream()), "rf_layernorm_fwd")
        if ev is not None:
            PROFILE.end("layernorm_fwd", ev, 8.0 * rows * cols, 4.0 * rows * cols * (4 if r2 is not None else 3))
        ctx.save_for_backward(xhat, rstd, gamma)
        ctx.has_res = residual is not None
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        xhat, rstd, gamma = ctx.saved_tensors
        rows, cols = xhat.shape
        dy2 = dy.reshape(rows, cols).contiguous()
        dx = torch.empty_like(xhat)
        dg = torch.empty(cols, device=dy.device, dtype=torch.float32)
        db = torch.empty(cols, device=dy.device, dtype=torch.float32)
        parts = _hip.lib().rf_layernorm_bwd_parts(rows)
        ws = torch.empty(parts * 2 * cols, device=dy.device, dtype=torch.float32)
        ev = PROFILE.begin() if PROFILE.on else None
        check(_hip.lib().rf_layernorm_bwd(ptr(dy2), ptr(xhat), ptr(rstd), ptr(gamma), ptr(dx), ptr(dg),
                                          ptr(db), ptr(ws), rows, cols, _stream()), "rf_layernorm_bwd")
        if ev is not None:
            PROFILE.end("layernorm_bwd", ev, 12.0 * rows * cols, 4.0 * rows * cols * 3)
        dx = dx.view(ctx.xshape)
        return dx, (dx if ctx.has_res else None), dg, db, None


def add_layer_norm(x, residual, gamma, beta, eps: float = 1e-5):
    return _AddLayerNorm.apply(x, residual, gamma, beta, eps)


class _Unfold3(torch.autograd.Function):
    """(B,L,C) -> (B,L+2p-2,3C) circular im2col for the k=3 sequence convolutions."""

    @staticmethod
    def forward(ctx, x, pad: int):
        _req(x, "unfold3.x")
        x = x.contiguous()
        B, L, C = x.shape
        cols = torch.empty(B, L + 2 * pad - 2, 3 * C, device=x.device, dtype=torch.float32)
        check(_hip.lib().rf_unfold3_circular(ptr(x), ptr(cols), B, L, C, pad, _stream()), "rf_unfold3")
        ctx.dims = (B, L, C, pad)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        B, L, C, pad = ctx.dims
        dcols = dcols.contiguous()
        dx = torch.empty(B, L, C, device=dcols.device, dtype=torch.float32)
        check(_hip.lib().rf_fold3_circular(ptr(dcols), ptr(dx), B, L, C, pad, _stream()), "rf_fold3")
        return dx, None


def circular_conv3(x, weight, bias=None, pad: int = 1):
    """Conv1d(k=3, padding_mode='circular') on channels-last sequences: weight (d, c, 3)."""
    cols = _Unfold3.apply(x, pad)
    w2 = weight.permute(0, 2, 1).reshape(weight.shape[0], -1)  # [d][t*C + c]
    return linear(cols, w2, bias)


class _BnEluPool(torch.autograd.Function):
    """BatchNorm1d -> ELU -> MaxPool1d(3,2,1) over (B,L,C) (Informer distilling tail)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, training):
        x = x.contiguous()
        B, L, C = x.shape
        Lout = (L - 1) // 2 + 1
        y = torch.empty(B, Lout, C, device=x.device, dtype=torch.float32)
        arg = torch.empty(B, Lout, C, device=x.device, dtype=torch.int32)
        check(_hip.lib().rf_bn_elu_pool_fwd(ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(y),
                                            ptr(arg), B, L, C, eps, _stream()), "rf_bn_elu_pool_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, var, arg)
        ctx.eps, ctx.training = eps, training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, var, arg = ctx.saved_tensors
        B, L, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(C, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32)
        check(_hip.lib().rf_bn_elu_pool_bwd(ptr(dy), ptr(arg), ptr(x), ptr(mean), ptr(var), ptr(gamma),
                                            ptr(beta), ptr(dx), ptr(dg), ptr(db), None, B, L, C, ctx.eps,
                                            1 if ctx.training else 0, _stream()), "rf_bn_elu_pool_bwd")
        return dx, dg, db, None, None, None, None


def bn_stats(x3: torch.Tensor):
    B, L, C = x3.shape
    mean = torch.empty(C, device=x3.device, dtype=torch.float32)
    var = torch.empty(C, device=x3.device, dtype=torch.float32)
    check(_hip.lib().rf_bn_stats(ptr(x3), ptr(mean), ptr(var), B * L, C, _stream()), "rf_bn_stats")
    return mean, var


def bn_elu_pool(x, gamma, beta, running_mean, running_var, num_batches_tracked, training: bool,
                momentum: float = 0.1, eps: float = 1e-5):
    x = x.contiguous()
    if training:
        mean, var = bn_stats(x.detach())
        n = x.shape[0] * x.shape[1]
        with torch.no_grad():  # running-stat update (unbiased variance), nn.BatchNorm1d semantics
            running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1 - momentum).add_(var, alpha=momentum * n / max(n - 1, 1))
            if num_batches_tracked is not None:
                num_batches_tracked.add_(1)
    else:
        mean, var = running_mean, running_var
    return _BnEluPool.apply(x, gamma, beta, mean, var, eps, training)


class TopSelection:
    """Debug / test hook around the ProbSparse top-u selection (discontinuous in its inputs).
    ``record``: list collecting the (B,H,u) ascending selections each call made.
    ``forced``: list of selections to impose, consumed in call order (teacher forcing)."""
    record: Optional[list] = None
    forced: Optional[list] = None

    def merge_forced(self, n_calls: int, n_layers: int):
        """``n_calls`` reference encoder calls (each ``n_layers`` selections) run as one batched call:
        turn their queued selections [call][layer] into per-layer batch-concatenated ones."""
        if self.forced is None or n_calls == 1:
            return
        head, rest = self.forced[: n_calls * n_layers], self.forced[n_calls * n_layers:]
        self.forced = [torch.cat([head[c * n_layers + l] for c in range(n_calls)], dim=0)
                       for l in range(n_layers)] + rest

    def split_record(self, n_calls: int, n_layers: int):
        """Inverse bookkeeping for ``record``: re-emit a batched call's selections in reference order."""
        if self.record is None or n_calls == 1:
            return
        tail = self.record[-n_layers:]
        del self.record[-n_layers:]
        for c in range(n_calls):
            for l in range(n_layers):
                self.record.append(tail[l].chunk(n_calls, dim=0)[c])


TOPS = TopSelection()


def prob_sizes(L_Q: int, L_K: int, factor: int):
    """(sample_k, n_top) = (min(c*ceil(ln L_K), L_K), min(c*ceil(ln L_Q), L_Q))."""
    U = factor * int(math.ceil(math.log(L_K)))
    u = factor * int(math.ceil(math.log(L_Q)))
    return (U if U < L_K else L_K), (u if u < L_Q else L_Q)


class _Attention(torch.autograd.Function):
    """Attention core on projected row-major matrices.  ``a`` holds Q at column ``q_off``; ``b`` holds
    K and V at ``k_off`` / ``v_off`` (``a is b`` for a packed self-attention QKV projection).  Gradients
    come back in the same packed layouts, so the projection backward is one GEMM per packed matrix.
    mode 0 full, 1 ProbSparse, 2 ProbSparse masked."""

    @staticmethod
    def forward(ctx, a, b, offs, index_sample, dims, mode, n_top, out_layout, scale, forced_top, idx_group=0):
        B, H, LQ, LK, E = dims
        q_off, k_off, v_off = offs
        _req(a, "attention.q")
        _req(b, "attention.kv")
        assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
        assert a.shape[0] == B * LQ and b.shape[0] == B * LK
        shape = (B, LQ, H, E) if out_layout == 0 else (B, H, LQ, E)
        out = torch.empty(shape, device=a.device, dtype=torch.float32)
        top, sample_k = None, 0
        if mode != 0:
            if forced_top is None and TOPS.forced is not None:
                forced_top = TOPS.forced.pop(0).to(device=a.device, dtype=torch.int32).contiguous()
                assert tuple(forced_top.shape) == (B, H, n_top), (tuple(forced_top.shape), (B, H, n_top))
            top = forced_top if forced_top is not None else \
                torch.empty(B, H, n_top, device=a.device, dtype=torch.int32)
            sample_k = index_sample.shape[-1] if index_sample is not None else 0
        ev = PROFILE.begin() if PROFILE.on else None
        check(_hip.lib().rf_attn_fwd(a.data_ptr() + 4 * q_off, b.data_ptr() + 4 * k_off,
                                     b.data_ptr() + 4 * v_off, a.stride(0), b.stride(0), b.stride(0),
                                     ptr(out), out_layout, ptr(index_sample), idx_group, ptr(top),
                                     1 if forced_top is not None else 0, B, H, LQ, LK, E, sample_k, n_top,
                                     mode, scale, _stream()), "rf_attn_fwd")
        if ev is not None:
            u = LQ if mode == 0 else n_top  # sample stage + active rows (QK^T and AV)
            PROFILE.end("attn_fwd", ev, B * H * (2.0 * LQ * sample_k * E + 4.0 * u * LK * E),
                        4.0 * B * H * E * (2 * LQ + 2 * LK) + 4.0 * LQ * sample_k)
        if top is not None and TOPS.record is not None:
            TOPS.record.append(top.clone())
        ctx.save_for_backward(a, b, top if top is not None else a)
        ctx.cfg = (dims, offs, mode, n_top, out_layout, scale, a.data_ptr() == b.data_ptr())
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, top = ctx.saved_tensors
        (B, H, LQ, LK, E), (q_off, k_off, v_off), mode, n_top, out_layout, scale, same = ctx.cfg
        dout = dout.contiguous()
        da = torch.empty(a.shape, device=dout.device, dtype=torch.float32)
        db = da if same else torch.empty(b.shape, device=dout.device, dtype=torch.float32)
        ev = PROFILE.begin() if PROFILE.on else None
        check(_hip.lib().rf_attn_bwd(a.data_ptr() + 4 * q_off, b.data_ptr() + 4 * k_off,
                                     b.data_ptr() + 4 * v_off, a.stride(0), b.stride(0), b.stride(0),
                                     ptr(dout), out_layout, ptr(top) if mode != 0 else None,
                                     da.data_ptr() + 4 * q_off, db.data_ptr() + 4 * k_off,
                                     db.data_ptr() + 4 * v_off, da.stride(0), db.stride(0), db.stride(0),
                                     B, H, LQ, LK, E, n_top, mode, scale, _stream()), "rf_attn_bwd")
        if ev is not None:
            u = LQ if mode == 0 else n_top
            PROFILE.end("attn_bwd", ev, B * H * 10.0 * u * LK * E, 4.0 * B * H * E * (4 * LQ + 4 * LK))
        return da, (None if same else db), None, None, None, None, None, None, None, None, None


def attention(a, b, offs, dims, mode: int, *, index_sample=None, n_top: int = 0, out_layout: int = 0,
              scale: Optional[float] = None, forced_top=None, idx_group: int = 0):
    """Returns ctx in (B,LQ,H,E) [out_layout 0] or (B,H,LQ,E) [out_layout 1: Informer's un-transposed
    layout, layers/SelfAttentionFamily.py:165].  Every column of ``a`` / ``b`` must be one of Q/K/V."""
    B, H, LQ, LK, E = dims
    scale = scale or 1.0 / math.sqrt(E)
    if a is b:
        assert a.shape[1] == 3 * H * E
    else:
        assert a.shape[1] == H * E and b.shape[1] == 2 * H * E
    if index_sample is not None and index_sample.dim() == 3:
        assert idx_group > 0 and index_sample.shape[0] * idx_group == B, (index_sample.shape, idx_group, B)
    return _Attention.apply(a, b, offs, index_sample, dims, mode, n_top, out_layout, scale, forced_top, idx_group)
