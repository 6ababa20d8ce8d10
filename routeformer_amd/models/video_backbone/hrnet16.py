"""Frozen HRNet-16 conv encoder ("InverseForm" backbone of the reference) on the HIP kernels.

Plugin contract = ``routeformer/models/video_backbone/config.py:45-52`` / ``InverseForm.py:140-176``:
``HRNet16Backbone(configs)``, ``.output_feature_shape == (240, 8, 8)``,
``forward(images (N,3,H,W)) -> (N,240,8,8)``.  The state_dict keys are those of the reference trunk
(``video_backbone._Backbone.*``; architecture: ``inverse_form_layers/hrnetv2.py:282-500`` with
``config.py:177-206``), so the Qualcomm ``hr16s_4k_slim`` checkpoint the reference downloads maps
onto it key-for-key (``load_inverseform_checkpoint``).

MI355X design: inference-only, so BatchNorm2d(eval) is folded into the conv weights once; activations
are NHWC fp32 so each 3x3/1x1 convolution is an implicit GEMM on the matrix cores
(``rf_conv2d_nhwc``: M = pixels, N = C_out, K = k*k*C_in) with bias / residual / ReLU in the epilogue;
the trunk output is pooled straight into the (N,65,240) token layout the frame encoder consumes.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from routeformer_amd import _hip
from routeformer_amd import kernels as K
from routeformer_amd._hip import check, ptr

from .config import VideoBackboneConfig, VideoBackboneModule

BRANCH_CH = (16, 32, 64, 128)
STAGES = (("stage2", 2, 1), ("stage3", 3, 3), ("stage4", 4, 2))  # name, branches, modules
BLOCKS_PER_BRANCH = 2


class _Tree(nn.Module):
    """Nested empty modules so parameters get dotted names identical to the reference's."""

    def put(self, path: str, tensor: torch.Tensor, buffer: bool = False):
        node = self
        *parents, leaf = path.split(".")
        for name in parents:
            if name not in node._modules:
                node.add_module(name, _Tree())
            node = node._modules[name]
        if buffer:
            node.register_buffer(leaf, tensor)
        else:
            node.register_parameter(leaf, nn.Parameter(tensor, requires_grad=False))


def _conv_units() -> List[Tuple[str, Optional[str], int, int, int]]:
    """(conv key, bn key or None, c_in, c_out, ksize) for every convolution of the HRNet16 trunk."""
    u: List[Tuple[str, Optional[str], int, int, int]] = [
        ("conv0", None, 3, 3, 2), ("conv1", "bn1", 3, 64, 3), ("conv2", "bn2", 64, 64, 3)]
    for b, cin in ((0, 64), (1, 256)):  # layer1: two Bottlenecks (planes 64, expansion 4)
        p = f"layer1.{b}"
        u += [(p + ".conv1", p + ".bn1", cin, 64, 1), (p + ".conv2", p + ".bn2", 64, 64, 3),
              (p + ".conv3", p + ".bn3", 64, 256, 1)]
        if b == 0:
            u.append((p + ".downsample.0", p + ".downsample.1", 64, 256, 1))
    u += [("transition1.0.0", "transition1.0.1", 256, 16, 3),
          ("transition1.1.0.0", "transition1.1.0.1", 256, 32, 3),
          ("transition2.2.0.0", "transition2.2.0.1", 32, 64, 3),
          ("transition3.3.0.0", "transition3.3.0.1", 64, 128, 3)]
    for stage, nb, nmod in STAGES:
        for m in range(nmod):
            p = f"{stage}.{m}"
            for br in range(nb):
                c = BRANCH_CH[br]
                for k in range(BLOCKS_PER_BRANCH):
                    q = f"{p}.branches.{br}.{k}"
                    u += [(q + ".conv1", q + ".bn1", c, c, 3), (q + ".conv2", q + ".bn2", c, c, 3)]
            for i in range(nb):
                for j in range(nb):
                    q = f"{p}.fuse_layers.{i}.{j}"
                    if j > i:
                        u.append((q + ".0", q + ".1", BRANCH_CH[j], BRANCH_CH[i], 1))
                    elif j < i:
                        for k in range(i - j):
                            cout = BRANCH_CH[i] if k == i - j - 1 else BRANCH_CH[j]
                            u.append((f"{q}.{k}.0", f"{q}.{k}.1", BRANCH_CH[j], cout, 3))
    return u


UNITS = _conv_units()


import os as _os
# cross-resolution fuse / final concat + pool without intermediate maps (csrc/fuse.hip); RF_FUSE_SUM=0: one up-sample launch
# per (output branch, source branch) pair as before
FUSE_SUM = _os.environ.get("RF_FUSE_SUM", "1") != "0"
GROUP_BRANCHES = _os.environ.get("RF_CONV_GROUP", "1") != "0"  # one launch per conv step of an HRNet module's branches


def pack_conv3x3_weights(w_khwc: torch.Tensor) -> torch.Tensor:
    """(cout, 3, 3, cin) fp32 device tensor -> the bf16 fragment-order buffer rf_conv3x3_bf16 consumes."""
    cout, _, _, cin = w_khwc.shape
    out = torch.empty(int(_hip.lib().rf_conv3x3_packed_elems(cin, cout)), device=w_khwc.device, dtype=torch.bfloat16)
    w = w_khwc.contiguous().float()
    check(_hip.lib().rf_conv3x3_pack_bf16(ptr(w), ptr(out), cin, cout, K._stream()), "rf_conv3x3_pack_bf16")
    return out


def pack_conv3x3s2_weights(w_khwc: torch.Tensor) -> torch.Tensor:
    """(cout, 3, 3, cin) fp32 device tensor -> the bf16 fragment-order buffer rf_conv3x3s2_bf16 consumes (the two stem
    convolutions: cin 4 / 64, cout 64)."""
    cout, _, _, cin = w_khwc.shape
    out = torch.empty(int(_hip.lib().rf_conv3x3s2_packed_elems(cin, cout)), device=w_khwc.device, dtype=torch.bfloat16)
    w = w_khwc.contiguous().float()
    check(_hip.lib().rf_conv3x3s2_pack_bf16(ptr(w), ptr(out), cin, cout, K._stream()), "rf_conv3x3s2_pack_bf16")
    return out


STEM_S2 = _os.environ.get("RF_STEM_S2", "1") != "0"  # measurement switches: the stride-2 raster-window kernel for the stem pair ...
# BasicBlock pairs in one launch, the intermediate map in LDS (rf_conv3x3_pair_group_bf16; bit-identical).  Measured no faster
# (tools/probes/pair_time.py: 16 ch @ 28 x 28 23 vs 26 us, 32 ch @ 14 x 14 24 vs 18 us, the grouped low-resolution branches
# equal; step 5.24 -> 5.32 / 5.41 ms at C2, unchanged at C5): conv1 runs on the tile + halo, i.e. three latency-bound k-loop
# passes per workgroup at a third of the occupancy.  Off by default; RF_CONV_PAIR=1 turns it on.
CONV_PAIR = _os.environ.get("RF_CONV_PAIR", "0") == "1"
CONV_S2 = _os.environ.get("RF_CONV_S2", "1") != "0"  # ... and for every stride-2 3x3 convolution it supports


def pack_pointwise_weights(w: torch.Tensor) -> torch.Tensor:
    """(cout, cin) fp32 device tensor -> the bf16 fragment-order buffer rf_pointwise_bf16 keeps in registers."""
    cout, cin = w.shape
    out = torch.empty(int(_hip.lib().rf_pointwise_packed_elems(cin, cout)), device=w.device, dtype=torch.bfloat16)
    w = w.contiguous().float()
    check(_hip.lib().rf_pointwise_pack_bf16(ptr(w), ptr(out), cin, cout, K._stream()), "rf_pointwise_pack_bf16")
    return out


class HRNet16Backbone(VideoBackboneModule):
    def __init__(self, configs: Optional[VideoBackboneConfig] = None):
        super().__init__()
        self.configs = configs
        self._Backbone = _Tree()
        for conv, bn, cin, cout, k in UNITS:
            # reference init (hrnetv2.py:502-512): conv ~ N(0, 0.001), BN weight 1 / bias 0
            self._Backbone.put(conv + ".weight", torch.randn(cout, cin, k, k) * 0.001)
            if bn is not None:
                self._Backbone.put(bn + ".weight", torch.ones(cout))
                self._Backbone.put(bn + ".bias", torch.zeros(cout))
                self._Backbone.put(bn + ".running_mean", torch.zeros(cout), buffer=True)
                self._Backbone.put(bn + ".running_var", torch.ones(cout), buffer=True)
                self._Backbone.put(bn + ".num_batches_tracked", torch.tensor(0, dtype=torch.long), buffer=True)
        self._folded: Optional[Dict[str, tuple]] = None
        self._folded_key = None
        self._fidx_cache: Dict[tuple, torch.Tensor] = {}
        self.token_cache = None  # optional token_cache.TokenCache (persistent backbone-feature cache)

    # ---- plugin contract ---------------------------------------------------------------------
    @property
    def output_feature_shape(self) -> tuple:
        return (240, 8, 8)

    def train(self, mode: bool = True):
        """Frozen encoder: BatchNorm stays in eval mode (InverseForm.py:69-71)."""
        super().train(mode)
        return self

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        """(N,3,H,W) -> (N,240,8,8).  The result is a channels-last *view* of the pooled NHWC map, so
        the ``permute(0,2,3,1).reshape`` the caller applies (routeformer.py:478-480) is free."""
        tok = self.encode_tokens(images.unsqueeze(0), None)
        return tok[:, :64, :].reshape(images.shape[0], 8, 8, 240).permute(0, 3, 1, 2)

    # ---- checkpoint compatibility --------------------------------------------------------------
    def load_inverseform_checkpoint(self, path: str):
        """Load the Qualcomm checkpoint the reference uses (key remapping of InverseForm.py:94-133:
        strip 'module.'/'model.' and 'backbone.' prefixes, keep shape-matching trunk entries)."""
        raw = torch.load(path, map_location="cpu")["state_dict"]
        own = self._Backbone.state_dict()
        picked = {}
        for k, v in raw.items():
            parts = k.split(".")
            while parts and parts[0] in ("module", "modules", "model", "backbone"):
                parts = parts[1:]
            k2 = ".".join(parts)
            if k2 in own and own[k2].shape == v.shape:
                picked[k2] = v
        own.update(picked)
        self._Backbone.load_state_dict(own)
        self._folded = None
        return len(picked)

    # ---- weight preparation (BN folding, NHWC filter layout) ------------------------------------
    # ---- backbone-feature cache ----------------------------------------------------------------------
    def fingerprint(self) -> int:
        """63-bit digest of the trunk's weights and buffers (what the cached tokens depend on besides the frames and the
        arithmetic mode): cache keys are namespaced with it, as the reference's torchcache keys on the module hash."""
        import hashlib
        key = (tuple(p._version for p in self._Backbone.parameters()), tuple(b._version for b in self._Backbone.buffers()),
               id(next(self._Backbone.parameters())), self._weights_epoch())
        hit = self.__dict__.get("_fingerprint")
        if hit is None or hit[0] != key:
            h = hashlib.blake2b(digest_size=8)
            for name, t in sorted(self._Backbone.state_dict().items()):
                h.update(name.encode())
                h.update(t.detach().cpu().contiguous().numpy().tobytes())
            hit = self.__dict__["_fingerprint"] = (key, int.from_bytes(h.digest(), "little") & 0x7FFFFFFFFFFFFFFF)
        return hit[1]

    @property
    def token_cache(self):
        return self.__dict__.get("_token_cache_obj")

    @token_cache.setter
    def token_cache(self, cache):
        if cache is not None:
            cache.bind(self.fingerprint())  # (re-attach the cache after loading other weights into the trunk)
        self.__dict__["_token_cache_obj"] = cache

    def _weights_epoch(self) -> int:
        """Part of the cache keys below.  The fused optimizer kernels rewrite parameters through raw pointers and never
        bump ``_version`` (the engine counts those writes in ``kernels.WEIGHTS_EPOCH``): as soon as a trunk parameter is
        trainable or lives in an engine's flat buffer (``train_backbone=True``, InverseForm.py:72-78 -- not built yet), the
        folded weights and the token-cache namespace must follow that counter too.  Frozen trunk: constant 0,
        so an optimizer step elsewhere does not invalidate anything."""
        from routeformer_amd import kernels as K
        live = any(p.requires_grad or hasattr(p, "_rf_grad") for p in self._Backbone.parameters())
        return K.WEIGHTS_EPOCH if live else 0

    def _prepare(self, device):
        key = (str(device), tuple(p._version for p in self._Backbone.parameters()),
               tuple(b._version for b in self._Backbone.buffers()), id(next(self._Backbone.parameters())), self._weights_epoch())
        if self._folded is not None and self._folded_key == key:
            return self._folded
        sd = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in self._Backbone.state_dict().items()}
        folded = {}
        with torch.no_grad():
            for conv, bn, cin, cout, k in UNITS:
                w = sd[conv + ".weight"]
                if bn is not None:
                    scale = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + 1e-5)
                    bias = (sd[bn + ".bias"] - sd[bn + ".running_mean"] * scale).contiguous()
                    w = w * scale.view(-1, 1, 1, 1)
                else:
                    bias = None
                if conv == "conv0":
                    folded[conv] = (w.contiguous(), None, cin, cout, k, None)
                    continue
                cin_p = (cin + 3) // 4 * 4  # conv1 reads the 4-channel (zero-padded) stem output
                wk = torch.zeros(cout, k, k, cin_p, device=device, dtype=torch.float32)
                wk[..., :cin] = w.permute(0, 2, 3, 1)
                wb = None  # bf16 copy in MFMA fragment order for the raster-window 3x3 kernel
                if k == 3 and bias is not None and _hip.lib().rf_conv3x3_bf16_supported(cin_p, cout) and conv not in ("conv1", "conv2"):
                    wb = pack_conv3x3_weights(wk)  # (the stride-2 kernel reads the same fragment order for these shapes)
                elif k == 3 and bias is not None and _hip.lib().rf_conv3x3s2_packed_elems(cin_p, cout) > 0:
                    wb = pack_conv3x3s2_weights(wk)  # stride-2 only shapes: the stem pair, the fuse layers' / transitions' chains
                elif k == 1 and bias is not None and _hip.lib().rf_pointwise_bf16_supported(cin_p, cout):
                    wb = pack_pointwise_weights(wk.view(cout, cin_p))  # streaming 1x1 kernel (bf16 maps)
                folded[conv] = (wk.contiguous(), bias, cin_p, cout, k, wb)
        self._folded, self._folded_key = folded, key
        return folded

    # ---- execution --------------------------------------------------------------------------------
    @staticmethod
    def _act_dtype():
        """Storage type of the trunk's activation maps: bf16 in the bf16 matrix-core mode (every convolution rounds
        its input to bf16 there anyway -- the maps just stop carrying the other 16 bits through HBM), else fp32."""
        return torch.bfloat16 if K._PRECISION == 1 else torch.float32

    @staticmethod
    def _act_code(t):
        return 1 if (t.dtype if isinstance(t, torch.Tensor) else t) == torch.bfloat16 else 0

    def _conv(self, W, unit, x, stride=1, relu=False, residual=None):
        w, b, cin, cout, k, wb = W[unit]
        N, H, Wd, C = x.shape
        assert C == cin, (unit, C, cin)
        pad = 1 if k == 3 else 0
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (Wd + 2 * pad - k) // stride + 1
        y = torch.empty(N, Ho, Wo, cout, device=x.device, dtype=x.dtype)
        act = self._act_code(x)
        assert residual is None or residual.dtype == x.dtype
        ev = K.PROFILE.begin() if K.PROFILE.on else None
        fast = (wb is not None and k == 3 and stride == 1 and K._PRECISION == 1 and _hip.lib().rf_conv3x3_bf16_supported(cin, cout))
        pw = wb is not None and k == 1 and stride == 1 and K._PRECISION == 1 and act == 1
        s2 = (CONV_S2 and wb is not None and k == 3 and stride == 2 and K._PRECISION == 1 and act == 1
              and (STEM_S2 if unit in ("conv1", "conv2") else True)
              and H % 2 == 0 and Wd % 2 == 0 and _hip.lib().rf_conv3x3s2_bf16_supported(cin, cout, Wd))
        if s2:  # stride-2 raster window on bf16 maps (the stem pair, fuse-layer / transition chains): no im2col index arithmetic
            fn, cargs = _hip.lib().rf_conv3x3s2_bf16, (ptr(x), ptr(wb), ptr(b), ptr(residual), ptr(y), N, H, Wd, cin, cout,
                                                       1 if relu else 0)
        elif pw:  # 1x1 over bf16 maps: streaming GEMM, weights in registers
            fn, cargs = _hip.lib().rf_pointwise_bf16, (ptr(x), ptr(wb), ptr(b), ptr(residual), ptr(y), N * H * Wd, cin, cout,
                                                       1 if relu else 0)
        elif fast:  # 3x3/s1 on the bf16 matrix cores straight out of an LDS raster window
            fn, cargs = _hip.lib().rf_conv3x3_bf16, (ptr(x), ptr(wb), ptr(b), ptr(residual), ptr(y), act, N, H, Wd, cin, cout,
                                                     1 if relu else 0)
        else:
            fn, cargs = _hip.lib().rf_conv2d_nhwc, (ptr(x), ptr(w), ptr(b), ptr(residual), ptr(y), act, N, H, Wd, cin, cout, k,
                                                    stride, pad, Ho, Wo, cout, cout, 1 if relu else 0, K._PRECISION)
        check(fn(*cargs, K._stream()), "trunk convolution")
        if ev is not None:  # algorithmic work: one read of x / w (/ residual), one write of y
            M = N * Ho * Wo
            tag = f"conv3x3s2_kernel<{cin}, {cout}>" if s2 else f"pointwise_kernel<{cin}, {cout}>" if pw else \
                f"conv3x3_kernel<{cin}, {cout}, {'__bf16' if act else 'float'}>" if fast else \
                f"gemm2_kernel<{K._PRECISION}, 3, 0, {1 if cout <= 16 else (2 if cout <= 32 else 0)}>"
            es = x.element_size()
            keep = (x, w, b, wb, residual, y)
            K.PROFILE.end(tag, ev, 2.0 * M * cout * k * k * cin,
                          es * (x.numel() + M * cout * (2 if residual is not None else 1)) + 4.0 * w.numel(),
                          replay=lambda f=fn, a=cargs, kp=keep: f(*a, K._stream()))
        return y

    def _conv_group(self, W, units, xs, residuals):
        """relu(conv3x3(x_b) + bias_b [+ residual_b]) for every branch b in one launch."""
        n = len(units)
        arr = (_hip.ConvEntry * n)()
        ys = []
        for i, (unit, x) in enumerate(zip(units, xs)):
            w, b, cin, cout, k, wb = W[unit]
            N, H, Wd, C = x.shape
            assert C == cin and k == 3 and wb is not None and x.is_contiguous()
            y = torch.empty(N, H, Wd, cout, device=x.device, dtype=x.dtype)
            r = None if residuals is None else residuals[i]
            assert r is None or (r.dtype == x.dtype and r.shape == y.shape and r.is_contiguous())
            e = arr[i]
            e.x, e.w_packed, e.bias, e.residual, e.y = ptr(x), ptr(wb), ptr(b), ptr(r), ptr(y)
            e.N, e.H, e.W, e.cin, e.cout, e.relu = N, H, Wd, cin, cout, 1
            ys.append(y)
        ev = K.PROFILE.begin() if K.PROFILE.on else None
        code = self._act_code(xs[0])
        check(_hip.lib().rf_conv3x3_group_bf16(arr, n, code, K._stream()), "rf_conv3x3_group_bf16")
        if ev is not None:
            es = xs[0].element_size()
            keep = (arr, list(xs), ys, residuals)
            K.PROFILE.end("conv3x3_group_kernel", ev,
                          float(sum(2.0 * y.numel() * 9 * x.shape[-1] for x, y in zip(xs, ys))),
                          float(sum(es * (x.numel() + y.numel() * (2 if residuals is not None else 1)) for x, y in zip(xs, ys))),
                          replay=lambda a=arr, n_=n, c_=code, kp=keep: _hip.lib().rf_conv3x3_group_bf16(a, n_, c_, K._stream()))
        return ys

    @classmethod
    def _upsample(cls, x, size, *, addend=None, out=None, ldy=None, accumulate=False, relu=False):
        """``out``: tensor, or a raw device address of maps stored in x's dtype."""
        N, Hi, Wi, C = x.shape
        Ho, Wo = size
        if out is None:
            out = torch.empty(N, Ho, Wo, C, device=x.device, dtype=x.dtype)
            ldy = C
        assert (addend is None or addend.dtype == x.dtype) and (not isinstance(out, torch.Tensor) or out.dtype == x.dtype)
        check(_hip.lib().rf_upsample_bilinear_nhwc(ptr(x), ptr(addend), out.data_ptr() if isinstance(out, torch.Tensor)
                                                   else out, cls._act_code(x), N, Hi, Wi, C, Ho, Wo, ldy,
                                                   1 if accumulate else 0, 1 if relu else 0, K._stream()),
              "rf_upsample_bilinear_nhwc")
        return out

    @classmethod
    def _add(cls, a, b, relu):
        out = torch.empty_like(a)
        assert a.dtype == b.dtype
        check(_hip.lib().rf_add_relu(ptr(a), ptr(b), ptr(out), cls._act_code(a), a.numel(), 1 if relu else 0,
                                     K._stream()), "rf_add_relu")
        return out

    def _pair_ok(self, W, p, x) -> bool:
        """Can BasicBlock ``p`` on map ``x`` take the fused pair kernel (bf16 maps, packed weights, window fits LDS)?"""
        u1, u2 = W.get(p + ".conv1"), W.get(p + ".conv2")
        return (CONV_PAIR and K._PRECISION == 1 and x.dtype == torch.bfloat16 and u1 is not None and u2 is not None
                and u1[5] is not None and u2[5] is not None and u1[2] == u1[3] == u2[2] == u2[3] == x.shape[-1] and u1[4] == 3
                and bool(_hip.lib().rf_conv3x3_pair_supported(x.shape[-1], x.shape[2])))

    def _conv_pairs(self, W, blocks, xs):
        """relu(conv2(relu(conv1 x_b + b1)) + b2 + x_b) for the BasicBlocks ``blocks`` (unit prefixes) of up to four independent
        maps in ONE launch, the intermediate maps in LDS (rf_conv3x3_pair_group_bf16; hrnetv2.py:45-61)."""
        n = len(blocks)
        arr = (_hip.ConvPairEntry * n)()
        ys = []
        for i, (p, x) in enumerate(zip(blocks, xs)):
            (w1, b1, c, _, _, wb1), (w2, b2, _, _, _, wb2) = W[p + ".conv1"], W[p + ".conv2"]
            N, H, Wd, C = x.shape
            assert C == c and x.is_contiguous() and x.dtype == torch.bfloat16
            y = torch.empty_like(x)
            e = arr[i]
            e.x, e.w1_packed, e.bias1, e.w2_packed, e.bias2, e.y = ptr(x), ptr(wb1), ptr(b1), ptr(wb2), ptr(b2), ptr(y)
            e.N, e.H, e.W, e.c = N, H, Wd, C
            ys.append(y)
        ev = K.PROFILE.begin() if K.PROFILE.on else None
        check(_hip.lib().rf_conv3x3_pair_group_bf16(arr, n, K._stream()), "rf_conv3x3_pair_group_bf16")
        if ev is not None:
            keep = (arr, list(xs), ys)
            K.PROFILE.end("conv3x3_pair16_kernel" if (n == 1 and xs[0].shape[-1] == 16) else "conv3x3_pair_group_kernel", ev,
                          float(sum(2 * 2.0 * x.numel() * 9 * x.shape[-1] for x in xs)), float(sum(2 * 2 * x.numel() for x in xs)),
                          replay=lambda a=arr, n_=n, kp=keep: _hip.lib().rf_conv3x3_pair_group_bf16(a, n_, K._stream()))
        return ys

    def _basic(self, W, p, x):
        if self._pair_ok(W, p, x):
            return self._conv_pairs(W, [p], [x])[0]
        y = self._conv(W, p + ".conv1", x, relu=True)
        return self._conv(W, p + ".conv2", y, relu=True, residual=x)

    def _bottleneck(self, W, p, x):
        y = self._conv(W, p + ".conv1", x, relu=True)
        y = self._conv(W, p + ".conv2", y, relu=True)
        res = self._conv(W, p + ".downsample.0", x) if (p + ".downsample.0") in W else x
        return self._conv(W, p + ".conv3", y, relu=True, residual=res)

    def _module(self, W, p, xs):
        """One HighResolutionModule: per-branch BasicBlocks, then the cross-resolution fuse
        (hrnetv2.py:250-277), summing terms in the reference's order j = 0..nb-1, ReLU on the last."""
        nb = len(xs)
        xs = list(xs)
        if GROUP_BRANCHES and nb > 2 and K._PRECISION == 1 and all(W[f"{p}.branches.{b}.0.conv1"][5] is not None for b in range(nb)):
            # the same conv step of the LOW-RESOLUTION branches in one launch (rf_conv3x3_group_bf16): each is a few
            # dozen latency-bound workgroups (17 us for 42 workgroups at 4x4) that would otherwise leave the chip idle
            # one after the other.  The 28x28 branch keeps its own launch: a group shares one LDS size and one
            # register budget, and with the 128-channel window every 16-channel workgroup loses half its occupancy
            # (all four together: trunk 2.66 -> 3.00 ms).
            lo = list(range(1, nb))
            for k in range(BLOCKS_PER_BRANCH):
                if all(self._pair_ok(W, f"{p}.branches.{b}.{k}", xs[b]) for b in range(nb)):
                    # fused BasicBlock pairs: the high-resolution branch in its own launch, the others together
                    x0 = self._conv_pairs(W, [f"{p}.branches.0.{k}"], [xs[0]])[0]
                    xs = [x0] + self._conv_pairs(W, [f"{p}.branches.{b}.{k}" for b in lo], [xs[b] for b in lo])
                    continue
                y0 = self._conv(W, f"{p}.branches.0.{k}.conv1", xs[0], relu=True)
                ys = self._conv_group(W, [f"{p}.branches.{b}.{k}.conv1" for b in lo], [xs[b] for b in lo], None)
                x0 = self._conv(W, f"{p}.branches.0.{k}.conv2", y0, relu=True, residual=xs[0])
                xs = [x0] + self._conv_group(W, [f"{p}.branches.{b}.{k}.conv2" for b in lo], ys, [xs[b] for b in lo])
        else:
            for b in range(nb):
                for k in range(BLOCKS_PER_BRANCH):
                    xs[b] = self._basic(W, f"{p}.branches.{b}.{k}", xs[b])
        if FUSE_SUM:
            return self._fuse_grouped(W, p, xs)
        outs = []
        for i in range(nb):
            size = tuple(xs[i].shape[1:3])
            acc, owned = None, False  # owned: acc is a private buffer that may be updated in place
            for j in range(nb):
                last = j == nb - 1
                q = f"{p}.fuse_layers.{i}.{j}"
                if j < i:  # stride-2 3x3 chain; the running sum rides in the last conv's epilogue
                    t = xs[j]
                    for k in range(i - j):
                        final = k == i - j - 1
                        t = self._conv(W, f"{q}.{k}.0", t, stride=2, relu=not final,
                                       residual=acc if final else None)
                    acc, owned = t, True
                elif j == i:
                    if acc is None:
                        acc, owned = xs[j], False
                    else:
                        acc, owned = self._add(acc, xs[j], relu=last), True
                else:  # 1x1 conv + BN at low resolution, bilinear up-sample into the sum
                    t = self._conv(W, q + ".0", xs[j])
                    if owned:
                        self._upsample(t, size, out=acc, ldy=acc.shape[-1], accumulate=True, relu=last)
                    else:
                        acc, owned = self._upsample(t, size, addend=acc, relu=last), True
            outs.append(acc)
        return outs

    def _fuse_grouped(self, W, p, xs):
        """The cross-resolution fuse (hrnetv2.py:250-271) with ONE launch for all the up-sampled terms of a module:
        out_i = relu(sum_{j<i} stride-2 chains + x_i + sum_{j>i} up(conv1x1_j(x_j))), terms in the reference's order.
        The chains keep their running sum in the last conv's epilogue; x_i and the up-sampled terms of every output
        branch i < nb - 1 are added by rf_fuse_upsample_sum (no read-modify-write pass per (i, j) pair, no add launch)."""
        nb = len(xs)
        outs, ents = [None] * nb, []
        for i in range(nb):
            acc = None
            for j in range(i):
                t = xs[j]
                for k in range(i - j):
                    final = k == i - j - 1
                    t = self._conv(W, f"{p}.fuse_layers.{i}.{j}.{k}.0", t, stride=2, relu=not final,
                                   residual=acc if final else None)
                acc = t
            if i == nb - 1:  # lowest resolution: no up-sampled terms, ReLU after adding x_i
                outs[i] = xs[i] if acc is None else self._add(acc, xs[i], relu=True)
                continue
            srcs = [self._conv(W, f"{p}.fuse_layers.{i}.{j}.0", xs[j]) for j in range(i + 1, nb)]
            out = torch.empty_like(xs[i])
            outs[i] = out
            ents.append((acc, xs[i], srcs, out))
        if ents:
            arr = (_hip.FuseEntry * len(ents))()
            keep = []
            for e, (acc, xi, srcs, out) in zip(arr, ents):
                N, Ho, Wo, C = out.shape
                first, second = (acc, xi) if acc is not None else (xi, None)
                assert first.is_contiguous() and (second is None or second.is_contiguous()) and first.shape == out.shape
                e.base, e.base2, e.out = ptr(first), ptr(second), ptr(out)
                e.N, e.Ho, e.Wo, e.C, e.n_src, e.relu = N, Ho, Wo, C, len(srcs), 1
                for s_i, t in enumerate(srcs):
                    assert t.is_contiguous() and t.shape[0] == N and t.shape[3] == C and t.dtype == out.dtype
                    e.src[s_i], e.Hi[s_i], e.Wi[s_i] = ptr(t), t.shape[1], t.shape[2]
                keep.append((first, second, srcs))
            ev = K.PROFILE.begin() if K.PROFILE.on else None
            check(_hip.lib().rf_fuse_upsample_sum(arr, len(ents), self._act_code(xs[0]), K._stream()), "rf_fuse_upsample_sum")
            if ev is not None:
                es = xs[0].element_size()
                code = self._act_code(xs[0])
                K.PROFILE.end("fuse_upsample_sum_kernel", ev, 0.0,
                              float(sum(es * (o.numel() * (2 + (a is not None)) + sum(t.numel() for t in sr))
                                        for a, _, sr, o in ents)),
                              replay=lambda a_=arr, n_=len(ents), c_=code, kp=(keep, ents): _hip.lib().rf_fuse_upsample_sum(
                                  a_, n_, c_, K._stream()))
        return outs

    def encode_tokens(self, video, frame_idx: Optional[torch.Tensor]) -> torch.Tensor:
        """video (B,T,3,H,W) fp16/fp32 -- or a LIST of such clips (several camera streams batched through
        the trunk at once) -- + frame indices (F,) -> tokens (S*B*F, 65, 240), stream-major, each with the
        trailing constant -1 row (routeformer.py:478-487).  Frame gather, dtype cast and conv0 are one kernel."""
        videos = list(video) if isinstance(video, (list, tuple)) else [video]
        return self.encode_clips([(v, frame_idx) for v in videos])

    def encode_clips(self, clips, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``_encode_clips_uncached`` behind the optional ``token_cache`` (token_cache.TokenCache, the counterpart of
        the reference's ``@torchcache(persistent=True)``, video_backbone/__init__.py:14-32): frames whose content has
        been seen are served from HBM, a pass with unknown frames runs the trunk once and stores its tokens.  Inside a
        stream capture (the engine's trunk graphs) the trunk always runs: the engine consults the cache itself."""
        cache = self.token_cache
        if cache is None or torch.cuda.is_current_stream_capturing():
            return self._encode_clips_uncached(clips, out)
        cache.bind(self.fingerprint())  # (cheap when the weights have not changed)
        clips = [((v if v.dtype in (torch.float16, torch.float32, torch.uint8) else v.float()).contiguous(), fi) for v, fi in clips]
        return cache.tokens_for(clips, self._encode_clips_uncached, out)

    def _encode_clips_uncached(self, clips, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """clips: [(video (B,T,3,H,W), frame_idx (F,) or None)] -- clips may differ in B, T and frame
        indices (e.g. the history and the target window of one training item) but share H x W.
        One trunk pass over ALL selected frames -> tokens (sum_i B_i*F_i, 65, 240), clip-major."""
        vids, fidx, counts = [], [], []
        for v, fi in clips:
            if not v.is_cuda:
                raise _hip.HipLibraryError("HRNet16Backbone runs on the GPU only; there is no CPU path")
            v = (v if v.dtype in (torch.float16, torch.float32, torch.uint8) else v.float()).contiguous()
            B, T, C3, H, Wd = v.shape
            assert C3 == 3 and H % 2 == 0 and Wd % 2 == 0 and v.shape[3:] == clips[0][0].shape[3:]
            dev = v.device
            if fi is None:
                fi = torch.arange(T, dtype=torch.int32)
            key = (tuple(fi.tolist()), str(dev))
            if key not in self._fidx_cache:  # cached: no host->device copy inside a captured step
                self._fidx_cache[key] = fi.to(device=dev, dtype=torch.int32)
            vids.append(v)
            fidx.append(self._fidx_cache[key])
            counts.append(B * fidx[-1].numel())
        H, Wd = vids[0].shape[3:]
        dev = vids[0].device
        N = sum(counts)
        W = self._prepare(dev)
        adt = self._act_dtype()
        act, es = self._act_code(adt), (2 if adt == torch.bfloat16 else 4)
        x = torch.empty(N, H // 2, Wd // 2, 4, device=dev, dtype=adt)
        off = 0
        for v, fi, n in zip(vids, fidx, counts):
            check(_hip.lib().rf_stem_conv0(ptr(v), {torch.float16: 0, torch.float32: 1, torch.uint8: 2}[v.dtype], ptr(fi), ptr(W["conv0"][0]),
                                           x.data_ptr() + es * off * (H // 2) * (Wd // 2) * 4, act, v.shape[0], v.shape[1],
                                           fi.numel(), H, Wd, K._stream()), "rf_stem_conv0")
            off += n
        tokens = out if out is not None else torch.empty(N, 65, 240, device=dev, dtype=torch.float32)
        assert tuple(tokens.shape) == (N, 65, 240) and tokens.is_contiguous()
        # (Measured dead end, profiles/r03/trunk_chunks.txt: running the rest of the trunk over 2 / 4 / 8 image slices, so
        #  that its chip-filling launches leave wave slots to the latency-bound transformer chain on the other streams of
        #  the step: 6.79 -> 6.77 / 8.19 / 13.3 ms per step -- the low-resolution half of the trunk is latency-bound
        #  itself and every slice pays its ~90 launches again.)
        self._trunk_body(W, x, tokens, adt, act, es)
        return tokens

    def _trunk_body(self, W, x, tokens, adt, act, es):
        """Everything after conv0 for a contiguous slice of images: (n, H/2, W/2, 4) maps -> tokens (n, 65, 240)."""
        N, dev = x.shape[0], x.device
        x = self._conv(W, "conv1", x, stride=2, relu=True)
        x = self._conv(W, "conv2", x, stride=2, relu=True)
        x = self._bottleneck(W, "layer1.0", x)
        x = self._bottleneck(W, "layer1.1", x)
        xs = [self._conv(W, "transition1.0.0", x, relu=True),
              self._conv(W, "transition1.1.0.0", x, stride=2, relu=True)]
        for stage, nb, nmod in STAGES:
            if len(xs) < nb:
                xs.append(self._conv(W, f"transition{nb - 1}.{nb - 1}.0.0", xs[-1], stride=2, relu=True))
            for m in range(nmod):
                xs = self._module(W, f"{stage}.{m}", xs)
        Hf, Wf = xs[0].shape[1:3]
        if FUSE_SUM and len(xs) <= 4 and sum(t.shape[-1] for t in xs) == 240:
            # up-sample + concat + AdaptiveAvgPool + token layout in one launch: the 240-channel map is never stored
            import ctypes
            n = len(xs)
            maps = (ctypes.c_void_p * n)(*[ptr(t.contiguous()) for t in xs])
            dims = [(ctypes.c_int32 * n)(*[t.shape[k] for t in xs]) for k in (1, 2, 3)]
            ev = K.PROFILE.begin() if K.PROFILE.on else None
            check(_hip.lib().rf_concat_pool_tokens(maps, dims[0], dims[1], dims[2], n, act, ptr(tokens), N, K._stream()),
                  "rf_concat_pool_tokens")
            if ev is not None:
                K.PROFILE.end("concat_pool_tokens_kernel", ev, 0.0, float(sum(es * t.numel() for t in xs) + 4 * tokens.numel()))
            return tokens
        feats = torch.empty(N, Hf, Wf, 240, device=dev, dtype=adt)
        off = 0
        for t in xs:  # concat along channels; identity-scale "upsample" copies branch 0
            c = t.shape[-1]
            self._upsample(t, (Hf, Wf), out=feats.data_ptr() + es * off, ldy=240)
            off += c
        check(_hip.lib().rf_avgpool8_tokens(ptr(feats), act, ptr(tokens), N, Hf, Wf, 240, K._stream()),
              "rf_avgpool8_tokens")
        return tokens
