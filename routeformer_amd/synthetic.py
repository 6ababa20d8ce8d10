"""Seeded synthetic weights and GEM-shaped batches (no datasets / checkpoints exist offline).

The same generator feeds (i) the golden-fixture script, which loads these weights into the
*reference* model in the build container, (ii) the parity tests, which load them into the CPU
oracle and into the HIP-backed product model, and (iii) ``bench.py``.  Tensors are generated on the
CPU with a per-key ``torch.Generator`` so that a key's values depend only on ``(seed, key, shape)``
-- not on iteration order or on which other keys exist.

Batch statistics follow SURVEY.md 8(d): per-step GPS motion ~ N((1.83, 0), 0.91) metres
(reference constants ``experiments/full_comparison.py:128``), videos U[0,1) in fp16
(``routeformer/io/dataset.py:1506-1523``), gaze U[0,1) at 200 Hz (``io/dataset.py:190``).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Mapping, Optional

import torch

_GPS_MEAN = (1.8332362885457094, 0.0)
_GPS_STD = 0.9090128501056961
GAZE_HZ = 200


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2**63 - 1))
    return g


def synth_tensor(key: str, like: torch.Tensor, seed: int) -> Optional[torch.Tensor]:
    """Value for one state_dict entry; ``None`` means "keep what the module already holds"
    (deterministic buffers such as the sinusoidal ``pe`` table)."""
    shape, leaf = tuple(like.shape), key.rsplit(".", 1)[-1]
    g = _gen(seed, key)
    if leaf == "pe":
        return None
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=like.dtype)
    if leaf == "running_var":
        return torch.rand(shape, generator=g) + 0.5
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if key.endswith("_embedding") and len(shape) == 3:  # the four (1,1,E) stream embeddings
        return torch.randn(shape, generator=g)
    if len(shape) == 1:
        if leaf == "weight":  # LayerNorm / BatchNorm scale
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    # conv2d stacks (the frozen HRNet trunk) get a 0.65 gain so ~70 residual convs keep O(1) features
    gain = 0.65 if len(shape) == 4 else 1.0
    return torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))


def synth_state_dict(template: Mapping[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    """Fill every entry of ``template`` (a module's ``state_dict()``) from ``seed``."""
    out = {}
    for k, v in template.items():
        t = synth_tensor(k, v, seed)
        out[k] = v.detach().clone() if t is None else t.to(v.dtype)
    return out


def state_dict_digest(sd: Mapping[str, torch.Tensor], keys: Optional[Iterable[str]] = None) -> float:
    """Cheap order-independent checksum used by fixtures to prove "same weights"."""
    tot = 0.0
    for k in (keys if keys is not None else sd.keys()):
        v = sd[k]
        if v.is_floating_point():
            tot += float(v.double().abs().sum()) * ((zlib.crc32(k.encode()) % 97) + 1)
    return tot


def synth_gps(B: int, T: int, seed: int, key: str = "gps") -> torch.Tensor:
    g = _gen(seed, key)
    step = torch.randn(B, T, 2, generator=g) * _GPS_STD + torch.tensor(_GPS_MEAN)
    return torch.cumsum(step, dim=1).to(torch.float32)


def synth_video(B: int, T: int, H: int, W: int, seed: int, key: str) -> torch.Tensor:
    g = _gen(seed, key)
    return torch.rand(B, T, 3, H, W, generator=g).to(torch.float16)


def synth_gaze(B: int, T: int, seed: int, output_fps: int = 5, key: str = "gaze") -> torch.Tensor:
    g = _gen(seed, key)
    n = T // output_fps * GAZE_HZ if T >= output_fps else T * GAZE_HZ // output_fps
    return torch.rand(B, n, 2, generator=g).to(torch.float32)


def synth_batch(B: int, T: int, seed: int, H: int = 224, W: int = 224, *,
                streams=("left_video", "right_video", "front_video"), gaze: bool = True,
                output_fps: int = 5, tag: str = "train") -> Dict[str, torch.Tensor]:
    """One ``Data`` dict (reference schema ``routeformer/io/dataset.py:43-62``)."""
    batch = {"gps": synth_gps(B, T, seed, f"{tag}.gps")}
    for s in streams:
        batch[s] = synth_video(B, T, H, W, seed, f"{tag}.{s}")
    if gaze:
        batch["gaze"] = synth_gaze(B, T, seed, output_fps, f"{tag}.gaze")
    return batch


def synth_item(B: int, T: int, P: int, seed: int, H: int = 224, W: int = 224, **kw):
    """``Item{train, target}``; the target GPS continues the input track."""
    train = synth_batch(B, T, seed, H, W, tag="train", **kw)
    target = synth_batch(B, P, seed, H, W, tag="target", **kw)
    g = _gen(seed, "target.step")
    step = torch.randn(B, P, 2, generator=g) * _GPS_STD + torch.tensor(_GPS_MEAN)
    target["gps"] = (train["gps"][:, -1:, :] + torch.cumsum(step, dim=1)).to(torch.float32)
    return {"train": train, "target": target}
