"""CPU restatement of the ingest helpers -- TEST INFRASTRUCTURE ONLY (see routeformer_oracle.py's header).

``resize_area``: the box-filter definition of ``cv2.INTER_AREA`` down-scaling used by
``routeformer/io/dataset.py:1463-1492`` (``cv2.resize(frame, target, None, None, None, cv2.INTER_AREA)``).
**Parity unpinned**: OpenCV (``cv2``, a third-party dependency of the reference, ``opencv-python`` in its
``pyproject.toml``) is not installed in the build container and the reference holds no fixture for this path, so the
restatement follows OpenCV's published algorithm (area-weighted mean of the covered source pixels, result rounded to
nearest-even and saturated; the exact-half scale takes OpenCV's integer 2 x 2 fast path, which rounds a half up) and is checked only against itself and against exact integer-factor block means.
``TokenCacheModel``: what the device hash table must do, as a dict."""
import numpy as np


def resize_area(frames: np.ndarray, factor: float) -> np.ndarray:
    """uint8 (..., H, W) -> uint8 (..., int(H*factor), int(W*factor)); float64 accumulation."""
    H, W = frames.shape[-2:]
    h, w = int(H * factor), int(W * factor)
    if H == 2 * h and W == 2 * w:  # OpenCV's 2 x 2 fast path (resize.cpp, ResizeAreaFastVec): rounds a half UP
        f = frames.astype(np.int32)
        return ((f[..., 0::2, 0::2] + f[..., 0::2, 1::2] + f[..., 1::2, 0::2] + f[..., 1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sy, sx = H / h, W / w

    def weights(n_src, n_dst, s):
        Wm = np.zeros((n_dst, n_src))
        for o in range(n_dst):
            a, b = o * s, (o + 1) * s
            for i in range(int(np.floor(a)), min(n_src, int(np.ceil(b - 1e-9)))):
                Wm[o, i] = min(i + 1, b) - max(i, a)
        return Wm

    Wy, Wx = weights(H, h, sy), weights(W, w, sx)
    acc = np.einsum("oh,...hw,pw->...op", Wy, frames.astype(np.float64), Wx) / (sy * sx)
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


class TokenCacheModel:
    """key -> slot with first-come slot numbers and a capacity: the semantics of rf_cache_lookup / rf_cache_insert."""

    def __init__(self, capacity):
        self.capacity, self.map = capacity, {}

    def lookup(self, keys):
        return [self.map.get(int(k), -1) for k in keys]

    def insert(self, keys):
        for k in keys:
            if int(k) not in self.map and len(self.map) < self.capacity:
                self.map[int(k)] = len(self.map)
        return self.lookup(keys)
