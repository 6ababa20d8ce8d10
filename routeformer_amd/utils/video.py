"""Dataset-side video ingest on the GPU (``routeformer/io/dataset.py:1439-1523``)."""
import torch

from routeformer_amd import _hip
from routeformer_amd._hip import check, ptr


def resize_area(video: torch.Tensor, factor: float) -> torch.Tensor:
    """``GEMDataset._apply_scaling`` for a down-scaling factor (``cv2.resize(frame, (int(W*factor), int(H*factor)),
    interpolation=cv2.INTER_AREA)``, io/dataset.py:1463-1492) on raw uint8 frames (..., H, W) resident in HBM: every
    output pixel is the coverage-weighted mean of the source pixels under it.  Returns uint8 (..., h, w)."""
    if not video.is_cuda:
        raise _hip.HipLibraryError("resize_area runs on the GPU only; there is no CPU path")
    assert video.dtype == torch.uint8 and 0.0 < factor <= 1.0
    v = video.contiguous()
    H, W = v.shape[-2:]
    h, w = int(H * factor), int(W * factor)  # the reference's target_resolution (io/dataset.py:1470-1473)
    out = torch.empty(v.shape[:-2] + (h, w), dtype=torch.uint8, device=v.device)
    n = v.numel() // (H * W)
    check(_hip.lib().rf_resize_area(ptr(v), ptr(out), n, H, W, h, w, torch.cuda.current_stream().cuda_stream), "rf_resize_area")
    return out
