"""Motion-feature helpers (``routeformer/utils/vector.py:6-111``, ``utils/filter.py:5-43``).

Elementwise, O(B*T) work on (B,T,2) tensors: these stay as torch tensor plumbing on whatever device
the inputs live on (negligible, inputs carry no gradient)."""
import torch


def rotate(tensor: torch.Tensor, angle: torch.Tensor) -> torch.Tensor:
    """Rotate (B,L,2) vectors by per-sample angles (B,1[,1]) with R = [[c,-s],[s,c]], in fp32."""
    dtype = tensor.dtype
    c = torch.cos(angle.float()).reshape(-1, 1)
    s = torch.sin(angle.float()).reshape(-1, 1)
    x, y = tensor[..., 0].float(), tensor[..., 1].float()
    return torch.stack([c * x - s * y, s * x + c * y], dim=-1).to(dtype)


def estimate_angle(tensor: torch.Tensor) -> torch.Tensor:
    t = tensor.float()
    return torch.atan2(t[..., 1], t[..., 0]).unsqueeze(-1)


def estimate_angle_and_norm(tensor: torch.Tensor):
    t = tensor.float()
    return torch.atan2(t[..., 1], t[..., 0]).unsqueeze(-1), torch.linalg.vector_norm(t, dim=-1, keepdim=True)


def median_downsampler(tensor: torch.Tensor, target_length: int) -> torch.Tensor:
    """(B,T,C) -> (B,target,C): lower median of consecutive windows of T//target samples."""
    B, T, C = tensor.shape
    if target_length >= T:
        raise ValueError("Target length must be less than the current time steps.")
    w = T // target_length
    if tensor.is_cuda and tensor.dtype == torch.float32 and w <= 1024:  # one selection launch instead of a sort + gather
        from routeformer_amd import kernels as K
        return K.median_windows(tensor, target_length)
    win = tensor[:, : w * target_length].reshape(B, target_length, w, C)
    return win.sort(dim=2).values[:, :, (w - 1) // 2, :]
