"""ADE / FDE (``routeformer/score/error.py:10-51``)."""
import torch


def ade(predicted_trajectory, ground_truth_trajectory):
    """Mean L2 distance over the last dimension."""
    assert predicted_trajectory.shape == ground_truth_trajectory.shape, \
        "Predicted and ground truth trajectories must be of the same shape"
    return torch.linalg.vector_norm(predicted_trajectory - ground_truth_trajectory, dim=-1).mean()


def fde(predicted_trajectory, ground_truth_trajectory):
    """Norm of the difference of the LAST element along dim 0 (the reference indexes the batch
    dimension; its trainer calls this on batch-of-one slices, full_comparison.py:668-671)."""
    assert predicted_trajectory.shape == ground_truth_trajectory.shape, \
        "Predicted and ground truth trajectories must be of the same shape"
    return torch.linalg.vector_norm(predicted_trajectory[-1] - ground_truth_trajectory[-1])


__all__ = ["ade", "fde"]
