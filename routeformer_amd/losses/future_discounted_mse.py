"""``FutureDiscountedLoss(discount_factor, epsilon, loss_function)(y_pred, y_true)`` -- API and
quirks of ``routeformer/losses/future_discounted_mse.py:9-95``: gamma^t weights over dim 1, the
discount switches when ``current_epoch`` hits a key of the dict, and the smooth-L1 path still
evaluates ``abs(err) < epsilon`` (so ``epsilon=None`` raises) although it ignores the result."""
from typing import Dict, Union

import torch
import torch.nn.functional as F
from torch import nn


class FutureDiscountedLoss(nn.Module):
    current_epoch = 0  # set by the training harness (Lightning provides it in the reference)

    def __init__(self, discount_factor: Union[float, Dict[int, float]] = 0.9, epsilon: float = None,
                 loss_function: str = "mse"):
        super().__init__()
        if loss_function not in ("mae", "mse", "smooth_l1"):
            raise ValueError(f"Unknown loss function {loss_function}")
        is_dict = isinstance(discount_factor, dict)
        self.current_discount_factor = discount_factor[0] if is_dict else discount_factor
        self.discount_factor_dict = discount_factor if is_dict else {}
        self.epsilon = epsilon
        self.loss_function = loss_function

    def discount(self) -> float:
        """The discount in force at ``current_epoch`` (switches, stickily, when the epoch is a key)."""
        if self.current_epoch in self.discount_factor_dict:
            self.current_discount_factor = self.discount_factor_dict[self.current_epoch]
        return self.current_discount_factor

    def forward(self, y_pred, y_true):
        self.discount()
        extra = y_pred.dim() - 2
        assert extra >= 0
        steps = torch.arange(y_pred.shape[1], device=y_pred.device)
        weights = torch.pow(self.current_discount_factor, steps).view(1, -1, *([1] * extra))
        err = y_pred - y_true
        err = torch.where(err.abs() < self.epsilon, torch.zeros_like(err), err)
        if self.loss_function == "mae":
            return (err.abs() * weights).mean()
        if self.loss_function == "mse":
            return (err.pow(2) * weights).mean()
        return (F.smooth_l1_loss(y_pred, y_true, reduction="none") * weights).mean()
