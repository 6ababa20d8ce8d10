from .future_discounted_mse import FutureDiscountedLoss

__all__ = ["FutureDiscountedLoss"]
