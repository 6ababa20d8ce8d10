"""Linear warm-up + cosine annealing, stepped once per epoch -- interface of
``routeformer/optimizers/lr_scheduler.py:27-139`` as the driver uses it
(``experiments/full_comparison.py:702-709``: ``warmup_epochs=2, max_epochs=EPOCHS, interval="epoch"``).

Works on anything that exposes ``param_groups`` (a list of dicts with an ``"lr"`` entry): a
``torch.optim.Optimizer`` or the engine's ``FusedAdamW`` (one group; the learning rate is a scalar
kernel argument of ``rf_adamw_clip``, so changing it costs nothing and needs no re-capture)."""
import math
from typing import List


class LinearWarmupCosineAnnealingLR:
    def __init__(self, optimizer, warmup_epochs: int, max_epochs: int, warmup_start_lr: float = 0.0,
                 eta_min: float = 0.0, last_epoch: int = -1):
        self.optimizer = optimizer
        self.warmup_epochs, self.max_epochs = warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self._last_lr: List[float] = [g["lr"] for g in optimizer.param_groups]
        self.step()  # torch's _LRScheduler performs one step on construction (epoch 0 values)

    def get_lr(self) -> List[float]:
        """Chainable form: next learning rates from the current ones (what ``step()`` applies)."""
        e, w, m = self.last_epoch, self.warmup_epochs, self.max_epochs
        groups = self.optimizer.param_groups
        if e == w:
            return list(self.base_lrs)
        if e == 0:
            return [self.warmup_start_lr] * len(self.base_lrs)
        if e < w:
            return [g["lr"] + (b - self.warmup_start_lr) / (w - 1) for b, g in zip(self.base_lrs, groups)]
        if (e - 1 - m) % (2 * (m - w)) == 0:
            return [g["lr"] + (b - self.eta_min) * (1 - math.cos(math.pi / (m - w))) / 2
                    for b, g in zip(self.base_lrs, groups)]
        return [(1 + math.cos(math.pi * (e - w) / (m - w))) / (1 + math.cos(math.pi * (e - w - 1) / (m - w)))
                * (g["lr"] - self.eta_min) + self.eta_min for g in groups]

    def _get_closed_form_lr(self) -> List[float]:
        e, w, m = self.last_epoch, self.warmup_epochs, self.max_epochs
        if e < w:
            return [self.warmup_start_lr + e * (b - self.warmup_start_lr) / max(1, w - 1) for b in self.base_lrs]
        return [self.eta_min + 0.5 * (b - self.eta_min) * (1 + math.cos(math.pi * (e - w) / (m - w)))
                for b in self.base_lrs]

    def step(self, epoch=None):
        if epoch is None:
            self.last_epoch += 1
            values = self.get_lr()
        else:
            self.last_epoch = epoch
            values = self._get_closed_form_lr()
        for g, lr in zip(self.optimizer.param_groups, values):
            g["lr"] = lr
        self._last_lr = list(values)

    def get_last_lr(self) -> List[float]:
        return self._last_lr

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, state):
        self.__dict__.update(state)
