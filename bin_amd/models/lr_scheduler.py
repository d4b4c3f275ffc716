"""Restartable LR schedules (reference models/lr_scheduler.py:10-66), host-side scalar logic."""
import math
from collections import Counter, defaultdict

from torch.optim.lr_scheduler import _LRScheduler


class MultiStepLR_Restart(_LRScheduler):
    """Multi-step decay with optional warm restarts (reference lr_scheduler.py:10-34)."""

    def __init__(self, optimizer, milestones, restarts=None, weights=None, gamma=0.1, clear_state=False,
                 last_epoch=-1):
        self.milestones = Counter(milestones)
        self.gamma = gamma
        self.clear_state = clear_state
        self.restarts = [v + 1 for v in (restarts if restarts else [0])]
        self.restart_weights = weights if weights else [1]
        assert len(self.restarts) == len(self.restart_weights), "restarts and their weights do not match."
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        if self.last_epoch in self.restarts:
            if self.clear_state:
                self.optimizer.state = defaultdict(dict)
            wgt = self.restart_weights[self.restarts.index(self.last_epoch)]
            return [g["initial_lr"] * wgt for g in self.optimizer.param_groups]
        if self.last_epoch not in self.milestones:
            return [g["lr"] for g in self.optimizer.param_groups]
        return [g["lr"] * self.gamma ** self.milestones[self.last_epoch] for g in self.optimizer.param_groups]


class CosineAnnealingLR_Restart(_LRScheduler):
    """Cosine annealing with restarts (reference lr_scheduler.py:37-66)."""

    def __init__(self, optimizer, T_period, restarts=None, weights=None, eta_min=0, last_epoch=-1):
        self.T_period = T_period
        self.T_max = self.T_period[0]
        self.eta_min = eta_min
        self.restarts = [v + 1 for v in (restarts if restarts else [0])]
        self.restart_weights = weights if weights else [1]
        self.last_restart = 0
        assert len(self.restarts) == len(self.restart_weights), "restarts and their weights do not match."
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        e = self.last_epoch
        if e == 0:
            return self.base_lrs
        if e in self.restarts:
            self.last_restart = e
            self.T_max = self.T_period[self.restarts.index(e) + 1]
            wgt = self.restart_weights[self.restarts.index(e)]
            return [g["initial_lr"] * wgt for g in self.optimizer.param_groups]
        if (e - self.last_restart - 1 - self.T_max) % (2 * self.T_max) == 0:
            return [g["lr"] + (b - self.eta_min) * (1 - math.cos(math.pi / self.T_max)) / 2
                    for b, g in zip(self.base_lrs, self.optimizer.param_groups)]
        num = 1 + math.cos(math.pi * (e - self.last_restart) / self.T_max)
        den = 1 + math.cos(math.pi * ((e - self.last_restart) - 1) / self.T_max)
        return [num / den * (g["lr"] - self.eta_min) + self.eta_min for g in self.optimizer.param_groups]
