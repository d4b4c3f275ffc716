"""Restartable learning-rate schedules, stated as closed-form curves of `last_epoch`.

Behavioural counterpart of the reference's `models/lr_scheduler.py:10-66` (class names, constructor arguments and the
attributes a `.state` pickle carries are fixed by that file; `tests/golden/g6_lr.npz` pins the curves to it).

Both schedules are described by ONE function, `shape(e)`: the multiplier of a group's starting rate at epoch `e`, as if
nobody else touched the optimizer.  It is found by locating the restart cycle `e` falls into (bisect over the restart
epochs) and evaluating the cycle's curve at the offset into it.  `get_lr()` then moves each group from the rate it HAS
to the rate it should have by the ratio `shape(e) / shape(e-1)` — so a rate somebody else set between two steps (the
warm-up in `BaseModel.update_learning_rate`) carries through exactly as it does in the reference — and lands on
`initial_lr * weight` exactly at a restart.
"""
import math
from bisect import bisect_right
from collections import Counter, defaultdict

from torch.optim.lr_scheduler import _LRScheduler


def _restart_table(restarts, weights):
    """(epochs, weights) of the restart points; the schedulers count an epoch as begun one step after the listed one."""
    epochs = [v + 1 for v in (restarts if restarts else [0])]
    weights = list(weights) if weights else [1]
    assert len(epochs) == len(weights), "restarts and their weights do not match."
    return epochs, weights


class _RestartSchedule(_LRScheduler):
    """Shared machinery: cycle lookup and the ratio step."""

    floor = 0.0                         # the rate a curve decays towards (eta_min of the cosine)

    def _cycle(self, e):
        """(index of the cycle epoch e lies in: 0 = before any restart, start epoch of that cycle, its weight)."""
        order = sorted(range(len(self.restarts)), key=lambda i: self.restarts[i])
        starts = [self.restarts[i] for i in order]
        k = bisect_right(starts, e)
        if k == 0:
            return 0, 0, 1
        return order[k - 1] + 1, starts[k - 1], self.restart_weights[order[k - 1]]

    def curve(self, cycle, t, start):   # multiplier at offset t into `cycle` (which began at epoch `start`)
        raise NotImplementedError

    def shape(self, e):
        cycle, start, wgt = self._cycle(e)
        return wgt * self.curve(cycle, e - start, start)

    def _on_restart(self, cycle):
        pass

    def get_lr(self):
        e = self.last_epoch
        groups = self.optimizer.param_groups
        if e <= 0:
            return [g["initial_lr"] for g in groups]
        cycle, start, wgt = self._cycle(e)
        if e == start and cycle > 0:
            self._on_restart(cycle)
            return [g["initial_lr"] * wgt for g in groups]
        num, den = self.curve(cycle, e - start, start), self.curve(cycle, e - 1 - start, start)
        if den == 0.0:
            # the curve sat on its floor, a ratio says nothing: the reference leaves the floor by an INCREMENT on the rate
            # the group has now (lr_scheduler.py:57-61) — the first step of the unweighted curve — so a rate somebody
            # rescaled at the bottom keeps its offset
            return [g["lr"] + (b - self.floor) * num for b, g in zip(self.base_lrs, groups)]
        return self._ratio_step(groups, num, den)

    def _ratio_step(self, groups, num, den):
        return [self.floor + (g["lr"] - self.floor) * (num / den) for g in groups]


class MultiStepLR_Restart(_RestartSchedule):
    """rate(e) = initial_lr * weight(cycle) * gamma ** #{milestones m : cycle start < m <= e}."""

    def __init__(self, optimizer, milestones, restarts=None, weights=None, gamma=0.1, clear_state=False,
                 last_epoch=-1):
        self.milestones = Counter(milestones)
        self.gamma = gamma
        self.clear_state = clear_state
        self.restarts, self.restart_weights = _restart_table(restarts, weights)
        super().__init__(optimizer, last_epoch)

    def _passed(self, start, e):
        return sum(n for m, n in self.milestones.items() if start < m <= e)

    def curve(self, cycle, t, start):
        return self.gamma ** self._passed(start, start + t)

    def _on_restart(self, cycle):
        if self.clear_state:
            self.optimizer.state = defaultdict(dict)

    def _ratio_step(self, groups, num, den):
        # gamma ** (integer number of milestones falling ON this epoch): exact where num / den would round
        e = self.last_epoch
        hit = self.milestones.get(e, 0)
        return [g["lr"] * self.gamma ** hit if hit else g["lr"] for g in groups]


class CosineAnnealingLR_Restart(_RestartSchedule):
    """rate(e) = eta_min + (initial_lr * weight(cycle) - eta_min) * (1 + cos(pi * t / T_cycle)) / 2, t = e - cycle start."""

    def __init__(self, optimizer, T_period, restarts=None, weights=None, eta_min=0, last_epoch=-1):
        self.T_period = T_period
        self.T_max = self.T_period[0]
        self.eta_min = eta_min
        self.floor = eta_min
        self.restarts, self.restart_weights = _restart_table(restarts, weights)
        self.last_restart = 0
        super().__init__(optimizer, last_epoch)

    def curve(self, cycle, t, start):
        period = self.T_period[cycle]
        if (t - period) % (2 * period) == 0:        # bottom of the cosine: exactly the floor
            return 0.0
        return (1.0 + math.cos(math.pi * t / period)) / 2.0

    def _on_restart(self, cycle):       # kept as attributes because .state pickles carry them
        self.last_restart = self.last_epoch
        self.T_max = self.T_period[cycle]
