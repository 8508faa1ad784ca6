"""One-cycle learning-rate schedule -- same API as the reference's utils/practices.py:6-41."""
import math


def annealing_linear(start, end, pct):
    return start + pct * (end - start)


def annealing_cos(start, end, pct):
    "Cosine anneal from `start` to `end` as pct goes from 0.0 to 1.0."
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


class OneCycleScheduler(object):
    """(0, pct_start): linear warm-up lr_max/div_factor -> lr_max; (pct_start, 1): cosine decay to
    lr_max/div_factor/1e4."""

    def __init__(self, lr_max, div_factor=25., pct_start=0.3):
        self.lr_max = lr_max
        self.div_factor = div_factor
        self.pct_start = pct_start
        self.lr_low = self.lr_max / self.div_factor

    def step(self, pct):
        if pct <= self.pct_start:
            return annealing_linear(self.lr_low, self.lr_max, pct / self.pct_start)
        return annealing_cos(self.lr_max, self.lr_low / 1e4,
                             (pct - self.pct_start) / (1 - self.pct_start))


def adjust_learning_rate(optimizer, lr):
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr
    return lr
