"""Host-side utilities of the trainer.  `Metric`, `MetricDict` and `TrackedTime` are pickled inside `.training`
checkpoints by class path + attribute dict, so they live at `ssdn.utils.utils` and keep the
attribute names of /root/reference/ssdn/ssdn/utils/utils.py:72-166."""
import math
import os
import time
from collections import OrderedDict
from contextlib import contextmanager

import torch


def compute_ramped_lrate(i, iteration_count, ramp_up_fraction, ramp_down_fraction, learning_rate):
    """Cosine ramp-up over the first `ramp_up_fraction` of the run, squared-cosine ramp-down over the last
    `ramp_down_fraction` (utils/utils.py:18-37).  NOTE the trainer passes the two configured fractions swapped
    (train.py:276-282) -- kept, see ssdn.train.DenoiserTrainer.learning_rate."""
    lr = learning_rate
    if ramp_up_fraction > 0.0 and i <= iteration_count * ramp_up_fraction:
        t = (i / ramp_up_fraction) / iteration_count
        lr = lr * (0.5 - math.cos(t * math.pi) / 2)
    if ramp_down_fraction > 0.0:
        start = iteration_count * (1 - ramp_down_fraction)
        if i >= start:
            t = ((i - start) / ramp_down_fraction) / iteration_count
            lr = lr * (0.5 + math.cos(t * math.pi) / 2) ** 2
    return lr


@contextmanager
def cd(newdir: str):
    prev = os.getcwd()
    os.chdir(os.path.expanduser(newdir))
    try:
        yield
    finally:
        os.chdir(prev)


class TrackedTime:
    """Accumulates wall-clock time between successive update() calls."""

    def __init__(self):
        self.total = 0
        self.last_time = None

    def update(self):
        now = time.time()
        if self.last_time is not None:
            self.total += now - self.last_time
        self.last_time = now

    def forget(self):
        self.last_time = None


def seconds_to_dhms(seconds: float, trim: bool = True) -> str:
    units = ((seconds // 86400, "d"), (seconds // 3600 % 24, "h"), (seconds // 60 % 60, "m"), (seconds % 60, "s"))
    out = ""
    for value, suffix in units:
        if trim and value < 1:
            continue
        trim = False
        out += "{:02}{}".format(int(value), suffix)
    return out


class Metric:
    """Running mean over batches: `total` holds the sum over samples of the per-sample value, `n` the sample count."""

    def __init__(self, batched: bool = True, collapse: bool = True):
        self.reset()
        self.batched = batched
        self.collapse = collapse

    def add(self, value: torch.Tensor):
        count = value.shape[0] if self.batched else 1
        if self.collapse:
            dims = list(range(1 if self.batched else 0, value.dim()))
            if dims:
                value = value.mean(dim=dims)
        if self.batched:
            value = value.sum(dim=0)
        self.total = value if self.total is None else self.total + value
        self.n += count

    def __add__(self, value):
        self.add(value)
        return self

    def accumulated(self, reset: bool = False):
        if self.n == 0:
            return None
        acc = self.total / self.n
        if reset:
            self.reset()
        return acc

    def reset(self):
        self.total = None
        self.n = 0

    def empty(self) -> bool:
        return self.n == 0


class MetricDict(OrderedDict):
    def __missing__(self, key):
        value = self[key] = Metric()
        return value


def separator(cols: int = 100) -> str:
    return "#" * cols


def list_constants(clazz, private: bool = False):
    """Values of the UPPER_CASE attributes of a class / Enum, in name order (utils/utils.py:40-54: the CLI's `choices`)."""
    import re
    pat = re.compile(r"^{}[A-Z0-9_]*$".format("" if private else "[A-Z]"))
    return [clazz.__dict__[n] if n in clazz.__dict__ else getattr(clazz, n) for n in dir(clazz) if pat.match(n)]
