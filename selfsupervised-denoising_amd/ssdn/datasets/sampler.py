"""Sampling order of the training / test loaders (drop-in for /root/reference/ssdn/ssdn/datasets/sampler.py:13-111).
`SamplingOrder.state_dict()` ({"order", "index"}) is stored in `.training` checkpoints under "train_order_iter"."""
from typing import Dict, Iterator, List, Optional

import torch
from torch.utils.data import Sampler


class SamplingOrder:
    """A fixed list of dataset indexes with a cursor; iterating consumes it."""

    def __init__(self, order: List[int], index: int = 0):
        self.order, self.index = order, index

    def __iter__(self):
        return self

    def __len__(self) -> int:
        return len(self.order)

    def __next__(self) -> int:
        if self.index >= len(self.order):
            raise StopIteration()
        v = self.order[self.index]
        self.index += 1
        return v

    def state_dict(self) -> Dict:
        return {"order": self.order, "index": self.index}

    @staticmethod
    def from_state_dict(state_dict: Dict) -> "SamplingOrder":
        return SamplingOrder(state_dict["order"], state_dict["index"])


class FixedLengthSampler(Sampler):
    """`num_samples` indexes: the dataset is looped as often as needed, every pass a fresh permutation when shuffled, so no
    sample is used more than once more than any other."""

    def __init__(self, data_source, num_samples: Optional[int] = None, shuffled: bool = False):
        self.data_source, self._num_samples, self.shuffled = data_source, num_samples, shuffled
        self._next_iter = None
        self._last_iter = None

    @property
    def num_samples(self) -> int:
        return len(self.data_source) if self._num_samples is None else self._num_samples

    def sampler(self) -> Iterator[int]:
        n, left = len(self.data_source), self.num_samples
        if self.shuffled:
            while left > 0:
                take = min(left, n)
                for idx in torch.randperm(n)[:take]:
                    yield int(idx)
                left -= take
        else:
            for i in range(left):
                yield i % n

    def __iter__(self):
        if self._next_iter is None:
            self._last_iter = SamplingOrder(list(self.sampler()))
            return self._last_iter
        return self._next_iter

    def __len__(self) -> int:
        return self.num_samples

    def for_next_iter(self, iter_order: SamplingOrder):
        self._next_iter = iter_order
        self._last_iter = iter_order

    def last_iter(self) -> SamplingOrder:
        return self._last_iter
