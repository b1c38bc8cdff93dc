"""Running average of solver iteration counts (reference: src/auxilliary/utils.py:10-45)."""

__all__ = ["Averager"]


class Averager:
    def __init__(self):
        self.reset()

    @property
    def value(self):
        return self._average

    @property
    def n_samples(self):
        return self._n_samples

    def update(self, x):
        self._n_samples += 1
        self._average += (x - self._average) / self._n_samples

    def reset(self):
        self._n_samples = 0
        self._average = 0

    def __repr__(self):
        return f"{self.value} (averaged over {self.n_samples} samples)"
