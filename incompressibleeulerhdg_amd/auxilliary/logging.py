"""Wall-clock timers with the labels of the reference (src/auxilliary/logging.py:11-60):
``timestep``, ``bdm_projection``, ``pressure_solve``, ``tentative_velocity_solve``.

Every C-ABI call is synchronous on return, so host timers bracket device-synchronised regions.
"""

import time
from collections import defaultdict
from contextlib import ContextDecorator

import numpy as np

__all__ = ["PerformanceLog", "log_summary"]


class PerformanceLog(ContextDecorator):
    """Context manager and decorator that records elapsed seconds under a label."""

    data = defaultdict(list)

    def __init__(self, label):
        self.label = label

    def __enter__(self):
        self._t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        PerformanceLog.data[self.label].append(time.perf_counter() - self._t0)
        return False

    @classmethod
    def reset(cls):
        cls.data.clear()


def log_summary(file=None):
    """Print ncall / total / avg / std per label, sorted by total time."""
    if not PerformanceLog.data:
        return
    rows = []
    for label, t in PerformanceLog.data.items():
        t = np.asarray(t)
        rows.append((label, len(t), t.sum(), t.mean(), t.std()))
    print(f"{'timer':>32s} : {'ncall':>6s}    {'total':>10s} {'avg':>10s} {'std':>10s}", file=file)
    print(77 * "-", file=file)
    for label, n, tot, avg, std in sorted(rows, key=lambda r: -r[2]):
        print(f"{label:>32s} : {n:6d}    {tot:10.4e} {avg:10.4e} {std:10.4e}", file=file)
