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
    # aggregated samples (ncall, total, sum of squares) reported by the engine's device-side section timers when a
    # whole step runs as one fused call (hdg_get_timers): same labels, merged by log_summary
    aggregates = defaultdict(lambda: [0, 0.0, 0.0])

    def __init__(self, label):
        self.label = label

    def __enter__(self):
        self._t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        PerformanceLog.data[self.label].append(time.perf_counter() - self._t0)
        return False

    @classmethod
    def add_aggregate(cls, label, ncall, total, sumsq):
        a = cls.aggregates[label]
        a[0] += int(ncall)
        a[1] += float(total)
        a[2] += float(sumsq)

    @classmethod
    def reset(cls):
        cls.data.clear()
        cls.aggregates.clear()


def log_summary(file=None):
    """Print ncall / total / avg / std per label, sorted by total time."""
    if not PerformanceLog.data and not PerformanceLog.aggregates:
        return
    rows = []
    for label in set(PerformanceLog.data) | set(PerformanceLog.aggregates):
        t = np.asarray(PerformanceLog.data.get(label, []), dtype=float)
        an, atot, asq = PerformanceLog.aggregates.get(label, (0, 0.0, 0.0))
        n, tot, sq = len(t) + an, t.sum() + atot, (t * t).sum() + asq
        if n == 0:
            continue
        avg = tot / n
        rows.append((label, n, tot, avg, np.sqrt(max(sq / n - avg * avg, 0.0))))
    print(f"{'timer':>32s} : {'ncall':>6s}    {'total':>10s} {'avg':>10s} {'std':>10s}", file=file)
    print(77 * "-", file=file)
    for label, n, tot, avg, std in sorted(rows, key=lambda r: -r[2]):
        print(f"{label:>32s} : {n:6d}    {tot:10.4e} {avg:10.4e} {std:10.4e}", file=file)
