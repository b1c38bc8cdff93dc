"""Model problems as node-evaluated arrays (reference: src/model_problems.py:10-105).

Only the manufactured Taylor-Green vortex is in scope.
"""

import numpy as np

from .mesh import Function

__all__ = ["TaylorGreen", "SeparableForcing"]


class SeparableForcing:
    """f(t) = g(t) * profile; callable like the reference's ``f_rhs`` lambdas (model_problems.py:76-79)
    and additionally exposes the factorisation so that the engine never re-uploads the profile."""

    def __init__(self, profile, g):
        self.profile = profile
        self.g = g

    def scale(self, t):
        return float(self.g(t))

    def __call__(self, t):
        return self.scale(t) * self.profile


class TaylorGreen:
    """Taylor-Green vortex with manufactured time dependence (model_problems.py:38-105)."""

    def __init__(self, V_Q, V_p, forcing="exponential", kappa=0.5):
        assert forcing in ("exponential", "constant"), "Forcing must be 'constant' or 'exponential'"
        self.V_Q, self.V_p = V_Q, V_p
        self.kappa = kappa
        self.forcing = forcing
        S = lambda z: np.sin((z - 0.5) * np.pi)
        C = lambda z: np.cos((z - 0.5) * np.pi)
        # model_problems.py:56-65 (the code's p_s, not the README's: SURVEY.md C-10)
        self.Q_stationary = lambda x, y: (-C(x) * S(y), S(x) * C(y))
        self.p_stationary = lambda x, y: (S(x) ** 2 + S(y) ** 2) / 2
        self._Qs = V_Q.interpolate(self.Q_stationary)
        self._ps = V_p.interpolate(self.p_stationary)

    def initial_condition(self):
        return self.Q_stationary, self.p_stationary

    def f_rhs(self):
        """Forcing as a function of time; kappa == 0 is treated as zero forcing."""
        k = self.kappa
        if k == 0:
            return SeparableForcing(self._Qs, lambda t: 0.0)
        if self.forcing == "exponential":
            return SeparableForcing(self._Qs, lambda t: -k * np.exp(-k * t))
        return SeparableForcing(self._Qs, lambda t: -k)

    def solution(self, t, integrate_pressure=None):
        """Interpolant of the stationary fields scaled by the time factors (model_problems.py:87-105)."""
        k = self.kappa
        if self.forcing == "exponential":
            Q = np.exp(-k * t) * self._Qs
            p = np.exp(-2 * k * t) * self._ps
        else:
            Q = (1.0 - k * t) * self._Qs
            p = (1.0 - k * t) ** 2 * self._ps
        if integrate_pressure is not None:
            p = p - integrate_pressure(p)  # model_problems.py:104: no division by the volume
        return Function(self.V_Q, Q, "velocity_exact"), Function(self.V_p, p, "pressure_exact")
