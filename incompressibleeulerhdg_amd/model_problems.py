"""Model problems as node-evaluated arrays (reference: src/model_problems.py:10-105).

The manufactured Taylor-Green vortex (model_problems.py:38-105), the Kelvin-Helmholtz instability on the unit disk
(:108-131) and the double-layer shear flow on the periodic square (:134-196).
"""

import numpy as np

from .mesh import Function

__all__ = ["TaylorGreen", "KelvinHelmholtz", "DoubleLayerShearFlow", "SeparableForcing"]


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


class KelvinHelmholtz:
    """Kelvin-Helmholtz instability on the circular mesh (model_problems.py:108-131): rigid rotation (-y, x) inside
    r < r_max = 0.5, fluid at rest outside, zero pressure, zero forcing; no exact solution."""

    def __init__(self, V_Q, V_p, r_max=0.5):
        self.V_Q, self.V_p = V_Q, V_p
        self.r_max = r_max
        inside = lambda x, y: x ** 2 + y ** 2 < r_max ** 2  # conditional(x**2 + y**2 < r_max**2, (-y, x), (0, 0))
        self.Q_stationary = lambda x, y: (np.where(inside(x, y), -y, 0.0), np.where(inside(x, y), x, 0.0))
        self.p_stationary = lambda x, y: 0.0 * x

    def initial_condition(self):
        return self.Q_stationary, self.p_stationary

    def f_rhs(self):
        """Zero forcing (model_problems.py:129-131)."""
        return None

    def solution(self, t, integrate_pressure=None):
        return None


class DoubleLayerShearFlow:
    """Double layer shear flow (model_problems.py:134-196; Guzman, Shu, Sequeira, IMA J. Numer. Anal. 37 (2017)) on the
    periodic square [0, 2 pi]^2: two tanh shear layers of width rho perturbed by a vertical velocity of magnitude delta;
    the initial pressure is the 28-term Fourier series of the reference, its coefficients by scipy.integrate.quad."""

    def __init__(self, V_Q, V_p, rho=np.pi / 15, delta=0.05):
        from scipy import integrate

        self.V_Q, self.V_p = V_Q, V_p
        self.rho, self.delta = rho, delta
        self.Q_initial = lambda x, y: (np.where(y <= np.pi, np.tanh((y - np.pi / 2) / rho), np.tanh((1.5 * np.pi - y) / rho)),
                                       delta * np.sin(x))
        kmax = 28  # number of Fourier coefficients (model_problems.py:164)
        coef = []
        for k in range(kmax):
            c = integrate.quad(
                lambda z: np.where(z <= 0.0, 1 - np.tanh((np.pi + 2 * z) / (4 * np.pi * rho)) ** 2,
                                   -1 + np.tanh((np.pi - 2 * z) / (4 * np.pi * rho)) ** 2) / (np.pi ** 2 * rho),
                -np.pi, +np.pi, weight="sin", wvar=2 * k + 1, epsabs=1e-12, epsrel=1e-12)[0]
            coef.append(c / (1 + (2 * k + 1) ** 2))
        self._coef = coef

        def p_initial(x, y):
            acc = 0.0
            for k, c in enumerate(self._coef):
                acc = acc + c * np.sin((2 * k + 1) * (y - np.pi))
            return acc * delta * np.cos(x)

        self.p_initial = p_initial

    def initial_condition(self):
        return self.Q_initial, self.p_initial

    def f_rhs(self):
        """Zero forcing (model_problems.py:194-196)."""
        return None
