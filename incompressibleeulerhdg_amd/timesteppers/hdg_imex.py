"""HDG IMEX timesteppers (reference: src/timesteppers/hdg_imex.py:22-1038).

Class names, constructor arguments, method names and the tableau numbers are those of the reference;
the numerics run on the MI355X through the C-ABI (include/hdg_mi355x.h).
"""

from abc import abstractmethod

import numpy as np

from .. import _lib
from ..auxilliary.logging import PerformanceLog
from ..auxilliary.utils import Averager
from ..mesh import Function
from .common import IncompressibleEuler

__all__ = [
    "IncompressibleEulerHDGIMEX",
    "IncompressibleEulerHDGIMEXImplicit",
    "IncompressibleEulerHDGIMEXARS2_232",
    "IncompressibleEulerHDGIMEXARS3_443",
    "IncompressibleEulerHDGIMEXSSP2_332",
    "IncompressibleEulerHDGIMEXSSP3_433",
]


def _key_to_int(key):
    """'stage_i' / 'final_stage' / 'pressure_reconstruction' (hdg_imex.py:258-263) -> C-ABI key."""
    if key == "final_stage":
        return _lib.HDG_KEY_FINAL_STAGE
    if key == "pressure_reconstruction":
        return _lib.HDG_KEY_PRESSURE_RECONSTRUCTION
    if key.startswith("stage_"):
        return int(key[len("stage_"):])
    raise KeyError(key)


class IncompressibleEulerHDGIMEX(IncompressibleEuler):
    """Abstract base class of the IMEX timesteppers (hdg_imex.py:22-660)."""

    def __init__(self, mesh, degree, dt, flux="upwind", use_projection_method=True, n_richardson=2,
                 label=None, callbacks=None, **engine_options):
        super().__init__(mesh, degree, dt, label, **engine_options)
        self.flux = flux
        self.use_projection_method = use_projection_method
        assert self.flux in ["upwind", "centered"]
        self.alpha_penalty = 1  # hdg_imex.py:56
        self.tau = 1  # hdg_imex.py:58
        self.n_richardson = n_richardson
        self.callbacks = [] if callbacks is None else callbacks
        self.niter_tentative = Averager()
        self.niter_pressure = Averager()
        self.niter_final_pressure = Averager()
        self.niter_pressure_reconstruction = Averager()
        self._create_engine(
            flux=flux, use_projection_method=use_projection_method, n_richardson=n_richardson,
            tau=self.tau, alpha_penalty=self.alpha_penalty, nstages=self.nstages,
            a_expl=self._a_expl, a_impl=self._a_impl, b_expl=self._b_expl, b_impl=self._b_impl,
            c_expl=self._c_expl,
        )

    # -- tableau (hdg_imex.py:283-311) -----------------------------------------------------------
    @property
    @abstractmethod
    def nstages(self):
        """number of stages s"""

    @property
    @abstractmethod
    def _a_expl(self):
        """s x s matrix with explicit coefficients"""

    @property
    @abstractmethod
    def _a_impl(self):
        """s x s matrix with implicit coefficients"""

    @property
    @abstractmethod
    def _b_expl(self):
        """explicit final-stage weights"""

    @property
    @abstractmethod
    def _b_impl(self):
        """implicit final-stage weights"""

    @property
    @abstractmethod
    def _c_expl(self):
        """fractional times of the explicit evaluations"""

    # -- solves (hdg_imex.py:257-281) ------------------------------------------------------------
    @PerformanceLog("pressure_solve")
    def pressure_solve(self, key):
        """Solve the pressure correction equation; returns the condensed Krylov iteration count."""
        return self._engine.pressure_solve(_key_to_int(key))

    @PerformanceLog("tentative_velocity_solve")
    def tentative_velocity_solve(self, key):
        """Compute the tentative velocity; returns the Krylov iteration count."""
        return self._engine.tentative_solve(_key_to_int(key))

    def _reconstruct_trace(self, state=None):
        """hdg_imex.py:450-469 (always acts on _current_state, its only use in the reference)."""
        self._engine.reconstruct_trace()

    def _shift_pressure(self, which):
        """hdg_imex.py:471-478; `which` selects _update / _stage_state[i] / _current_state."""
        self._engine.shift_pressure(which)

    # -- forcing -------------------------------------------------------------------------------
    def _set_forcing(self, slot, f_rhs, t):
        if f_rhs is None or (isinstance(f_rhs, (int, float)) and f_rhs == 0):
            self._engine.set_forcing_scale(slot, 0.0)
        elif hasattr(f_rhs, "profile") and hasattr(f_rhs, "scale"):
            if self._forcing_profile is not f_rhs.profile:
                self._engine.set_forcing_profile(f_rhs.profile)
                self._forcing_profile = f_rhs.profile
            self._engine.set_forcing_scale(slot, f_rhs.scale(t))
        else:
            self._engine.set_forcing_nodal(slot, self._as_nodal_velocity(f_rhs(t)))

    # -- time loop (hdg_imex.py:505-660) ----------------------------------------------------------
    def solve(self, Q_initial, p_initial, q_initial, f_rhs, T_final, warmup=False, fused=False):
        """Propagate the solution to T_final with nt timesteps; returns (Q, p).

        ``fused=True`` runs each step as one device-resident ``hdg_step`` call instead of the
        per-solve calls that mirror the reference's loop (identical results, no per-solve timers).
        """
        eng = self._engine
        tracer = self._init_tracer(q_initial)  # hdg_imex.py:523-529
        s = self.nstages
        nt = self.get_timesteps(T_final, warmup)
        self._forcing_profile = None
        eng.set_state(self._as_nodal_velocity(Q_initial), self._as_nodal_pressure(p_initial))
        self._reconstruct_trace()
        for a in (self.niter_tentative, self.niter_pressure, self.niter_final_pressure,
                  self.niter_pressure_reconstruction):
            a.reset()
        eng.iteration_stats(reset=True)
        eng.timers(reset=True)
        for callback in self.callbacks:
            callback.reset()
            Q, p, _ = eng.get_field(_lib.HDG_STATE_CURRENT, lam=False)
            callback(Function(self._V_Q, Q, "Q"), Function(self._V_p, p, "p"), 0, q_tracer=self.q_tracer)
        for k in range(nt):
            with PerformanceLog("timestep"):
                tn = k * self._dt
                for i in range(s):
                    self._set_forcing(i, f_rhs, tn + self._c_expl[i] * self._dt)
                self._set_forcing(s, f_rhs, tn + self._dt)  # _b_new (hdg_imex.py:629)
                if fused:
                    eng.step()
                else:
                    eng.begin_step()
                    if tracer:
                        eng.tracer_begin_step()  # self._q[0].assign(q_tracer), hdg_imex.py:560
                    for i in range(1, s):
                        with PerformanceLog("bdm_projection"):
                            eng.project_bdm(i - 1, i - 1)
                        if self.use_projection_method:
                            for _ in range(self.n_richardson):
                                its = self.tentative_velocity_solve(f"stage_{i:d}")
                                self.niter_tentative.update(its)
                                its = self.pressure_solve(f"stage_{i:d}")
                                self.niter_pressure.update(its)
                                self._shift_pressure(_lib.HDG_STATE_UPDATE)
                                eng.stage_update(i)
                        else:
                            with PerformanceLog("unsplit_solve"):
                                its = eng.unsplit_solve(i)  # hdg_imex.py:600-620
                            self.niter_tentative.update(its)
                        self._shift_pressure(i)
                        if tracer:
                            eng.tracer_stage(i)  # hdg_imex.py:622-623
                    its = self.pressure_solve("final_stage")
                    self.niter_final_pressure.update(its)
                    its = self.pressure_solve("pressure_reconstruction")
                    self.niter_pressure_reconstruction.update(its)
                    eng.finish_step()
                    if tracer:
                        eng.tracer_finish_step()  # hdg_imex.py:638-639
            if self.callbacks:
                Q, p, _ = eng.get_field(_lib.HDG_STATE_CURRENT, lam=False)
                qt = self._tracer_function() if tracer else None
                for callback in self.callbacks:
                    callback(Function(self._V_Q, Q, "Q"), Function(self._V_p, p, "p"), tn + self._dt, q_tracer=qt)
        if fused:
            # per-solve breakdown of the fused steps from the engine's device-side timers (same labels as the
            # host timers of the per-solve path; "timestep" is already timed on the host)
            for label, (n, tot, sq) in eng.timers(reset=True).items():
                if label != "timestep" and n:
                    PerformanceLog.add_aggregate(label, n, tot, sq)
            sums, cnt = eng.iteration_stats()
            for a, sm, c in zip((self.niter_tentative, self.niter_pressure, self.niter_final_pressure,
                                 self.niter_pressure_reconstruction), sums, cnt):
                a._n_samples, a._average = int(c), (sm / c if c else 0)
        print("average number of solver iterations")
        print(40 * "-")
        print(f"  tentative velocity its      : {self.niter_tentative.value:8.2f}")
        if self.use_projection_method:
            print(f"  pressure its                : {self.niter_pressure.value:8.2f}")
            print(f"  final pressure its          : {self.niter_final_pressure.value:8.2f}")
        print(f"  pressure reconstruction its : {self.niter_pressure_reconstruction.value:8.2f}")
        print()
        Q, p, _ = eng.get_field(_lib.HDG_STATE_CURRENT, lam=False)
        if tracer:
            self._tracer_function()  # the final tracer field: self.q_tracer (the reference returns (Q, p) only)
        return Function(self._V_Q, Q, "Q"), Function(self._V_p, p, "p")


#######################################################################################
#       S P E C I F I C     I M E X     T I M E S T E P P E R S                       #
#   tableau numbers are data of the reference (hdg_imex.py:702-1038), quirks included   #
#######################################################################################


def _make(label_, s_, a_expl_, a_impl_, b_expl_, b_impl_, c_expl_, doc):
    class _T(IncompressibleEulerHDGIMEX):
        def __init__(self, mesh, degree, dt, flux="upwind", use_projection_method=True, n_richardson=2,
                     callbacks=None, **engine_options):
            super().__init__(mesh, degree, dt, flux, use_projection_method, n_richardson,
                             label=label_, callbacks=callbacks, **engine_options)

        nstages = property(lambda self: s_)
        _a_expl = property(lambda self: np.asarray(a_expl_, dtype=float))
        _a_impl = property(lambda self: np.asarray(a_impl_, dtype=float))
        _b_expl = property(lambda self: np.asarray(b_expl_, dtype=float))
        _b_impl = property(lambda self: np.asarray(b_impl_, dtype=float))
        _c_expl = property(lambda self: np.asarray(c_expl_, dtype=float))

    _T.__doc__ = doc
    return _T


_g = 1 - 1 / np.sqrt(2)
_d = -2 / 3 * np.sqrt(2)
_al, _be, _et = 0.24169426078821, 0.06042356519705, 0.12915286960590
_de = 1 / 2 - _al - _be - _et

IncompressibleEulerHDGIMEXImplicit = _make(
    "HDG IMEX Implicit", 2, [[0, 0], [1, 0]], [[0, 0], [0, 1]], [1, 0], [0, 1], [0, 1],
    "IMEX implementation of the first order implicit method (hdg_imex.py:668-729)")
IncompressibleEulerHDGIMEXARS2_232 = _make(
    "HDG IMEX ARS2(2,3,2)", 3,
    [[0, 0, 0], [_g, 0, 0], [_d, 1 - _d, 0]], [[0, 0, 0], [0, _g, 0], [0, 1 - _g, _g]],
    [0, 1 - _g, _g], [0, 1 - _g, _g], [0, _g, 1],
    "IMEX ARS2(2,3,2) timestepper (hdg_imex.py:732-799)")
IncompressibleEulerHDGIMEXARS3_443 = _make(
    "HDG IMEX ARS3(4,4,3)", 5,
    [[0, 0, 0, 0, 0], [1 / 2, 0, 0, 0, 0], [11 / 18, 1 / 18, 0, 0, 0], [5 / 6, -5 / 6, 1 / 2, 0, 0],
     [1 / 4, 7 / 4, 3 / 4, -7 / 4, 0]],
    [[0, 0, 0, 0, 0], [0, 1 / 2, 0, 0, 0], [0, 1 / 6, 1 / 2, 0, 0], [0, -1 / 2, 1 / 2, 1 / 2, 0],
     [0, 3 / 2, -3 / 2, 1 / 2, 1 / 2]],
    [1 / 4, 7 / 4, 3 / 4, -7 / 4, 0], [0, 3 / 2, -3, 2, 1 / 2, 1 / 2], [0, 1 / 2, 2 / 3, 1 / 2, 1],
    "IMEX ARS3(4,4,3) timestepper (hdg_imex.py:802-879); b_impl has 6 entries as written")
IncompressibleEulerHDGIMEXSSP2_332 = _make(
    "HDG IMEX SSP2(3,3,2)", 3,
    [[0, 0, 0], [1 / 2, 0, 0], [1 / 2, 1 / 2, 0]], [[1 / 4, 0, 0], [0, 1 / 4, 0], [1 / 3, 1 / 3, 1 / 3]],
    [1 / 3, 1 / 3, 1 / 3], [1 / 3, 1 / 3, 1 / 3], [0, 1, 1 / 2],
    "IMEX SSP2(3,3,2) timestepper (hdg_imex.py:882-949); c_expl = [0, 1, 1/2] as written")
IncompressibleEulerHDGIMEXSSP3_433 = _make(
    "HDG IMEX SSP3(4,3,3)", 4,
    [[0, 0, 0, 0], [0, 0, 0, 0], [0, 1, 0, 0], [0, 1 / 4, 1 / 4, 0]],
    [[_al, 0, 0, 0], [-_al, _al, 0, 0], [0, 1 - _al, _al, 0], [_be, _et, _de, _al]],
    [0, 1 / 6, 1 / 6, 2 / 3], [0, 1 / 6, 1 / 6, 2 / 3], [0, 0, 1, 1 / 2],
    "IMEX SSP3(4,3,3) timestepper (hdg_imex.py:952-1038)")
for _n, _c in list(globals().items()):
    if _n.startswith("IncompressibleEulerHDGIMEX") and isinstance(_c, type) and _c is not IncompressibleEulerHDGIMEX:
        _c.__name__ = _c.__qualname__ = _n
