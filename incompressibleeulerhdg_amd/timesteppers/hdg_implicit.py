"""First-order implicit HDG timestepper (reference: src/timesteppers/hdg_implicit.py:10-197)."""

from .. import _lib
from ..auxilliary.logging import PerformanceLog
from ..auxilliary.utils import Averager
from ..mesh import Function
from .common import IncompressibleEuler

__all__ = ["IncompressibleEulerHDGImplicit"]


class IncompressibleEulerHDGImplicit(IncompressibleEuler):
    """First order in time; Chorin's projection method (hdg_implicit.py:101-150).

    ``n_richardson`` is accepted and ignored so that the reference driver's call shape
    (driver.py:220-228) works.  ``use_projection_method=False`` selects the
    monolithic (u, phi, lambda) solve of hdg_implicit.py:151-186.
    """

    def __init__(self, mesh, degree, dt, flux="upwind", use_projection_method=True, callbacks=None,
                 n_richardson=None, **engine_options):
        # the fully implicit step carries the whole dt as implicit weight (four times SSP2(3,3,2)'s a_ii dt): the spectrum
        # of the tentative-velocity operator is too wide for the Chebyshev iteration to pay at k >= 2 (k = 1 / 2 / 3 at
        # 256^2 / 512^2 / 512^2, ms per step with tent_solver 0 / 1: 5.17 / 4.31, 10.80 / 11.17, 15.64 / 16.08)
        engine_options.setdefault("tent_solver", 1 if degree <= 1 else 0)
        super().__init__(mesh, degree, dt, label="HDG Implicit", **engine_options)
        self.flux = flux
        assert self.flux in ["upwind", "centered"]
        self.use_projection_method = use_projection_method
        self.callbacks = [] if callbacks is None else callbacks
        self.alpha = 1  # hdg_implicit.py:41
        self.tau = 1  # hdg_implicit.py:43
        self.niter_tentative = Averager()
        self.niter_pressure = Averager()
        self._create_engine(flux=flux, use_projection_method=use_projection_method, n_richardson=1, tau=self.tau,
                            alpha_penalty=self.alpha, nstages=1, a_expl=[[0]], a_impl=[[1]], b_expl=[1],
                            b_impl=[1], c_expl=[0])

    def solve(self, Q_initial, p_initial, q_initial, f_rhs, T_final, warmup=False):
        eng = self._engine
        tracer = self._init_tracer(q_initial)  # hdg_implicit.py:72-78
        nt = self.get_timesteps(T_final, warmup)
        eng.set_state(self._as_nodal_velocity(Q_initial), self._as_nodal_pressure(p_initial))
        profile = None
        for callback in self.callbacks:
            callback.reset()
            Q, p, _ = eng.get_field(_lib.HDG_STATE_CURRENT, lam=False)
            callback(Function(self._V_Q, Q), Function(self._V_p, p), 0, q_tracer=self.q_tracer)
        for k in range(nt):
            with PerformanceLog("timestep"):
                t = k * self._dt  # forcing at the START of the step (hdg_implicit.py:100)
                if f_rhs is None or (isinstance(f_rhs, (int, float)) and f_rhs == 0):
                    eng.set_forcing_scale(0, 0.0)
                elif hasattr(f_rhs, "profile"):
                    if profile is not f_rhs.profile:
                        eng.set_forcing_profile(f_rhs.profile)
                        profile = f_rhs.profile
                    eng.set_forcing_scale(0, f_rhs.scale(t))
                else:
                    eng.set_forcing_nodal(0, self._as_nodal_velocity(f_rhs(t)))
                it_t, it_p = eng.implicit_step()
                self.niter_tentative.update(it_t)
                self.niter_pressure.update(it_p)
            if self.callbacks:
                Q, p, _ = eng.get_field(_lib.HDG_STATE_CURRENT, lam=False)
                qt = self._tracer_function() if tracer else None
                for callback in self.callbacks:
                    callback(Function(self._V_Q, Q), Function(self._V_p, p), (k + 1) * self._dt, q_tracer=qt)
        Q, p, _ = eng.get_field(_lib.HDG_STATE_CURRENT, lam=False)
        if tracer:
            self._tracer_function()
        return Function(self._V_Q, Q, "velocity"), Function(self._V_p, p, "pressure")
