"""Common functionality of the timesteppers (reference: src/timesteppers/common.py:15-144).

Same class surface as the reference; the bodies call the HIP engine through the ctypes C-ABI.
"""

from abc import ABC, abstractmethod

import numpy as np

from .._lib import Engine
from ..mesh import Function, FunctionSpace

__all__ = ["IncompressibleEuler"]


class IncompressibleEuler(ABC):
    """Abstract base class for timesteppers of the incompressible Euler equations.

    The reference constructor builds the 1/h_F facet field, V_BDM and the inverse DOF multiplicity
    (common.py:36-70); here those live inside the engine's operator tables (edge lengths are closed
    form on the structured mesh; the BDM averaging is built into the projection kernel).
    """

    def __init__(self, mesh, degree, dt, label=None, **engine_options):
        self._mesh = mesh
        self.degree = degree
        self._dt = dt
        self._label = label
        self._engine_options = engine_options
        self._engine = None
        # common.py:72-73
        self.domain_volume = float(mesh.volume) if getattr(mesh, "general", False) else float(getattr(mesh, "L", 1.0)) ** 2

    # -- engine and function spaces ------------------------------------------------------------
    def _create_engine(self, **kw):
        if getattr(self._mesh, "general", False):
            # general affine triangulation: per-element geometry (hdg_create_general)
            opts = dict(vertices=self._mesh.vertices, cells=self._mesh.cells, degree=self.degree, dt=self._dt)
        else:
            opts = dict(nx=self._mesh.nx, ny=self._mesh.ny, degree=self.degree, dt=self._dt,
                        periodic=getattr(self._mesh, "periodic", False), length=getattr(self._mesh, "L", 1.0))
        opts.update(kw)
        opts.update(self._engine_options)
        self._engine = Engine(**opts)
        xq, xp = self._engine.node_coordinates()
        k = self.degree
        self._V_Q = FunctionSpace(self._mesh, "DG", k + 1, xq, value_size=2)
        self._V_p = FunctionSpace(self._mesh, "DG", k, xp)
        self._V_q = self._V_p  # tracer space DG_k (hdg_imex.py:68)
        for V in (self._V_Q, self._V_p):
            V._engine = self._engine  # device operations on a Function (vorticity callback)
        self._V_trace = ("DGT", k, self._engine.n_edges * self._engine.n_l)
        self._V = (self._V_Q, self._V_p, self._V_trace)
        return self._engine

    def _as_nodal_velocity(self, Q):
        """Accept a callable (x, y) -> (ux, uy) [the reference passes UFL expressions], a Function or
        an array."""
        if callable(Q):
            return self._V_Q.interpolate(Q)
        if isinstance(Q, Function):
            return np.asarray(Q.dat.data, dtype=float)
        return np.asarray(Q, dtype=float)

    def _as_nodal_pressure(self, p):
        if callable(p):
            return self._V_p.interpolate(p)
        if isinstance(p, Function):
            return np.asarray(p.dat.data, dtype=float)
        return np.asarray(p, dtype=float)

    # -- reference API -------------------------------------------------------------------------
    def get_timesteps(self, t_final, warmup):
        """Number of timesteps (common.py:75-84)."""
        nt = 1 if warmup else int(np.round(t_final / self._dt))
        assert warmup or (abs(nt * self._dt - t_final) < 1.0e-12)
        return nt

    @property
    def label(self):
        return self._label

    def project_bdm(self, Q):
        """Project a velocity from the DG space to the BDM space (common.py:91-108).

        Returns Q* as a Function on the broken space [P_{k+1}]^2 (same function, continuous normals,
        zero normal component on the boundary)."""
        out = self._engine.project_bdm_nodal(self._as_nodal_velocity(Q))
        return Function(self._V_Q, out, "Q_star")

    def _init_tracer(self, q_initial):
        """q_initial (expression / array / None, driver.py:340-344) -> the engine's tracer state; returns whether a
        tracer is advected."""
        if q_initial is None or q_initial is False:
            self._engine.set_tracer(None)
            self.q_tracer = None
            return False
        self._engine.set_tracer(self._as_nodal_pressure(q_initial))
        self.q_tracer = Function(self._V_q, self._engine.get_tracer(), "tracer")
        return True

    def _tracer_function(self):
        self.q_tracer = Function(self._V_q, self._engine.get_tracer(), "tracer")
        return self.q_tracer

    @abstractmethod
    def solve(self, Q_initial, p_initial, q_initial, f_rhs, T_final, warmup=False):
        """Propagate the solution to T_final; returns the final velocity and pressure (common.py:131-144)."""
