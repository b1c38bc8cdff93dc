"""Callbacks invoked after every timestep (reference: src/auxilliary/callbacks.py:11-85).

``AnimationCallback(filename)`` writes velocity, pressure, the CG_{k+1} vorticity and (if present) the tracer of every
time level into one VTK collection.  The vorticity projection of ``vorticity_solver`` (callbacks.py:43-69) runs on the
device (``hdg_vorticity``: weak curl of the broken velocity, continuous mass matrix solved by Jacobi-preconditioned CG).
"""
from abc import ABC, abstractmethod

from ..mesh import Function, FunctionSpace
from ..output import VTKFile

__all__ = ["Callback", "AnimationCallback"]


class Callback(ABC):
    """Abstract base class (callbacks.py:11-27)."""

    @abstractmethod
    def __call__(self, Q, p, t, q_tracer=None):
        """Invoke the callback for velocity / pressure (and tracer) fields at time t."""

    @abstractmethod
    def reset(self):
        """Reset callback"""


class AnimationCallback(Callback):
    """Save fields to disk (callbacks.py:30-85)."""

    def __init__(self, filename):
        self.filename = filename
        self._V_vort = None
        self.reset()

    def reset(self):
        """Re-open file (callbacks.py:39-41)."""
        self.outfile = VTKFile(self.filename, mode="w")

    def vorticity(self, Q):
        """omega in CG_{k+1} with (tau, omega) = -(eps : grad tau x Q) dx + tau eps : (n x Q) ds (callbacks.py:43-69),
        returned on the node set of the velocity space so that it can be written next to the broken fields."""
        V_Q = Q.function_space()
        eng = V_Q._engine
        if self._V_vort is None or self._V_vort.mesh() is not V_Q.mesh():
            self._V_vort = FunctionSpace(V_Q.mesh(), "CG", V_Q.degree, V_Q.coordinates)
        omega_cg = eng.vorticity(Q.dat.data)
        return Function(self._V_vort, eng.cg_to_broken(omega_cg), "vorticity")

    def __call__(self, Q, p, t, q_tracer=None):
        fields = [Q, p, self.vorticity(Q)]
        if q_tracer is not None:
            fields.append(q_tracer)
        self.outfile.write(*fields, time=t)
