"""Minimal stand-ins for the Firedrake objects the reference's call sites pass around.

The reference builds ``UnitSquareMesh(nx, nx, quadrilateral=False)`` (src/driver.py:181) and hands it
to the timestepper constructors; ``driver.py`` and ``model_problems.py`` then read the function spaces
``_V_Q`` / ``_V_p`` (driver.py:311-313,332).  These classes carry exactly that information.
"""

import numpy as np

__all__ = ["UnitSquareMesh", "PeriodicSquareMesh", "FunctionSpace", "Function"]


class UnitSquareMesh:
    """nx x ny squares on the unit square, each split into two triangles."""

    def __init__(self, nx, ny=None, quadrilateral=False):
        if quadrilateral:
            raise NotImplementedError("only triangular meshes (driver.py:181)")
        self.nx = int(nx)
        self.ny = int(nx if ny is None else ny)
        self.periodic = False
        self.L = 1.0

    def num_cells(self):
        return 2 * self.nx * self.ny


class PeriodicSquareMesh(UnitSquareMesh):
    """Doubly periodic square of side L, ``PeriodicSquareMesh(nx, nx, L=2 * pi, quadrilateral=False)`` (src/driver.py:182-183):
    the same triangulation, edges on opposite sides identified."""

    def __init__(self, nx, ny=None, L=1.0, quadrilateral=False):
        super().__init__(nx, ny, quadrilateral)
        self.periodic = True
        self.L = float(L)


class _Dat:
    def __init__(self, data):
        self.data = data

    @property
    def data_ro(self):
        return self.data


class FunctionSpace:
    """A broken Lagrange space; ``coordinates`` are the physical node positions (n_nodes, 2)."""

    def __init__(self, mesh, family, degree, coordinates, value_size=1):
        self._mesh = mesh
        self.family = family
        self.degree = degree
        self.coordinates = coordinates
        self.value_size = value_size

    def mesh(self):
        return self._mesh

    def dim(self):
        return self.coordinates.shape[0] * self.value_size

    def interpolate(self, expr):
        """Evaluate ``expr(x, y)`` at the nodes (Firedrake: Function(V).interpolate(expr))."""
        x, y = self.coordinates[:, 0], self.coordinates[:, 1]
        val = expr(x, y)
        if self.value_size == 1:
            return np.ascontiguousarray(np.broadcast_to(val, x.shape), dtype=float)
        return np.ascontiguousarray(np.stack([np.broadcast_to(v, x.shape) for v in val], axis=-1), dtype=float)


class Function:
    """Array-backed function; ``.dat.data`` has the reference's layout."""

    def __init__(self, space, data, name=None):
        self._space = space
        self.dat = _Dat(np.asarray(data))
        self._name = name

    def function_space(self):
        return self._space

    def rename(self, name):
        self._name = name

    def name(self):
        return self._name
