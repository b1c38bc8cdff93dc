"""Minimal stand-ins for the Firedrake objects the reference's call sites pass around.

The reference builds ``UnitSquareMesh(nx, nx, quadrilateral=False)`` (src/driver.py:181) and hands it
to the timestepper constructors; ``driver.py`` and ``model_problems.py`` then read the function spaces
``_V_Q`` / ``_V_p`` (driver.py:311-313,332).  These classes carry exactly that information.
"""

import numpy as np

__all__ = ["UnitSquareMesh", "PeriodicSquareMesh", "TriangleMesh", "UnitDiskMesh", "FunctionSpace", "Function"]


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


class TriangleMesh:
    """Any conforming affine triangulation: vertices (nv, 2), cells (nc, 3) vertex numbers.  The engine derives edges and
    numbering from these two arrays (include/hdg_mi355x.h, hdg_create_general); per-element geometry replaces the two shared
    element shapes of the square meshes."""

    general = True
    periodic = False

    def __init__(self, vertices, cells):
        self.vertices = np.ascontiguousarray(vertices, dtype=float)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        if self.vertices.ndim != 2 or self.vertices.shape[1] != 2 or self.cells.ndim != 2 or self.cells.shape[1] != 3:
            raise ValueError("vertices (nv, 2) and cells (nc, 3) expected")
        v = self.vertices[self.cells]
        det = (v[:, 1, 0] - v[:, 0, 0]) * (v[:, 2, 1] - v[:, 0, 1]) - (v[:, 2, 0] - v[:, 0, 0]) * (v[:, 1, 1] - v[:, 0, 1])
        if np.any(det == 0.0):
            raise ValueError("degenerate triangle")
        self.volume = float(np.sum(np.abs(det)) / 2.0)
        self.nx = self.ny = 0

    def num_cells(self):
        return len(self.cells)

    def refined(self):
        """Uniform refinement: every triangle into four through its edge midpoints."""
        X = [tuple(p) for p in self.vertices]
        mid = {}

        def midpoint(i, j):
            key = (min(i, j), max(i, j))
            if key not in mid:
                mid[key] = len(X)
                X.append(tuple(0.5 * (self.vertices[i] + self.vertices[j])))
            return mid[key]

        cells = []
        for (i, j, k) in self.cells:
            a, b, c = midpoint(i, j), midpoint(j, k), midpoint(k, i)
            cells += [(i, a, c), (a, j, b), (c, b, k), (a, b, c)]
        return TriangleMesh(np.array(X), np.array(cells))


class UnitDiskMesh(TriangleMesh):
    """``UnitDiskMesh(refinement_level=...)`` of the reference's driver (src/driver.py:184-185), the mesh of the
    Kelvin-Helmholtz set-up.  Construction as remembered from Firedrake's utility mesh of that name (Firedrake cannot be run
    here: unpinned): the square [-1, 1]^2 cut into 8 triangles around the origin, `refinement_level` uniform refinements, then
    every vertex farther than 2^-(level+1) from the origin is moved radially to x * max(|x_1|, |x_2|) / |x|, which turns the
    squares concentric with the origin into circles."""

    def __init__(self, refinement_level=0):
        X = np.array([[0, 0], [1, 0], [1, 1], [0, 1], [-1, 1], [-1, 0], [-1, -1], [0, -1], [1, -1]], dtype=float)
        C = np.array([[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 4, 5], [0, 5, 6], [0, 6, 7], [0, 7, 8], [0, 8, 1]])
        m = TriangleMesh(X, C)
        for _ in range(int(refinement_level)):
            m = m.refined()
        Y = m.vertices.copy()
        r = np.linalg.norm(Y, axis=1)
        move = r > 1.0 / (1 << (int(refinement_level) + 1))
        Y[move] *= (np.max(np.abs(Y[move]), axis=1) / r[move])[:, None]
        super().__init__(Y, m.cells)
        self.refinement_level = int(refinement_level)


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
