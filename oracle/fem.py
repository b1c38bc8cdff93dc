"""TEST INFRASTRUCTURE ONLY -- finite-element machinery for the CPU oracle.

This file is part of ``oracle/``: a CPU restatement (numpy/scipy) of the arithmetic that the
reference delegates to Firedrake (UFL -> TSFC -> PyOP2, FIAT elements).  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; the product
(``incompressibleeulerhdg_amd``) never imports it.

PARITY UNPINNED: Firedrake is not importable in the build container and the reference ships no
tests, fixtures or golden vectors.  The conventions fixed here (node sets,
mesh diagonal, numbering) are the build's own documented choice.

Conventions
-----------
* Mesh (reference: ``UnitSquareMesh(nx, nx, quadrilateral=False)``, src/driver.py:181):
  nx x nx squares of side h = 1/nx, each split by the diagonal from (x_{i+1}, y_j) to
  (x_i, y_{j+1}) into a lower-left triangle ``L`` with vertices (i,j),(i+1,j),(i,j+1) and an
  upper-right triangle ``U`` with vertices (i+1,j+1),(i,j+1),(i+1,j).
  Cell number c = 2*(j*nx + i) + s, s = 0 (L) or 1 (U).
* Edges: horizontal H(i,j), i<nx, j<=nx, number j*nx+i, from (x_i,y_j) to (x_{i+1},y_j);
  vertical V(i,j), i<=nx, j<nx, number N_H + j*(nx+1)+i, from (x_i,y_j) to (x_i,y_{j+1});
  diagonal D(i,j), number N_H+N_V + j*nx+i, from (x_{i+1},y_j) to (x_i,y_{j+1}).
* Nodal bases: Lagrange polynomials on the "recursive GLL" node family (identical to the
  equispaced lattice up to degree 2); lattice order ``for b in 0..n: for a in 0..n-b`` with the
  node at reference point (xi, eta) ~ (a, b)/n.  Trace (DGT_k) nodes: the k+1 GLL points along the
  edge direction given above.
* Array layout at the product boundary (what ``Function.dat.data`` looks like in the reference):
  velocity (N_c*n_u, 2) cell-major / node / component fastest, pressure (N_c*n_p,),
  trace (N_e*n_l,).
"""

import numpy as np
from numpy.polynomial.legendre import leggauss
from scipy.special import roots_jacobi

__all__ = [
    "gll_points",
    "lattice",
    "triangle_nodes",
    "edge_nodes",
    "gauss_legendre_01",
    "triangle_quadrature",
    "PolySpace2D",
    "PolySpace1D",
    "Mesh",
    "TriMesh",
    "unit_disk_mesh",
]


# --------------------------------------------------------------------------------------
# node sets
# --------------------------------------------------------------------------------------
def gll_points(n):
    """The n+1 Gauss-Lobatto-Legendre points on [0, 1] (n >= 1); midpoint for n == 0."""
    if n == 0:
        return np.array([0.5])
    if n == 1:
        return np.array([0.0, 1.0])
    # interior GLL points are the roots of P_n' = roots of Jacobi P^{(1,1)}_{n-1}
    x, _ = roots_jacobi(n - 1, 1.0, 1.0)
    return np.concatenate([[0.0], 0.5 * (np.sort(x) + 1.0), [1.0]])


def lattice(n):
    """Multi-indices (a, b) of the degree-n triangle lattice in the documented order."""
    return [(a, b) for b in range(n + 1) for a in range(n + 1 - b)]


def _recursive_bary(alpha, family):
    """Recursive node construction (barycentric coordinates of lattice index alpha).

    The barycentric coordinate vector of a node is the weighted average of the nodes of the
    same construction on each facet-simplex, with weights taken from the 1D family.
    """
    d = len(alpha)
    n = sum(alpha)
    if d == 1:
        return np.array([1.0])
    xn = family(n)
    if d == 2:
        return np.array([xn[alpha[0]], xn[alpha[1]]])
    b = np.zeros(d)
    wsum = 0.0
    for i in range(d):
        rest = alpha[:i] + alpha[i + 1 :]
        w = xn[n - alpha[i]]
        br = _recursive_bary(rest, family)
        full = np.insert(br, i, 0.0)
        b += w * full
        wsum += w
    return b / wsum


def triangle_nodes(n, variant="gll"):
    """Reference coordinates (xi, eta) of the degree-n Lagrange nodes, lattice order."""
    if n == 0:
        return np.array([[1.0 / 3.0, 1.0 / 3.0]])
    pts = []
    for a, b in lattice(n):
        if variant == "equispaced":
            pts.append((a / n, b / n))
        elif variant == "gll":
            bary = _recursive_bary((n - a - b, a, b), gll_points)
            pts.append((bary[1], bary[2]))
        else:
            raise ValueError(variant)
    return np.asarray(pts)


def edge_nodes(n, variant="gll"):
    """Parameters t in [0,1] of the degree-n trace nodes along an edge."""
    if variant == "equispaced":
        return np.array([0.5]) if n == 0 else np.linspace(0.0, 1.0, n + 1)
    return gll_points(n)


# --------------------------------------------------------------------------------------
# quadrature
# --------------------------------------------------------------------------------------
def gauss_legendre_01(npts):
    """Gauss-Legendre rule with npts points on [0, 1]; exact to degree 2*npts-1."""
    x, w = leggauss(npts)
    return 0.5 * (x + 1.0), 0.5 * w


def triangle_quadrature(degree):
    """Collapsed Gauss-Jacobi rule on the reference triangle, exact to the given degree."""
    m = max(1, (degree + 2) // 2)
    xa, wa = leggauss(m)  # for xi direction
    xb, wb = roots_jacobi(m, 1.0, 0.0)  # weight (1-x) for the collapsed direction
    pts = []
    wts = []
    for i in range(m):
        for j in range(m):
            eta = 0.5 * (xb[j] + 1.0)
            xi = 0.5 * (xa[i] + 1.0) * (1.0 - eta)
            pts.append((xi, eta))
            wts.append(wa[i] * wb[j] / 8.0)
    return np.asarray(pts), np.asarray(wts)


# --------------------------------------------------------------------------------------
# polynomial spaces with nodal bases, via a (centred) monomial Vandermonde matrix
# --------------------------------------------------------------------------------------
class PolySpace2D:
    """Nodal Lagrange basis of P_n on the reference triangle."""

    def __init__(self, n, variant="gll"):
        self.n = n
        self.nodes = triangle_nodes(n, variant)
        self.ndof = len(self.nodes)
        self._exps = [(p, q) for d in range(n + 1) for q in range(d + 1) for p in [d - q]]
        V = self._mono(self.nodes)[0]
        self._coef = np.linalg.inv(V)  # monomial coefficients of the nodal basis functions

    def _mono(self, pts, deriv=0):
        """Monomials (and derivatives) in centred coordinates at pts[..., 2]."""
        x = pts[..., 0] - 1.0 / 3.0
        y = pts[..., 1] - 1.0 / 3.0
        ne = len(self._exps)
        val = np.empty(pts.shape[:-1] + (ne,))
        out = [val]
        if deriv >= 1:
            g = np.zeros(pts.shape[:-1] + (ne, 2))
            out.append(g)
        if deriv >= 2:
            H = np.zeros(pts.shape[:-1] + (ne, 2, 2))
            out.append(H)

        def pw(z, e):
            return z**e if e >= 0 else np.zeros_like(z)

        for m, (p, q) in enumerate(self._exps):
            val[..., m] = pw(x, p) * pw(y, q)
            if deriv >= 1:
                g[..., m, 0] = p * pw(x, p - 1) * pw(y, q)
                g[..., m, 1] = q * pw(x, p) * pw(y, q - 1)
            if deriv >= 2:
                H[..., m, 0, 0] = p * (p - 1) * pw(x, p - 2) * pw(y, q)
                H[..., m, 1, 1] = q * (q - 1) * pw(x, p) * pw(y, q - 2)
                H[..., m, 0, 1] = H[..., m, 1, 0] = p * q * pw(x, p - 1) * pw(y, q - 1)
        return out

    def tabulate(self, pts, deriv=0):
        """Basis values [..., ndof], reference gradients [..., ndof, 2], Hessians [..., ndof,2,2]."""
        out = self._mono(np.asarray(pts, dtype=float), deriv)
        res = [out[0] @ self._coef]
        if deriv >= 1:
            res.append(np.einsum("...md,mk->...kd", out[1], self._coef))
        if deriv >= 2:
            res.append(np.einsum("...mde,mk->...kde", out[2], self._coef))
        return res if deriv > 0 else res[0]


class PolySpace1D:
    """Nodal Lagrange basis of P_n on [0, 1]."""

    def __init__(self, n, variant="gll"):
        self.n = n
        self.nodes = edge_nodes(n, variant)
        self.ndof = n + 1
        V = np.vander(self.nodes - 0.5, n + 1, increasing=True)
        self._coef = np.linalg.inv(V)

    def tabulate(self, t):
        t = np.asarray(t, dtype=float)
        return np.vander(t.ravel() - 0.5, self.n + 1, increasing=True).reshape(
            t.shape + (self.n + 1,)
        ) @ self._coef


# --------------------------------------------------------------------------------------
# mesh
# --------------------------------------------------------------------------------------
class Mesh:
    """Structured triangulation of the unit square (see module docstring).

    ``periodic=True``: the doubly periodic square of side L (reference: ``PeriodicSquareMesh(nx, nx, L=2 pi)``,
    src/driver.py:182-183).  Same cells; the edges on the top / right side are identified with those on the bottom /
    left side, so there are nx*nx edges of each type, no boundary edges, and neighbour relations wrap around."""

    def __init__(self, nx, periodic=False, L=1.0):
        self.nx = nx
        self.periodic = bool(periodic)
        self.L = float(L)
        h = self.L / nx
        self.h = h
        nc = 2 * nx * nx
        self.ncells = nc
        v = np.zeros((nc, 3, 2))
        for j in range(nx):
            for i in range(nx):
                c = 2 * (j * nx + i)
                v[c] = [(i * h, j * h), ((i + 1) * h, j * h), (i * h, (j + 1) * h)]
                v[c + 1] = [((i + 1) * h, (j + 1) * h), (i * h, (j + 1) * h), ((i + 1) * h, j * h)]
        self.cell_vertices = v
        # affine maps x = v0 + J xi
        self.J = np.stack([v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]], axis=-1)  # [c, x-comp, ref-dir]
        self.detJ = np.abs(np.linalg.det(self.J))
        self.Jinv = np.linalg.inv(self.J)
        # edges
        nrow = nx if self.periodic else nx + 1  # rows of H edges = columns of V edges
        NH = nx * nrow
        NV = nx * nrow
        ND = nx * nx
        self.nedges = NH + NV + ND
        a = np.zeros((self.nedges, 2))
        b = np.zeros((self.nedges, 2))
        cp = -np.ones(self.nedges, dtype=int)
        cm = -np.ones(self.nedges, dtype=int)

        def cell(i, j, s):
            if self.periodic:
                return 2 * ((j % nx) * nx + (i % nx)) + s
            return 2 * (j * nx + i) + s if (0 <= i < nx and 0 <= j < nx) else -1

        for j in range(nrow):
            for i in range(nx):
                e = j * nx + i
                a[e] = (i * h, j * h)
                b[e] = ((i + 1) * h, j * h)
                cp[e], cm[e] = cell(i, j, 0), cell(i, j - 1, 1)
        for j in range(nx):
            for i in range(nrow):
                e = NH + j * nrow + i
                a[e] = (i * h, j * h)
                b[e] = (i * h, (j + 1) * h)
                cp[e], cm[e] = cell(i, j, 0), cell(i - 1, j, 1)
        for j in range(nx):
            for i in range(nx):
                e = NH + NV + j * nx + i
                a[e] = ((i + 1) * h, j * h)
                b[e] = (i * h, (j + 1) * h)
                cp[e], cm[e] = cell(i, j, 0), cell(i, j, 1)
        # make '+' always an existing cell
        swap = cp < 0
        cp[swap], cm[swap] = cm[swap], cp[swap]
        self.edge_a, self.edge_b = a, b
        self.edge_plus, self.edge_minus = cp, cm
        self.edge_len = np.linalg.norm(b - a, axis=1)
        # normal pointing out of the '+' cell
        t = (b - a) / self.edge_len[:, None]
        nrm = np.stack([t[:, 1], -t[:, 0]], axis=1)
        centroid = v[cp].mean(axis=1)
        mid = 0.5 * (a + b)
        flip = np.einsum("ed,ed->e", nrm, mid - centroid) < 0
        nrm[flip] *= -1
        self.edge_normal_plus = nrm
        self.interior = cm >= 0
        self.volume = float(np.sum(self.detJ) / 2.0)

    def ref_coords(self, cells, x):
        """Reference coordinates in `cells` ([m]) of physical points x ([m, q, 2])."""
        d = x - self.cell_vertices[cells, 0][:, None, :]
        if self.periodic:  # the point may be given in the other copy of the cell
            d = (d + 0.5 * self.L) % self.L - 0.5 * self.L
        return np.einsum("mrd,mqd->mqr", self.Jinv[cells], d)


# --------------------------------------------------------------------------------------
# general affine triangle meshes (the step the HIP path has NOT taken yet)
# --------------------------------------------------------------------------------------


class TriMesh:
    """Conforming triangulation given by vertex coordinates [nv, 2] and cells [nc, 3] (vertex numbers).  Same attributes
    as ``Mesh`` (per-cell affine map, edges with their two cells, outward normal of the '+' cell), so that
    ``HDGDiscretisation(mesh=...)`` runs on it unchanged.  Edge numbering: order of first appearance while walking the
    cells; edge direction: from the lower to the higher vertex number."""

    periodic = False

    def __init__(self, vertices, cells):
        X = np.asarray(vertices, dtype=float)
        C = np.asarray(cells, dtype=int)
        self.vertices, self.cells = X, C
        self.nx, self.L, self.h = 1, 1.0, None  # (only used to scale coordinate keys)
        self.ncells = nc = len(C)
        v = X[C]  # [nc, 3, 2]
        self.cell_vertices = v
        self.J = np.stack([v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]], axis=-1)
        self.detJ = np.abs(np.linalg.det(self.J))
        self.Jinv = np.linalg.inv(self.J)
        edges = {}
        ea, eb, cp, cm = [], [], [], []
        for c in range(nc):
            for l in range(3):
                i0, i1 = C[c, l], C[c, (l + 1) % 3]
                key = (min(i0, i1), max(i0, i1))
                if key in edges:
                    e = edges[key]
                    if cm[e] >= 0:
                        raise ValueError("an edge with more than two cells")
                    cm[e] = c
                else:
                    edges[key] = len(ea)
                    ea.append(key[0]); eb.append(key[1]); cp.append(c); cm.append(-1)
        self.nedges = len(ea)
        self.edge_vertices = np.stack([np.array(ea), np.array(eb)], axis=1)
        a, b = X[np.array(ea)], X[np.array(eb)]
        cp, cm = np.array(cp), np.array(cm)
        self.edge_a, self.edge_b = a, b
        self.edge_plus, self.edge_minus = cp, cm
        self.edge_len = np.linalg.norm(b - a, axis=1)
        t = (b - a) / self.edge_len[:, None]
        nrm = np.stack([t[:, 1], -t[:, 0]], axis=1)
        flip = np.einsum("ed,ed->e", nrm, 0.5 * (a + b) - v[cp].mean(axis=1)) < 0
        nrm[flip] *= -1
        self.edge_normal_plus = nrm
        self.interior = cm >= 0
        self.volume = float(np.sum(self.detJ) / 2.0)

    def ref_coords(self, cells, x):
        d = x - self.cell_vertices[cells, 0][:, None, :]
        return np.einsum("mrd,mqd->mqr", self.Jinv[cells], d)

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
        return TriMesh(np.array(X), np.array(cells))


def unit_disk_mesh(refinement_level=0):
    """The unit disk as the reference builds it (``UnitDiskMesh(refinement_level)``, src/driver.py:184-185).

    Restated from Firedrake's utility mesh of that name AS REMEMBERED (Firedrake is not available here: PARITY UNPINNED, and
    the construction itself could not be compared with the library): the square [-1, 1]^2 cut into 8 triangles around the
    origin, `refinement_level` uniform refinements, then every vertex x farther than 2^-(level+1) from the origin moved
    radially to  x * max(|x_1|, |x_2|) / |x|  (squares concentric with the origin become circles)."""
    X = np.array([[0, 0], [1, 0], [1, 1], [0, 1], [-1, 1], [-1, 0], [-1, -1], [0, -1], [1, -1]], dtype=float)
    C = np.array([[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 4, 5], [0, 5, 6], [0, 6, 7], [0, 7, 8], [0, 8, 1]])
    m = TriMesh(X, C)
    for _ in range(refinement_level):
        m = m.refined()
    Y = m.vertices.copy()
    r = np.linalg.norm(Y, axis=1)
    move = r > 1.0 / (1 << (refinement_level + 1))
    Y[move] *= (np.max(np.abs(Y[move]), axis=1) / r[move])[:, None]
    return TriMesh(Y, m.cells)
