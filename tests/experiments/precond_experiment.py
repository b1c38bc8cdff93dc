"""MEASURED numbers for DESIGN.md section 9 ("an advection-aware component for the tentative-velocity
preconditioner ... or a DESIGN section 9 row with measured numbers instead of 'reasoning only'").

Not collected by pytest (no test_ prefix); run by hand:  python tests/experiments/precond_experiment.py
Uses the numpy oracle's assembled matrices (test infrastructure), never the product.

System of a tentative-velocity solve (hdg_imex.py:223-255): A = M - gamma F(Q*), gamma = a_ii dt, dt = 0.25 / nx, Q* = BDM
projection of the Taylor-Green velocity.  GMRES (no restart, rtol 1e-10 on the preconditioned residual) iteration counts for
  none        unpreconditioned (mass-scaled)
  ilu0        ILU(0) of the assembled matrix -- what the reference configures (hdg_imex.py:224-228: pc_type ilu)
  bjac        element block-Jacobi of (mass + penalty): tent_precond = 0
  hybrid      Pi + Dinv (I - Pi): the product's default, tent_precond = 2
  hyb+cellA   hybrid with the element blocks of the FULL operator (advection included) on the conforming part
  hyb+gs1     hybrid, conforming part through one cell-block Gauss-Seidel sweep of A in the lexicographic (row) ordering
  hyb+sgs     the same with a symmetric (forward + backward) sweep
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import hdg_oracle as orc  # noqa: E402


def gmres_its(A, M, b, rtol=1e-10, maxit=400):
    """left-preconditioned full GMRES, returns the iteration count to ||M r|| <= rtol ||M r0||"""
    n = len(b)
    r = M(b)
    beta = np.linalg.norm(r)
    V = [r / beta]
    H = np.zeros((maxit + 1, maxit))
    for j in range(maxit):
        w = M(A @ V[j])
        for i in range(j + 1):
            H[i, j] = V[i] @ w
            w = w - H[i, j] * V[i]
        H[j + 1, j] = np.linalg.norm(w)
        V.append(w / H[j + 1, j])
        e1 = np.zeros(j + 2)
        e1[0] = beta
        y, res, *_ = np.linalg.lstsq(H[: j + 2, : j + 1], e1, rcond=None)
        rn = np.linalg.norm(H[: j + 2, : j + 1] @ y - e1)
        if rn <= rtol * beta:
            return j + 1
    return maxit


def run(k, nx):
    d = orc.HDGDiscretisation(nx, k)
    tg = orc.TaylorGreen(d)
    Q0, _ = tg.initial_condition()
    Qstar = d.project_bdm(Q0)
    dt = 0.25 / nx
    gamma = 0.25 * dt  # SSP2(3,3,2): a_ii = 1/4
    F = d.assemble_f_impl(Qstar, "upwind")
    MQ = d.MQ.tocsc()
    A = (MQ - gamma * F).tocsc()
    n = A.shape[0]
    n2 = 2 * d.nu
    nc = d.mesh.ncells
    # cell blocks (the oracle's velocity dofs are cell-major: rows c*n2 .. (c+1)*n2)
    blocks = lambda Mx: [Mx[c * n2:(c + 1) * n2, c * n2:(c + 1) * n2].toarray() for c in range(nc)]
    Minv = spla.splu(MQ)
    # BDM projection as a matrix (columns = images of unit vectors)
    P = np.stack([d.project_bdm(np.eye(n)[:, c].reshape(-1, 2)).ravel() for c in range(n)], axis=1)
    # symmetric part without advection: mass + penalty.  F = adv + penalty; the penalty part is F(Q* = 0)
    Fpen = d.assemble_f_impl(np.zeros_like(Qstar), "upwind")
    Dmp = sp.block_diag([np.linalg.inv(B) for B in blocks((MQ - gamma * Fpen).tocsc())]).tocsc()
    DA = sp.block_diag([np.linalg.inv(B) for B in blocks(A)]).tocsc()
    rng = np.random.default_rng(0)
    b = MQ @ rng.standard_normal(n)
    I = np.eye(n)
    res = {}
    res["none"] = gmres_its(A, lambda r: Minv.solve(r), b)
    try:
        ilu = spla.spilu(A, fill_factor=1.0, drop_tol=0.0)
        res["ilu0"] = gmres_its(A, ilu.solve, b)
    except Exception as exc:  # noqa: BLE001
        res["ilu0"] = f"failed: {exc}"
    res["bjac"] = gmres_its(A, lambda r: Dmp @ r, b)
    # in the product the residual lives in the orthonormal basis: r_hat = M^-1 r; Pi acts on coefficient vectors
    hyb = lambda r: P @ Minv.solve(r) + Dmp @ (r - MQ @ (P @ Minv.solve(r)))
    res["hybrid"] = gmres_its(A, hyb, b)
    hybA = lambda r: P @ (DA @ (MQ @ (P @ Minv.solve(r)))) + Dmp @ (r - MQ @ (P @ Minv.solve(r)))
    res["hyb+cellA"] = gmres_its(A, hybA, b)
    # cell-block Gauss-Seidel sweeps of A (lower / upper block triangle by cell number = mesh row order)
    Ad = A.toarray()
    cell = np.arange(n) // n2
    Lw = np.where(cell[:, None] >= cell[None, :], Ad, 0.0)
    Up = np.where(cell[:, None] <= cell[None, :], Ad, 0.0)
    Dg = np.where(cell[:, None] == cell[None, :], Ad, 0.0)
    Lwi, Upi = np.linalg.inv(Lw), np.linalg.inv(Up)
    gs1 = lambda r: P @ (Lwi @ (MQ @ (P @ Minv.solve(r)))) + Dmp @ (r - MQ @ (P @ Minv.solve(r)))
    res["hyb+gs1"] = gmres_its(A, gs1, b)
    sgs = lambda r: P @ (Upi @ (Dg @ (Lwi @ (MQ @ (P @ Minv.solve(r)))))) + Dmp @ (r - MQ @ (P @ Minv.solve(r)))
    res["hyb+sgs"] = gmres_its(A, sgs, b)
    return res


if __name__ == "__main__":
    for k, nx in ((1, 8), (1, 12), (2, 6), (2, 8), (2, 12)):
        print(f"k={k} nx={nx}", run(k, nx), flush=True)
