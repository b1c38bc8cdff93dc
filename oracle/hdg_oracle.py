"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the HDG / HDG-IMEX timestep hot path.

A numpy/scipy restatement of the weak forms and the time loop of the reference
(eikehmueller/IncompressibleEulerHDG):

* src/timesteppers/hdg_imex.py:29-660   (forms :313-413, solve loop :505-660, tableaux :668-1038)
* src/timesteppers/hdg_implicit.py:52-197
* src/timesteppers/common.py:23-108     (1/h_F, BDM projection, timestep count)
* src/model_problems.py:38-105          (Taylor-Green manufactured vortex)
* src/driver.py:365-381                 (L2 error norms)

The forms are assembled exactly as the UFL is written ('+'/'-' restrictions on interior facets,
``ds`` on boundary facets, ``dx`` on cells) into global scipy.sparse matrices in NODAL bases and
solved with sparse direct LU, i.e. the converged limit of the reference's PETSc solvers.  This is
deliberately a different formulation from the product (which is modal, matrix-free, statically
condensed and iterative), so agreement between the two is a real cross-check.

PARITY UNPINNED: Firedrake cannot be imported here and the reference has no
tests or golden data, so this oracle is pinned only by (i) the analytic vortex and its convergence
rates, (ii) basis-independent discrete invariants (tests/test_oracle_*.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product never does.
"""

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .fem import Mesh, PolySpace1D, PolySpace2D, gauss_legendre_01, triangle_quadrature

__all__ = ["HDGDiscretisation", "TaylorGreen", "OracleHDGIMEX", "OracleHDGImplicit", "TABLEAUX"]


def _coo(rows, cols, vals, shape):
    return sp.coo_matrix(
        (np.asarray(vals).ravel(), (np.asarray(rows).ravel(), np.asarray(cols).ravel())),
        shape=shape,
    ).tocsr()


class HDGDiscretisation:
    """Spaces [DG_{k+1}]^2 x DG_k x DGT_k on the structured mesh and all forms of the hot path.

    Reference: hdg_imex.py:65-70 (spaces), :313-365 (forms), common.py:36-57 (1/h_F).
    """

    def __init__(self, nx, degree, variant="gll", tau=1.0, alpha_penalty=1.0, periodic=False, L=1.0, mesh=None):
        # mesh: any object with the attributes of fem.Mesh (fem.TriMesh: general affine triangles, e.g. fem.unit_disk_mesh);
        # nothing below depends on the mesh being structured
        self.mesh = mesh = Mesh(nx, periodic=periodic, L=L) if mesh is None else mesh
        self.k = k = degree
        self.tau = float(tau)  # hdg_imex.py:58
        self.alpha = float(alpha_penalty)  # hdg_imex.py:56
        self.PU = PolySpace2D(k + 1, variant)
        self.PP = PolySpace2D(k, variant)
        self.PL = PolySpace1D(k, variant)
        self.nu, self.np_, self.nl = self.PU.ndof, self.PP.ndof, self.PL.ndof
        nc, ne = mesh.ncells, mesh.nedges
        self.NQ = nc * self.nu * 2
        self.NP = nc * self.np_
        self.NL = ne * self.nl
        self.N = self.NQ + self.NP + self.NL

        # ---- cell quadrature data (exact for degree 3k+4: covers every polynomial cell integrand)
        qp, qw = triangle_quadrature(3 * k + 4)
        self.cq_w = qw
        Uv, Ug, UH = self.PU.tabulate(qp, deriv=2)
        Pv, Pg = self.PP.tabulate(qp, deriv=1)
        self.cU = Uv  # [q, nu]
        self.cP = Pv
        # physical gradients per cell: grad phi = J^{-T} grad_hat phi
        Ji = mesh.Jinv  # [c, r, d]  (xi_r = Ji[r,d] (x_d - v0_d))
        self.cUg = np.einsum("crd,qmr->cqmd", Ji, Ug)  # [c, q, nu, 2]
        self.cPg = np.einsum("crd,qmr->cqmd", Ji, Pg)
        self.cUH = np.einsum("crd,cse,qmrs->cqmde", Ji, Ji, UH)  # [c,q,nu,2,2]
        self.cwdet = mesh.detJ[:, None] * qw[None, :]  # [c, q]

        # ---- incidences (cell, edge) with outward normals: first all '+' sides, then '-' sides
        ep, em = mesh.edge_plus, mesh.edge_minus
        eint = np.nonzero(mesh.interior)[0]
        self.inc_cell = np.concatenate([ep, em[eint]])
        self.inc_edge = np.concatenate([np.arange(ne), eint])
        self.inc_normal = np.concatenate([mesh.edge_normal_plus, -mesh.edge_normal_plus[eint]])
        self.eint = eint
        self.ebnd = np.nonzero(~mesh.interior)[0]

        # ---- edge quadrature: "exact" rule for polynomial integrands and the specific rule that
        # UFL/FIAT would pick for the degree-(3k+3) upwind integrand
        self.eq_exact = gauss_legendre_01(2 * k + 4)
        self.eq_upwind = gauss_legendre_01(-(-(3 * k + 4) // 2))

        # ---- dof maps
        self.dofQ = (np.arange(nc * self.nu * 2)).reshape(nc, self.nu, 2)
        self.dofP = np.arange(nc * self.np_).reshape(nc, self.np_)
        self.dofL = np.arange(ne * self.nl).reshape(ne, self.nl)

        self._build_constant_operators()

    # ---------------------------------------------------------------- helpers
    def edge_tab(self, edges, cells, rule):
        """Tabulate velocity/pressure/trace bases of `cells` on `edges` at the rule's points."""
        m = self.mesh
        t, w = rule
        x = m.edge_a[edges][:, None, :] + t[None, :, None] * (m.edge_b[edges] - m.edge_a[edges])[:, None, :]
        xi = m.ref_coords(cells, x)
        U = self.PU.tabulate(xi)  # [m, q, nu]
        P = self.PP.tabulate(xi)
        L = np.broadcast_to(self.PL.tabulate(t), (len(edges),) + (len(t), self.nl))
        wl = w[None, :] * m.edge_len[edges][:, None]
        return U, P, L, wl

    def edge_tab_grad(self, edges, cells, rule):
        m = self.mesh
        t, w = rule
        x = m.edge_a[edges][:, None, :] + t[None, :, None] * (m.edge_b[edges] - m.edge_a[edges])[:, None, :]
        xi = m.ref_coords(cells, x)
        Uv, Ug = self.PU.tabulate(xi, deriv=1)
        Ugp = np.einsum("mrd,mqkr->mqkd", m.Jinv[cells], Ug)
        return Uv, Ugp

    def node_coords(self, space):
        """Physical coordinates of the nodes of a cell space: [nc, ndof, 2]."""
        m = self.mesh
        return m.cell_vertices[:, 0][:, None, :] + np.einsum("cdr,nr->cnd", m.J, space.nodes)

    def interpolate_velocity(self, fn):
        """Nodal interpolation into [DG_{k+1}]^2: fn(x, y) -> (ux, uy)."""
        X = self.node_coords(self.PU)
        ux, uy = fn(X[..., 0], X[..., 1])
        return np.stack([ux, uy], axis=-1).reshape(-1, 2)

    def interpolate_pressure(self, fn):
        X = self.node_coords(self.PP)
        return fn(X[..., 0], X[..., 1]).reshape(-1)

    # ---------------------------------------------------------------- constant operators
    def _build_constant_operators(self):
        m = self.mesh
        nc, ne = m.ncells, m.nedges
        nu, np_, nl = self.nu, self.np_, self.nl
        tau = self.tau
        # --- mass matrices
        Mu_loc = np.einsum("cq,qa,qb->cab", self.cwdet, self.cU, self.cU)  # [c, nu, nu]
        rows, cols, vals = [], [], []
        for d in range(2):
            rows.append(np.broadcast_to(self.dofQ[:, :, None, d], Mu_loc.shape))
            cols.append(np.broadcast_to(self.dofQ[:, None, :, d], Mu_loc.shape))
            vals.append(Mu_loc)
        self.MQ = _coo(np.stack(rows), np.stack(cols), np.stack(vals), (self.NQ, self.NQ))
        Mp_loc = np.einsum("cq,qa,qb->cab", self.cwdet, self.cP, self.cP)
        self.MP = _coo(
            np.broadcast_to(self.dofP[:, :, None], Mp_loc.shape),
            np.broadcast_to(self.dofP[:, None, :], Mp_loc.shape),
            Mp_loc,
            (self.NP, self.NP),
        )
        self.int_p = np.asarray(self.MP.sum(axis=0)).ravel()  # integral functional on DG_k

        # --- (psi, div u)_K : [NP x NQ]
        Bdiv = np.einsum("cq,qa,cqbd->cabd", self.cwdet, self.cP, self.cUg)  # [c, np, nu, 2]
        self.Bdiv = _coo(
            np.broadcast_to(self.dofP[:, :, None, None], Bdiv.shape),
            np.broadcast_to(self.dofQ[:, None, :, :], Bdiv.shape),
            Bdiv,
            (self.NP, self.NQ),
        )

        # --- incidence-based facet blocks
        ic, ie, inorm = self.inc_cell, self.inc_edge, self.inc_normal
        U, P, L, wl = self.edge_tab(ie, ic, self.eq_exact)
        # <lambda, w.n_K>_{dK}: rows w (NQ), cols lambda (NL)
        Ct = np.einsum("mq,mqa,md,mqb->madb", wl, U, inorm, L)  # [m, nu, 2, nl]
        self.CT = _coo(
            np.broadcast_to(self.dofQ[ic][:, :, :, None], Ct.shape),
            np.broadcast_to(self.dofL[ie][:, None, None, :], Ct.shape),
            Ct,
            (self.NQ, self.NL),
        )
        # tau <phi, psi>_{dK}
        T = tau * np.einsum("mq,mqa,mqb->mab", wl, P, P)
        self.T = _coo(
            np.broadcast_to(self.dofP[ic][:, :, None], T.shape),
            np.broadcast_to(self.dofP[ic][:, None, :], T.shape),
            T,
            (self.NP, self.NP),
        )
        # tau <lambda, psi>_{dK}: rows psi, cols lambda
        E = tau * np.einsum("mq,mqa,mqb->mab", wl, P, L)
        self.Et = _coo(
            np.broadcast_to(self.dofP[ic][:, :, None], E.shape),
            np.broadcast_to(self.dofL[ie][:, None, :], E.shape),
            E,
            (self.NP, self.NL),
        )
        # tau <lambda, mu>_{dK} summed over incidences
        Lm = tau * np.einsum("mq,mqa,mqb->mab", wl, L, L)
        self.Lm = _coo(
            np.broadcast_to(self.dofL[ie][:, :, None], Lm.shape),
            np.broadcast_to(self.dofL[ie][:, None, :], Lm.shape),
            Lm,
            (self.NL, self.NL),
        )
        # integral functional on DGT_k, every edge counted once (used to make inconsistent data consistent)
        mult = np.where(np.repeat(m.interior, nl), 2.0, 1.0)
        self.int_l = np.asarray(self.Lm.sum(axis=0)).ravel() / tau / mult
        self.skeleton_length = float(np.sum(m.edge_len))
        # trace mass used by _reconstruct_trace (hdg_imex.py:462): 2 tau lam+ mu+ dS + tau lam mu ds
        # = tau * (sum over incidences), i.e. self.Lm.

        # --- pressure gradient g(w,p,lambda) = p div w dx - sum_K <lambda, w.n_K>   (hdg_imex.py:333-340)
        self.G_p = self.Bdiv.T.tocsr()  # [NQ x NP]
        self.G_l = (-self.CT).tocsr()  # [NQ x NL]

        # --- mixed Poisson operator (hdg_imex.py:123-127):
        #   w-row:  (w,u) - (phi, div w) + <lambda, w.n>
        #   psi-row: (psi, div u) + tau<phi - lambda, psi>
        #   mu-row:  <u.n + tau(phi - lambda), mu>
        self.K_mp = sp.bmat(
            [
                [self.MQ, -self.Bdiv.T, self.CT],
                [self.Bdiv, self.T, -self.Et],
                [self.CT.T, self.Et.T, -self.Lm],
            ],
            format="csc",
        )

        # --- weak divergence (hdg_imex.py:353-365): rows psi (NP), cols Q (NQ)
        self.Wdiv = self._assemble_weak_divergence()

        # --- null-space vector (hdg_imex.py:480-489) and bordered mixed-Poisson factorisation
        # Bordering: the column is the LEFT null vector (0, -1, 1) (never in the range of K), the row fixes the mean of phi.
        # (A column equal to the right null vector (0, 1, 1) is orthogonal to the left one when NP == NL -- the periodic
        # mesh at k = 1 -- and the bordered matrix is then singular.)
        yl = np.concatenate([np.zeros(self.NQ), -np.ones(self.NP), np.ones(self.NL)])
        cvec = np.concatenate([np.zeros(self.NQ), self.int_p, np.zeros(self.NL)])
        Kb = sp.bmat([[self.K_mp, sp.csc_matrix(yl[:, None])], [sp.csc_matrix(cvec[None, :]), None]], format="csc")
        self._lu_mp = spla.splu(Kb)

    def _assemble_weak_divergence(self):
        m = self.mesh
        rows, cols, vals = [], [], []
        # cell term psi div(Q) dx
        rows.append(self.Bdiv)
        W = self.Bdiv.copy()
        # interior facets:  -2 avg(psi n.Q) + 2 avg(psi n) . avg(Q)
        e = self.eint
        cp, cm = m.edge_plus[e], m.edge_minus[e]
        n = m.edge_normal_plus[e]
        Up, Pp, _, wl = self.edge_tab(e, cp, self.eq_exact)
        Um, Pm, _, _ = self.edge_tab(e, cm, self.eq_exact)
        sides = [(cp, Up, Pp, n), (cm, Um, Pm, -n)]
        blocks = []
        for a, (ca, _, Pa, na) in enumerate(sides):
            for b, (cb, Ub, _, _) in enumerate(sides):
                coef = 0.5 - (1.0 if a == b else 0.0)
                blk = coef * np.einsum("mq,mqa,md,mqb->mabd", wl, Pa, na, Ub)  # [m, np, nu, 2]
                blocks.append(
                    _coo(
                        np.broadcast_to(self.dofP[ca][:, :, None, None], blk.shape),
                        np.broadcast_to(self.dofQ[cb][:, None, :, :], blk.shape),
                        blk,
                        (self.NP, self.NQ),
                    )
                )
        # boundary facets: - psi n.Q ds
        e = self.ebnd
        c = m.edge_plus[e]
        n = m.edge_normal_plus[e]
        Ub, Pb, _, wl = self.edge_tab(e, c, self.eq_exact)
        blk = -np.einsum("mq,mqa,md,mqb->mabd", wl, Pb, n, Ub)
        blocks.append(
            _coo(
                np.broadcast_to(self.dofP[c][:, :, None, None], blk.shape),
                np.broadcast_to(self.dofQ[c][:, None, :, :], blk.shape),
                blk,
                (self.NP, self.NQ),
            )
        )
        for b in blocks:
            W = W + b
        return W.tocsr()

    # ---------------------------------------------------------------- mixed Poisson solve
    def solve_mixed_poisson(self, rQ=None, rP=None, rL=None):
        """Solve the (singular) hybridised mixed Poisson system; returns (u, phi, lambda) with
        zero-mean phi (the constant is removed again by the callers' _shift_pressure)."""
        rP = np.zeros(self.NP) if rP is None else rP
        rL = np.zeros(self.NL) if rL is None else rL
        # The operator is singular: left null vector (0, -1, 1), right null vector (0, 1, 1).  Data that
        # are not orthogonal to the left null vector (hdg_implicit.py:145 uses the BROKEN divergence;
        # the reference hands that to a direct solver with no null-space information, SURVEY.md C-5,
        # so its result is ill defined) are made consistent by a uniform flux correction on the
        # skeleton: r_lambda(mu) -= c * int_E mu ds.  This is basis independent and is exactly what
        # projecting the condensed right-hand side orthogonally to the constants does.
        c = (np.sum(rL) - np.sum(rP)) / self.skeleton_length
        rL = rL - c * self.int_l
        r = np.concatenate([np.zeros(self.NQ) if rQ is None else rQ, rP, rL, [0.0]])
        x = self._lu_mp.solve(r)
        return x[: self.NQ], x[self.NQ : self.NQ + self.NP], x[self.NQ + self.NP : self.N]

    # ---------------------------------------------------------------- f_impl (hdg_imex.py:313-331)
    def assemble_f_impl(self, Qstar, flux="upwind"):
        """Matrix F with  w^T F Q = f_impl(w, Q, Q*)  for the broken field Q* ([nc*nu, 2] nodal)."""
        m = self.mesh
        nu = self.nu
        Qs = Qstar.reshape(m.ncells, nu, 2)
        blocks = []
        # cell: - inner(outer(w,Q*), grad Q) dx = - w_i Q*_j d_j Q_i
        Qs_q = np.einsum("qa,cad->cqd", self.cU, Qs)  # [c, q, 2]
        adv = np.einsum("cqd,cqbd->cqb", Qs_q, self.cUg)  # Q*.grad(phi_b)
        loc = -np.einsum("cq,qa,cqb->cab", self.cwdet, self.cU, adv)
        for d in range(2):
            blocks.append(
                _coo(
                    np.broadcast_to(self.dofQ[:, :, None, d], loc.shape),
                    np.broadcast_to(self.dofQ[:, None, :, d], loc.shape),
                    loc,
                    (self.NQ, self.NQ),
                )
            )
        # interior facets
        e = self.eint
        cp, cm = m.edge_plus[e], m.edge_minus[e]
        n = m.edge_normal_plus[e]
        hinv = 1.0 / m.edge_len[e]
        sgn = [1.0, -1.0]
        for rule, kind in ((self.eq_exact, "poly"), (self.eq_upwind, "upwind")):
            if kind == "upwind" and flux != "upwind":
                continue
            Up, _, _, wl = self.edge_tab(e, cp, rule)
            Um, _, _, _ = self.edge_tab(e, cm, rule)
            qn = np.einsum("mqa,mad,md->mq", Up, Qs[cp], n)  # Q*(+).n(+)
            sides = [(cp, Up, n), (cm, Um, -n)]
            for a, (ca, Ua, na) in enumerate(sides):
                for b, (cb, Ub, nb) in enumerate(sides):
                    if kind == "poly":
                        # (Q*+ . n+) (Q+ - Q-) . avg(w)
                        loc = 0.5 * sgn[b] * np.einsum("mq,mq,mqa,mqb->mab", wl, qn, Ua, Ub)
                        for d in range(2):
                            blocks.append(
                                _coo(
                                    np.broadcast_to(self.dofQ[ca][:, :, None, d], loc.shape),
                                    np.broadcast_to(self.dofQ[cb][:, None, :, d], loc.shape),
                                    loc,
                                    (self.NQ, self.NQ),
                                )
                            )
                        # - alpha 4 avg(1/hF) avg(Q.n) avg(w.n)
                        pen = -self.alpha * np.einsum(
                            "m,mq,mqa,mc,mqb,md->macbd", hinv, wl, Ua, na, Ub, nb
                        )
                        blocks.append(
                            _coo(
                                np.broadcast_to(self.dofQ[ca][:, :, :, None, None], pen.shape),
                                np.broadcast_to(self.dofQ[cb][:, None, None, :, :], pen.shape),
                                pen,
                                (self.NQ, self.NQ),
                            )
                        )
                    else:
                        # - |Q*+ . n+| (Q+ - Q-) . (w+ - w-)
                        loc = -sgn[a] * sgn[b] * np.einsum("mq,mq,mqa,mqb->mab", wl, np.abs(qn), Ua, Ub)
                        for d in range(2):
                            blocks.append(
                                _coo(
                                    np.broadcast_to(self.dofQ[ca][:, :, None, d], loc.shape),
                                    np.broadcast_to(self.dofQ[cb][:, None, :, d], loc.shape),
                                    loc,
                                    (self.NQ, self.NQ),
                                )
                            )
        # boundary facets: - alpha (1/hF) (Q.n)(w.n) ds
        e = self.ebnd
        c = m.edge_plus[e]
        n = m.edge_normal_plus[e]
        Ub, _, _, wl = self.edge_tab(e, c, self.eq_exact)
        pen = -self.alpha * np.einsum("m,mq,mqa,mc,mqb,md->macbd", 1.0 / m.edge_len[e], wl, Ub, n, Ub, n)
        blocks.append(
            _coo(
                np.broadcast_to(self.dofQ[c][:, :, :, None, None], pen.shape),
                np.broadcast_to(self.dofQ[c][:, None, None, :, :], pen.shape),
                pen,
                (self.NQ, self.NQ),
            )
        )
        F = blocks[0]
        for b in blocks[1:]:
            F = F + b
        return F.tocsr()

    # ---------------------------------------------------------------- BDM projection (common.py:91-108)
    def project_bdm(self, Q):
        """Q* in BDM_{k+1}, returned as a broken [P_{k+1}]^2 nodal field [nc*nu, 2].

        Per cell, Q*|_K is the unique member of [P_{k+1}(K)]^2 with
          (i)   edge normal moments against P_{k+1}(e) equal to the AVERAGE of the two sides'
                (INC-interpolation times inverse multiplicity, common.py:101-104),
          (ii)  zero on boundary edges (DirichletBC, common.py:106-107),
          (iii) interior moments against the Nedelec (first kind) space of degree k equal to
                those of Q|_K.
        """
        m = self.mesh
        k, nu = self.k, self.nu
        nc = m.ncells
        Qc = Q.reshape(nc, nu, 2)
        PE = PolySpace1D(k + 1, "equispaced")  # any basis of P_{k+1}(e) gives the same Q*
        nE = k + 2
        # rows: 3 edges x (k+2) normal moments + k(k+2) interior moments
        ndn = k * (k + 2)
        A = np.zeros((nc, 2 * nu, nu, 2))
        rhs = np.zeros((nc, 2 * nu))
        # --- edge moments.  local edge order per cell taken from incidences
        ic, ie, inorm = self.inc_cell, self.inc_edge, self.inc_normal
        U, _, _, wl = self.edge_tab(ie, ic, self.eq_exact)
        Eb = PE.tabulate(self.eq_exact[0])  # [q, nE]
        mom = np.einsum("mq,qr,mqa,md->mrad", wl, Eb, U, inorm)  # functional rows [m, nE, nu, 2]
        own = np.einsum("mrad,mad->mr", mom, Qc[ic])  # own-side normal moments (outward normal)
        # slot of this incidence within its cell (0..2)
        order = np.argsort(ic, kind="stable")
        slot = np.empty(len(ic), dtype=int)
        slot[order] = np.arange(len(ic)) % 3
        # target: average with the other side (whose outward normal is opposite), zero on boundary
        ne = m.nedges
        nint = len(self.eint)
        other = np.full(len(ic), -1)
        # '+' incidence index of edge e is e; '-' incidence index is ne + position in eint
        pos = -np.ones(ne, dtype=int)
        pos[self.eint] = np.arange(nint)
        plus_idx = np.arange(ne)
        minus_idx = ne + pos
        other[plus_idx[self.eint]] = minus_idx[self.eint]
        other[ne:] = self.eint
        target = np.zeros_like(own)
        has = other >= 0
        target[has] = 0.5 * (own[has] - own[other[has]])
        for s in range(3):
            sel = slot == s
            A[ic[sel], s * nE : (s + 1) * nE] = mom[sel]
            rhs[ic[sel], s * nE : (s + 1) * nE] = target[sel]
        # --- interior moments against ND_k = [P_{k-1}]^2 + (-y, x) * homogeneous P_{k-1}
        if ndn > 0:
            qp = triangle_quadrature(3 * k + 4)[0]
            X = m.cell_vertices[:, 0][:, None, :] + np.einsum("cdr,qr->cqd", m.J, qp)
            xc = X - m.cell_vertices.mean(axis=1)[:, None, :]
            tests = []
            for d in range(k):
                for q in range(d + 1):
                    mono = xc[..., 0] ** (d - q) * xc[..., 1] ** q
                    tests.append(np.stack([mono, 0 * mono], -1))
                    tests.append(np.stack([0 * mono, mono], -1))
            for q in range(k):
                mono = xc[..., 0] ** (k - 1 - q) * xc[..., 1] ** q
                tests.append(np.stack([-xc[..., 1] * mono, xc[..., 0] * mono], -1))
            Tq = np.stack(tests, axis=1)  # [c, ndn, q, 2]
            assert Tq.shape[1] == ndn
            rowsI = np.einsum("cq,crqd,qa->crad", self.cwdet, Tq, self.cU)
            A[:, 3 * nE :] = rowsI
            rhs[:, 3 * nE :] = np.einsum("crad,cad->cr", rowsI, Qc)
        sol = np.linalg.solve(A.reshape(nc, 2 * nu, 2 * nu), rhs[..., None])[..., 0]
        return sol.reshape(nc * nu, 2)

    # ---------------------------------------------------------------- trace reconstruction (hdg_imex.py:450-469)
    def reconstruct_trace(self, Q, p):
        ic, ie, inorm = self.inc_cell, self.inc_edge, self.inc_normal
        U, P, L, wl = self.edge_tab(ie, ic, self.eq_exact)
        Qc = Q.reshape(-1, self.nu, 2)[ic]
        pc = p.reshape(-1, self.np_)[ic]
        g = np.einsum("mqa,mad,md->mq", U, Qc, inorm) + self.tau * np.einsum("mqa,ma->mq", P, pc)
        b = np.zeros(self.NL)
        np.add.at(b, self.dofL[ie], np.einsum("mq,mq,mqa->ma", wl, g, L))
        return spla.spsolve(self.Lm.tocsc(), b)

    # ---------------------------------------------------------------- pressure reconstruction RHS (hdg_imex.py:201-207)
    def pressure_reconstruction_rhs(self, Q, b_new):
        """(rP, rL) for  weak_divergence(psi, -b_new + dot(grad(Q), Q)) - mu n.b_new ds."""
        m = self.mesh
        nu = self.nu
        Qc = Q.reshape(m.ncells, nu, 2)
        bc_ = b_new.reshape(m.ncells, nu, 2)
        # cell: psi * div(v),  v_i = -b_i + Q_j d_j Q_i ;  div v = -div b + d_iQ_j d_jQ_i + Q_j d_j d_i Q_i
        gQ = np.einsum("cqad,cai->cqid", self.cUg, Qc)  # d_d Q_i
        HQ = np.einsum("cqade,cai->cqide", self.cUH, Qc)  # d_d d_e Q_i
        Qq = np.einsum("qa,cai->cqi", self.cU, Qc)
        divb = np.einsum("cqad,cad->cq", self.cUg, bc_)
        divv = -divb + np.einsum("cqij,cqji->cq", gQ, gQ) + np.einsum("cqj,cqiji->cq", Qq, HQ)
        rP = np.zeros(self.NP)
        np.add.at(rP, self.dofP, np.einsum("cq,qa,cq->ca", self.cwdet, self.cP, divv))
        # facet terms need v on both sides
        rule = gauss_legendre_01(2 * self.k + 4)

        def vside(e, c):
            Uv, Ug = self.edge_tab_grad(e, c, rule)
            Qe = np.einsum("mqa,mai->mqi", Uv, Qc[c])
            gQe = np.einsum("mqad,mai->mqid", Ug, Qc[c])
            be = np.einsum("mqa,mai->mqi", Uv, bc_[c])
            return -be + np.einsum("mqid,mqd->mqi", gQe, Qe), be

        e = self.eint
        cp, cm = m.edge_plus[e], m.edge_minus[e]
        n = m.edge_normal_plus[e]
        vp, _ = vside(e, cp)
        vm, _ = vside(e, cm)
        _, Pp, _, wl = self.edge_tab(e, cp, rule)
        _, Pm, _, _ = self.edge_tab(e, cm, rule)
        vavg = 0.5 * (vp + vm)
        # -2 avg(psi n.v) + 2 avg(psi n).avg(v)
        tp = np.einsum("mq,mqa,mq->ma", wl, Pp, np.einsum("md,mqd->mq", n, vavg - vp))
        tm = np.einsum("mq,mqa,mq->ma", wl, Pm, np.einsum("md,mqd->mq", -n, vavg - vm))
        np.add.at(rP, self.dofP[cp], tp)
        np.add.at(rP, self.dofP[cm], tm)
        e = self.ebnd
        c = m.edge_plus[e]
        n = m.edge_normal_plus[e]
        vb, bb = vside(e, c)
        _, Pb, Lb, wl = self.edge_tab(e, c, rule)
        np.add.at(rP, self.dofP[c], -np.einsum("mq,mqa,mq->ma", wl, Pb, np.einsum("md,mqd->mq", n, vb)))
        rL = np.zeros(self.NL)
        np.add.at(rL, self.dofL[e], -np.einsum("mq,mqa,mq->ma", wl, Lb, np.einsum("md,mqd->mq", n, bb)))
        return rP, rL

    # ---------------------------------------------------------------- pressure shift (hdg_imex.py:471-478)
    def shift_pressure(self, p, lam):
        pbar = float(self.int_p @ p) / self.mesh.volume
        return p - pbar, lam - pbar

    # ---------------------------------------------------------------- norms (driver.py:376-377)
    def l2_norm_velocity(self, Q):
        q = Q.reshape(-1)
        return float(np.sqrt(q @ (self.MQ @ q)))

    def l2_norm_pressure(self, p):
        return float(np.sqrt(p @ (self.MP @ p)))


# ======================================================================================
# model problem (src/model_problems.py:38-105)
# ======================================================================================
class TaylorGreen:
    """Manufactured time-dependent Taylor-Green vortex; nodal restatement."""

    def __init__(self, disc, forcing="exponential", kappa=0.5):
        assert forcing in ("exponential", "constant")
        self.disc = disc
        self.forcing = forcing
        self.kappa = kappa
        S = lambda z: np.sin((z - 0.5) * np.pi)
        C = lambda z: np.cos((z - 0.5) * np.pi)
        self.Qs = disc.interpolate_velocity(lambda x, y: (-C(x) * S(y), S(x) * C(y)))
        self.ps = disc.interpolate_pressure(lambda x, y: (S(x) ** 2 + S(y) ** 2) / 2)

    def initial_condition(self):
        return self.Qs.copy(), self.ps.copy()

    def f_rhs(self, t):
        """Nodal forcing at time t (model_problems.py:71-80; kappa == 0 treated as zero forcing)."""
        if self.kappa == 0:
            return np.zeros_like(self.Qs)
        if self.forcing == "exponential":
            return -self.kappa * np.exp(-self.kappa * t) * self.Qs
        return -self.kappa * self.Qs

    def solution(self, t):
        if self.forcing == "exponential":
            Q = np.exp(-self.kappa * t) * self.Qs
            p = np.exp(-2 * self.kappa * t) * self.ps
        else:
            Q = (1.0 - self.kappa * t) * self.Qs
            p = (1.0 - self.kappa * t) ** 2 * self.ps
        p = p - float(self.disc.int_p @ p)  # model_problems.py:104 (no division by the volume)
        return Q, p


class KelvinHelmholtz:
    """Kelvin-Helmholtz instability on the unit disk (model_problems.py:108-131): solid-body rotation (-y, x) inside the
    radius 0.5 (a conditional, interpolated at the nodes), fluid at rest outside, p = 0, no forcing."""

    def __init__(self, disc, r_max=0.5):
        self.disc = disc
        self.Qs = disc.interpolate_velocity(lambda x, y: (np.where(x ** 2 + y ** 2 < r_max ** 2, -y, 0.0),
                                                          np.where(x ** 2 + y ** 2 < r_max ** 2, x, 0.0)))
        self.ps = np.zeros(disc.NP)

    def initial_condition(self):
        return self.Qs.copy(), self.ps.copy()

    def f_rhs(self, t):
        return np.zeros_like(self.Qs)


# ======================================================================================
# tableaux -- literal values of hdg_imex.py:702-1038 (including the quirks)
# ======================================================================================
def _tableaux():
    g = 1 - 1 / np.sqrt(2)
    d = -2 / 3 * np.sqrt(2)
    al, be, et = 0.24169426078821, 0.06042356519705, 0.12915286960590
    de = 1 / 2 - al - be - et
    return {
        "imex_implicit": dict(
            label="HDG IMEX Implicit",
            a_expl=[[0, 0], [1, 0]],
            a_impl=[[0, 0], [0, 1]],
            b_expl=[1, 0],
            b_impl=[0, 1],
            c_expl=[0, 1],
        ),
        "imex_ars2_232": dict(
            label="HDG IMEX ARS2(2,3,2)",
            a_expl=[[0, 0, 0], [g, 0, 0], [d, 1 - d, 0]],
            a_impl=[[0, 0, 0], [0, g, 0], [0, 1 - g, g]],
            b_expl=[0, 1 - g, g],
            b_impl=[0, 1 - g, g],
            c_expl=[0, g, 1],
        ),
        "imex_ars3_443": dict(
            label="HDG IMEX ARS3(4,4,3)",
            a_expl=[
                [0, 0, 0, 0, 0],
                [1 / 2, 0, 0, 0, 0],
                [11 / 18, 1 / 18, 0, 0, 0],
                [5 / 6, -5 / 6, 1 / 2, 0, 0],
                [1 / 4, 7 / 4, 3 / 4, -7 / 4, 0],
            ],
            a_impl=[
                [0, 0, 0, 0, 0],
                [0, 1 / 2, 0, 0, 0],
                [0, 1 / 6, 1 / 2, 0, 0],
                [0, -1 / 2, 1 / 2, 1 / 2, 0],
                [0, 3 / 2, -3 / 2, 1 / 2, 1 / 2],
            ],
            b_expl=[1 / 4, 7 / 4, 3 / 4, -7 / 4, 0],
            b_impl=[0, 3 / 2, -3, 2, 1 / 2, 1 / 2],  # 6 entries as written (hdg_imex.py:874)
            c_expl=[0, 1 / 2, 2 / 3, 1 / 2, 1],
        ),
        "imex_ssp2_332": dict(
            label="HDG IMEX SSP2(3,3,2)",
            a_expl=[[0, 0, 0], [1 / 2, 0, 0], [1 / 2, 1 / 2, 0]],
            a_impl=[[1 / 4, 0, 0], [0, 1 / 4, 0], [1 / 3, 1 / 3, 1 / 3]],
            b_expl=[1 / 3, 1 / 3, 1 / 3],
            b_impl=[1 / 3, 1 / 3, 1 / 3],
            c_expl=[0, 1, 1 / 2],
        ),
        "imex_ssp3_433": dict(
            label="HDG IMEX SSP3(4,3,3)",
            a_expl=[[0, 0, 0, 0], [0, 0, 0, 0], [0, 1, 0, 0], [0, 1 / 4, 1 / 4, 0]],
            a_impl=[[al, 0, 0, 0], [-al, al, 0, 0], [0, 1 - al, al, 0], [be, et, de, al]],
            b_expl=[0, 1 / 6, 1 / 6, 2 / 3],
            b_impl=[0, 1 / 6, 1 / 6, 2 / 3],
            c_expl=[0, 0, 1, 1 / 2],
        ),
    }


TABLEAUX = _tableaux()


# ======================================================================================
# IMEX timestepper (hdg_imex.py:22-660)
# ======================================================================================
class OracleHDGIMEX:
    """Direct-solver restatement of IncompressibleEulerHDGIMEX (all five tableaux)."""

    def __init__(self, disc, dt, tableau="imex_ssp2_332", flux="upwind", use_projection_method=True, n_richardson=2):
        self.disc = d = disc
        self.dt = dt
        tb = TABLEAUX[tableau]
        self.label = tb["label"]
        self.a_expl = np.asarray(tb["a_expl"], dtype=float)
        self.a_impl = np.asarray(tb["a_impl"], dtype=float)
        self.b_expl = np.asarray(tb["b_expl"], dtype=float)
        self.b_impl = np.asarray(tb["b_impl"], dtype=float)
        self.c_expl = np.asarray(tb["c_expl"], dtype=float)
        self.nstages = len(self.c_expl)
        self.flux = flux
        self.use_projection_method = use_projection_method
        self.n_richardson = n_richardson
        s = self.nstages
        nn = d.mesh.ncells * d.nu
        # persistent stage vectors (hdg_imex.py:72-88; never reset: SURVEY.md C-3)
        self.stage_Q = [np.zeros((nn, 2)) for _ in range(s)]
        self.stage_p = [np.zeros(d.NP) for _ in range(s)]
        self.stage_l = [np.zeros(d.NL) for _ in range(s)]
        self.Qstar = [np.zeros((nn, 2)) for _ in range(s - 1)]
        self.Q_tent = [np.zeros((nn, 2)) for _ in range(s)]
        self.b_rhs = [np.zeros((nn, 2)) for _ in range(s)]
        self.Q = np.zeros((nn, 2))
        self.p = np.zeros(d.NP)
        self.lam = np.zeros(d.NL)
        self.trace = {}  # optional capture of intermediates for the parity tests

    # dual residual vectors r_i(w) (hdg_imex.py:367-391)
    def _residual(self, i):
        d = self.disc
        assert 0 < i < self.nstages
        r = d.MQ @ self.stage_Q[0].ravel()
        for j in range(1, i):  # column 0 is never read
            if self.a_impl[i, j] != 0:
                r = r + (self.a_impl[i, j] / self.a_impl[j, j]) * (d.MQ @ self.stage_Q[j].ravel() - self._residual(j))
        for j in range(i):
            if self.a_expl[i, j] != 0:
                r = r + self.dt * self.a_expl[i, j] * (d.MQ @ self.b_rhs[j].ravel())
        return r

    # hdg_imex.py:393-413
    def _final_residual(self):
        d = self.disc
        r = d.MQ @ self.stage_Q[0].ravel()
        for i in range(1, self.nstages):
            if self.b_impl[i] != 0:
                r = r + (self.b_impl[i] / self.a_impl[i, i]) * (d.MQ @ self.stage_Q[i].ravel() - self._residual(i))
        for i in range(self.nstages):
            if self.b_expl[i] != 0:
                r = r + self.dt * self.b_expl[i] * (d.MQ @ self.b_rhs[i].ravel())
        return r

    def set_initial_condition(self, Q0, p0):
        d = self.disc
        self.Q = Q0.copy()
        self.p = p0 - float(d.int_p @ p0) / d.mesh.volume  # hdg_imex.py:522
        self.lam = d.reconstruct_trace(self.Q, self.p)  # hdg_imex.py:534

    def step(self, f_rhs, tn):
        """One pass of the loop body hdg_imex.py:551-637."""
        d = self.disc
        dt = self.dt
        s = self.nstages
        for i in range(s):
            self.b_rhs[i] = f_rhs(tn + self.c_expl[i] * dt)
        self.stage_Q[0], self.stage_p[0], self.stage_l[0] = self.Q.copy(), self.p.copy(), self.lam.copy()
        for i in range(1, s):
            self.Qstar[i - 1] = d.project_bdm(self.stage_Q[i - 1])
            adt = self.a_impl[i, i] * dt
            F = d.assemble_f_impl(self.Qstar[i - 1], self.flux)
            ri = self._residual(i)
            if self.use_projection_method:
                A = (d.MQ - adt * F).tocsc()
                lu = spla.splu(A)
                for _ in range(self.n_richardson):
                    Qi = self.stage_Q[i].ravel()
                    rhs = ri - d.MQ @ Qi + adt * (F @ Qi + d.G_p @ self.stage_p[i] + d.G_l @ self.stage_l[i])
                    dQ = lu.solve(rhs)
                    self.Q_tent[i] = dQ.reshape(-1, 2)
                    rP = -(1.0 / adt) * (d.Wdiv @ dQ)
                    du, dp, dl = d.solve_mixed_poisson(rP=rP)
                    dp, dl = d.shift_pressure(dp, dl)
                    self.stage_Q[i] = self.stage_Q[i] + self.Q_tent[i] + adt * du.reshape(-1, 2)
                    self.stage_p[i] = self.stage_p[i] + dp
                    self.stage_l[i] = self.stage_l[i] + dl
            else:
                # monolithic stage solve (hdg_imex.py:600-620)
                Ktop = sp.bmat(
                    [
                        [d.MQ - adt * F, -adt * d.G_p, -adt * d.G_l],
                        [d.Bdiv, d.T, -d.Et],
                        [d.CT.T, d.Et.T, -d.Lm],
                    ],
                    format="csc",
                )
                x = _solve_singular(d, Ktop, np.concatenate([ri, np.zeros(d.NP + d.NL)]))
                self.stage_Q[i] = x[: d.NQ].reshape(-1, 2)
                self.stage_p[i] = x[d.NQ : d.NQ + d.NP]
                self.stage_l[i] = x[d.NQ + d.NP :]
            self.stage_p[i], self.stage_l[i] = d.shift_pressure(self.stage_p[i], self.stage_l[i])
        # final stage (hdg_imex.py:624): full mixed Poisson with the velocity-row RHS r^{n+1}
        u, phi, lam = d.solve_mixed_poisson(rQ=self._final_residual())
        self.Q = u.reshape(-1, 2)
        # pressure reconstruction (hdg_imex.py:629-637)
        b_new = f_rhs(tn + dt)
        rP, rL = d.pressure_reconstruction_rhs(self.Q, b_new)
        _, p, lam = d.solve_mixed_poisson(rP=rP, rL=rL)
        self.p, self.lam = d.shift_pressure(p, lam)

    def solve(self, Q0, p0, f_rhs, T_final, warmup=False):
        nt = 1 if warmup else int(np.round(T_final / self.dt))
        assert warmup or abs(nt * self.dt - T_final) < 1e-12  # common.py:83
        self.set_initial_condition(Q0, p0)
        for k in range(nt):
            self.step(f_rhs, k * self.dt)
        return self.Q, self.p


def _solve_singular(d, K, rhs):
    """Direct solve of a system whose null space is the constant (phi, lambda) shift."""
    yl = np.concatenate([np.zeros(d.NQ), -np.ones(d.NP), np.ones(d.NL)])  # left null vector (see _build_constant_operators)
    cvec = np.concatenate([np.zeros(d.NQ), d.int_p, np.zeros(d.NL)])
    Kb = sp.bmat([[K, sp.csc_matrix(yl[:, None])], [sp.csc_matrix(cvec[None, :]), None]], format="csc")
    return spla.spsolve(Kb, np.concatenate([rhs, [0.0]]))[:-1]


# ======================================================================================
# first-order implicit timestepper (hdg_implicit.py:10-197)
# ======================================================================================
class OracleHDGImplicit:
    def __init__(self, disc, dt, flux="upwind", use_projection_method=True):
        self.disc = disc
        self.dt = dt
        self.flux = flux
        self.use_projection_method = use_projection_method
        self.label = "HDG Implicit"

    def solve(self, Q0, p0, f_rhs, T_final, warmup=False):
        d, dt = self.disc, self.dt
        nt = 1 if warmup else int(np.round(T_final / dt))
        assert warmup or abs(nt * dt - T_final) < 1e-12
        Q = Q0.copy()
        p = p0 - float(d.int_p @ p0) / d.mesh.volume
        for k in range(nt):
            Qstar = d.project_bdm(Q)  # hdg_implicit.py:98
            f = f_rhs(k * dt)  # forcing at the START of the step (hdg_implicit.py:100)
            F = d.assemble_f_impl(Qstar, self.flux)
            rhs = d.MQ @ Q.ravel() + dt * (d.MQ @ f.ravel())
            if self.use_projection_method:
                Qt = spla.spsolve((d.MQ - dt * F).tocsc(), rhs)  # hdg_implicit.py:103-129
                rP = -(1.0 / dt) * (d.Bdiv @ Qt)  # broken divergence only (hdg_implicit.py:145)
                u, phi, _ = d.solve_mixed_poisson(rP=rP)
                Q = (Qt + dt * u).reshape(-1, 2)  # hdg_implicit.py:150
            else:
                K = sp.bmat(
                    [
                        [d.MQ - dt * F, -dt * d.G_p, -dt * d.G_l],
                        [d.Bdiv, d.T, -d.Et],
                        [d.CT.T, d.Et.T, -d.Lm],
                    ],
                    format="csc",
                )
                x = _solve_singular(d, K, np.concatenate([rhs, np.zeros(d.NP + d.NL)]))
                Q = x[: d.NQ].reshape(-1, 2)
                phi = x[d.NQ : d.NQ + d.NP]
            p = phi - float(d.int_p @ phi) / d.mesh.volume  # hdg_implicit.py:189-190
        return Q, p
