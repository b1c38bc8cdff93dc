"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy/scipy) of the passive-tracer transport and of the vorticity
diagnostic of the reference, on top of oracle/hdg_oracle.py.  Imported only by tests/.

Restated reference lines (paths relative to the reference's src/):
  timesteppers/common.py:110-129      _tracer_advection(chi, q, u, project_onto_cg=True):
                                        u_ = L2 projection of u onto [CG_{deg u}]^2,  un = (u_.n + |u_.n|)/2,
                                        T(chi; q, u_) = q div(chi u_) dx - (chi+ - chi-)(un+ q+ - un- q-) dS
  timesteppers/hdg_imex.py:415-448    _tracer_residual(chi, i) = chi q_0 dx + dt sum_{j<i} a_expl[i,j] T(chi; q_j, P(Q_i))
                                      (the velocity of stage i for EVERY j, as written),
                                      _tracer_final_residual = chi q_0 dx + dt sum_i b_expl[i] T(chi; q_i, P(Q_i))
  timesteppers/hdg_imex.py:560,622-623,638-639   q_0 <- q^n; solve after each stage; final solve -> q^{n+1}
  timesteppers/hdg_implicit.py:93-96,192-193     q^{n+1} = q^n + dt M^-1 T(.; q^n, P(Q^n))  (the projection is taken when
                                                 the form is built, i.e. of the velocity at the START of the step)
  auxilliary/callbacks.py:43-69       vorticity omega in CG_{k+1}:  (tau, omega) = -(eps : grad(tau) x Q) dx + tau eps : (n x Q) ds,
                                      eps = [[0, 1], [-1, 0]], i.e. the weak 2-D curl d_x Q_y - d_y Q_x of the broken velocity

PARITY UNPINNED like the rest of the oracle (no Firedrake).  Choices that cannot be observed offline: the continuous
space uses the same node family as the broken one (so its dofs are the coincident nodes); the upwind facet integral,
whose integrand |u_.n| is not polynomial, uses ceil((3k+4)/2) Gauss points per edge (the rule of the velocity
upwind term); Firedrake's `project` solves the mass system iteratively (rtol 1e-8 by default),
here it is solved exactly.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

__all__ = ["TracerOracle"]


class TracerOracle:
    def __init__(self, disc):
        self.d = d = disc
        m = d.mesh
        nc, nu = m.ncells, d.nu
        # ---- continuous space CG_{k+1}: identify coincident nodes of the broken space
        X = d.node_coords(d.PU).reshape(-1, 2)
        key = np.round(X * (m.nx * 1e6 / m.L)).astype(np.int64)
        if m.periodic:
            key = key % int(round(m.nx * 1e6))
        _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
        self.cg_of_dg = inv.reshape(-1)  # [nc*nu] -> CG dof
        self.ncg = int(self.cg_of_dg.max()) + 1
        self.cg_coords = X[first]
        self.R = sp.csr_matrix((np.ones(nc * nu), (np.arange(nc * nu), self.cg_of_dg)), shape=(nc * nu, self.ncg))
        # scalar mass matrix of DG_{k+1} (block of MQ for one component)
        idx = np.arange(nc * nu) * 2
        self.MU = d.MQ.tocsr()[idx][:, idx].tocsc()
        self.MCG = (self.R.T @ self.MU @ self.R).tocsc()
        self._lu_cg = spla.splu(self.MCG)
        self._lu_mp = spla.splu(d.MP.tocsc())

    # ------------------------------------------------------------------ L2 projection onto [CG_{k+1}]^2 (common.py:119-122)
    def cg_project(self, u):
        """u: broken velocity [nc*nu, 2] nodal -> the projection, returned in the SAME broken nodal layout."""
        out = np.empty_like(u)
        for c in range(2):
            out[:, c] = self.R @ self._lu_cg.solve(self.R.T @ (self.MU @ u[:, c]))
        return out

    # ------------------------------------------------------------------ T(chi; q, u_) (common.py:123-129), dual vector [NP]
    def tracer_form(self, q, u_):
        d, m = self.d, self.d.mesh
        nc, nu, np_ = m.ncells, d.nu, d.np_
        qc = q.reshape(nc, np_)
        uc = u_.reshape(nc, nu, 2)
        # cell term: q (grad chi . u_ + chi div u_)
        qq = np.einsum("qa,ca->cq", d.cP, qc)  # [c, q]
        uq = np.einsum("qa,cad->cqd", d.cU, uc)  # [c, q, 2]
        divu = np.einsum("cqad,cad->cq", d.cUg, uc)
        out = np.einsum("cq,cq,cqid,cqd->ci", d.cwdet, qq, d.cPg, uq) + np.einsum("cq,cq,qi,cq->ci", d.cwdet, qq, d.cP, divu)
        out = out.reshape(-1).copy()
        # interior facets: -(chi+ - chi-)(un+ q+ - un- q-)
        e = d.eint
        cp, cm = m.edge_plus[e], m.edge_minus[e]
        n = m.edge_normal_plus[e]
        rule = d.eq_upwind
        Up, Pp, _, wl = d.edge_tab(e, cp, rule)
        Um, Pm, _, _ = d.edge_tab(e, cm, rule)
        u_p = np.einsum("mqa,mad->mqd", Up, uc[cp])
        u_m = np.einsum("mqa,mad->mqd", Um, uc[cm])
        unp = np.einsum("mqd,md->mq", u_p, n)
        unm = -np.einsum("mqd,md->mq", u_m, n)
        unp, unm = 0.5 * (unp + np.abs(unp)), 0.5 * (unm + np.abs(unm))
        qp = np.einsum("mqa,ma->mq", Pp, qc[cp])
        qm = np.einsum("mqa,ma->mq", Pm, qc[cm])
        flux = wl * (unp * qp - unm * qm)  # [m, q]
        np.subtract.at(out, d.dofP[cp], np.einsum("mq,mqi->mi", flux, Pp))
        np.add.at(out, d.dofP[cm], np.einsum("mq,mqi->mi", flux, Pm))
        return out

    def tracer_tendency(self, q, u_broken):
        """M^-1 T(.; q, P(u)): what one explicit term adds per unit dt."""
        return self._lu_mp.solve(self.tracer_form(q, self.cg_project(u_broken)))

    # ------------------------------------------------------------------ vorticity (callbacks.py:43-69)
    def vorticity(self, Q):
        """CG_{k+1} vorticity of the broken velocity Q [nc*nu, 2]; returns (values at the CG dofs, dof coordinates)."""
        d, m = self.d, self.d.mesh
        nc, nu = m.ncells, d.nu
        Qc = Q.reshape(nc, nu, 2)
        Qq = np.einsum("qa,cad->cqd", d.cU, Qc)
        # -(d_x tau Q_y - d_y tau Q_x) dx, tau = nodal basis of P_{k+1} on the cell
        b = -np.einsum("cq,cqi,cq->ci", d.cwdet, d.cUg[..., 0], Qq[..., 1]) + np.einsum("cq,cqi,cq->ci", d.cwdet, d.cUg[..., 1], Qq[..., 0])
        b = b.reshape(-1)
        e = d.ebnd
        c = m.edge_plus[e]
        n = m.edge_normal_plus[e]
        Ub, _, _, wl = d.edge_tab(e, c, d.eq_exact)
        Qe = np.einsum("mqa,mad->mqd", Ub, Qc[c])
        cross = n[:, None, 0] * Qe[..., 1] - n[:, None, 1] * Qe[..., 0]
        dofU = np.arange(nc * nu).reshape(nc, nu)
        np.add.at(b, dofU[c], np.einsum("mq,mq,mqi->mi", wl, cross, Ub))
        return self._lu_cg.solve(self.R.T @ b), self.cg_coords


def imex_with_tracer(o, tr, Q0, p0, q0, f_rhs, T_final):
    """OracleHDGIMEX.solve with the passive tracer carried along (hdg_imex.py:560,622-623,638-639)."""
    d = o.disc
    nt = int(np.round(T_final / o.dt))
    o.set_initial_condition(Q0, p0)
    q = q0.copy()
    s = o.nstages
    for k in range(nt):
        qs = [q.copy()] + [None] * (s - 1)

        def after_stage(i):
            ui = tr.cg_project(o.stage_Q[i])
            acc = qs[0].copy()
            for j in range(i):
                if o.a_expl[i, j] != 0:
                    acc = acc + o.dt * o.a_expl[i, j] * tr._lu_mp.solve(tr.tracer_form(qs[j], ui))
            qs[i] = acc

        _step_with_hook(o, f_rhs, k * o.dt, after_stage)
        qn = qs[0].copy()
        for i in range(s):
            if o.b_expl[i] != 0:
                qn = qn + o.dt * o.b_expl[i] * tr._lu_mp.solve(tr.tracer_form(qs[i], tr.cg_project(o.stage_Q[i])))
        q = qn
    return o.Q, o.p, q


def _step_with_hook(o, f_rhs, tn, after_stage):
    """OracleHDGIMEX.step (projection method) with a callback after every stage (stage_Q[i] final)."""
    import scipy.sparse.linalg as spla_

    d, dt, s = o.disc, o.dt, o.nstages
    for i in range(s):
        o.b_rhs[i] = f_rhs(tn + o.c_expl[i] * dt)
    o.stage_Q[0], o.stage_p[0], o.stage_l[0] = o.Q.copy(), o.p.copy(), o.lam.copy()
    for i in range(1, s):
        o.Qstar[i - 1] = d.project_bdm(o.stage_Q[i - 1])
        adt = o.a_impl[i, i] * dt
        F = d.assemble_f_impl(o.Qstar[i - 1], o.flux)
        ri = o._residual(i)
        lu = spla_.splu((d.MQ - adt * F).tocsc())
        for _ in range(o.n_richardson):
            Qi = o.stage_Q[i].ravel()
            rhs = ri - d.MQ @ Qi + adt * (F @ Qi + d.G_p @ o.stage_p[i] + d.G_l @ o.stage_l[i])
            dQ = lu.solve(rhs)
            o.Q_tent[i] = dQ.reshape(-1, 2)
            du, dp, dl = d.solve_mixed_poisson(rP=-(1.0 / adt) * (d.Wdiv @ dQ))
            dp, dl = d.shift_pressure(dp, dl)
            o.stage_Q[i] = o.stage_Q[i] + o.Q_tent[i] + adt * du.reshape(-1, 2)
            o.stage_p[i] = o.stage_p[i] + dp
            o.stage_l[i] = o.stage_l[i] + dl
        o.stage_p[i], o.stage_l[i] = d.shift_pressure(o.stage_p[i], o.stage_l[i])
        after_stage(i)
    u, phi, lam = d.solve_mixed_poisson(rQ=o._final_residual())
    o.Q = u.reshape(-1, 2)
    rP, rL = d.pressure_reconstruction_rhs(o.Q, f_rhs(tn + dt))
    _, p, lam = d.solve_mixed_poisson(rP=rP, rL=rL)
    o.p, o.lam = d.shift_pressure(p, lam)


def implicit_with_tracer(d, tr, dt, Q0, p0, q0, f_rhs, T_final, flux="upwind"):
    """OracleHDGImplicit.solve (projection method) with the tracer (hdg_implicit.py:93-96,192-193)."""
    from oracle.hdg_oracle import OracleHDGImplicit

    nt = int(np.round(T_final / dt))
    Q = Q0.copy()
    p = p0 - float(d.int_p @ p0) / d.mesh.volume
    q = q0.copy()
    one = OracleHDGImplicit(d, dt, flux=flux)
    for k in range(nt):
        dq = dt * tr.tracer_tendency(q, Q)  # velocity at the START of the step
        Q, p = one.solve(Q, p, lambda t, k=k: f_rhs(k * dt), dt)
        q = q + dq
    return Q, p, q
