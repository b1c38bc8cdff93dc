"""The C++/OpenMP CPU twin (oracle/cpu_twin, test infrastructure and the bench's CPU baseline) against the numpy
oracle: operators at round-off level, whole IMEX steps at the two-converged-solvers tolerance.
Runs without a GPU.  The twin is then the CPU leg of the GPU parity tests at sizes the numpy oracle cannot reach
(tests/test_gpu_cpu_twin.py)."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

TOL = 2e-8


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _twin(k, nx, tableau="imex_ssp2_332", **kw):
    from oracle.cpu_twin import CpuTwin
    from oracle.hdg_oracle import TABLEAUX

    tb = TABLEAUX[tableau]
    return CpuTwin(nx=nx, degree=k, dt=kw.pop("dt", 0.25 / nx), nstages=len(tb["c_expl"]), a_expl=tb["a_expl"], a_impl=tb["a_impl"],
                   b_expl=tb["b_expl"], b_impl=tb["b_impl"], c_expl=tb["c_expl"], **kw)


@pytest.mark.parametrize("k,nx", [(1, 5), (2, 4), (3, 3)])
def test_cpu_twin_operators_match_the_numpy_oracle(k, nx):
    from oracle.hdg_oracle import HDGDiscretisation

    d = HDGDiscretisation(nx, k)
    t = _twin(k, nx)
    assert (t.n_cells, t.n_edges, t.n_u, t.n_p, t.n_l) == (d.mesh.ncells, d.mesh.nedges, d.nu, d.np_, d.nl)
    rng = np.random.default_rng(7)
    Q, x = rng.standard_normal(t.shape_Q), rng.standard_normal(t.shape_Q)
    assert _rel(t.project_bdm_nodal(Q), d.project_bdm(Q)) < 1e-11
    Qstar = d.project_bdm(Q)
    for flux in ("upwind", "centered"):
        tf = _twin(k, nx, flux=flux)
        gamma = 0.3 / nx
        F = d.assemble_f_impl(Qstar, flux)
        ref = x.ravel() - gamma * spla.spsolve(d.MQ.tocsc(), F @ x.ravel())
        assert _rel(tf.apply_advection(Qstar, x, gamma).ravel(), ref) < 5e-11
    Mi = spla.splu(d.MP.tocsc())
    assert _rel(t.apply_weak_divergence(Q), Mi.solve(d.Wdiv @ Q.ravel())) < 1e-11
    assert np.max(np.abs(t.apply_trace_operator(np.ones(t.shape_l)))) < 1e-9  # constants are in the null space


@pytest.mark.parametrize("k,nx,tableau,R", [(1, 6, "imex_ssp2_332", 2), (2, 4, "imex_ssp2_332", 2), (1, 5, "imex_ars3_443", 1),
                                            (2, 4, "imex_ssp3_433", 2)])
def test_cpu_twin_steps_match_the_numpy_oracle(k, nx, tableau, R):
    from oracle import hdg_oracle as orc

    dt, nsteps = 0.25 / nx, 2
    d = orc.HDGDiscretisation(nx, k)
    tg = orc.TaylorGreen(d)
    o = orc.OracleHDGIMEX(d, dt, tableau, n_richardson=R)
    oQ, op = o.solve(*tg.initial_condition(), tg.f_rhs, nsteps * dt)
    t = _twin(k, nx, tableau, n_richardson=R)
    t.set_state(*tg.initial_condition())
    t.reconstruct_trace()
    c = orc.TABLEAUX[tableau]["c_expl"]
    for n in range(nsteps):
        for i, ci in enumerate(c):
            t.set_forcing_nodal(i, tg.f_rhs(n * dt + ci * dt))
        t.set_forcing_nodal(len(c), tg.f_rhs((n + 1) * dt))
        t.step()
    Q, p, lam = t.get_state()
    assert _rel(Q, oQ) < TOL and _rel(p, op) < TOL and _rel(lam, o.lam) < TOL
    sums, cnt = t.iteration_stats()
    assert np.all(cnt > 0) and np.all(sums / cnt < 80)


@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_shared_tables_give_the_oracle_matrices_entry_by_entry(k):
    """The twin shares ONE file with the product: the host tables csrc/hdg_tables.hpp (bases, quadrature, local matrices).
    The large-mesh twin-vs-GPU comparisons (tests/test_gpu_cpu_twin.py) therefore cannot see a table error.  This test
    can: every operator the tables feed is probed with unit vectors on a 2 x 2 mesh (8 cells: both shapes, interior and
    boundary edges of every orientation) and the resulting global matrices are compared ENTRY BY ENTRY with the numpy
    oracle's independently assembled nodal matrices -- BDM projection (N, Lift), advection operator for both fluxes
    (cell / edge tabulations, quadrature weights, normals, penalty), weak divergence (D0, N, Pt), condensed trace operator
    (A^-1, W, Y, S_K) -- at every degree the product supports, k = 4 included."""
    import scipy.sparse as sp

    from oracle.hdg_oracle import HDGDiscretisation

    nx = 2
    d = HDGDiscretisation(nx, k)
    t = _twin(k, nx)
    nq = int(np.prod(t.shape_Q))
    eye_q = np.eye(nq)

    def columns(fn, n, shape):
        return np.stack([np.asarray(fn(col.reshape(shape))).ravel() for col in np.eye(n)], axis=1)

    # BDM projection: a linear map on the nodal velocity vector
    P_twin = columns(t.project_bdm_nodal, nq, t.shape_Q)
    P_ref = np.stack([d.project_bdm(eye_q[:, c].reshape(t.shape_Q)).ravel() for c in range(nq)], axis=1)
    assert np.max(np.abs(P_twin - P_ref)) < 1e-11 * max(1.0, np.max(np.abs(P_ref)))
    # advection operator I - gamma M^-1 F(Q*) for a fixed random conforming Q*
    rng = np.random.default_rng(11)
    Qstar = d.project_bdm(rng.standard_normal(t.shape_Q))
    Mi = spla.splu(d.MQ.tocsc())
    gamma = 0.2 / nx
    for flux in ("upwind", "centered"):
        tf = _twin(k, nx, flux=flux)
        A_twin = columns(lambda x: tf.apply_advection(Qstar, x, gamma), nq, t.shape_Q)
        A_ref = np.eye(nq) - gamma * Mi.solve(d.assemble_f_impl(Qstar, flux).toarray())
        assert np.max(np.abs(A_twin - A_ref)) < 5e-11 * np.max(np.abs(A_ref)), flux
    # weak divergence
    W_twin = columns(t.apply_weak_divergence, nq, t.shape_Q)
    W_ref = spla.splu(d.MP.tocsc()).solve(d.Wdiv.toarray())
    assert np.max(np.abs(W_twin - W_ref)) < 1e-11 * np.max(np.abs(W_ref))
    # condensed trace operator -S in the Riesz representation of the orthonormal edge basis (single edge mass matrix)
    n1 = d.NQ + d.NP
    Kmp = d.K_mp.tocsc()
    S = Kmp[n1:, n1:].toarray() - Kmp[n1:, :n1] @ spla.splu(Kmp[:n1, :n1].tocsc()).solve(Kmp[:n1, n1:].toarray())
    mult = np.where(np.repeat(d.mesh.interior, d.nl), 2.0, 1.0)
    Mtr = (sp.diags(1.0 / mult) @ (d.Lm.tocsc() / d.tau)).tocsc()
    T_ref = spla.splu(Mtr).solve(-S)
    nl = int(np.prod(t.shape_l))
    T_twin = columns(t.apply_trace_operator, nl, t.shape_l)
    assert np.max(np.abs(T_twin - T_ref)) < 1e-10 * np.max(np.abs(T_ref))
