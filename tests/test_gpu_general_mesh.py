"""General affine triangulations in the PRODUCT (UnitDiskMesh / Kelvin-Helmholtz,
src/driver.py:184-185, src/model_problems.py:108-131): the HIP path with per-element geometry (hdg_create_general) against
the numpy oracle on the same triangulation (oracle/fem.py TriMesh) -- on the unit disk, on the structured mesh handed over
as a general mesh, and on a deliberately irregular mesh with mixed cell orientations.

Operators at round-off level, whole HDG-IMEX steps at the two-converged-solvers tolerance.  The
reference's UnitDiskMesh construction is restated from memory on both sides (Firedrake cannot run here): parity unpinned."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu
RTOL = 1e-10
TOL = 2e-8


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _structured(nx):
    xs = np.linspace(0.0, 1.0, nx + 1)
    X = np.array([(x, y) for y in xs for x in xs])
    vid = lambda i, j: j * (nx + 1) + i
    cells = []
    for j in range(nx):
        for i in range(nx):
            cells.append((vid(i, j), vid(i + 1, j), vid(i, j + 1)))
            cells.append((vid(i + 1, j + 1), vid(i, j + 1), vid(i + 1, j)))
    return X, np.array(cells)


def _irregular():
    """a small mesh with a perturbed interior vertex, a clockwise cell and cells whose first vertex differs"""
    X = np.array([[0.0, 0.0], [1.0, 0.0], [2.1, -0.1], [0.1, 1.0], [1.2, 0.9], [2.0, 1.1], [0.0, 2.0], [0.9, 2.1], [2.2, 2.0]])
    C = np.array([[0, 1, 4], [0, 4, 3], [1, 2, 4], [4, 2, 5], [3, 4, 7], [6, 3, 7], [7, 4, 8], [5, 8, 4]])
    C[3] = C[3][[0, 2, 1]]  # clockwise
    return X, C


def _mesh(kind):
    from incompressibleeulerhdg_amd.mesh import TriangleMesh, UnitDiskMesh
    from oracle import fem

    if kind == "disk1":
        pm = UnitDiskMesh(1)
    elif kind == "disk2":
        pm = UnitDiskMesh(2)
    elif kind == "square4":
        pm = TriangleMesh(*_structured(4))
    else:
        pm = TriangleMesh(*_irregular())
    return pm, fem.TriMesh(pm.vertices, pm.cells)


def _engine(pm, k, dt=0.01, **kw):
    from incompressibleeulerhdg_amd._lib import Engine
    from oracle.hdg_oracle import TABLEAUX

    tb = TABLEAUX["imex_ssp2_332"]
    return Engine(vertices=pm.vertices, cells=pm.cells, degree=k, dt=dt, nstages=3, a_expl=tb["a_expl"], a_impl=tb["a_impl"],
                  b_expl=tb["b_expl"], b_impl=tb["b_impl"], c_expl=tb["c_expl"], **kw)


@pytest.mark.parametrize("kind,k", [("irregular", 1), ("irregular", 2), ("disk1", 2), ("square4", 2), ("disk1", 3), ("irregular", 4)])
def test_general_mesh_operators(hip_lib, kind, k):
    from incompressibleeulerhdg_amd import _lib
    from oracle.hdg_oracle import HDGDiscretisation

    pm, om = _mesh(kind)
    d = HDGDiscretisation(0, k, mesh=om)
    e = _engine(pm, k)
    assert (e.n_cells, e.n_edges, e.n_u, e.n_p, e.n_l) == (om.ncells, om.nedges, d.nu, d.np_, d.nl)
    ev, ec = e.general_topology()
    assert np.array_equal(ev, om.edge_vertices) and np.array_equal(ec[:, 0], om.edge_plus) and np.array_equal(ec[:, 1], om.edge_minus)
    xq, xp = e.node_coordinates()
    assert np.allclose(xq, d.node_coords(d.PU).reshape(-1, 2), atol=1e-13) and np.allclose(xp, d.node_coords(d.PP).reshape(-1, 2), atol=1e-13)
    rng = np.random.default_rng(100 + k)
    Q, x = rng.standard_normal(e.shape_Q), rng.standard_normal(e.shape_Q)
    p, lam = rng.standard_normal(e.shape_p), rng.standard_normal(e.shape_l)
    # conversions, norms, integrals
    e.set_field(1, Q, p, lam)
    Q2, p2, l2 = e.get_field(1)
    assert _rel(Q2, Q) < 1e-12 and _rel(p2, p) < 1e-12 and _rel(l2, lam) < 1e-12
    nq, npr = e.l2_norms(Q, p)
    assert abs(nq - d.l2_norm_velocity(Q)) < 1e-11 * nq and abs(npr - d.l2_norm_pressure(p)) < 1e-11 * npr
    assert abs(e.integrate_pressure(p) - d.int_p @ p) < 1e-12 * max(1.0, abs(d.int_p @ p))
    # BDM projection (common.py:91-108)
    assert _rel(e.project_bdm_nodal(Q), d.project_bdm(Q)) < RTOL
    # advection operator (hdg_imex.py:313-331), both fluxes
    Qstar = d.project_bdm(Q)
    gamma = 0.05
    for flux in ("upwind", "centered"):
        ef = _engine(pm, k, flux=flux)
        F = d.assemble_f_impl(Qstar, flux)
        ref = x.ravel() - gamma * spla.spsolve(d.MQ.tocsc(), F @ x.ravel())
        assert _rel(ef.apply_advection(Qstar, x, gamma).ravel(), ref) < RTOL, flux
    # weak / broken divergence
    Mi = spla.splu(d.MP.tocsc())
    assert _rel(e.apply_weak_divergence(Q), Mi.solve(d.Wdiv @ Q.ravel())) < RTOL
    assert _rel(e.apply_weak_divergence(Q, broken=True), Mi.solve(d.Bdiv @ Q.ravel())) < RTOL
    # condensed trace operator vs the oracle's dense Schur complement
    n1 = d.NQ + d.NP
    Kmp = d.K_mp.tocsc()
    S = Kmp[n1:, n1:].toarray() - Kmp[n1:, :n1] @ spla.splu(Kmp[:n1, :n1].tocsc()).solve(Kmp[n1:, :n1].T.toarray() * 0 + Kmp[:n1, n1:].toarray())
    mult = np.where(np.repeat(om.interior, d.nl), 2.0, 1.0)
    Mtr = (sp.diags(1.0 / mult) @ (d.Lm.tocsc() / d.tau)).tocsc()
    assert _rel(e.apply_trace_operator(lam), spla.spsolve(Mtr, -S @ lam)) < 1e-9
    assert np.max(np.abs(e.apply_trace_operator(np.ones(e.shape_l)))) < 1e-9
    # trace reconstruction and the pressure shift
    e.set_state(Q, p)
    e.reconstruct_trace()
    _, p_dev, l_dev = e.get_field(_lib.HDG_STATE_CURRENT)
    p0 = p - (d.int_p @ p) / om.volume
    assert _rel(p_dev, p0) < RTOL and _rel(l_dev, d.reconstruct_trace(Q, p0)) < RTOL
    e.set_field(1, Q, p, lam)
    e.shift_pressure(1)
    _, p1, l1 = e.get_field(1)
    ps, ls = d.shift_pressure(p, lam)
    assert _rel(p1, ps) < RTOL and _rel(l1, ls) < RTOL


def _smooth(seed):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1.0, 1.0, size=(3, 6))
    Q0 = lambda x, y: (a[0, 0] * np.sin(2 * x + a[0, 1]) * np.cos(1.5 * y) + a[0, 2] * y, a[0, 3] * np.cos(1.7 * x) * np.sin(2 * y + a[0, 4]) + a[0, 5] * x)
    p0 = lambda x, y: a[1, 0] * np.cos(2 * x + a[1, 1]) * np.sin(1.3 * y + a[1, 2])
    f = lambda t: (lambda x, y: (a[2, 0] * np.sin(3 * t + x + a[2, 1] * y), a[2, 2] * np.cos(2 * t - y + a[2, 3] * x)))
    return Q0, p0, f


@pytest.mark.parametrize("kind,k,tableau,R", [("disk1", 1, "imex_ssp2_332", 2), ("irregular", 2, "imex_ssp2_332", 2), ("disk1", 2, "imex_ars3_443", 1),
                                              ("square4", 1, "imex_ssp3_433", 2), ("disk2", 1, "imex_ssp2_332", 2)])
def test_general_mesh_whole_steps_on_smooth_random_data(hip_lib, kind, k, tableau, R):
    """Two HDG-IMEX steps on data for which nothing cancels (velocity neither divergence free nor tangential, unrelated
    pressure, time-dependent non-gradient forcing): every piece of the step -- residuals, tentative velocity (GMRES + element
    block-Jacobi), condensation / CG / back-substitution, pressure reconstruction with its boundary term -- against the
    oracle's direct solves on the same triangulation; per-solve path and fused path."""
    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd import timesteppers as ts_mod
    from oracle import hdg_oracle as orc

    cls = {"imex_ssp2_332": ts_mod.IncompressibleEulerHDGIMEXSSP2_332, "imex_ars3_443": ts_mod.IncompressibleEulerHDGIMEXARS3_443,
           "imex_ssp3_433": ts_mod.IncompressibleEulerHDGIMEXSSP3_433}[tableau]
    pm, om = _mesh(kind)
    dt, nsteps = 0.02, 2
    Q0, p0, f = _smooth(7 + k)
    d = orc.HDGDiscretisation(0, k, mesh=om)
    o = orc.OracleHDGIMEX(d, dt, tableau, n_richardson=R)
    oQ, op = o.solve(d.interpolate_velocity(Q0), d.interpolate_pressure(p0), lambda t: d.interpolate_velocity(f(t)), nsteps * dt)
    assert _rel(oQ, d.interpolate_velocity(Q0)) > 1e-3  # the step really changes the velocity
    for fused in (False, True):
        ts = cls(pm, k, dt, use_projection_method=True, n_richardson=R)
        Q, p = ts.solve(Q0, p0, None, f, nsteps * dt, fused=fused)
        lam = ts._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
        assert _rel(Q.dat.data, oQ) < TOL and _rel(p.dat.data, op) < TOL and _rel(lam, o.lam) < TOL, fused
        sQ, sp_, sl = ts._engine.get_field(ts.nstages - 1)
        assert _rel(sQ, o.stage_Q[-1]) < TOL and _rel(sp_, o.stage_p[-1]) < TOL and _rel(sl, o.stage_l[-1]) < TOL


def test_kelvin_helmholtz_on_the_unit_disk(hip_lib):
    """The reference's third set-up end to end: UnitDiskMesh(refinement_level) + KelvinHelmholtz (driver.py:184-185,336-337;
    model_problems.py:108-131) through the product's classes, against the oracle; plus the invariants of the discretisation on
    non-uniform triangles: zero-mean pressure, Q* a fixed point of the projection."""
    from incompressibleeulerhdg_amd.mesh import UnitDiskMesh
    from incompressibleeulerhdg_amd.model_problems import KelvinHelmholtz
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332
    from oracle import fem
    from oracle import hdg_oracle as orc

    level, k = 2, 1
    pm = UnitDiskMesh(level)
    assert pm.num_cells() == 8 * 4 ** level and 2.8 < pm.volume < np.pi
    dt = 0.0125
    ts = IncompressibleEulerHDGIMEXSSP2_332(pm, k, dt, use_projection_method=True, n_richardson=2)
    kh = KelvinHelmholtz(ts._V_Q, ts._V_p)
    Q, p = ts.solve(*kh.initial_condition(), None, kh.f_rhs(), 2 * dt, fused=True)
    om = fem.unit_disk_mesh(level)
    d = orc.HDGDiscretisation(0, k, mesh=om)
    okh = orc.KelvinHelmholtz(d)
    oQ, op = orc.OracleHDGIMEX(d, dt, "imex_ssp2_332").solve(*okh.initial_condition(), okh.f_rhs, 2 * dt)
    assert _rel(Q.dat.data, oQ) < TOL and np.max(np.abs(p.dat.data - op)) < TOL * max(np.max(np.abs(op)), 1.0)
    e = ts._engine
    assert abs(e.integrate_pressure(p.dat.data)) < 1e-10
    Qs = e.project_bdm_nodal(Q.dat.data)
    assert _rel(e.project_bdm_nodal(Qs), Qs) < 1e-10
    sums, cnt = e.iteration_stats()
    assert np.all(cnt > 0) and np.all(sums / cnt < 400)


def test_driver_runs_kelvin_helmholtz(hip_lib, capsys, tmp_path):
    from incompressibleeulerhdg_amd import driver

    driver.main(["--problem", "kelvinhelmholtz", "--refinement", "1", "--degree", "1", "--dt", "0.02", "--tfinal", "0.04",
                 "--use_projection_method", "--fused", "--output", str(tmp_path / "solution.pvd")])
    out = capsys.readouterr().out
    assert "model problem = kelvinhelmholtz" in out and "refinement level = 1" in out
    assert (tmp_path / "solution.pvd").exists()


def test_general_mesh_preconditioners_are_mesh_independent(hip_lib, monkeypatch):
    """The two-level preconditioners of the general path (hdg_amg.hpp; hdg_imex.py:139-167 GTMG + GAMG, :224-228 for the
    tentative velocity): the P1 coarse space with its algebraic V-cycle keeps the CG count of the condensed system bounded
    under refinement of the disk, the edge block-Jacobi alone doubles it per level; the hybrid preconditioner of the
    tentative velocity at least halves the GMRES count of the element block-Jacobi; and none of them changes the answer."""
    from incompressibleeulerhdg_amd.mesh import UnitDiskMesh
    from incompressibleeulerhdg_amd.model_problems import KelvinHelmholtz
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    def run(level, k, env):
        for name in ("HDG_GENERAL_NO_COARSE", "HDG_GENERAL_BLOCK_JACOBI"):
            monkeypatch.delenv(name, raising=False)
        for name in env:
            monkeypatch.setenv(name, "1")
        ts = IncompressibleEulerHDGIMEXSSP2_332(UnitDiskMesh(level), k, 0.005, use_projection_method=True, n_richardson=2)
        kh = KelvinHelmholtz(ts._V_Q, ts._V_p)
        Q, p = ts.solve(*kh.initial_condition(), None, kh.f_rhs(), 0.01, fused=True)
        sums, cnt = ts._engine.iteration_stats()
        return Q.dat.data.copy(), p.dat.data.copy(), sums / cnt

    its = {}
    for level in (2, 3, 4):
        Q1, p1, its[level, "two-level"] = run(level, 1, ())
        Q0, p0, its[level, "one-level"] = run(level, 1, ("HDG_GENERAL_NO_COARSE", "HDG_GENERAL_BLOCK_JACOBI"))
        assert _rel(Q1, Q0) < TOL and np.max(np.abs(p1 - p0)) < TOL * max(np.max(np.abs(p0)), 1.0)
    cg = lambda level, which: its[level, which][1]      # tentative, pressure, final pressure, reconstruction
    gm = lambda level, which: its[level, which][0]
    assert cg(4, "two-level") < 20 and cg(4, "two-level") < cg(2, "two-level") + 8
    assert cg(4, "one-level") > 3.0 * cg(2, "one-level") and cg(4, "one-level") > 8 * cg(4, "two-level")
    assert gm(4, "two-level") < 0.5 * gm(4, "one-level")
    # k = 2: same behaviour
    _, _, i2 = run(3, 2, ())
    assert i2[1] < 20


@pytest.mark.parametrize("kind,k", [("disk1", 1), ("irregular", 2)])
def test_general_mesh_fully_implicit_stepper(hip_lib, kind, k):
    """IncompressibleEulerHDGImplicit (hdg_implicit.py:92-190, projection branch) on a general triangulation: two steps on
    smooth random data against the oracle's direct solves."""
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGImplicit
    from oracle import hdg_oracle as orc

    pm, om = _mesh(kind)
    dt = 0.02
    Q0, p0, f = _smooth(21 + k)
    d = orc.HDGDiscretisation(0, k, mesh=om)
    oQ, op = orc.OracleHDGImplicit(d, dt).solve(d.interpolate_velocity(Q0), d.interpolate_pressure(p0),
                                                lambda t: d.interpolate_velocity(f(t)), 2 * dt)
    ts = IncompressibleEulerHDGImplicit(pm, k, dt, flux="upwind", use_projection_method=True, n_richardson=2)
    Q, p = ts.solve(Q0, p0, None, f, 2 * dt)
    assert _rel(Q.dat.data, oQ) < TOL and _rel(p.dat.data, op) < TOL


@pytest.mark.parametrize("kind,k,stepper", [("disk1", 1, "implicit"), ("irregular", 2, "implicit"), ("disk1", 2, "imex_ssp2_332"),
                                            ("square4", 1, "imex_ars3_443")])
def test_general_mesh_monolithic_solves(hip_lib, kind, k, stepper):
    """use_projection_method=False -- the reference driver's DEFAULT (driver.py:97-102) -- on a general triangulation: the
    monolithic (u, phi, lambda) stage system (hdg_imex.py:600-620, hdg_implicit.py:151-186) through flexible GMRES with the
    block preconditioner (a second set of tau-dependent operators and its own coarse hierarchy), against the oracle's
    bordered sparse LU."""
    from incompressibleeulerhdg_amd import timesteppers as ts_mod
    from oracle import hdg_oracle as orc

    pm, om = _mesh(kind)
    dt = 0.02
    Q0, p0, f = _smooth(31 + k)
    d = orc.HDGDiscretisation(0, k, mesh=om)
    fo = lambda t: d.interpolate_velocity(f(t))
    if stepper == "implicit":
        oQ, op = orc.OracleHDGImplicit(d, dt, use_projection_method=False).solve(d.interpolate_velocity(Q0), d.interpolate_pressure(p0), fo, 2 * dt)
        ts = ts_mod.IncompressibleEulerHDGImplicit(pm, k, dt, use_projection_method=False)
    else:
        oQ, op = orc.OracleHDGIMEX(d, dt, stepper, use_projection_method=False).solve(d.interpolate_velocity(Q0), d.interpolate_pressure(p0), fo, 2 * dt)
        cls = {"imex_ssp2_332": ts_mod.IncompressibleEulerHDGIMEXSSP2_332, "imex_ars3_443": ts_mod.IncompressibleEulerHDGIMEXARS3_443}[stepper]
        ts = cls(pm, k, dt, use_projection_method=False)
    Q, p = ts.solve(Q0, p0, None, f, 2 * dt)
    assert _rel(Q.dat.data, oQ) < TOL and _rel(p.dat.data, op) < TOL
    sums, cnt = ts._engine.iteration_stats()
    assert np.all(sums[cnt > 0] / cnt[cnt > 0] < 200)


@pytest.mark.parametrize("kind,k", [("disk1", 1), ("irregular", 2), ("disk2", 2), ("square4", 3), ("irregular", 4)])
def test_general_mesh_continuous_space_and_tracer_operator(hip_lib, kind, k):
    """CG_{k+1} on a general triangulation (common.py:110-129, callbacks.py:43-69): dof set, L2 projection of a broken velocity,
    vorticity and the tracer transport operator against the oracle's restatement (oracle/tracer_oracle.py on the TriMesh)."""
    from oracle.hdg_oracle import HDGDiscretisation
    from oracle.tracer_oracle import TracerOracle

    pm, om = _mesh(kind)
    d = HDGDiscretisation(0, k, mesh=om)
    tr = TracerOracle(d)
    e = _engine(pm, k)
    assert e.cg_size() == tr.ncg
    key = lambda X: {tuple(np.round(x * 1e6).astype(np.int64)) for x in X}
    assert key(e.cg_coordinates()) == key(tr.cg_coords) and len(key(e.cg_coordinates())) == tr.ncg
    rng = np.random.default_rng(50 + k)
    u = rng.standard_normal(e.shape_Q)
    P = e.cg_project_nodal(u)
    assert _rel(P, tr.cg_project(u)) < 1e-10
    assert _rel(e.cg_project_nodal(P), P) < 1e-10
    cont = d.interpolate_velocity(lambda x, y: (x ** (k + 1) - y, x * y ** k + 1.0))
    assert _rel(e.cg_project_nodal(cont), cont) < 1e-10
    Q = rng.standard_normal(e.shape_Q)
    w, xy = tr.vorticity(Q)
    wd = e.vorticity(Q)
    order = lambda X: np.lexsort((np.round(X[:, 1] * 1e6), np.round(X[:, 0] * 1e6)))
    assert _rel(wd[order(e.cg_coordinates())], w[order(xy)]) < 1e-10
    assert _rel(e.cg_to_broken(wd), tr.R @ w) < 1e-10
    # rigid rotation (-y, x): vorticity 2 everywhere, also on the polygonal boundary
    rot = d.interpolate_velocity(lambda x, y: (-y, x))
    assert np.max(np.abs(e.vorticity(rot) - 2.0)) < 1e-9
    q = rng.standard_normal(e.shape_p)
    assert _rel(e.apply_tracer_advection(q, u, project=True), tr.tracer_tendency(q, u)) < 1e-10
    uc = tr.cg_project(u)
    assert _rel(e.apply_tracer_advection(q, uc, project=False), tr._lu_mp.solve(tr.tracer_form(q, uc))) < 1e-10


@pytest.mark.parametrize("kind,k,tableau", [("disk1", 1, "imex_ssp2_332"), ("irregular", 2, "imex_ars3_443"), ("disk1", 2, "implicit")])
def test_general_mesh_steps_with_tracer(hip_lib, kind, k, tableau):
    """Whole steps with a passive tracer on a general triangulation (hdg_imex.py:415-448,560,622-623,638-639;
    hdg_implicit.py:93-96,192-193) against the oracle."""
    from incompressibleeulerhdg_amd import timesteppers as tsm
    from oracle import hdg_oracle as orc
    from oracle.tracer_oracle import TracerOracle, imex_with_tracer, implicit_with_tracer

    pm, om = _mesh(kind)
    dt = 0.02
    Q0, p0, f = _smooth(41 + k)
    q0 = lambda x, y: np.sin(1.5 * x + 0.3) * np.cos(1.2 * y) + 0.2 * x
    d = orc.HDGDiscretisation(0, k, mesh=om)
    tr = TracerOracle(d)
    fo = lambda t: d.interpolate_velocity(f(t))
    args = (d.interpolate_velocity(Q0), d.interpolate_pressure(p0), d.interpolate_pressure(q0), fo, 2 * dt)
    if tableau == "implicit":
        oQ, op, oq = implicit_with_tracer(d, tr, dt, *args)
        ts = tsm.IncompressibleEulerHDGImplicit(pm, k, dt, use_projection_method=True)
    else:
        oQ, op, oq = imex_with_tracer(orc.OracleHDGIMEX(d, dt, tableau), tr, *args)
        cls = {"imex_ssp2_332": tsm.IncompressibleEulerHDGIMEXSSP2_332, "imex_ars3_443": tsm.IncompressibleEulerHDGIMEXARS3_443}[tableau]
        ts = cls(pm, k, dt, use_projection_method=True)
    Q, p = ts.solve(Q0, p0, q0, f, 2 * dt)
    assert _rel(Q.dat.data, oQ) < TOL and _rel(p.dat.data, op) < TOL and _rel(ts.q_tracer.dat.data, oq) < TOL
    assert _rel(oq, d.interpolate_pressure(q0)) > 1e-3  # the tracer moved


def test_driver_kelvin_helmholtz_with_tracer_and_animation(hip_lib, tmp_path, capsys, monkeypatch):
    """The reference driver's Kelvin-Helmholtz run with its DEFAULT (monolithic) solve, --animation (CG vorticity) and
    --tracer_advection on the disk (driver.py:97-102,165-176,184-187,336-344)."""
    from incompressibleeulerhdg_amd import driver

    monkeypatch.chdir(tmp_path)
    driver.main(["--problem", "kelvinhelmholtz", "--refinement", "1", "--degree", "1", "--dt", "0.02", "--tfinal", "0.04",
                 "--tracer_advection", "--animation", "--output", ""])
    out = capsys.readouterr().out
    assert "use projection method = False" in out and "advect tracer = True" in out
    assert (tmp_path / "evolution.pvd").read_text().count("<DataSet") == 3
    vtu = (tmp_path / "evolution_2.vtu").read_text()
    for name in ('Name="Q"', 'Name="p"', 'Name="vorticity"', 'Name="tracer"'):
        assert name in vtu, name
