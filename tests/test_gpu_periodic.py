"""Doubly periodic square (PeriodicSquareMesh(nx, nx, L = 2 pi), src/driver.py:182-183) and the double-layer shear flow
(src/model_problems.py:134-196): SURVEY.md section 8(f) row 2, first step.  Operators and whole steps against the oracle
on the periodic mesh (oracle/fem.py: Mesh(periodic=True)); same tolerances as on the unit square."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu
TOL = 2e-8
L = 2 * np.pi


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _setup(k, nx, tableau="imex_ssp2_332", **kw):
    from incompressibleeulerhdg_amd._lib import Engine
    from oracle.hdg_oracle import TABLEAUX, HDGDiscretisation

    d = HDGDiscretisation(nx, k, periodic=True, L=L)
    tb = TABLEAUX[tableau]
    e = Engine(nx=nx, degree=k, dt=0.25 * L / nx, nstages=len(tb["c_expl"]), a_expl=tb["a_expl"], a_impl=tb["a_impl"], b_expl=tb["b_expl"],
               b_impl=tb["b_impl"], c_expl=tb["c_expl"], periodic=True, length=L, **kw)
    return d, e


@pytest.mark.parametrize("k,nx", [(1, 4), (1, 6), (2, 4), (3, 4), (2, 10)])
def test_periodic_operators(hip_lib, k, nx):
    from incompressibleeulerhdg_amd import _lib

    d, e = _setup(k, nx)
    assert e.n_cells == d.mesh.ncells and e.n_edges == d.mesh.nedges == 3 * nx * nx
    xq, xp = e.node_coordinates()
    assert np.allclose(xq, d.node_coords(d.PU).reshape(-1, 2), atol=1e-13) and np.allclose(xp, d.node_coords(d.PP).reshape(-1, 2), atol=1e-13)
    rng = np.random.default_rng(21)
    Q, x = rng.standard_normal(e.shape_Q), rng.standard_normal(e.shape_Q)
    p, lam = rng.standard_normal(e.shape_p), rng.standard_normal(e.shape_l)
    assert _rel(e.project_bdm_nodal(Q), d.project_bdm(Q)) < 1e-11
    Qstar = d.project_bdm(Q)
    gamma = 0.3 * d.mesh.h
    for flux in ("upwind", "centered"):
        _, ef = _setup(k, nx, flux=flux)
        F = d.assemble_f_impl(Qstar, flux)
        ref = x.ravel() - gamma * spla.spsolve(d.MQ.tocsc(), F @ x.ravel())
        assert _rel(ef.apply_advection(Qstar, x, gamma).ravel(), ref) < 5e-11
    Mi = spla.splu(d.MP.tocsc())
    assert _rel(e.apply_weak_divergence(Q), Mi.solve(d.Wdiv @ Q.ravel())) < 1e-11
    # condensed operator against the oracle's dense Schur complement
    n1 = d.NQ + d.NP
    K = d.K_mp.tocsc()
    S = K[n1:, n1:].toarray() - K[n1:, :n1] @ spla.splu(K[:n1, :n1].tocsc()).solve(K[:n1, n1:].toarray())
    Mtr = (d.Lm.tocsc() / d.tau) @ sp.diags(np.full(d.NL, 0.5))  # every edge is interior: single edge mass
    assert _rel(e.apply_trace_operator(lam), spla.spsolve(Mtr.tocsc(), -S @ lam)) < 1e-10
    assert np.max(np.abs(e.apply_trace_operator(np.ones(e.shape_l)))) < 1e-9
    # trace reconstruction and pressure shift with the domain volume L^2
    e.set_state(Q, p)
    e.reconstruct_trace()
    _, p_dev, l_dev = e.get_field(_lib.HDG_STATE_CURRENT)
    p0 = p - (d.int_p @ p) / d.mesh.volume
    assert _rel(p_dev, p0) < 1e-11 and _rel(l_dev, d.reconstruct_trace(Q, p0)) < 1e-11


def _shear(rho=np.pi / 15, delta=0.05):
    Q0 = lambda x, y: (np.where(y <= np.pi, np.tanh((y - np.pi / 2) / rho), np.tanh((1.5 * np.pi - y) / rho)), delta * np.sin(x))
    p0 = lambda x, y: delta * np.cos(x) * np.sin(y - np.pi) * 0.3
    return Q0, p0


@pytest.mark.parametrize("k,nx,tableau", [(1, 6, "imex_ssp2_332"), (2, 4, "imex_ssp2_332"), (1, 4, "imex_ars3_443"), (2, 6, "imex_ars2_232")])
def test_periodic_imex_steps(hip_lib, k, nx, tableau):
    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd import timesteppers as tsm
    from incompressibleeulerhdg_amd.mesh import PeriodicSquareMesh
    from oracle import hdg_oracle as orc

    cls = {"imex_ssp2_332": tsm.IncompressibleEulerHDGIMEXSSP2_332, "imex_ars3_443": tsm.IncompressibleEulerHDGIMEXARS3_443,
           "imex_ars2_232": tsm.IncompressibleEulerHDGIMEXARS2_232}[tableau]
    d = orc.HDGDiscretisation(nx, k, periodic=True, L=L)
    dt, nsteps = 0.25 * d.mesh.h, 2
    Q0, p0 = _shear()
    f = lambda t: (lambda x, y: (0.1 * np.cos(y) * np.cos(t), 0.2 * np.sin(x + y)))
    o = orc.OracleHDGIMEX(d, dt, tableau)
    oQ, op = o.solve(d.interpolate_velocity(Q0), d.interpolate_pressure(p0), lambda t: d.interpolate_velocity(f(t)), nsteps * dt)
    for fused in (False, True):
        ts = cls(PeriodicSquareMesh(nx, nx, L=L), k, dt)
        Q, p = ts.solve(Q0, p0, None, f, nsteps * dt, fused=fused)
        lam = ts._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
        assert _rel(Q.dat.data, oQ) < TOL and _rel(p.dat.data, op) < TOL and _rel(lam, o.lam) < TOL, fused
    assert _rel(oQ, d.interpolate_velocity(Q0)) > 1e-3


@pytest.mark.parametrize("k,nx", [(2, 16), (1, 18), (3, 16)])
def test_periodic_steps_through_the_tiled_trace_preconditioner(hip_lib, k, nx):
    """The LDS-tiled trace preconditioner on the periodic square (hdg_trace_tile.hpp: wrapped columns, the strip's own
    opposite rows as ghost rows; nx >= 16): whole IMEX steps against the oracle, on a mesh of whole tiles and on one whose
    last tile column is partial (18 columns: wrapped halo columns and masked surplus columns in the same tile)."""
    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd import timesteppers as tsm
    from incompressibleeulerhdg_amd.mesh import PeriodicSquareMesh
    from oracle import hdg_oracle as orc

    d = orc.HDGDiscretisation(nx, k, periodic=True, L=L)
    dt, nsteps = 0.25 * d.mesh.h, 2
    Q0, p0 = _shear()
    f = lambda t: (lambda x, y: (0.1 * np.cos(y) * np.cos(t), 0.2 * np.sin(x + y)))
    o = orc.OracleHDGIMEX(d, dt, "imex_ssp2_332")
    oQ, op = o.solve(d.interpolate_velocity(Q0), d.interpolate_pressure(p0), lambda t: d.interpolate_velocity(f(t)), nsteps * dt)
    ts = tsm.IncompressibleEulerHDGIMEXSSP2_332(PeriodicSquareMesh(nx, nx, L=L), k, dt)
    assert ts._engine.kernel_forms()["trace_precond"] == 1  # the tile kernels
    Q, p = ts.solve(Q0, p0, None, f, nsteps * dt, fused=True)
    lam = ts._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
    assert _rel(Q.dat.data, oQ) < TOL and _rel(p.dat.data, op) < TOL and _rel(lam, o.lam) < TOL
    sums, cnt = ts._engine.iteration_stats()
    assert np.all((sums / np.maximum(cnt, 1))[1:] < 25)


@pytest.mark.parametrize("k,nx", [(2, 64), (1, 128), (2, 80)])
def test_periodic_fused_vcycle_legs_equal_the_per_level_kernels(hip_lib, tmp_path, k, nx):
    """The vertex-grid V-cycle of the periodic square with the fused LDS-tile legs (k_p1_down / k_p1_up, PER = true: wrapped
    loads, every vertex interior) for n > 32 and the tail n <= 32 as one dense product, carrying the p / x half of the CG update
    as side jobs -- against the per-level kernels (HDG_MG_NO_FUSE, read once per process: two worker processes).  The legs form
    every vertex value by the same expression in the same order, the dense tail is the same linear map to rounding: fields at
    1e-9, same CG counts.  64: one leg level above the tail; 128: two; 80: tiles that wrap inside their halo (80 = 2.5 tiles)."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for tag, env in (("fused", {}), ("levels", {"HDG_MG_NO_FUSE": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, os.path.join(here, "periodic_worker.py"), str(k), str(nx), "2", out], env=dict(os.environ, **env),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        res[tag] = np.load(out)
    for name in ("Q", "p", "lam"):
        assert _rel(res["fused"][name], res["levels"][name]) < 1e-9, name
    assert np.all(np.abs(res["fused"]["its"][1:] - res["levels"]["its"][1:]) <= 1.0), (res["fused"]["its"], res["levels"]["its"])


@pytest.mark.parametrize("k,nx", [(1, 4), (2, 4), (3, 4), (2, 6)])
def test_periodic_continuous_space_tracer_operator_vorticity(hip_lib, k, nx):
    """CG_{k+1} on the periodic square (nx x ny corners, indices wrap): projection (common.py:119-122), tracer transport
    form (common.py:110-129) and vorticity (callbacks.py:43-69) against oracle/tracer_oracle.py on the periodic mesh."""
    from oracle.tracer_oracle import TracerOracle

    d, e = _setup(k, nx)
    tr = TracerOracle(d)
    assert e.cg_size() == tr.ncg == ((k + 1) * nx) ** 2
    key = lambda X: {tuple(np.round(x * nx / L * 1e6).astype(np.int64)) for x in X}
    assert key(e.cg_coordinates()) == key(tr.cg_coords) and len(key(e.cg_coordinates())) == tr.ncg
    rng = np.random.default_rng(5)
    u = rng.standard_normal(e.shape_Q)
    P = e.cg_project_nodal(u)
    assert _rel(P, tr.cg_project(u)) < 1e-10 and _rel(e.cg_project_nodal(P), P) < 1e-10
    cont = d.interpolate_velocity(lambda x, y: (np.sin(x) * np.cos(2 * y), np.cos(x + y)))  # smooth and periodic
    assert _rel(e.cg_project_nodal(cont), tr.cg_project(cont)) < 1e-10
    q = rng.standard_normal(e.shape_p)
    assert _rel(e.apply_tracer_advection(q, u, project=True), tr.tracer_tendency(q, u)) < 1e-10
    uc = tr.cg_project(u)
    assert _rel(e.apply_tracer_advection(q, uc, project=False), tr._lu_mp.solve(tr.tracer_form(q, uc))) < 1e-10
    Q = rng.standard_normal(e.shape_Q)
    w, xy = tr.vorticity(Q)
    wd = e.vorticity(Q)
    order = lambda X: np.lexsort((np.round(X[:, 1] * nx / L * 1e6), np.round(X[:, 0] * nx / L * 1e6)))
    assert _rel(wd[order(e.cg_coordinates())], w[order(xy)]) < 1e-10
    assert _rel(e.cg_to_broken(wd), tr.R @ w) < 1e-10


@pytest.mark.parametrize("k,nx,tableau", [(1, 6, "imex_ssp2_332"), (2, 4, "imex_ars3_443")])
def test_periodic_imex_steps_with_tracer(hip_lib, k, nx, tableau):
    """Shear flow + passive tracer on the periodic square (driver.py:182-183,340-344), fused and per-call paths."""
    from incompressibleeulerhdg_amd import timesteppers as tsm
    from incompressibleeulerhdg_amd.mesh import PeriodicSquareMesh
    from oracle import hdg_oracle as orc
    from oracle.tracer_oracle import TracerOracle, imex_with_tracer

    cls = {"imex_ssp2_332": tsm.IncompressibleEulerHDGIMEXSSP2_332, "imex_ars3_443": tsm.IncompressibleEulerHDGIMEXARS3_443}[tableau]
    d = orc.HDGDiscretisation(nx, k, periodic=True, L=L)
    dt, nsteps = 0.25 * d.mesh.h, 2
    Q0, p0 = _shear()
    q0 = lambda x, y: np.sin(x) * np.sin(y)
    f = lambda t: (lambda x, y: (0.1 * np.cos(y) * np.cos(t), 0.2 * np.sin(x + y)))
    o = orc.OracleHDGIMEX(d, dt, tableau)
    oQ, op, oq = imex_with_tracer(o, TracerOracle(d), d.interpolate_velocity(Q0), d.interpolate_pressure(p0), d.interpolate_pressure(q0),
                                  lambda t: d.interpolate_velocity(f(t)), nsteps * dt)
    for fused in (False, True):
        ts = cls(PeriodicSquareMesh(nx, nx, L=L), k, dt)
        Q, p = ts.solve(Q0, p0, q0, f, nsteps * dt, fused=fused)
        assert _rel(Q.dat.data, oQ) < TOL and _rel(p.dat.data, op) < TOL and _rel(ts.q_tracer.dat.data, oq) < TOL, fused
    assert _rel(oq, d.interpolate_pressure(q0)) > 1e-3  # the tracer moved


@pytest.mark.parametrize("proj", [True, False])
def test_periodic_implicit_stepper(hip_lib, proj):
    from incompressibleeulerhdg_amd.mesh import PeriodicSquareMesh
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGImplicit
    from oracle import hdg_oracle as orc

    k, nx = 1, 6
    d = orc.HDGDiscretisation(nx, k, periodic=True, L=L)
    dt = 0.25 * d.mesh.h
    Q0, p0 = _shear()
    zero = lambda t: np.zeros((d.mesh.ncells * d.nu, 2))
    oQ, op = orc.OracleHDGImplicit(d, dt, use_projection_method=proj).solve(d.interpolate_velocity(Q0), d.interpolate_pressure(p0), zero, 2 * dt)
    ts = IncompressibleEulerHDGImplicit(PeriodicSquareMesh(nx, nx, L=L), k, dt, use_projection_method=proj)
    Q, p = ts.solve(Q0, p0, None, None, 2 * dt)
    assert _rel(Q.dat.data, oQ) < TOL and _rel(p.dat.data, op) < TOL


def test_driver_shear_flow(hip_lib, tmp_path, capsys, monkeypatch):
    """`--problem shear` (driver.py:182-183,334-335): periodic mesh, DoubleLayerShearFlow; mesh-independent Krylov counts."""
    from incompressibleeulerhdg_amd import driver

    monkeypatch.chdir(tmp_path)
    rc = driver.main(["--problem", "shear", "--nx", "32", "--degree", "2", "--dt", "0.02", "--tfinal", "0.06", "--use_projection_method",
                      "--fused", "--output", "shear.pvd", "--tracer_advection", "--animation"])
    out = capsys.readouterr().out
    assert rc == 0 and "model problem = shear" in out and (tmp_path / "shear.pvd").exists()
    assert "advect tracer = True" in out and (tmp_path / "evolution.pvd").exists()  # vorticity + tracer frames on the periodic mesh
    its = float(out.split("pressure its                :")[1].split()[0])
    assert 0 < its < 25, out


def test_periodic_properties_at_scale(hip_lib):
    """k = 2 on 256^2 periodic: momentum is conserved by the periodic discretisation without forcing up to solver tolerance
    (no boundaries), the pressure mean stays zero, iteration counts stay mesh independent."""
    from incompressibleeulerhdg_amd.mesh import PeriodicSquareMesh
    from incompressibleeulerhdg_amd.model_problems import DoubleLayerShearFlow
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    nx, k = 256, 2
    dt = 0.25 * L / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(PeriodicSquareMesh(nx, nx, L=L), k, dt)
    mp = DoubleLayerShearFlow(ts._V_Q, ts._V_p)
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), 2 * dt, fused=True)
    e = ts._engine
    assert abs(e.integrate_pressure(p.dat.data)) < 1e-9
    assert np.all(np.isfinite(Q.dat.data)) and np.max(np.abs(Q.dat.data)) < 1.2
    sums, cnt = e.iteration_stats()
    its = sums / np.maximum(cnt, 1)
    assert its[0] < 80 and np.all(its[1:] < 25), its
