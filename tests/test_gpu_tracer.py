"""Passive tracer transport and the continuous-space diagnostics against the
oracle's restatement (oracle/tracer_oracle.py) of common.py:110-129, hdg_imex.py:415-448,560,622-623,638-639,
hdg_implicit.py:93-96,192-193 and callbacks.py:43-69.

Tolerances: single operator applications 1e-10 relative (the continuous mass matrix is solved by CG to 1e-13 on the
device, exactly in the oracle); whole steps 2e-8 like the velocity / pressure fields."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 2e-8


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _setup(k, nx, tableau="imex_ssp2_332", **kw):
    from incompressibleeulerhdg_amd._lib import Engine
    from oracle.hdg_oracle import TABLEAUX, HDGDiscretisation
    from oracle.tracer_oracle import TracerOracle

    d = HDGDiscretisation(nx, k)
    tb = TABLEAUX[tableau]
    e = Engine(nx=nx, degree=k, dt=0.25 / nx, nstages=len(tb["c_expl"]), a_expl=tb["a_expl"], a_impl=tb["a_impl"], b_expl=tb["b_expl"],
               b_impl=tb["b_impl"], c_expl=tb["c_expl"], **kw)
    return d, TracerOracle(d), e


@pytest.mark.parametrize("k,nx", [(1, 5), (2, 4), (3, 3), (4, 2), (2, 9)])
def test_continuous_space_projection_and_vorticity(hip_lib, k, nx):
    d, tr, e = _setup(k, nx)
    assert e.cg_size() == tr.ncg == ((k + 1) * nx + 1) ** 2
    # same dofs: every oracle dof position appears exactly once among the device's
    key = lambda X: {tuple(np.round(x * nx * 1e6).astype(np.int64)) for x in X}
    assert key(e.cg_coordinates()) == key(tr.cg_coords) and len(key(e.cg_coordinates())) == tr.ncg
    rng = np.random.default_rng(3)
    u = rng.standard_normal(e.shape_Q)
    P = e.cg_project_nodal(u)
    assert _rel(P, tr.cg_project(u)) < 1e-10
    assert _rel(e.cg_project_nodal(P), P) < 1e-10  # idempotent
    cont = d.interpolate_velocity(lambda x, y: (x ** (k + 1) - y, x * y ** k + 1.0))
    assert _rel(e.cg_project_nodal(cont), cont) < 1e-10  # a continuous P_{k+1} field is reproduced
    # vorticity (callbacks.py:43-69): compare dof by dof through the coordinates
    Q = rng.standard_normal(e.shape_Q)
    w, xy = tr.vorticity(Q)
    wd = e.vorticity(Q)
    order = lambda X: np.lexsort((np.round(X[:, 1] * nx * 1e6), np.round(X[:, 0] * nx * 1e6)))
    assert _rel(wd[order(e.cg_coordinates())], w[order(xy)]) < 1e-10
    # ... and on the broken node set (what the VTK writer gets)
    wb = e.cg_to_broken(wd)
    assert _rel(wb, (tr.R @ w)) < 1e-10


@pytest.mark.parametrize("k,nx", [(1, 5), (2, 4), (3, 3), (4, 2), (1, 17)])
def test_tracer_transport_operator(hip_lib, k, nx):
    d, tr, e = _setup(k, nx)
    rng = np.random.default_rng(4)
    q, u = rng.standard_normal(e.shape_p), rng.standard_normal(e.shape_Q)
    assert _rel(e.apply_tracer_advection(q, u, project=True), tr.tracer_tendency(q, u)) < 1e-10
    uc = tr.cg_project(u)
    assert _rel(e.apply_tracer_advection(q, uc, project=False), tr._lu_mp.solve(tr.tracer_form(q, uc))) < 1e-10
    # consistency with the transport equation: M^-1 T(q, u) -> -u . grad q for smooth q, solenoidal tangential u
    S = lambda z: np.sin((z - 0.5) * np.pi)
    C = lambda z: np.cos((z - 0.5) * np.pi)
    uu = d.interpolate_velocity(lambda x, y: (-C(x) * S(y), S(x) * C(y)))
    qq = d.interpolate_pressure(lambda x, y: x * (1 - y) + 0.3 * y)
    ex = d.interpolate_pressure(lambda x, y: -((-C(x) * S(y)) * (1 - y) + (S(x) * C(y)) * (0.3 - x)))
    err = d.l2_norm_pressure(e.apply_tracer_advection(qq, uu) - ex) / d.l2_norm_pressure(ex)
    assert err < (0.4 if k == 1 and nx < 10 else 0.1), err


def _q0(x, y):
    return np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y)  # driver.py:342


@pytest.mark.parametrize("k,nx,tableau", [(1, 6, "imex_ssp2_332"), (2, 4, "imex_ssp2_332"), (1, 5, "imex_ars3_443"), (2, 4, "imex_ssp3_433"),
                                          (1, 5, "imex_implicit")])
def test_imex_steps_with_tracer(hip_lib, k, nx, tableau):
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd import timesteppers as tsm
    from oracle import hdg_oracle as orc
    from oracle.tracer_oracle import TracerOracle, imex_with_tracer

    cls = {"imex_ssp2_332": tsm.IncompressibleEulerHDGIMEXSSP2_332, "imex_ars3_443": tsm.IncompressibleEulerHDGIMEXARS3_443,
           "imex_ssp3_433": tsm.IncompressibleEulerHDGIMEXSSP3_433, "imex_implicit": tsm.IncompressibleEulerHDGIMEXImplicit}[tableau]
    dt, nsteps = 0.25 / nx, 2
    d = orc.HDGDiscretisation(nx, k)
    tr = TracerOracle(d)
    tg = orc.TaylorGreen(d)
    o = orc.OracleHDGIMEX(d, dt, tableau)
    oQ, op, oq = imex_with_tracer(o, tr, *tg.initial_condition(), d.interpolate_pressure(_q0), tg.f_rhs, nsteps * dt)
    for fused in (False, True):
        ts = cls(UnitSquareMesh(nx, nx), k, dt)
        mp = TaylorGreen(ts._V_Q, ts._V_p)
        Q, p = ts.solve(*mp.initial_condition(), _q0, mp.f_rhs(), nsteps * dt, fused=fused)
        assert _rel(Q.dat.data, oQ) < TOL and _rel(p.dat.data, op) < TOL
        assert _rel(ts.q_tracer.dat.data, oq) < TOL, fused
    assert _rel(oq, d.interpolate_pressure(_q0)) > 1e-3  # the tracer moved


@pytest.mark.parametrize("k,nx", [(1, 6), (2, 4)])
def test_implicit_stepper_with_tracer(hip_lib, k, nx):
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGImplicit
    from oracle import hdg_oracle as orc
    from oracle.tracer_oracle import TracerOracle, implicit_with_tracer

    dt = 0.25 / nx
    d = orc.HDGDiscretisation(nx, k)
    tg = orc.TaylorGreen(d)
    oQ, op, oq = implicit_with_tracer(d, TracerOracle(d), dt, *tg.initial_condition(), d.interpolate_pressure(_q0), tg.f_rhs, 3 * dt)
    ts = IncompressibleEulerHDGImplicit(UnitSquareMesh(nx, nx), k, dt)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    Q, p = ts.solve(*mp.initial_condition(), _q0, mp.f_rhs(), 3 * dt)
    assert _rel(Q.dat.data, oQ) < TOL and _rel(p.dat.data, op) < TOL and _rel(ts.q_tracer.dat.data, oq) < TOL


def test_driver_tracer_and_animation(hip_lib, tmp_path, capsys, monkeypatch):
    """driver.py:165-176,187,340-344: --tracer_advection and --animation (evolution.pvd with velocity, pressure, CG vorticity
    and tracer of every time level)."""
    from incompressibleeulerhdg_amd import driver

    monkeypatch.chdir(tmp_path)
    rc = driver.main(["--nx", "8", "--degree", "1", "--dt", "0.05", "--tfinal", "0.1", "--use_projection_method", "--tracer_advection",
                      "--animation", "--output", ""])
    out = capsys.readouterr().out
    assert rc == 0 and "advect tracer = True" in out
    pvd = (tmp_path / "evolution.pvd").read_text()
    assert pvd.count("<DataSet") == 3  # t = 0 and two steps
    vtu = (tmp_path / "evolution_2.vtu").read_text()
    for name in ('Name="Q"', 'Name="p"', 'Name="vorticity"', 'Name="tracer"'):
        assert name in vtu, name


def test_tracer_properties_at_scale(hip_lib):
    """k = 2 on 256^2 (beyond the oracle): total tracer mass is conserved by the upwind transport with a tangential
    velocity up to the boundary flux of the projected velocity, the CG mass solve converges, and the field stays bounded."""
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    nx, k = 256, 2
    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    q0 = lambda x, y: 1.0 + _q0(x, y)
    ts.solve(*mp.initial_condition(), q0, mp.f_rhs(), 3 * dt, fused=True)
    e = ts._engine
    m0 = e.integrate_pressure(ts._V_p.interpolate(q0))
    m1 = e.integrate_pressure(ts.q_tracer.dat.data)
    assert abs(m1 - m0) < 1e-8 * abs(m0), (m0, m1)
    assert np.max(np.abs(ts.q_tracer.dat.data)) < 2.5
