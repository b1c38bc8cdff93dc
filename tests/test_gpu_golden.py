"""GPU parity against the committed golden vectors (tests/golden/*.npz, written by the oracle).

Tolerance 2e-8 relative max-norm: solver-tolerance level (see tests/test_gpu_timestep.py)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "imex_*.npz"))))
def test_imex_golden(hip_lib, path):
    from incompressibleeulerhdg_amd import _lib, timesteppers as ts
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen

    g = np.load(path)
    name = os.path.basename(path)[len("imex_"):-len(".npz")]
    tab, rest = name.rsplit("_k", 1)
    k, nx, R, n = [int(x[1:] if x[0] in "nRx" else x) for x in rest.replace("nx", "x").split("_")]
    cls = {"imex_implicit": ts.IncompressibleEulerHDGIMEXImplicit, "imex_ars2_232": ts.IncompressibleEulerHDGIMEXARS2_232,
           "imex_ars3_443": ts.IncompressibleEulerHDGIMEXARS3_443, "imex_ssp2_332": ts.IncompressibleEulerHDGIMEXSSP2_332,
           "imex_ssp3_433": ts.IncompressibleEulerHDGIMEXSSP3_433}[tab]
    dt = float(g["dt"])
    t = cls(UnitSquareMesh(nx, nx), k, dt, n_richardson=R)
    mp = TaylorGreen(t._V_Q, t._V_p)
    Q, p = t.solve(*mp.initial_condition(), None, mp.f_rhs(), n * dt, fused=True)
    lam = t._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
    assert _relerr(Q.dat.data, g["Q"]) < 2e-8 and _relerr(p.dat.data, g["p"]) < 2e-8 and _relerr(lam, g["lam"]) < 2e-8
    Qe, pe = mp.solution(n * dt, t._engine.integrate_pressure)
    eq, ep = t._engine.l2_norms(Q.dat.data - Qe.dat.data, p.dat.data - pe.dat.data)
    assert abs(eq - float(g["err_Q"])) < 1e-9 and abs(ep - float(g["err_p"])) < 1e-9


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "implicit_*.npz"))))
def test_implicit_golden(hip_lib, path):
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGImplicit

    g = np.load(path)
    k, nx = [int(x) for x in os.path.basename(path).replace("implicit_proj_k", "").replace("_n4.npz", "").split("_nx")]
    dt = float(g["dt"])
    t = IncompressibleEulerHDGImplicit(UnitSquareMesh(nx, nx), k, dt)
    mp = TaylorGreen(t._V_Q, t._V_p)
    Q, p = t.solve(*mp.initial_condition(), None, mp.f_rhs(), 4 * dt)
    assert _relerr(Q.dat.data, g["Q"]) < 2e-8 and _relerr(p.dat.data, g["p"]) < 2e-8


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "tracer_*.npz")) + glob.glob(os.path.join(GOLD, "periodic_*.npz"))))
def test_tracer_and_periodic_golden(hip_lib, path):
    """Passive tracer on the unit square (common.py:110-129, hdg_imex.py:415-448) and shear layer + forcing + tracer +
    vorticity on the doubly periodic square (driver.py:182-183, callbacks.py:43-69) against the committed vectors."""
    import sys

    sys.path.insert(0, GOLD)
    from make_golden import periodic_case

    from incompressibleeulerhdg_amd import _lib, timesteppers as ts
    from incompressibleeulerhdg_amd.mesh import PeriodicSquareMesh, UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen

    g = np.load(path)
    kind, rest = os.path.basename(path)[:-len(".npz")].split("_", 1)
    tab, rest = rest.rsplit("_k", 1)
    k, nx, n = [int(x.lstrip("nx")) for x in rest.split("_")]
    cls = {"imex_ars3_443": ts.IncompressibleEulerHDGIMEXARS3_443, "imex_ssp2_332": ts.IncompressibleEulerHDGIMEXSSP2_332}[tab]
    dt = float(g["dt"])
    for fused in (True, False):
        if kind == "tracer":
            t = cls(UnitSquareMesh(nx, nx), k, dt)
            mp = TaylorGreen(t._V_Q, t._V_p)
            q0 = lambda x, y: np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y)
            Q, p = t.solve(*mp.initial_condition(), q0, mp.f_rhs(), n * dt, fused=fused)
        else:
            t = cls(PeriodicSquareMesh(nx, nx, L=2 * np.pi), k, dt)
            Q0, p0, q0, f = periodic_case()
            Q, p = t.solve(Q0, p0, q0, f, n * dt, fused=fused)
            lam = t._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
            assert _relerr(lam, g["lam"]) < 2e-8
            w, xy = t._engine.vorticity(), t._engine.cg_coordinates()
            order = np.lexsort((np.round(xy[:, 1] * 1e6), np.round(xy[:, 0] * 1e6)))
            assert _relerr(w[order], g["vorticity_sorted"]) < 5e-8
        assert _relerr(Q.dat.data, g["Q"]) < 2e-8 and _relerr(p.dat.data, g["p"]) < 2e-8, fused
        assert _relerr(t.q_tracer.dat.data, g["q"]) < 2e-8, fused


def test_operator_golden(hip_lib):
    from incompressibleeulerhdg_amd._lib import Engine

    g = np.load(os.path.join(GOLD, "operators_k2_nx3.npz"))
    e = Engine(nx=3, degree=2, dt=0.1, nstages=2, a_expl=[[0, 0], [1, 0]], a_impl=[[0, 0], [0, 1]], b_expl=[1, 0],
               b_impl=[0, 1], c_expl=[0, 1])
    assert _relerr(e.project_bdm_nodal(g["Q"]), g["Qstar"]) < 1e-11
    e.set_state(g["Q"], g["p"])
    e.reconstruct_trace()
    lam = e.get_field(0, Q=False, p=False)[2]
    # set_state removes the pressure mean; lambda shifts by the same constant
    shift = e.integrate_pressure(g["p"])
    assert _relerr(lam, g["lam"] - shift) < 1e-11


def test_engine_config_holds_the_reference_tableaux(hip_lib):
    """The hdg_config an engine is built with (what crosses the C-ABI) carries the tableau of the reference fixture
    (tests/golden/tableaux_reference.json, extracted from hdg_imex.py:668-1038) bit for bit, for all five classes."""
    import json

    from incompressibleeulerhdg_amd import timesteppers as ts
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh

    with open(os.path.join(GOLD, "tableaux_reference.json")) as f:
        classes = json.load(f)["classes"]
    for cname, e in classes.items():
        t = getattr(ts, cname)(UnitSquareMesh(4, 4), 1, 0.01)
        cfg = t._engine.cfg
        s = e["nstages"]
        assert cfg.nstages == s and t.label == e["label"]
        ref = {k: np.array([float.fromhex(h) for h in e[k]["hex"]]).reshape(e[k]["shape"]) for k in ("a_expl", "a_impl", "b_expl", "b_impl", "c_expl")}
        assert np.array_equal(np.array(cfg.a_expl[:s * s]).reshape(s, s), ref["a_expl"])
        assert np.array_equal(np.array(cfg.a_impl[:s * s]).reshape(s, s), ref["a_impl"])
        assert np.array_equal(np.array(cfg.b_expl[:s]), ref["b_expl"]) and np.array_equal(np.array(cfg.c_expl[:s]), ref["c_expl"])
        nb = len(ref["b_impl"])  # ARS3(4,4,3): 6 entries as written (the first 5 are read)
        assert np.array_equal(np.array(cfg.b_impl[:nb]), ref["b_impl"])
