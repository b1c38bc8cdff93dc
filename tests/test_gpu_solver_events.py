"""Residual replacement of the condensed (trace) CG and its counters (hdg_get_solver_events).

The reference's KSP for the condensed system (hdg_imex.py:136-137: rtol 1e-12; PETSc max_it 10000) converges or raises;
SURVEY.md 5.3: "Krylov divergence / max-it is an error code, never silent".  The single-reduction CG of the library watches
for a stalled recurrence residual / p.Ap <= 0, then recomputes the TRUE residual, restarts the recurrences and ends only at
rtol or at the rounding floor of the true residual (32 eps |x|) -- every such event is counted."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 2e-8


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _pair(k, nx, nsteps, fused):
    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332
    from oracle import hdg_oracle as orc

    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, use_projection_method=True, n_richardson=2)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), nsteps * dt, fused=fused)
    lam = ts._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
    d = orc.HDGDiscretisation(nx, k)
    tg = orc.TaylorGreen(d)
    o = orc.OracleHDGIMEX(d, dt, "imex_ssp2_332")
    oQ, op = o.solve(*tg.initial_condition(), tg.f_rhs, nsteps * dt)
    return ts, (Q.dat.data, p.dat.data, lam), (oQ, op, o.lam)


def test_default_runs_need_no_replacement(hip_lib):
    ts, got, ref = _pair(2, 8, 2, True)
    for a, b in zip(got, ref):
        assert _rel(a, b) < TOL
    ev = ts._engine.solver_events()
    assert ev["cg_residual_replacements"] == 0 and ev["cg_floor_exits"] == 0 and ev["sstep_gmres_fallbacks"] == 0, ev


@pytest.mark.parametrize("k,nx,at", [(1, 8, 3), (2, 8, 5), (3, 4, 2)])
def test_forced_residual_replacement_leaves_the_answer_alone(hip_lib, k, nx, at):
    """HDG_CG_FORCE_REPLACE=n (test hook, read when an engine is built): the n-th iteration of every condensed solve replaces the
    residual by b - T x and restarts the recurrences.  The solves still converge to rtol 1e-12, the fields match the oracle at
    2e-8, and every replacement is counted (one per solve that got that far, none of them a floor exit)."""
    os.environ["HDG_CG_FORCE_REPLACE"] = str(at)
    try:
        ts, got, ref = _pair(k, nx, 2, True)
    finally:
        del os.environ["HDG_CG_FORCE_REPLACE"]
    for a, b, name in zip(got, ref, "Qpl"):
        assert _rel(a, b) < TOL, name
    ev = ts._engine.solver_events()
    sums, cnt = ts._engine.iteration_stats()
    nsolves = int(cnt[1] + cnt[2] + cnt[3])
    assert 0 < ev["cg_residual_replacements"] <= nsolves, (ev, nsolves)
    assert ev["cg_floor_exits"] == 0, ev
    # reset works
    ts._engine.solver_events(reset=True)
    assert ts._engine.solver_events()["cg_residual_replacements"] == 0


@pytest.mark.parametrize("k,nx", [(1, 16), (2, 12)])
def test_resolving_a_converged_system_ends_at_the_rounding_floor(hip_lib, k, nx):
    """The situation of C5's third step in miniature: a condensed solve whose warm start already IS the solution (the stage
    pressure solve repeated with the same right-hand side).  Its initial residual is rounding noise, so a reduction by 1e-12 is
    not attainable: the solve must terminate -- converged, or through residual replacement and the rounding-floor exit, both
    counted -- and must leave the solution where it was (1e-10 of the update's size; the oracle's update agrees at 2e-8)."""
    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, use_projection_method=True, n_richardson=2)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    e = ts._engine
    rng = np.random.default_rng(123456789)
    xq, _ = e.node_coordinates()
    # a tentative velocity that is NOT divergence free: the stage solve has a proper right-hand side
    Qt = np.stack([np.sin(2.0 * xq[:, 0]) * np.cos(1.5 * xq[:, 1]), np.cos(xq[:, 0] + 0.3) * np.sin(2.5 * xq[:, 1])], axis=1)
    Qt += 1e-3 * rng.standard_normal(Qt.shape)
    e.set_field(100 + 1, Qt, None, None)
    its1 = e.pressure_solve(1)
    u1, p1, l1 = e.get_field(_lib.HDG_STATE_UPDATE)
    assert its1 > 3 and e.solver_events()["cg_residual_replacements"] == 0
    for rep in range(3):  # warm start = the converged solution
        its = e.pressure_solve(1)
        u2, p2, l2 = e.get_field(_lib.HDG_STATE_UPDATE)
        assert _rel(l2, l1) < 1e-10 and _rel(p2, p1) < 1e-9 and _rel(u2, u1) < 1e-9, rep
        assert its <= 100, its  # bounded: a stalled recurrence is replaced, a third confirmed drift would be an error
    ev = e.solver_events()
    print(f"k={k} nx={nx}: first solve {its1} iterations, events after three re-solves: {ev}")
    assert ev["cg_floor_exits"] <= ev["cg_residual_replacements"] <= 6, ev
