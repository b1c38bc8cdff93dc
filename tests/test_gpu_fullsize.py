"""Size-independent properties at the BASELINE sizes, where the oracle cannot run (C2: k=1, 256^2;
C3: k=2, 1024^2).  Each property holds for any correct implementation of the reference's forms."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stepper(k, nx, **kw):
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, 0.25 / nx, use_projection_method=True, n_richardson=2, **kw)
    return ts, TaylorGreen(ts._V_Q, ts._V_p, "exponential", 0.5)


@pytest.mark.parametrize("k,nx", [(1, 256), (2, 1024), (3, 512), (4, 200)])  # k >= 3: matrix-core lift, many tiles / partial tile
def test_operator_properties_at_baseline_size(hip_lib, k, nx):
    ts, mp = _stepper(k, nx)
    e = ts._engine
    rng = np.random.default_rng(123456789)
    x = rng.standard_normal(e.shape_Q)
    y = rng.standard_normal(e.shape_Q)
    # BDM projection: idempotent, and the projection of an interpolated smooth H(div) field is itself
    Px = e.project_bdm_nodal(x)
    PPx = e.project_bdm_nodal(Px)
    assert np.max(np.abs(PPx - Px)) < 1e-10 * np.max(np.abs(Px))
    Qs = ts._V_Q.interpolate(mp.Q_stationary)
    assert np.max(np.abs(e.project_bdm_nodal(Qs) - Qs)) < 1e-11
    # advection operator: linear in x; reduces to the identity for gamma = 0
    a, b = 0.7, -1.3
    gamma = 0.25 * 0.25 / nx
    lhs = e.apply_advection(Px, a * x + b * y, gamma)
    rhs = a * e.apply_advection(Px, x, gamma) + b * e.apply_advection(Px, y, gamma)
    assert np.max(np.abs(lhs - rhs)) < 1e-10 * np.max(np.abs(rhs))
    assert np.max(np.abs(e.apply_advection(Px, x, 0.0) - x)) < 1e-12 * np.max(np.abs(x))
    # condensed trace operator: constants are in the null space (hdg_imex.py:480-489); the weak divergence
    # of any field sums to zero against psi = 1 (consistency of the stage right-hand sides)
    lam = np.ones(e.shape_l)
    T1 = e.apply_trace_operator(lam)
    Tr = e.apply_trace_operator(rng.standard_normal(e.shape_l))
    assert np.max(np.abs(T1)) < 1e-9 * np.max(np.abs(Tr))
    wd = e.apply_weak_divergence(x)
    assert abs(e.integrate_pressure(wd)) < 1e-9 * np.max(np.abs(wd))
    # nodal <-> modal round trip through the device
    e.set_field(1, x, None, None)
    assert np.max(np.abs(e.get_field(1, p=False, lam=False)[0] - x)) < 1e-12 * np.max(np.abs(x))


@pytest.mark.parametrize("k,nx,nsteps", [(1, 256, 3), (2, 1024, 2), (3, 512, 1), (4, 256, 1), (4, 2048, 3)])
def test_timestep_properties_at_baseline_size(hip_lib, k, nx, nsteps):
    """C2 (k=1, 256^2), C3 (k=2, 1024^2) and C5 (k=4, 2048^2: 541 M unknowns on ONE MI355X) of BASELINE.json, plus the
    matrix-core kernels at k = 3, 4 on meshes with many tiles: the error against the exact vortex catches a wrong
    stencil / neighbour index at sizes the oracle cannot reach.  C5 runs three steps: in the third the second Richardson pass
    has a right-hand side so small against the pressure it corrects that the recurrence residual of the CG stalls a few units
    above rtol * |z0| (round 3 ended such a solve silently; round 4: residual replacement -- the TRUE residual is recomputed, the
    recurrences restart, and the solve ends only at rtol or at the rounding floor of the true residual, every such event
    counted: hdg_get_solver_events).  C2 / C3 / the 512^2 runs must not need it at all."""
    from incompressibleeulerhdg_amd import _lib

    ts, mp = _stepper(k, nx)
    e = ts._engine
    dt = 0.25 / nx
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), nsteps * dt, fused=True)
    # zero-mean pressure after every shift (hdg_imex.py:471-478)
    assert abs(e.integrate_pressure(p.dat.data)) < 1e-11
    # the manufactured vortex decays like exp(-kappa t): discretisation error only (h^{k+2}, dt)
    Qe, pe = mp.solution(nsteps * dt, e.integrate_pressure)
    eq, ep = e.l2_norms(Q.dat.data - Qe.dat.data, p.dat.data - pe.dat.data)
    nq, _ = e.l2_norms(Qe.dat.data, pe.dat.data)
    assert eq < 1e-6 * nq and ep < 1e-4
    # Q* used in the last stage is H(div) conforming with zero boundary flux: a fixed point of the projection
    Qstar = e.get_field(200 + 1, p=False, lam=False)[0]
    assert np.max(np.abs(e.project_bdm_nodal(Qstar) - Qstar)) < 1e-10 * np.max(np.abs(Qstar))
    # mesh-independent Krylov behaviour (the design claim behind the two-level preconditioners)
    sums, cnt = e.iteration_stats()
    its = sums / np.maximum(cnt, 1)
    assert its[0] < 70 and np.all(its[1:] < 20), its
    # residual replacements / rounding-floor exits of the condensed solves: none at C2 / C3 / 512^2; C5's third step may need them,
    # and then every floor exit follows a replacement (a floor exit is only taken on a freshly computed true residual)
    ev = e.solver_events()
    print(f"k={k} nx={nx}: solver events {ev}, iterations {its}")
    assert ev["sstep_gmres_fallbacks"] == 0, ev  # the s-step tail of the tentative-velocity solver never had to give up
    if nx < 2048:
        assert ev["cg_residual_replacements"] == 0 and ev["cg_floor_exits"] == 0, ev
    else:
        assert ev["cg_floor_exits"] <= ev["cg_residual_replacements"] <= 2 * int(cnt[1] + cnt[2] + cnt[3]), ev


@pytest.mark.parametrize("proj,nsteps", [(True, 2), (False, 1)])
def test_implicit_timestep_properties_at_C4_size(hip_lib, proj, nsteps):
    """BASELINE C4: IncompressibleEulerHDGImplicit, k = 3, 512 x 512 (hdg_implicit.py:92-190), Chorin projection and
    the monolithic (u, phi, lambda) solve, on one GPU.  Size-independent properties: zero-mean pressure, error
    against the exact vortex (first order in dt: the error of one or two steps is O(dt^2)), iteration bounds."""
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGImplicit

    k, nx = 3, 512
    dt = 0.25 / nx
    ts = IncompressibleEulerHDGImplicit(UnitSquareMesh(nx, nx), k, dt, use_projection_method=proj)
    mp = TaylorGreen(ts._V_Q, ts._V_p, "exponential", 0.5)
    e = ts._engine
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), nsteps * dt)
    assert abs(e.integrate_pressure(p.dat.data)) < 1e-11
    Qe, pe = mp.solution(nsteps * dt, e.integrate_pressure)
    eq, ep = e.l2_norms(Q.dat.data - Qe.dat.data, p.dat.data - pe.dat.data)
    nq, _ = e.l2_norms(Qe.dat.data, pe.dat.data)
    # time error of the first-order scheme after n steps ~ n dt^2 |Q_tt| ~ 1e-7; the pressure is first order in dt
    assert eq < 1e-5 * nq and ep < 1e-2, (eq / nq, ep)
    assert 0 < ts.niter_tentative.value < 120, ts.niter_tentative.value
    if proj:
        assert 0 < ts.niter_pressure.value < 25, ts.niter_pressure.value
        # the projection step leaves a velocity whose BDM projection is (almost) itself: normal jumps are O(dt)
        Qn = Q.dat.data
        assert np.max(np.abs(e.project_bdm_nodal(Qn) - Qn)) < 1e-3 * np.max(np.abs(Qn))
