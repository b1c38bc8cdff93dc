"""HIP engine vs the C++/OpenMP CPU twin (oracle/cpu_twin) at sizes the numpy oracle cannot reach.

The twin is pinned by the numpy oracle at small sizes (tests/test_cpu_twin.py, runs without a GPU); here it carries the
comparison to meshes with many workgroups / tiles / XCD bands: every operator at the single-application tolerance
(1e-11 relative) and whole timesteps at the two-converged-solvers tolerance (2e-8)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _pair(k, nx, tableau="imex_ssp2_332", R=2, flux="upwind"):
    from incompressibleeulerhdg_amd._lib import Engine
    from oracle.cpu_twin import CpuTwin
    from oracle.hdg_oracle import TABLEAUX

    tb = TABLEAUX[tableau]
    kw = dict(nx=nx, degree=k, dt=0.25 / nx, nstages=len(tb["c_expl"]), a_expl=tb["a_expl"], a_impl=tb["a_impl"], b_expl=tb["b_expl"],
              b_impl=tb["b_impl"], c_expl=tb["c_expl"], n_richardson=R, flux=flux)
    return Engine(**kw), CpuTwin(**kw), tb


@pytest.mark.parametrize("k,nx", [(1, 200), (2, 160), (3, 72), (4, 40)])
def test_operators_match_the_cpu_twin_on_large_meshes(hip_lib, k, nx):
    e, t, _ = _pair(k, nx)
    xq, xp = e.node_coordinates()
    tq, tp = t.node_coordinates()
    assert np.allclose(xq, tq, rtol=0, atol=1e-14) and np.allclose(xp, tp, rtol=0, atol=1e-14)
    rng = np.random.default_rng(31)
    Q, x, lam = rng.standard_normal(e.shape_Q), rng.standard_normal(e.shape_Q), rng.standard_normal(e.shape_l)
    Px = e.project_bdm_nodal(Q)
    assert _rel(Px, t.project_bdm_nodal(Q)) < 1e-11
    gamma = 0.3 / nx
    assert _rel(e.apply_advection(Px, x, gamma), t.apply_advection(Px, x, gamma)) < 1e-11
    # the operator part alone: (x - y) / gamma
    assert _rel((x - e.apply_advection(Px, x, gamma)) / gamma, (x - t.apply_advection(Px, x, gamma)) / gamma) < 1e-9
    assert _rel(e.apply_trace_operator(lam), t.apply_trace_operator(lam)) < 1e-11
    assert _rel(e.apply_weak_divergence(x), t.apply_weak_divergence(x)) < 1e-11


@pytest.mark.parametrize("k,nx,tableau,nsteps", [(2, 128, "imex_ssp2_332", 2), (1, 256, "imex_ssp2_332", 2), (3, 48, "imex_ars2_232", 1),
                                                 (2, 64, "imex_ars3_443", 1)])
def test_timesteps_match_the_cpu_twin_on_large_meshes(hip_lib, k, nx, tableau, nsteps):
    """Same data (Taylor-Green vortex plus a seeded smooth perturbation, so that the implicit terms do not cancel),
    same tableau: the two independently written solver stacks converge to the same fields."""
    from incompressibleeulerhdg_amd import _lib

    e, t, tb = _pair(k, nx, tableau)
    xq, xp = e.node_coordinates()
    S = lambda z: np.sin((z - 0.5) * np.pi)
    C = lambda z: np.cos((z - 0.5) * np.pi)
    Q0 = np.stack([-C(xq[:, 0]) * S(xq[:, 1]) + 0.1 * np.sin(2 * np.pi * xq[:, 1]), S(xq[:, 0]) * C(xq[:, 1]) + 0.05 * xq[:, 0] ** 2], axis=-1)
    p0 = (S(xp[:, 0]) ** 2 + S(xp[:, 1]) ** 2) / 2
    prof = np.stack([np.cos(np.pi * xq[:, 0]) * xq[:, 1], np.sin(np.pi * xq[:, 1]) + xq[:, 0]], axis=-1)
    s = len(tb["c_expl"])
    for eng in (e, t):
        eng.set_state(Q0, p0)
        eng.reconstruct_trace()
        eng.set_forcing_profile(prof)
        for n in range(nsteps):
            for sl in range(s + 1):
                eng.set_forcing_scale(sl, -0.5 * np.exp(-0.1 * (n + sl)))
            eng.step()
    Q, p, lam = e.get_field(_lib.HDG_STATE_CURRENT)
    tQ, tp_, tl = t.get_state()
    assert _rel(Q, tQ) < 2e-8 and _rel(p, tp_) < 2e-8 and _rel(lam, tl) < 2e-8
    assert _rel(tQ, Q0) > 1e-3  # the step did something
