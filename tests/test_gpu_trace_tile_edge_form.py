"""The tile kernels of the trace preconditioner with one thread per EDGE (hdg_trace_tile3.hpp; default at k = 4, where the
corner-per-thread form needs 274 VGPRs) against the corner-per-thread form (k <= 3) and against the row-stencil kernels (k = 4):
the same tiles, stages and arithmetic per edge -- only the partial sums of the five CG inner products are formed in another
order --, so whole steps agree far below the solver tolerance (hdg_imex.py:136-137: rtol 1e-12) with the same CG counts.
Meshes with full and partial tiles, a periodic one (wrapped halo columns), and k = 4 against the oracle at 2e-8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _step(k, nx, periodic, nsteps=1):
    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd.mesh import PeriodicSquareMesh, UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import DoubleLayerShearFlow, TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    L = 2 * np.pi if periodic else 1.0
    dt = 0.25 * L / nx
    mesh = PeriodicSquareMesh(nx, nx, L=L) if periodic else UnitSquareMesh(nx, nx)
    ts = IncompressibleEulerHDGIMEXSSP2_332(mesh, k, dt, use_projection_method=True, n_richardson=2)
    mp = (DoubleLayerShearFlow if periodic else TaylorGreen)(ts._V_Q, ts._V_p)
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), nsteps * dt, fused=True)
    lam = ts._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
    sums, cnt = ts._engine.iteration_stats()
    return Q.dat.data.copy(), p.dat.data.copy(), lam.copy(), sums / np.maximum(cnt, 1), ts._engine.kernel_forms()["trace_precond"]


@pytest.mark.parametrize("k,nx,periodic", [(1, 40, False), (2, 37, False), (3, 24, False), (2, 18, True), (4, 18, False), (4, 33, False), (4, 16, True)])
def test_edge_form_equals_the_other_form(hip_lib, k, nx, periodic, monkeypatch):
    res = {}
    for form in ("1", "0"):  # read when an engine is built
        monkeypatch.setenv("HDG_TRACE_TILE3", form)
        res[form] = _step(k, nx, periodic)
    assert res["1"][4] == 2 and res["0"][4] == (1 if k <= 3 else 0)  # k = 4 without the edge form: the row-stencil kernels
    for q in range(3):
        assert _rel(res["1"][q], res["0"][q]) < 1e-9, q
    assert np.all(np.abs(res["1"][3][1:] - res["0"][3][1:]) <= 1.0), (res["1"][3], res["0"][3])


def test_edge_form_is_the_default_at_degree_four_and_matches_the_oracle(hip_lib, monkeypatch):
    from oracle import hdg_oracle as orc

    monkeypatch.delenv("HDG_TRACE_TILE3", raising=False)
    k, nx = 4, 8
    Q, p, lam, its, form = _step(k, nx, False, nsteps=2)
    assert form == 2  # hdg_get_kernel_forms: tiles, one thread per edge
    d = orc.HDGDiscretisation(nx, k)
    tg = orc.TaylorGreen(d)
    o = orc.OracleHDGIMEX(d, 0.25 / nx, "imex_ssp2_332")
    oQ, op = o.solve(*tg.initial_condition(), tg.f_rhs, 2 * 0.25 / nx)
    assert _rel(Q, oQ) < 2e-8 and _rel(p, op) < 2e-8 and _rel(lam, o.lam) < 2e-8
