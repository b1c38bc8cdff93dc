"""The update of the condensed single-reduction CG in two halves (Engine::trace_cg_sr, one rank, tile preconditioner): r and s
at once, p and x as side jobs of the V-cycle's leg launches (hdg_kernels.hpp: SideXP, HDG_P1_SIDE_JOB) or on a second stream
(HDG_CG_XP_MODE=0), and the second reduction stage fused with the CG scalars (k_cg_sr_reduce_scalars).  Same arithmetic per
entry as the one-launch update k_cg_sr_update (HDG_CG_NO_SPLIT_UPDATE / HDG_CG_NO_FUSED_SCALARS / HDG_CG_FUSED_RUPDATE, read when an engine is built);
the only difference is that the step beyond the tested iterate is not taken: fields agree far below the solver tolerance
(hdg_imex.py:136-137: rtol 1e-12) with the same iteration counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


# (2, 128): legs at n = 128, 64 with more side rows than tile rows on the finest level; (1, 256): three leg levels;
# (2, 72): a finest level whose tiles are partial (73 vertices per row); (3, 66): k = 3 tables, n = 66 -> 33 (one leg level)
@pytest.mark.parametrize("k,nx", [(2, 128), (1, 256), (2, 72), (3, 66)])
def test_split_cg_update_equals_the_one_launch_update(hip_lib, k, nx, monkeypatch):
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    res = {}
    # "ride": the default (p / x on the V-cycle legs, s / r as a launch of their own); "fused_r": s / r inside the next pre
    # tile kernel (k_trace_pre_tile<K, true>: measured slower, off by default); "stream": p / x on a second stream; "one": the
    # one-launch update
    for tag, env in (("ride", {}), ("fused_r", {"HDG_CG_FUSED_RUPDATE": "1"}), ("stream", {"HDG_CG_XP_MODE": "0"}),
                     ("one", {"HDG_CG_NO_SPLIT_UPDATE": "1", "HDG_CG_NO_FUSED_SCALARS": "1"})):
        for name in ("HDG_CG_XP_MODE", "HDG_CG_NO_SPLIT_UPDATE", "HDG_CG_NO_FUSED_SCALARS", "HDG_CG_FUSED_RUPDATE"):
            monkeypatch.delenv(name, raising=False)
        for name, v in env.items():
            monkeypatch.setenv(name, v)
        ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, 0.25 / nx, use_projection_method=True, n_richardson=2)
        mp = TaylorGreen(ts._V_Q, ts._V_p)
        Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), 2 * 0.25 / nx, fused=True)
        sums, cnt = ts._engine.iteration_stats()
        res[tag] = (Q.dat.data.copy(), p.dat.data.copy(), sums / np.maximum(cnt, 1))
    for tag in ("ride", "fused_r", "stream"):
        assert _rel(res[tag][0], res["one"][0]) < 1e-9 and _rel(res[tag][1], res["one"][1]) < 1e-9, tag
        # the warm starts of later solves differ at 1e-12 (the dropped step): a CG count at the edge of its tolerance may move by
        # one.  (The tentative-velocity counts of the FIRST steps of a run are not compared: the spectral bounds come from an
        # opening Arnoldi cycle on smooth data, whose extreme Ritz values react to such perturbations -- the fields do not.)
        assert np.all(np.abs(res[tag][2][1:] - res["one"][2][1:]) <= 1.0), (tag, res[tag][2], res["one"][2])
