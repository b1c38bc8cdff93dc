"""Round-4 code paths against their alternatives and the oracle: the s-step minimal-residual tail of the tentative-velocity
solver (vs the GMRES(8) tail it replaces, with and without the LGMRES-type augmentation), the paired form of the advection
kernel (vs the gather form), the matrix-free lift on general meshes (vs the assembled operators).  Every alternative is the
same mathematics in another schedule: fields agree far below the solver tolerance, and the default agrees with the oracle at
2e-8 (tests/test_gpu_timestep.py runs the default path against the oracle for every tableau)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _worker(tmp_path, tag, k, nx, nsteps, env):
    out = str(tmp_path / f"{tag}.npz")
    r = subprocess.run([sys.executable, os.path.join(HERE, "mp_strip_worker.py"), "0", "1", "unused", str(k), str(nx), str(nsteps), out],
                       env=dict(os.environ, HDG_DEBUG="1", **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return np.load(out), r.stdout.decode()


@pytest.mark.parametrize("k,nx", [(2, 48), (3, 24), (2, 128)])
def test_sstep_tail_against_the_gmres_tail_and_the_oracle(hip_lib, tmp_path, k, nx):
    """The Chebyshev iteration hands its tail to s-step minimal-residual cycles (Engine::sstep_mr: power basis, one Gram pass,
    least squares on the host, one update pass) instead of a GMRES(8) cycle.  Same Krylov spaces, same stopping rule
    (hdg_imex.py:224-228: rtol 1e-10 on the preconditioned residual): the fields agree at 1e-9, the s-step path needs no more
    than a few iterations more per solve, and the small case agrees with the oracle at 2e-8."""
    res = {}
    for tag, env in (("sstep", {}), ("gmres", {"HDG_TAIL_GMRES": "1"}), ("aug", {"HDG_SSTEP_AUG": "1"}), ("aug2", {"HDG_SSTEP_AUG": "2", "HDG_SSTEP_MAX": "5"})):
        res[tag], log = _worker(tmp_path, tag, k, nx, 2, env)
        # (the GMRES-tail run may still show s-step cycles: a WHOLE solve that the Chebyshev iteration does not take -- its
        # ellipse predicts > 64 iterations -- runs as s-step cycles in either mode; the first solve of a run from smooth data,
        # whose opening Arnoldi cycle spans a nearly invariant subspace, can fall on either side of that line)
        assert "[sstep] cycle" in log if tag != "gmres" else "[gmres]" in log, tag
    for tag in ("gmres", "aug", "aug2"):
        for name in ("Q", "p", "lam"):
            assert _rel(res[tag][name], res["sstep"][name]) < 1e-9, (tag, name)
    assert res["sstep"]["its"][0] <= res["gmres"]["its"][0] + 6.0
    assert np.all(np.abs(res["sstep"]["its"][1:] - res["gmres"]["its"][1:]) <= 1.0)
    if nx <= 48 and k == 2:
        from oracle import hdg_oracle as orc

        d = orc.HDGDiscretisation(12, 2)  # the oracle's size: a separate small run of the default path
        small, _ = _worker(tmp_path, "small", 2, 12, 2, {})
        tg = orc.TaylorGreen(d)
        o = orc.OracleHDGIMEX(d, 0.25 / 12, "imex_ssp2_332")
        oQ, op = o.solve(*tg.initial_condition(), tg.f_rhs, 2 * 0.25 / 12)
        assert _rel(small["Q"], oQ) < 2e-8 and _rel(small["p"], op) < 2e-8 and _rel(small["lam"], o.lam) < 2e-8


@pytest.mark.parametrize("k,nx", [(1, 130), (2, 200), (2, 65)])
def test_paired_advection_kernel_equals_the_gather_form(hip_lib, k, nx, monkeypatch):
    """k_adv_pair (both triangles of 64 squares per workgroup, neighbour traces through LDS; HDG_ADV_PAIR=1, off by default: no
    gain measured) against k_adv_apply on meshes with full and partial blocks of 64 squares, both fluxes, plain and through a
    whole step: the neighbour trace is the same sum formed by another thread -- equal to rounding."""
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    rng = np.random.default_rng(123456789)
    out = {}
    for flux in ("upwind", "centered"):
        for pair in ("0", "1"):
            monkeypatch.setenv("HDG_ADV_PAIR", pair)  # read when an engine is built
            ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, 0.25 / nx, flux=flux)
            e = ts._engine
            if (flux, "x") not in out:
                out[flux, "x"] = rng.standard_normal(e.shape_Q)
                out[flux, "q"] = e.project_bdm_nodal(rng.standard_normal(e.shape_Q))
            out[flux, pair] = e.apply_advection(out[flux, "q"], out[flux, "x"], 0.3 / nx)
            if flux == "upwind":
                mp = TaylorGreen(ts._V_Q, ts._V_p)
                Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), 0.25 / nx, fused=True)
                out["step", pair] = (Q.dat.data.copy(), p.dat.data.copy())
        assert _rel(out[flux, "1"], out[flux, "0"]) < 1e-13, flux
    assert _rel(out["step", "1"][0], out["step", "0"][0]) < 1e-9 and _rel(out["step", "1"][1], out["step", "0"][1]) < 1e-9


@pytest.mark.parametrize("k,level", [(1, 3), (2, 3), (3, 2)])
def test_matrix_free_general_mesh_lift_equals_the_assembled_operators(hip_lib, k, level, monkeypatch):
    """k_g_lift (reference moment tables in LDS x per-cell geometry, one per-cell lifting matrix) against the CSR form of the BDM
    projection and of the hybrid preconditioner Pi + Dinv (I - Pi) on the unit disk (HDG_GENERAL_CSR_LIFT, read when an engine
    is built): the projection to 1e-12, whole Kelvin-Helmholtz steps (Chebyshev + s-step solver at k >= 2, GMRES at k = 1) to 1e-9
    with the same Krylov counts (the assembled path is compared with the oracle in tests/test_gpu_general_mesh.py)."""
    from incompressibleeulerhdg_amd.mesh import UnitDiskMesh
    from incompressibleeulerhdg_amd.model_problems import KelvinHelmholtz
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    rng = np.random.default_rng(123456789)
    res = {}
    x = None
    for tag in ("free", "csr"):
        if tag == "csr":
            monkeypatch.setenv("HDG_GENERAL_CSR_LIFT", "1")
        else:
            monkeypatch.delenv("HDG_GENERAL_CSR_LIFT", raising=False)
        ts = IncompressibleEulerHDGIMEXSSP2_332(UnitDiskMesh(level), k, 0.01, use_projection_method=True, n_richardson=2)
        e = ts._engine
        if x is None:
            x = rng.standard_normal(e.shape_Q)
        P = e.project_bdm_nodal(x)
        kh = KelvinHelmholtz(ts._V_Q, ts._V_p)
        Q, p = ts.solve(*kh.initial_condition(), None, kh.f_rhs(), 0.02, fused=True)
        sums, cnt = e.iteration_stats()
        res[tag] = (P, Q.dat.data.copy(), p.dat.data.copy(), sums / np.maximum(cnt, 1))
    assert _rel(res["free"][0], res["csr"][0]) < 1e-12
    assert _rel(res["free"][1], res["csr"][1]) < 1e-9 and np.max(np.abs(res["free"][2] - res["csr"][2])) < 1e-9 * max(np.max(np.abs(res["csr"][2])), 1.0)
    assert np.all(np.abs(res["free"][3] - res["csr"][3]) <= 1.0)
    # idempotent, as a projection must be
    assert _rel(e.project_bdm_nodal(res["csr"][0]), res["csr"][0]) < 1e-11
