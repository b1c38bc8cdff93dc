"""Edge cases and error behaviour of the C-ABI on the GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SSP2 = dict(nstages=3, a_expl=[[0, 0, 0], [0.5, 0, 0], [0.5, 0.5, 0]], a_impl=[[0.25, 0, 0], [0, 0.25, 0], [1 / 3, 1 / 3, 1 / 3]],
            b_expl=[1 / 3] * 3, b_impl=[1 / 3] * 3, c_expl=[0, 1, 0.5])


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("k,nx", [(1, 1), (1, 2), (1, 3), (2, 5), (1, 7), (4, 2)])
def test_small_and_odd_meshes_match_oracle(hip_lib, k, nx):
    """nx not a power of two (single or no multigrid coarsening), down to one square."""
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332
    from oracle import hdg_oracle as orc

    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, use_projection_method=True)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), 2 * dt, fused=True)
    d = orc.HDGDiscretisation(nx, k)
    tg = orc.TaylorGreen(d)
    oQ, op = orc.OracleHDGIMEX(d, dt, "imex_ssp2_332").solve(*tg.initial_condition(), tg.f_rhs, 2 * dt)
    assert _rel(Q.dat.data, oQ) < 2e-8 and _rel(p.dat.data, op) < 2e-8


def test_bad_arguments_return_error_codes(hip_lib):
    from incompressibleeulerhdg_amd._lib import Engine, HDGError

    for bad in (dict(degree=0), dict(degree=5), dict(nx=0), dict(dt=0.0), dict(ny=5)):
        kw = dict(nx=4, degree=1, dt=0.1, **SSP2)
        kw.update(bad)
        with pytest.raises(HDGError) as ei:
            Engine(**kw)
        assert ei.value.code == -1, bad
    e = Engine(nx=4, degree=1, dt=0.1, **SSP2)
    for call in (lambda: e.tentative_solve(0), lambda: e.tentative_solve(3), lambda: e.pressure_solve(7),
                 lambda: e.project_bdm(5, 0), lambda: e.get_field(999), lambda: e.set_forcing_scale(9, 1.0),
                 lambda: e.stage_update(0)):
        with pytest.raises(HDGError) as ei:
            call()
        assert ei.value.code == -1
    with pytest.raises(ValueError):
        e.set_state(np.zeros((3, 2)), np.zeros(5))  # wrong shapes are caught before crossing the ABI


def test_krylov_failure_is_an_error_not_silent(hip_lib):
    """Krylov max-it is an error code, never silent."""
    from incompressibleeulerhdg_amd._lib import Engine, HDGError

    e = Engine(nx=8, degree=1, dt=0.03, tent_maxit=2, **SSP2)
    rng = np.random.default_rng(0)
    e.set_state(rng.standard_normal(e.shape_Q), rng.standard_normal(e.shape_p))
    e.reconstruct_trace()
    for i in range(4):
        e.set_forcing_scale(i, 0.0)
    e.begin_step()
    e.project_bdm(0, 0)
    with pytest.raises(HDGError) as ei:
        e.tentative_solve(1)
    assert ei.value.code == -3
    e2 = Engine(nx=8, degree=1, dt=0.03, trace_maxit=1, **SSP2)
    e2.set_state(rng.standard_normal(e2.shape_Q), rng.standard_normal(e2.shape_p))
    e2.reconstruct_trace()
    for i in range(4):
        e2.set_forcing_scale(i, 0.0)
    e2.begin_step()
    with pytest.raises(HDGError) as ei:
        e2.pressure_solve(0)
    assert ei.value.code == -3


def test_zero_forcing_and_zero_state(hip_lib):
    """kappa == 0 (zero forcing, SURVEY C-6) and an all-zero state are fixed points."""
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    nx, k = 6, 1
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, 0.02)
    Q, p = ts.solve(lambda x, y: (0 * x, 0 * x), lambda x, y: 0 * x, None, 0, 0.04, fused=True)
    assert np.max(np.abs(Q.dat.data)) == 0.0 and np.max(np.abs(p.dat.data)) == 0.0
    mp = TaylorGreen(ts._V_Q, ts._V_p, "exponential", 0.0)
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), 0.04, fused=True)
    Qe, _ = mp.solution(0.04)
    assert ts._engine.l2_norms(Q.dat.data - Qe.dat.data)[0] < 5e-3  # stationary vortex preserved to discretisation error


def test_reproducible_bitwise(hip_lib):
    """Owner-computes gather + deterministic reductions: two runs give identical bits."""
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    out = []
    for _ in range(2):
        ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(16, 16), 2, 0.25 / 16)
        mp = TaylorGreen(ts._V_Q, ts._V_p)
        Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), 0.5 / 16, fused=True)
        out.append((Q.dat.data.copy(), p.dat.data.copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


def test_chebyshev_with_wrong_bounds_falls_back_to_gmres(hip_lib, tmp_path):
    """The Chebyshev phase of the tentative solver is driven by estimated spectral bounds.  With the
    upper bound deliberately halved (HDG_CHEB_FHI=0.5) the iteration diverges; the solver must notice,
    finish with GMRES and still deliver the converged answer (parity with the oracle)."""
    import os
    import subprocess
    import sys

    from oracle import hdg_oracle as orc

    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / "r.npz")
    k, nx, nsteps = 2, 8, 2
    env = dict(os.environ, HDG_CHEB_FHI="0.5", HDG_DEBUG="1")
    r = subprocess.run([sys.executable, os.path.join(here, "mp_strip_worker.py"), "0", "1", "unused", str(k), str(nx),
                        str(nsteps), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    got = np.load(out)
    d = orc.HDGDiscretisation(nx, k)
    tg = orc.TaylorGreen(d)
    dt = 0.25 / nx
    oQ, op = orc.OracleHDGIMEX(d, dt, "imex_ssp2_332").solve(*tg.initial_condition(), tg.f_rhs, nsteps * dt)
    assert _rel(got["Q"], oQ) < 2e-8 and _rel(got["p"], op) < 2e-8
    assert "falling back to GMRES" in r.stdout.decode()


def test_fused_vcycle_kernels_are_bitwise_equal_to_the_per_level_launches(hip_lib, tmp_path):
    """The P1 V-cycle of the trace preconditioner runs its legs as fused LDS-tile kernels (k_p1_down /
    k_p1_up, levels with n > 32) and its tail in one workgroup (k_p1_vcycle_tail; HDG_MG_NO_DENSE_TAIL).  Both are pure
    re-schedulings of the per-level launches: the whole time step must come out bit-identical with
    either switched off.  nx = 128, 96: fused levels n = 128, 64 / 96 with partial tiles at the far boundary;
    nx = 512: enough workgroups in flight that an in-place halo race between tiles shows (it did)."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    for nx in (128, 96, 512):
        res = {}
        for tag, extra in (("fused", {"HDG_MG_NO_DENSE_TAIL": "1"}), ("unfused", {"HDG_MG_NO_FUSE": "1", "HDG_MG_NO_DENSE_TAIL": "1"}),
                           ("plain", {"HDG_MG_NO_FUSE": "1", "HDG_MG_NO_TAIL": "1"}), ("dense", {})):
            out = str(tmp_path / f"{tag}{nx}.npz")
            r = subprocess.run([sys.executable, os.path.join(here, "mp_strip_worker.py"), "0", "1", "unused", "1", str(nx),
                                "1", out], env=dict(os.environ, **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               timeout=600)
            assert r.returncode == 0, r.stdout.decode()[-2000:]
            res[tag] = np.load(out)
        for name in ("Q", "p", "lam", "its"):
            assert np.array_equal(res["fused"][name], res["unfused"][name]), (nx, name)
            assert np.array_equal(res["fused"][name], res["plain"][name]), (nx, name)
        # round 4 (default): the tail as ONE dense matrix-vector product (k_p1_dense_tail) -- the same linear map built from the
        # tail kernel's action on unit vectors, another summation order: equal to rounding, same iteration counts
        for name in ("Q", "p", "lam"):
            assert _rel(res["dense"][name], res["fused"][name]) < 1e-10, (nx, name)
        assert np.all(np.abs(res["dense"]["its"][1:] - res["fused"]["its"][1:]) <= 0.5), nx


def test_alternative_trace_solver_paths_agree(hip_lib, tmp_path):
    """The fused smoother step (k_trace_smooth) and the device-resident CG scalars are re-schedulings of the same
    algorithm (different rounding order only): switching either off must give the same fields to well below the
    solver tolerance and the same CG iteration counts."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for tag, extra in (("default", {}), ("unfused_smoother", {"HDG_TRACE_NO_FUSE": "1"}),
                       ("host_scalars", {"HDG_CG_HOST_SCALARS": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, os.path.join(here, "mp_strip_worker.py"), "0", "1", "unused", "2", "48", "2", out],
                           env=dict(os.environ, **extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0, r.stdout.decode()[-2000:]
        res[tag] = np.load(out)
    for tag in ("unfused_smoother", "host_scalars"):
        for name in ("Q", "p", "lam"):
            assert _rel(res[tag][name], res["default"][name]) < 1e-9, (tag, name)
        assert np.all(np.abs(res[tag]["its"][1:] - res["default"]["its"][1:]) <= 1.0), tag


def test_section_timers_across_the_abi(hip_lib):
    """hdg_get_timers: the reference's PerformanceLog labels (logging.py:34-60; hdg_imex.py:257,274,551,564) for a
    fused step -- call counts follow the loop structure (s-1 projections, (s-1) R tentative solves, (s-1) R + 2
    mixed-Poisson solves per step), the parts add up to no more than the whole, and log_summary prints them."""
    import io

    from incompressibleeulerhdg_amd.auxilliary.logging import PerformanceLog, log_summary
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    nx, k, nsteps = 32, 2, 3
    dt = 0.25 / nx
    PerformanceLog.reset()
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, n_richardson=2)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    e = ts._engine
    e.set_state(ts._V_Q.interpolate(mp.Q_stationary), ts._V_p.interpolate(mp.p_stationary))
    e.reconstruct_trace()
    e.set_forcing_profile(mp.f_rhs().profile)
    e.timers(reset=True)
    for n in range(nsteps):
        for sl in range(4):
            e.set_forcing_scale(sl, -0.5)
        e.step()
    t = e.timers()
    assert t["timestep"][0] == nsteps and t["bdm_projection"][0] == 2 * nsteps
    assert t["tentative_velocity_solve"][0] == 4 * nsteps and t["pressure_solve"][0] == 6 * nsteps
    assert t["unsplit_solve"][0] == 0
    parts = t["bdm_projection"][1] + t["tentative_velocity_solve"][1] + t["pressure_solve"][1]
    assert 0 < parts <= t["timestep"][1] * 1.001
    assert parts > 0.5 * t["timestep"][1]  # the three sections are the bulk of a step
    assert e.timers(reset=True)["timestep"][0] == nsteps and e.timers()["timestep"][0] == 0
    # kernel-level brackets (hdg_set_kernel_timing): off by default; on, every launch of the two kernels of a
    # tentative-velocity iteration is timed in place: as many as iterations (+ the first residual of each solve), and their
    # sum stays inside the tentative-velocity section
    assert e.timers(kernels=True)["kernel_advection"][0] == 0
    e.set_kernel_timing(True)
    e.iteration_stats(reset=True)
    for sl in range(4):
        e.set_forcing_scale(sl, -0.5)
    e.step()
    tk = e.timers(reset=True, kernels=True)
    sums, cnt = e.iteration_stats()
    e.set_kernel_timing(False)
    # one label per FORM (round 4): residual-form operator / lift with the fused Chebyshev step (Chebyshev phase), plain operator /
    # plain lift (opening Arnoldi cycle, first residual of a solve, the s-step cycles of the tail).  Every iteration is one
    # operator + one lift; a solve adds its first residual (and one more when it opens with an Arnoldi cycle)
    n_adv = tk["kernel_advection"][0] + tk["kernel_advection_plain"][0]
    n_lift = tk["kernel_lift"][0] + tk["kernel_lift_plain"][0]
    assert tk["kernel_advection"][0] > 0 and tk["kernel_lift"][0] > 0 and tk["kernel_advection_plain"][0] > 0 and tk["kernel_lift_plain"][0] > 0
    assert sums[0] <= n_lift <= sums[0] + 3 * cnt[0], (n_lift, sums, cnt)
    assert sums[0] <= n_adv <= sums[0] + 3 * cnt[0], (n_adv, sums, cnt)
    assert 0 < sum(tk[lab][1] for lab in ("kernel_advection", "kernel_lift", "kernel_advection_plain", "kernel_lift_plain")) <= tk["tentative_velocity_solve"][1] * 1.001
    e.step()
    assert e.timers(kernels=True)["kernel_advection"][0] == 0  # switched off again
    e.timers(reset=True)
    # the class surface feeds them to the reference's summary table when the steps are fused
    ts.solve(*mp.initial_condition(), None, mp.f_rhs(), 2 * dt, fused=True)
    buf = io.StringIO()
    log_summary(file=buf)
    out = buf.getvalue()
    for label in ("timestep", "bdm_projection", "tentative_velocity_solve", "pressure_solve"):
        assert label in out, out


def test_bench_line_contract(hip_lib):
    """`python bench.py` prints ONE JSON line with the driver's keys, the roofline block (durations measured in place for
    the two kernels of a tentative-velocity iteration) and, unless switched off, the CPU baseline with the reference's
    timer labels; run here on a small mesh (the schema does not depend on the size)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k, extra in ((2, ["--no-cpu-baseline"]), (3, ["--no-cpu-baseline"])):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--nx", "64", "--degree", str(k), "--steps", "2", "--warmup", "1"] + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "config", "roofline", "timers"):
            assert key in d, key
        assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
        assert d["unit"] == "million DOF-updates/s" and d["value"] > 0 and abs(d["value"] - d["config"]["n_dof"] * 2 / (d["ms_per_step"] * 2e-3) / 1e6) < 1e-6 * d["value"]
        rf = d["roofline"]
        assert rf["bound"] == ("hbm" if k <= 2 else "mfma") and rf["unit"] == ("GB/s" if k <= 2 else "TFLOP/s")
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["launches_timed"] > 0 and rf["timing"].startswith("in place")
        assert set(d["timers"]) >= {"timestep", "tentative_velocity_solve", "pressure_solve", "bdm_projection"}
        assert d["timers"]["timestep"]["ncall"] == 2
        # round 4: residual replacements / floor exits of the condensed solves in the timed steps (none on a healthy run) and the
        # kernel forms as the engine reports them
        ev = d["config"]["solver_events"]
        assert ev["cg_residual_replacements"] == 0 and ev["cg_floor_exits"] == 0 and ev["sstep_gmres_fallbacks"] == 0 and ev["sstep_cycles"] >= 0
        assert d["config"]["kernel_forms"]["lift"] == (0 if k <= 2 else 2) and d["config"]["kernel_forms"]["trace_precond"] == 1
        assert "alt_stop_rule" not in d  # opt-in only (BENCH_ALT_STOP=1)


def test_rccl_selftest_loopback(hip_lib):
    """hdg_rccl_selftest: the grouped ncclSend / ncclRecv pattern of a halo exchange (with this rank as both neighbours), an
    all-reduce and an all-gather on a 1-rank RCCL communicator -- the part of the RCCL transport a one-GPU box can run
    (two ranks on one device are refused by RCCL; the multi-rank logic is covered by the shared-memory transport)."""
    import ctypes as C

    from incompressibleeulerhdg_amd import _lib

    lib = _lib.load_library()
    for n in (1, 1000, 1 << 20):
        err = C.c_double(-1.0)
        rc = lib.hdg_rccl_selftest(0, n, C.byref(err))
        assert rc == 0, lib.hdg_last_error(None).decode()
        assert err.value == 0.0, (n, err.value)
    assert lib.hdg_rccl_selftest(0, 0, C.byref(err)) != 0


def test_backward_error_stop_of_the_condensed_solves(hip_lib, monkeypatch):
    """HDG_TRACE_BACKWARD_TOL (experiment, off by default; bench.py's secondary number): the condensed solves of the projection
    method stop at |M r| <= 1e-15 |pressure trace| or at the reference's rtol, whichever comes first.  Same fields as the
    reference's rule to 1e-9 (measured: 2e-11 after four steps), fewer CG iterations, and the final-stage solve (vanishing right-hand side) ends at once."""
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    def run(tol):
        if tol:
            monkeypatch.setenv("HDG_TRACE_BACKWARD_TOL", tol)
        else:
            monkeypatch.delenv("HDG_TRACE_BACKWARD_TOL", raising=False)
        nx, k = 64, 2
        ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, 0.25 / nx, use_projection_method=True, n_richardson=2)
        mp = TaylorGreen(ts._V_Q, ts._V_p)
        Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), 4 * 0.25 / nx, fused=True)
        sums, cnt = ts._engine.iteration_stats()
        return Q.dat.data.copy(), p.dat.data.copy(), sums / cnt

    Q0, p0, i0 = run(None)
    Q1, p1, i1 = run("1e-15")
    # (two converged Krylov runs: the tentative-velocity solves stop at 1e-10 of their initial residual in either run)
    assert np.max(np.abs(Q1 - Q0)) < 1e-9 * np.max(np.abs(Q0)) and np.max(np.abs(p1 - p0)) < 1e-9 * max(np.max(np.abs(p0)), 1.0)
    assert i1[1] < i0[1] - 0.5 and i1[2] < i0[2] - 2.0 and i1[3] <= i0[3]  # (the first step has no scale yet: the reference rule)
