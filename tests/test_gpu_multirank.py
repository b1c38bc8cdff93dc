"""Strip-partitioned engine: P ranks (separate processes sharing the one GPU of the test box, shared-
memory transport) must reproduce the single-rank fields to Krylov tolerance (SURVEY.md section 8e,
"correctness gate") and take the same iteration counts up to reduction order."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_ranks(nranks, k, nx, nsteps, tmp_path, extra=(), env=None, want_logs=False):
    token = "/hdg_test_" + uuid.uuid4().hex[:12]
    procs, outs = [], []
    for r in range(nranks):
        out = str(tmp_path / f"rank{r}.npz")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "mp_strip_worker.py"), str(r), str(nranks), token,
                                       str(k), str(nx), str(nsteps), out, *extra],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      # HDG_OVERLAP=1: the interior / boundary split around every exchange (the default over RCCL;
                                      # off by default over the shared-memory transport these tests use) is what gets tested
                                      env=dict(os.environ, HDG_OVERLAP="1", **(env or {}))))
    logs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    if want_logs:
        return [np.load(o) for o in outs], logs
    return [np.load(o) for o in outs]


def _assemble(parts, k, nx):
    """Concatenate the strips into the global boundary numbering of a single rank."""
    P = len(parts)
    nyl = nx // P
    nu, npp, nl = (k + 2) * (k + 3) // 2, (k + 1) * (k + 2) // 2, k + 1
    Q = np.concatenate([d["Q"] for d in parts])  # cells are row-major in j, both shapes adjacent: strips concatenate
    p = np.concatenate([d["p"] for d in parts])
    NH_l, NV_l, ND_l = nx * (nyl + 1), (nx + 1) * nyl, nx * nyl
    H, V, D = [], [], []
    for r, d in enumerate(parts):
        lam = d["lam"].reshape(-1, nl)
        h = lam[:NH_l].reshape(nyl + 1, nx, nl)
        if r < P - 1:  # the top row of a lower rank duplicates the upper rank's bottom row
            upper = parts[r + 1]["lam"].reshape(-1, nl)[:NH_l].reshape(nyl + 1, nx, nl)
            assert np.allclose(h[-1], upper[0], rtol=0, atol=1e-13)
        H.append(h if r == P - 1 else h[:-1])
        V.append(lam[NH_l:NH_l + NV_l])
        D.append(lam[NH_l + NV_l:])
    lam = np.concatenate([np.concatenate(H).reshape(-1, nl), np.concatenate(V), np.concatenate(D)]).ravel()
    return Q, p, lam


# (3, 2, 30), (2, 1, 20), (3, 3, 27): strips of 10 / 10 / 9 rows -- the tiled trace preconditioner with tile rows that do not divide the
# strip, three ranks (a middle rank with two neighbours), the replicated coarse-grid path
@pytest.mark.parametrize("nranks,k,nx", [(2, 1, 8), (2, 2, 8), (4, 1, 8), (3, 2, 6), (2, 1, 6), (2, 3, 4), (2, 2, 96), (2, 3, 40), (4, 4, 24), (2, 1, 128), (4, 2, 256),
                                         (3, 2, 30), (2, 1, 20), (3, 3, 27)])
def test_strip_partition_matches_single_rank(hip_lib, tmp_path, nranks, k, nx):
    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    nsteps = 2
    parts = _run_ranks(nranks, k, nx, nsteps, tmp_path)
    Q, p, lam = _assemble(parts, k, nx)
    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, use_projection_method=True, n_richardson=2)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    Q1, p1 = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), nsteps * dt, fused=True)
    lam1 = ts._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert rel(Q, Q1.dat.data) < 2e-8 and rel(p, p1.dat.data) < 2e-8 and rel(lam, lam1) < 2e-8
    # global reductions agree on every rank
    for d in parts[1:]:
        assert abs(float(d["eq"]) - float(parts[0]["eq"])) < 1e-13 and np.allclose(d["its"], parts[0]["its"])
    s1, c1 = ts._engine.iteration_stats()
    its1 = s1 / np.maximum(c1, 1)
    # condensed CG: same counts up to reduction order.  Tentative velocity: the Chebyshev iteration is driven by
    # Ritz estimates (they move with the reduction order) and checks convergence every 4th iteration only
    assert np.all(np.abs(parts[0]["its"][1:] - its1[1:]) <= 1.0)
    assert abs(parts[0]["its"][0] - its1[0]) <= max(4.0, 0.25 * its1[0])


@pytest.mark.parametrize("nranks,k,nx,opts", [
    (2, 2, 16, {"tent_precond": 1}),                    # additive two-level preconditioner: Pi^T, then lift + block-Jacobi
    (2, 1, 16, {"tent_precond": 0}),                    # block-Jacobi only (no stencil in the preconditioner)
    (3, 2, 12, {"tent_solver": 0}),                     # GMRES(8) instead of the Chebyshev iteration
    (2, 2, 16, {"tent_solver": 0, "tent_precond": 1}),
    (2, 2, 16, {"trace_precond": 0}),                   # edge block-Jacobi instead of the multigrid V-cycle
    (4, 1, 16, {"gmres_restart": 4, "tent_solver": 0}),
])
def test_strip_partition_with_the_other_solver_paths(hip_lib, tmp_path, nranks, k, nx, opts):
    """The ghost-row bookkeeping (Engine::Flow) serves every solver configuration: each alternative preconditioner /
    Krylov method on several ranks reproduces the single-rank run with the same options."""
    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    nsteps = 2
    parts = _run_ranks(nranks, k, nx, nsteps, tmp_path, extra=tuple(f"opt:{a}={b}" for a, b in opts.items()))
    Q, p, lam = _assemble(parts, k, nx)
    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, use_projection_method=True, n_richardson=2, **opts)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    Q1, p1 = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), nsteps * dt, fused=True)
    lam1 = ts._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert rel(Q, Q1.dat.data) < 2e-8 and rel(p, p1.dat.data) < 2e-8 and rel(lam, lam1) < 2e-8


@pytest.mark.parametrize("nranks,k,nx,opts", [(2, 2, 16, {}), (4, 1, 16, {}), (2, 3, 40, {}), (2, 1, 128, {}), (2, 2, 16, {"tent_solver": 0}),
                                              (3, 2, 12, {"tent_precond": 1}), (2, 4, 16, {})])
def test_ghost_row_bookkeeping_self_check(hip_lib, tmp_path, nranks, k, nx, opts):
    """HDG_FLOW_CHECK: every exchange the bookkeeping (Engine::Flow) skips is carried out anyway and the received rows are
    compared with the ghost rows in place; a solver call fails if they differ by more than 1e-9 relative.  The rows a rank
    computes redundantly agree with its neighbour's to rounding (here: exactly), a stale validity would be an O(1) error."""
    import re

    parts, logs = _run_ranks(nranks, k, nx, 2, tmp_path, extra=tuple(f"opt:{a}={b}" for a, b in opts.items()),
                             env={"HDG_FLOW_CHECK": "1", "HDG_DEBUG": "1"}, want_logs=True)
    m = re.search(r"\[flow check\] (\d+) skipped exchanges verified, worst relative deviation ([0-9.eE+-]+)", logs[0])
    assert m, logs[0][-2000:]
    assert int(m.group(1)) > 20 and float(m.group(2)) < 1e-12, m.group(0)
    # the exchanges that were NOT skipped ran beside an interior launch (interior / boundary split on a second stream)
    m2 = re.search(r"\((\d+) of them beside an interior launch\)", logs[0])
    assert m2 and int(m2.group(1)) > 10, logs[0][-2000:]
    # round 4: strips of >= 5 rows run the LDS-tiled trace preconditioner of the one-GPU path (one exchange of r, 5 rows deep, per
    # CG iteration; k = 4: the edge-per-thread form, hdg_trace_tile3.hpp); shorter strips keep the row-stencil kernels
    m3 = re.search(r"tiled trace preconditioner applications (\d+)", logs[0])
    assert m3, logs[0][-2000:]
    assert (int(m3.group(1)) > 20) == (nx // nranks >= 5), (m3.group(0), k, nx, nranks)


def test_strip_partition_at_the_benchmark_size(hip_lib, tmp_path):
    """C3's mesh (k = 2, 1024^2) on 4 ranks: 256 rows per rank, 4-row halos, distributed finest multigrid level, every skipped
    exchange verified (HDG_FLOW_CHECK); one step against the single-rank run."""
    import re

    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    k, nx, nranks = 2, 1024, 4
    parts, logs = _run_ranks(nranks, k, nx, 1, tmp_path, env={"HDG_FLOW_CHECK": "1", "HDG_DEBUG": "1"}, want_logs=True)
    Q, p, lam = _assemble(parts, k, nx)
    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, use_projection_method=True, n_richardson=2)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    Q1, p1 = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), dt, fused=True)
    lam1 = ts._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert rel(Q, Q1.dat.data) < 2e-8 and rel(p, p1.dat.data) < 2e-8 and rel(lam, lam1) < 2e-8
    s1, c1 = ts._engine.iteration_stats()
    assert np.all(np.abs(parts[0]["its"][1:] - (s1 / np.maximum(c1, 1))[1:]) <= 1.0)
    m = re.search(r"\[flow check\] (\d+) skipped exchanges verified, worst relative deviation ([0-9.eE+-]+)", logs[0])
    # (round 3: > 100 per step with the row-stencil preconditioner; the tiled one exchanges r once per CG iteration and skips none)
    assert m and int(m.group(1)) > 50 and float(m.group(2)) < 1e-12, logs[0][-1500:]
    m3 = re.search(r"tiled trace preconditioner applications (\d+)", logs[0])
    assert m3 and int(m3.group(1)) >= 60, logs[0][-1500:]  # the strips run the one-GPU path's tile kernels


def test_strip_partition_unsplit(hip_lib, tmp_path):
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    k, nx = 1, 8
    parts = _run_ranks(2, k, nx, 1, tmp_path, extra=("unsplit",))
    Q, p, lam = _assemble(parts, k, nx)
    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, use_projection_method=False)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    Q1, p1 = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), dt, fused=True)
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert rel(Q, Q1.dat.data) < 2e-8 and rel(p, p1.dat.data) < 2e-8


def test_failed_time_kernel_leaves_halos_on(hip_lib, tmp_path):
    """hdg_time_kernel switches the halo exchanges off while it launches bare kernels; an unknown id (or a HIP
    error) must not leave them off -- every later stencil operator would silently read stale ghost rows."""
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    k, nx = 2, 8
    parts = _run_ranks(2, k, nx, 1, tmp_path, extra=("badtimer",))
    Q, p, lam = _assemble(parts, k, nx)
    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, use_projection_method=True, n_richardson=2)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    Q1, p1 = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), dt, fused=True)
    rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
    assert rel(Q, Q1.dat.data) < 2e-8 and rel(p, p1.dat.data) < 2e-8


def test_bench_self_launch_reports_the_ranks_that_ran(hip_lib):
    """`python bench.py --gpus N` outside torchrun starts the N ranks itself and reports n_gpus = ranks that ran.
    On a box with fewer devices than ranks the run is a labelled shared-device rehearsal, refused beyond 4 ranks
    per device (never a 1-rank number printed as N)."""
    import json

    import torch

    root = os.path.dirname(HERE)
    ndev = torch.cuda.device_count()
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--nx", "64", "--degree", "1", "--steps", "1",
           "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["n_devices"] == min(2, ndev)
    assert ("rehearsal" in line["config"]["parallelism"]) == (ndev < 2)
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    if ndev * 4 < 8:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--nx", "64", "--degree", "1",
                            "--steps", "1", "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode != 0 and not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
