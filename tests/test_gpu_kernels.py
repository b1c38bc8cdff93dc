"""GPU parity of the individual HIP operators against the CPU oracle (through the C-ABI).

The oracle assembles the reference's UFL forms as global sparse matrices in nodal bases
(oracle/hdg_oracle.py); the product evaluates the same operators matrix-free in modal bases on the
device.  Tolerances are floating-point tolerances for float64 operators applied once:
relative 1e-11 in max-norm ("<= 1e-11 relative per kernel").
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu

RTOL = 1e-11
CASES = [(1, 4), (1, 7), (2, 6), (3, 4), (4, 3)]


_DISC = {}


def _disc(nx, k):
    """The oracle's discretisation (read-only in these tests; building one takes 15-30 s at k >= 3, nx ~ 20)."""
    from oracle.hdg_oracle import HDGDiscretisation

    if (nx, k) not in _DISC:
        _DISC[(nx, k)] = HDGDiscretisation(nx, k)
    return _DISC[(nx, k)]


def _setup(k, nx, **kw):
    from incompressibleeulerhdg_amd._lib import Engine
    from oracle.hdg_oracle import TABLEAUX

    d = _disc(nx, k)
    tb = TABLEAUX["imex_ssp2_332"]
    e = Engine(nx=nx, degree=k, dt=0.25 / nx, nstages=3, a_expl=tb["a_expl"], a_impl=tb["a_impl"],
               b_expl=tb["b_expl"], b_impl=tb["b_impl"], c_expl=tb["c_expl"], **kw)
    return d, e


def _relerr(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("k,nx", CASES)
def test_sizes_and_nodes(hip_lib, k, nx):
    d, e = _setup(k, nx)
    assert e.n_cells == d.mesh.ncells and e.n_edges == d.mesh.nedges
    assert (e.n_u, e.n_p, e.n_l) == (d.nu, d.np_, d.nl)
    xq, xp = e.node_coordinates()
    assert np.allclose(xq, d.node_coords(d.PU).reshape(-1, 2), atol=1e-14)
    assert np.allclose(xp, d.node_coords(d.PP).reshape(-1, 2), atol=1e-14)


@pytest.mark.parametrize("k,nx", CASES)
def test_nodal_modal_roundtrip_and_norms(hip_lib, k, nx):
    d, e = _setup(k, nx)
    rng = np.random.default_rng(123456789)
    Q = rng.standard_normal(e.shape_Q)
    p = rng.standard_normal(e.shape_p)
    lam = rng.standard_normal(e.shape_l)
    e.set_field(1, Q, p, lam)
    Q2, p2, l2 = e.get_field(1)
    assert _relerr(Q2, Q) < 1e-12 and _relerr(p2, p) < 1e-12 and _relerr(l2, lam) < 1e-12
    nq, npr = e.l2_norms(Q, p)
    assert abs(nq - d.l2_norm_velocity(Q)) < 1e-11 * nq
    assert abs(npr - d.l2_norm_pressure(p)) < 1e-11 * npr
    assert abs(e.integrate_pressure(p) - d.int_p @ p) < 1e-12


@pytest.mark.parametrize("k,nx", CASES)
def test_project_bdm(hip_lib, k, nx):
    d, e = _setup(k, nx)
    rng = np.random.default_rng(1)
    Q = rng.standard_normal(e.shape_Q)
    assert _relerr(e.project_bdm_nodal(Q), d.project_bdm(Q)) < RTOL


@pytest.mark.parametrize("flux", ["upwind", "centered"])
@pytest.mark.parametrize("k,nx", CASES)
def test_advection_apply(hip_lib, k, nx, flux):
    d, e = _setup(k, nx, flux=flux)
    rng = np.random.default_rng(2)
    Qstar = d.project_bdm(rng.standard_normal(e.shape_Q))
    x = rng.standard_normal(e.shape_Q)
    gamma = 0.3 / nx
    F = d.assemble_f_impl(Qstar, flux)
    ref = x.ravel() - gamma * spla.spsolve(d.MQ.tocsc(), F @ x.ravel())
    got = e.apply_advection(Qstar, x, gamma)
    assert _relerr(got.ravel(), ref) < 5e-11


@pytest.mark.parametrize("k,nx", CASES)
def test_weak_divergence(hip_lib, k, nx):
    d, e = _setup(k, nx)
    rng = np.random.default_rng(3)
    Q = rng.standard_normal(e.shape_Q)
    Mi = spla.splu(d.MP.tocsc())
    assert _relerr(e.apply_weak_divergence(Q), Mi.solve(d.Wdiv @ Q.ravel())) < RTOL
    assert _relerr(e.apply_weak_divergence(Q, broken=True), Mi.solve(d.Bdiv @ Q.ravel())) < RTOL


def _oracle_schur(d):
    """-S = Lm + [C E] A^{-1} [C^T; -E^T] of the oracle's mixed Poisson matrix, dense."""
    n1 = d.NQ + d.NP
    K = d.K_mp.tocsc()
    A = K[:n1, :n1]
    G = K[:n1, n1:]
    H = K[n1:, :n1]
    L = K[n1:, n1:]
    S = L.toarray() - H @ spla.splu(A.tocsc()).solve(G.toarray())
    return -S


@pytest.mark.parametrize("k,nx", CASES)
def test_trace_operator(hip_lib, k, nx):
    d, e = _setup(k, nx)
    rng = np.random.default_rng(4)
    lam = rng.standard_normal(e.shape_l)
    T = _oracle_schur(d)
    Mtr = d.Lm.tocsc() / d.tau  # sum over incidences of edge mass ...
    # ... the Riesz map of the product's orthonormal edge basis is the SINGLE edge mass matrix
    mult = np.where(np.repeat(d.mesh.interior, d.nl), 2.0, 1.0)
    Mtr = sp.diags(1.0 / mult) @ Mtr
    ref = spla.spsolve(Mtr.tocsc(), T @ lam)
    assert _relerr(e.apply_trace_operator(lam), ref) < 1e-10
    # constants are in the null space (hdg_imex.py:480-489)
    assert np.max(np.abs(e.apply_trace_operator(np.ones(e.shape_l)))) < 1e-9


@pytest.mark.parametrize("k,nx", CASES)
def test_trace_reconstruction_and_shift(hip_lib, k, nx):
    from incompressibleeulerhdg_amd import _lib

    d, e = _setup(k, nx)
    rng = np.random.default_rng(5)
    Q = rng.standard_normal(e.shape_Q)
    p = rng.standard_normal(e.shape_p)
    e.set_state(Q, p)
    e.reconstruct_trace()
    _, p_dev, lam = e.get_field(_lib.HDG_STATE_CURRENT)
    p0 = p - (d.int_p @ p) / d.mesh.volume
    assert _relerr(p_dev, p0) < RTOL
    assert _relerr(lam, d.reconstruct_trace(Q, p0)) < RTOL
    # _shift_pressure on (p, lambda)
    lam_r = rng.standard_normal(e.shape_l)
    e.set_field(1, Q, p, lam_r)
    e.shift_pressure(1)
    _, p1, l1 = e.get_field(1)
    ps, ls = d.shift_pressure(p, lam_r)
    assert _relerr(p1, ps) < RTOL and _relerr(l1, ls) < RTOL


def test_two_lane_advection_kernel_all_degrees(hip_lib):
    """k_adv_apply2 (one velocity component per lane, used for k = 3 by default) against the oracle for every
    degree: HDG_ADV_SPLIT=1:4 routes k = 1..4 through it (the switch is read once per process -> worker)."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "adv_split_worker.py")], env=dict(os.environ, HDG_ADV_SPLIT="1:4"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert r.stdout.decode().count("e-") == 8


def test_time_kernel_ids_and_errors(hip_lib):
    """hdg_time_kernel (bench.py's roofline probe): every documented kernel id runs and returns a positive
    duration, for both preconditioner variants; an unknown id is an argument error, not a crash."""
    from incompressibleeulerhdg_amd import _lib

    for tp in (1, 2):
        d, e = _setup(2, 16, tent_precond=tp)
        rng = np.random.default_rng(5)
        e.set_state(rng.standard_normal(e.shape_Q), rng.standard_normal(e.shape_p))
        for kid in range(15):
            ms = e.time_kernel(kid, 3)
            assert 0.0 < ms < 50.0, (tp, kid, ms)
        with pytest.raises(_lib.HDGError):
            e.time_kernel(99, 1)


@pytest.mark.parametrize("k,nx", [(3, 20), (4, 18)])
def test_project_bdm_matrix_core_kernel_tiles(hip_lib, k, nx):
    """k >= 3 runs the BDM lift on the matrix cores (k_edge_lift_mfma: one wave per 16 cells): meshes with full
    tiles and a partial last tile (nx not a multiple of 16) must match the oracle (more tiles than waves: the
    full-size property tests, k = 3 at 512^2)."""
    d, e = _setup(k, nx)
    rng = np.random.default_rng(11)
    Q = rng.standard_normal(e.shape_Q)
    assert _relerr(e.project_bdm_nodal(Q), d.project_bdm(Q)) < RTOL


@pytest.mark.parametrize("flux", ["upwind", "centered"])
@pytest.mark.parametrize("k,nx", [(3, 20), (4, 18)])
def test_advection_matrix_core_kernel_tiles(hip_lib, k, nx, flux):
    """k_adv_mfma is the default advection operator at k >= 3 (BASELINE C4 / C5).  One wave owns 16 consecutive
    cells: nx = 20 / 18 gives a full tile plus a partial one per row (columns i >= 16, the has2 / column clamps)
    and several rows per XCD band, so the rowN / rowN0 neighbour indexing meets the oracle's assembled f_impl
    (hdg_imex.py:313-331) for both fluxes -- not just linearity, which a wrong neighbour index would pass."""
    d, e = _setup(k, nx, flux=flux)
    rng = np.random.default_rng(12)
    Qstar = d.project_bdm(rng.standard_normal(e.shape_Q))
    x = rng.standard_normal(e.shape_Q)
    gamma = 0.3 / nx
    F = d.assemble_f_impl(Qstar, flux)
    ref = x.ravel() - gamma * spla.spsolve(d.MQ.tocsc(), F @ x.ravel())
    got = e.apply_advection(Qstar, x, gamma)
    assert _relerr(got.ravel(), ref) < 5e-11
    # the operator part alone (the identity would hide a small relative error in F): (x - y) / gamma = M^-1 F x
    Fx = (x.ravel() - ref) / gamma
    assert _relerr((x.ravel() - got.ravel()) / gamma, Fx) < 1e-9


def test_matrix_core_kernels_at_k2_behind_the_switch(hip_lib):
    """HDG_MFMA_K2 (the north_star's "MFMA at k >= 2", measured and rejected -- 1.8-2x slower than
    the per-thread advection kernel at C3) routes k = 2 through k_adv_mfma<2> / k_edge_lift_mfma<2>.  The switch is read
    once per process, so the check runs in a child: BDM projection and advection operator (both fluxes) against the
    oracle on a mesh with a full and a partial 16-cell tile per row."""
    import os
    import subprocess
    import sys

    code = (
        "import sys, numpy as np, scipy.sparse.linalg as spla\n"
        "sys.path.insert(0, %r)\n"
        "from incompressibleeulerhdg_amd._lib import Engine\n"
        "from oracle.hdg_oracle import HDGDiscretisation, TABLEAUX\n"
        "k, nx = 2, 20\n"
        "d = HDGDiscretisation(nx, k)\n"
        "tb = TABLEAUX['imex_ssp2_332']\n"
        "rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))\n"
        "for flux in ('upwind', 'centered'):\n"
        "    e = Engine(nx=nx, degree=k, dt=0.25 / nx, flux=flux, nstages=3, a_expl=tb['a_expl'], a_impl=tb['a_impl'],\n"
        "               b_expl=tb['b_expl'], b_impl=tb['b_impl'], c_expl=tb['c_expl'])\n"
        "    rng = np.random.default_rng(13)\n"
        "    Q = rng.standard_normal(e.shape_Q)\n"
        "    assert rel(e.project_bdm_nodal(Q), d.project_bdm(Q)) < 1e-11\n"
        "    Qstar, x, gamma = d.project_bdm(Q), rng.standard_normal(e.shape_Q), 0.3 / nx\n"
        "    F = d.assemble_f_impl(Qstar, flux)\n"
        "    ref = x.ravel() - gamma * spla.spsolve(d.MQ.tocsc(), F @ x.ravel())\n"
        "    got = e.apply_advection(Qstar, x, gamma)\n"
        "    assert rel(got.ravel(), ref) < 5e-11\n"
        "    assert rel((x.ravel() - got.ravel()) / gamma, (x.ravel() - ref) / gamma) < 1e-9\n"
        "print('ok')\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HDG_MFMA_K2="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("mfma_condense", [False, True])
@pytest.mark.parametrize("k,nx", [(3, 20), (4, 18)])
def test_schur_matrix_core_kernels_tiles(hip_lib, k, nx, mfma_condense, monkeypatch):
    """The GEMM-shaped Schur kernels (hdg_schur_mfma.hpp: k_backsub_mfma, k_condense_mfma, k_pgrad_mfma,
    k_weak_div_mfma; default at k >= 3, BASELINE C4 / C5) own 16 consecutive cells / corners per wave: meshes with a full
    and a partial tile per row (the small meshes of the other tests have partial tiles only).  One stage iteration, the
    final stage and the pressure reconstruction of an HDG-IMEX step, piece by piece against the oracle's direct solves
    (hdg_imex.py:239-247 tentative right-hand side with the pressure gradient, :177-179 weak divergence, :128-135 SCPC
    elimination / back-substitution in all three right-hand-side forms, :201-207 with the boundary term)."""
    from incompressibleeulerhdg_amd import _lib
    from oracle import hdg_oracle as orc

    # condensation runs on the per-thread kernel by default (it is the faster one); HDG_MFMA_CONDENSE, read when an engine is
    # built, selects k_condense_mfma
    if mfma_condense:
        monkeypatch.setenv("HDG_MFMA_CONDENSE", "1")
    else:
        monkeypatch.delenv("HDG_MFMA_CONDENSE", raising=False)
    d, e = _setup(k, nx)
    dt = 0.25 / nx
    o = orc.OracleHDGIMEX(d, dt, "imex_ssp2_332")
    rng = np.random.default_rng(40 + k)
    TOL = 2e-8  # two converged solvers
    # operators first: weak divergence in both forms (hdg_imex.py:353-365; hdg_implicit.py:145)
    Qr = rng.standard_normal(e.shape_Q)
    Mi = spla.splu(d.MP.tocsc())
    assert _relerr(e.apply_weak_divergence(Qr), Mi.solve(d.Wdiv @ Qr.ravel())) < RTOL
    assert _relerr(e.apply_weak_divergence(Qr, broken=True), Mi.solve(d.Bdiv @ Qr.ravel())) < RTOL
    # state: current fields, persistent stage iterates, nodal forcing in every slot
    Q0, p0 = rng.standard_normal(e.shape_Q), rng.standard_normal(e.shape_p)
    e.set_state(Q0, p0)
    e.reconstruct_trace()
    o.set_initial_condition(Q0, p0)
    for i in (1, 2):
        o.stage_Q[i], o.stage_p[i], o.stage_l[i] = (rng.standard_normal(e.shape_Q), rng.standard_normal(e.shape_p),
                                                    rng.standard_normal(e.shape_l))
        e.set_field(i, o.stage_Q[i], o.stage_p[i], o.stage_l[i])
    for sl in range(3):
        o.b_rhs[sl] = rng.standard_normal(e.shape_Q)
        e.set_forcing_nodal(sl, o.b_rhs[sl])
    b_new = rng.standard_normal(e.shape_Q)
    e.set_forcing_nodal(3, b_new)
    e.begin_step()
    o.stage_Q[0], o.stage_p[0], o.stage_l[0] = o.Q.copy(), o.p.copy(), o.lam.copy()
    # stage 1, one Richardson iteration: tentative velocity (k_pgrad_mfma builds its right-hand side) ...
    e.project_bdm(0, 0)
    Qstar = d.project_bdm(o.stage_Q[0])
    adt = o.a_impl[1, 1] * dt
    F = d.assemble_f_impl(Qstar, "upwind")
    Qi = o.stage_Q[1].ravel()
    rhs = o._residual(1) - d.MQ @ Qi + adt * (F @ Qi + d.G_p @ o.stage_p[1] + d.G_l @ o.stage_l[1])
    dQ = spla.splu((d.MQ - adt * F).tocsc()).solve(rhs)
    e.tentative_solve(1)
    assert _relerr(e.get_field(101, p=False, lam=False)[0].ravel(), dQ) < TOL
    # ... and its pressure correction (k_weak_div_mfma, k_condense_mfma<.,false,true>, k_backsub_mfma<.,false,true>)
    du, dp, dl = d.solve_mixed_poisson(rP=-(1.0 / adt) * (d.Wdiv @ dQ))
    dp, dl = d.shift_pressure(dp, dl)
    e.pressure_solve(1)
    e.shift_pressure(_lib.HDG_STATE_UPDATE)
    gu, gp, gl = e.get_field(_lib.HDG_STATE_UPDATE)
    assert _relerr(gu.ravel(), du) < TOL and _relerr(gp, dp) < TOL and _relerr(gl, dl) < TOL
    # final stage: velocity-row right-hand side (k_condense_mfma<.,true,false>, k_backsub_mfma<.,true,false>)
    u, _, _ = d.solve_mixed_poisson(rQ=o._final_residual())
    e.pressure_solve(_lib.HDG_KEY_FINAL_STAGE)
    gQ = e.get_field(_lib.HDG_STATE_CURRENT, p=False, lam=False)[0]
    assert _relerr(gQ.ravel(), u) < TOL
    # pressure reconstruction: pressure- and trace-row right-hand side (k_condense_mfma with r_lambda)
    rP, rL = d.pressure_reconstruction_rhs(u.reshape(-1, 2), b_new)
    _, p, lam = d.solve_mixed_poisson(rP=rP, rL=rL)
    p, lam = d.shift_pressure(p, lam)
    e.pressure_solve(_lib.HDG_KEY_PRESSURE_RECONSTRUCTION)
    e.finish_step()
    _, gp, gl = e.get_field(_lib.HDG_STATE_CURRENT)
    assert _relerr(gp, p) < TOL and _relerr(gl, lam) < TOL


@pytest.mark.parametrize("periodic", [False, True])
@pytest.mark.parametrize("k,nx", [(1, 72), (2, 128), (2, 200), (1, 130)])
def test_paired_lift_kernel_equals_the_gather_form(hip_lib, monkeypatch, k, nx, periodic):
    """k_edge_lift_pair (both triangles of a square in one workgroup, edge moments through LDS, hdg_kernels.hpp) against the
    gather form k_edge_lift on meshes with full and partial 64-square blocks, with and without periodic wrap-around: the BDM
    projection and the hybrid preconditioner inside a tentative-velocity solve (same sums, two additions per moment reordered)."""
    from incompressibleeulerhdg_amd._lib import Engine
    from oracle.hdg_oracle import TABLEAUX

    tb = TABLEAUX["imex_ssp2_332"]

    def make(no_pair):
        if no_pair:
            monkeypatch.setenv("HDG_LIFT_NO_PAIR", "1")
        else:
            monkeypatch.delenv("HDG_LIFT_NO_PAIR", raising=False)
        return Engine(nx=nx, degree=k, dt=0.25 / nx, nstages=3, a_expl=tb["a_expl"], a_impl=tb["a_impl"], b_expl=tb["b_expl"],
                      b_impl=tb["b_impl"], c_expl=tb["c_expl"], periodic=periodic, length=1.0)

    ea, eb = make(False), make(True)
    rng = np.random.default_rng(5 + k)
    Q = rng.standard_normal(ea.shape_Q)
    Pa, Pb = ea.project_bdm_nodal(Q), eb.project_bdm_nodal(Q)
    assert np.max(np.abs(Pa - Pb)) < 1e-12 * np.max(np.abs(Pb))
    assert np.max(np.abs(Pa - Q)) > 1e-3  # the projection changes random data
    # one whole step: hybrid lift with and without the fused Chebyshev step inside the tentative-velocity solves
    S = lambda z: np.sin(2 * np.pi * z)
    C = lambda z: np.cos(2 * np.pi * z)
    xq, xp = ea.node_coordinates()
    Q0 = np.stack([S(xq[:, 0]) * C(xq[:, 1]), -C(xq[:, 0]) * S(xq[:, 1])], axis=-1)
    out = []
    for e in (ea, eb):
        e.set_state(Q0, 0.25 * (C(2 * xp[:, 0]) + C(2 * xp[:, 1])))
        e.reconstruct_trace()
        for sl in range(4):
            e.set_forcing_scale(sl, 0.0)
        e.step()
        out.append(e.get_field(0)[0].copy())
    assert np.max(np.abs(out[0] - out[1])) < 2e-8 * np.max(np.abs(out[1]))
