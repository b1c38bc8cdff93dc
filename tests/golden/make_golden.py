"""Generate the golden vectors under tests/golden/ with the CPU oracle (oracle/hdg_oracle.py).

The reference ships no fixtures and cannot be run here, so these vectors are
outputs of the build's own oracle: PARITY UNPINNED with respect to Firedrake.  They pin the oracle
against accidental change and give the GPU tests committed data to compare with.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import hdg_oracle as orc  # noqa: E402

CASES = [  # (k, nx, tableau, R, nsteps)
    (1, 4, "imex_ssp2_332", 2, 2),
    (1, 8, "imex_ssp2_332", 2, 2),
    (2, 4, "imex_ssp2_332", 2, 2),
    (1, 4, "imex_ars2_232", 1, 2),
    (1, 4, "imex_ars3_443", 2, 2),
    (1, 4, "imex_ssp3_433", 2, 2),
    (1, 4, "imex_implicit", 2, 3),
]


def periodic_case(rho=np.pi / 15, delta=0.05):
    """initial data and forcing of the periodic golden cases (shared with the tests that replay them)"""
    Q0 = lambda x, y: (np.where(y <= np.pi, np.tanh((y - np.pi / 2) / rho), np.tanh((1.5 * np.pi - y) / rho)), delta * np.sin(x))
    p0 = lambda x, y: delta * np.cos(x) * np.sin(y - np.pi) * 0.3
    q0 = lambda x, y: np.sin(x) * np.sin(y)
    f = lambda t: (lambda x, y: (0.1 * np.cos(y) * np.cos(t), 0.2 * np.sin(x + y)))
    return Q0, p0, q0, f


def main():
    for k, nx, tab, R, nsteps in CASES:
        d = orc.HDGDiscretisation(nx, k)
        tg = orc.TaylorGreen(d)
        dt = 0.25 / nx
        o = orc.OracleHDGIMEX(d, dt, tab, n_richardson=R)
        Q, p = o.solve(*tg.initial_condition(), tg.f_rhs, nsteps * dt)
        Qe, pe = tg.solution(nsteps * dt)
        name = f"imex_{tab}_k{k}_nx{nx}_R{R}_n{nsteps}.npz"
        np.savez_compressed(
            os.path.join(HERE, name), Q=Q, p=p, lam=o.lam, Qstar_last=o.Qstar[-1], stage_Q_last=o.stage_Q[-1],
            err_Q=d.l2_norm_velocity(Q - Qe), err_p=d.l2_norm_pressure(p - pe), dt=dt, nsteps=nsteps)
        print("wrote", name)
    # implicit projection (config C1 at reduced size)
    for k, nx in ((1, 8), (1, 16)):
        d = orc.HDGDiscretisation(nx, k)
        tg = orc.TaylorGreen(d)
        dt = 0.05
        Q, p = orc.OracleHDGImplicit(d, dt).solve(*tg.initial_condition(), tg.f_rhs, 4 * dt)
        Qe, pe = tg.solution(4 * dt)
        name = f"implicit_proj_k{k}_nx{nx}_n4.npz"
        np.savez_compressed(os.path.join(HERE, name), Q=Q, p=p, err_Q=d.l2_norm_velocity(Q - Qe),
                            err_p=d.l2_norm_pressure(p - pe), dt=dt, nsteps=4)
        print("wrote", name)
    # passive tracer carried along, unit square, Taylor-Green velocity
    from oracle.tracer_oracle import TracerOracle, imex_with_tracer

    q0 = lambda x, y: np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y)  # driver.py:342
    for k, nx, tab, nsteps in ((1, 4, "imex_ssp2_332", 2), (2, 4, "imex_ars3_443", 2)):
        d = orc.HDGDiscretisation(nx, k)
        tg = orc.TaylorGreen(d)
        dt = 0.25 / nx
        o = orc.OracleHDGIMEX(d, dt, tab)
        Q, p, q = imex_with_tracer(o, TracerOracle(d), *tg.initial_condition(), d.interpolate_pressure(q0), tg.f_rhs, nsteps * dt)
        name = f"tracer_{tab}_k{k}_nx{nx}_n{nsteps}.npz"
        np.savez_compressed(os.path.join(HERE, name), Q=Q, p=p, q=q, dt=dt, nsteps=nsteps)
        print("wrote", name)
    # doubly periodic square, shear-layer data with a non-gradient forcing, tracer (section 8(f) row 2 first step)
    L = 2 * np.pi
    for k, nx, tab, nsteps in ((1, 4, "imex_ssp2_332", 2), (2, 4, "imex_ssp2_332", 2)):
        d = orc.HDGDiscretisation(nx, k, periodic=True, L=L)
        dt = 0.25 * d.mesh.h
        Q0, p0, qq0, f = periodic_case()
        o = orc.OracleHDGIMEX(d, dt, tab)
        Q, p, q = imex_with_tracer(o, TracerOracle(d), d.interpolate_velocity(Q0), d.interpolate_pressure(p0), d.interpolate_pressure(qq0),
                                   lambda t: d.interpolate_velocity(f(t)), nsteps * dt)
        w, xy = TracerOracle(d).vorticity(Q)
        order = np.lexsort((np.round(xy[:, 1] * 1e6), np.round(xy[:, 0] * 1e6)))
        name = f"periodic_{tab}_k{k}_nx{nx}_n{nsteps}.npz"
        np.savez_compressed(os.path.join(HERE, name), Q=Q, p=p, q=q, lam=o.lam, vorticity_sorted=w[order], dt=dt, nsteps=nsteps)
        print("wrote", name)
    # operator-level vectors on seeded random data
    rng = np.random.default_rng(123456789)
    k, nx = 2, 3
    d = orc.HDGDiscretisation(nx, k)
    Q = rng.standard_normal((d.mesh.ncells * d.nu, 2))
    p = rng.standard_normal(d.NP)
    Qs = d.project_bdm(Q)
    F = d.assemble_f_impl(Qs, "upwind")
    np.savez_compressed(os.path.join(HERE, "operators_k2_nx3.npz"), Q=Q, p=p, Qstar=Qs, FQ=F @ Q.ravel(),
                        wdiv=d.Wdiv @ Q.ravel(), lam=d.reconstruct_trace(Q, p))
    print("wrote operators_k2_nx3.npz")


if __name__ == "__main__":
    main()
