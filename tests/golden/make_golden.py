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
