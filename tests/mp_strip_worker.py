"""Worker of tests/test_gpu_multirank.py: one rank of a strip-partitioned HDG-IMEX run.

usage: mp_strip_worker.py RANK NRANKS TOKEN K NX NSTEPS OUTFILE [unsplit] [badtimer] [opt:NAME=INT ...]
(opt: engine options of the timestepper classes, e.g. opt:tent_precond=1 opt:tent_solver=0 opt:trace_precond=0)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, nranks, token, k, nx, nsteps, out = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]),
                                               int(sys.argv[5]), int(sys.argv[6]), sys.argv[7])
    unsplit = "unsplit" in sys.argv[8:]
    badtimer = "badtimer" in sys.argv[8:]
    opts = {a[4:].split("=")[0]: int(a.split("=")[1]) for a in sys.argv[8:] if a.startswith("opt:")}
    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, use_projection_method=not unsplit,
                                            n_richardson=2, rank=rank, nranks=nranks, comm_backend="shm", comm_token=token, **opts)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    if badtimer:
        # a failed micro-benchmark call (unknown kernel id) must leave the halo exchanges switched ON: the step
        # below then still matches the single-rank run (hdg_time_kernel is not collective: no peer is involved)
        try:
            ts._engine.time_kernel(99, 1)
            raise SystemExit("time_kernel(99) did not fail")
        except _lib.HDGError:
            pass
        assert ts._engine.time_kernel(8, 2) > 0
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), nsteps * dt, fused=True)
    lam = ts._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
    Qe, pe = mp.solution(nsteps * dt, ts._engine.integrate_pressure)
    eq, ep = ts._engine.l2_norms(Q.dat.data - Qe.dat.data, p.dat.data - pe.dat.data)
    sums, cnt = ts._engine.iteration_stats()
    np.savez(out, Q=Q.dat.data, p=p.dat.data, lam=lam, eq=eq, ep=ep, its=sums / np.maximum(cnt, 1))


if __name__ == "__main__":
    main()
