"""Worker of tests/test_gpu_periodic.py: a few HDG-IMEX steps of the double-layer shear flow on the periodic square in a process of
its own (environment switches that the engine reads once per process).  usage: periodic_worker.py K NX NSTEPS OUTFILE"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    k, nx, nsteps, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    from incompressibleeulerhdg_amd import _lib
    from incompressibleeulerhdg_amd.mesh import PeriodicSquareMesh
    from incompressibleeulerhdg_amd.model_problems import DoubleLayerShearFlow
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    L = 2 * np.pi
    dt = 0.25 * L / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(PeriodicSquareMesh(nx, nx, L=L), k, dt)
    mp = DoubleLayerShearFlow(ts._V_Q, ts._V_p)
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), nsteps * dt, fused=True)
    lam = ts._engine.get_field(_lib.HDG_STATE_CURRENT, Q=False, p=False)[2]
    sums, cnt = ts._engine.iteration_stats()
    np.savez(out, Q=Q.dat.data, p=p.dat.data, lam=lam, its=sums / np.maximum(cnt, 1))


if __name__ == "__main__":
    main()
