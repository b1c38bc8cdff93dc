"""h-convergence of the HIP path on the manufactured Taylor-Green vortex ("errors decrease under h- and
dt-refinement"): HDG-IMEX SSP2(3,3,2), dt = 0.25/nx (so dt is refined with h), T = 0.125, exponential forcing, kappa = 0.5.
Prints the L2 errors of velocity and pressure the reference driver prints (driver.py:371-380) and the observed orders."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
from incompressibleeulerhdg_amd.model_problems import TaylorGreen
from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

T = 0.125
for k in (1, 2, 3, 4):
    prev = None
    for nx in ((16, 32, 64, 128, 256) if k <= 2 else (8, 16, 32, 64)):
        dt = 0.25 / nx
        ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt)
        mp = TaylorGreen(ts._V_Q, ts._V_p, "exponential", 0.5)
        Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), T, fused=True)
        Qe, pe = mp.solution(T, ts._engine.integrate_pressure)
        eq, ep = ts._engine.l2_norms(Q.dat.data - Qe.dat.data, p.dat.data - pe.dat.data)
        orders = "" if prev is None else f"   orders {np.log2(prev[0] / eq):5.2f} {np.log2(prev[1] / ep):5.2f}"
        print(f"k={k} nx={nx:4d}  velocity error {eq:.3e}  pressure error {ep:.3e}{orders}", flush=True)
        prev = (eq, ep)
        del ts
