"""C4's problem on one GPU (hdg_implicit, k = 3, 512^2, projection): ms/step and iterations for tent_solver 0 / 1."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
from incompressibleeulerhdg_amd.model_problems import TaylorGreen
from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGImplicit

k, nx = int(sys.argv[1]), int(sys.argv[2])
dt = 0.25 / nx
for solver in (0, 1):
    ts = IncompressibleEulerHDGImplicit(UnitSquareMesh(nx, nx), k, dt, use_projection_method=True, tent_solver=solver)
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    Q0, p0 = mp.initial_condition()
    ts.solve(Q0, p0, None, mp.f_rhs(), 2 * dt)  # warm-up
    ts._engine.iteration_stats(reset=True)
    ts._engine.timers(reset=True)
    ts.solve(Q0, p0, None, mp.f_rhs(), 6 * dt)
    t = ts._engine.timers()
    sums, cnt = ts._engine.iteration_stats()
    print(f"k={k} nx={nx} tent_solver={solver}: {1e3 * t['timestep'][1] / t['timestep'][0]:.2f} ms/step, tentative {1e3 * t['tentative_velocity_solve'][1] / max(t['tentative_velocity_solve'][0], 1):.2f} ms "
          f"({sums[0] / max(cnt[0], 1):.1f} its), pressure {1e3 * t['pressure_solve'][1] / max(t['pressure_solve'][0], 1):.2f} ms", flush=True)
    del ts
