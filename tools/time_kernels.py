"""Micro-timings of the hot kernels through hdg_time_kernel (HIP events on the engine's stream)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
from incompressibleeulerhdg_amd.model_problems import TaylorGreen
from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332
k, nx = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 1024)
ids = [int(a) for a in sys.argv[3:]] or [7, 6, 0, 9, 2, 5, 3, 1, 8]
ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, 0.25 / nx)
e = ts._engine
mp = TaylorGreen(ts._V_Q, ts._V_p)
e.set_state(ts._V_Q.interpolate(mp.Q_stationary), ts._V_p.interpolate(mp.p_stationary))
NQ = e.n_cells * 2 * e.n_u
names = {0: "adv", 1: "trace_apply", 2: "bdm", 3: "backsub", 4: "additive+cheb", 5: "liftT", 6: "hybrid lift+cheb", 7: "adv residual", 8: "triad", 9: "hybrid lift"}
nvec = {0: 3, 2: 2, 5: 2, 6: 4, 7: 4, 8: 3, 9: 2}
for kid in ids:
    ms = min(e.time_kernel(kid, 20) for _ in range(3))
    extra = f"  {8.0 * nvec[kid] * NQ / ms / 1e6:8.1f} GB/s algorithmic" if kid in nvec else ""
    print(f"k={k} nx={nx} id {kid:2d} {names[kid]:18s} {ms * 1e3:8.1f} us{extra}", flush=True)
