import sys, numpy as np
sys.path.insert(0, '.')
from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
from incompressibleeulerhdg_amd.model_problems import TaylorGreen
from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332
nx = int(sys.argv[1]); trp = int(sys.argv[2])
ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), 1, 0.25/nx, trace_precond=trp, trace_maxit=300)
mp = TaylorGreen(ts._V_Q, ts._V_p)
try:
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), float(sys.argv[3]) * 0.25/nx, fused=True)
    print(nx, trp, "ok", ts._engine.iteration_stats())
except Exception as e:
    print(nx, trp, "FAILED", e)
