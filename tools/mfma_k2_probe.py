import os, sys, json
sys.path.insert(0, os.getcwd())
from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
from incompressibleeulerhdg_amd.model_problems import TaylorGreen
from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332
nx, k = 1024, 2
dt = 0.25 / nx
ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt)
mp = TaylorGreen(ts._V_Q, ts._V_p)
e = ts._engine
e.set_state(ts._V_Q.interpolate(mp.Q_stationary), ts._V_p.interpolate(mp.p_stationary))
e.reconstruct_trace()
e.set_forcing_profile(mp.f_rhs().profile)
for sl in range(4):
    e.set_forcing_scale(sl, -0.5)
e.step()
out = {}
for name, kid in (("adv plain", 0), ("adv residual", 7), ("bdm lift", 2), ("hybrid lift (no epilogue)", 9)):
    out[name] = round(e.time_kernel(kid, 20) * 1e3, 1)
print("HDG_MFMA_K2 =", os.environ.get("HDG_MFMA_K2"), out)
