import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332
k, nx = 1, 512
dt = 0.25 / nx
ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt)
e = ts._engine
rng = np.random.default_rng(1)
x = rng.standard_normal(e.shape_Q); y = rng.standard_normal(e.shape_Q)
def rep(name, f, n=6):
    r0 = f(); worst = 0.0; nb = 0
    for _ in range(n):
        d = np.abs(f() - r0); worst = max(worst, d.max()); nb = max(nb, int((d > 0).sum()))
    print(name, "worst", worst, "entries", nb, flush=True)
def roundtrip():
    e.set_field(1, y, None, None)
    return e.get_field(1, p=False, lam=False)[0]
rep("roundtrip", roundtrip)
Px = e.project_bdm_nodal(x)
rep("bdm", lambda: e.project_bdm_nodal(y))
rep("adv gamma=0.25dt", lambda: e.apply_advection(Px, y, 0.25 * dt))
rep("adv gamma=0", lambda: e.apply_advection(Px, y, 0.0))
rep("adv Qstar=0", lambda: e.apply_advection(0 * Px, y, 0.25 * dt))
rep("weak_div", lambda: e.apply_weak_divergence(y))
