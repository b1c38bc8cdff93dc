"""Bitwise run-to-run comparison of single operators and solver pieces (same process, same engine)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from incompressibleeulerhdg_amd import _lib
from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
from incompressibleeulerhdg_amd.model_problems import TaylorGreen
from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332
k, nx = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 512)
dt = 0.25 / nx
ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt)
e = ts._engine
rng = np.random.default_rng(1)
x = rng.standard_normal(e.shape_Q); y = rng.standard_normal(e.shape_Q); lam = rng.standard_normal(e.shape_l)
def chk(name, f, n=3):
    ref = f()
    bad = 0.0
    for _ in range(n - 1):
        r = f()
        for a, b in zip(ref if isinstance(ref, tuple) else (ref,), r if isinstance(r, tuple) else (r,)):
            bad = max(bad, float(np.max(np.abs(np.asarray(a, dtype=float) - np.asarray(b, dtype=float)))))
    print(f"{name:40s} max run-to-run diff {bad:.3e}", flush=True)
chk("project_bdm_nodal", lambda: e.project_bdm_nodal(x))
Px = e.project_bdm_nodal(x)
chk("apply_advection", lambda: e.apply_advection(Px, y, 0.25 * dt))
chk("apply_trace_operator", lambda: e.apply_trace_operator(lam))
chk("apply_weak_divergence", lambda: e.apply_weak_divergence(x))
chk("l2_norms", lambda: e.l2_norms(x, rng.standard_normal(e.shape_p) * 0 + 1.0))
mp = TaylorGreen(ts._V_Q, ts._V_p)
Q0 = ts._V_Q.interpolate(mp.Q_stationary); p0 = ts._V_p.interpolate(mp.p_stationary)
def pieces(upto):
    e.set_state(Q0, p0); e.reconstruct_trace()
    e.set_forcing_profile(mp.f_rhs().profile)
    for sl in range(4): e.set_forcing_scale(sl, -0.5)
    for i in range(1, 3):
        e.set_field(i, np.zeros(e.shape_Q), np.zeros(e.shape_p), np.zeros(e.shape_l))
    e.begin_step(); e.project_bdm(0, 0)
    if upto == "bdm": return e.get_field(200, p=False, lam=False)[0]
    it = e.tentative_solve(1)
    if upto == "tent": return e.get_field(101, p=False, lam=False)[0], np.array([it])
    it2 = e.pressure_solve(1)
    if upto == "press": return e.get_field(_lib.HDG_STATE_UPDATE) + (np.array([it2]),)
for u in ("bdm", "tent", "press"):
    chk("step pieces up to " + u, lambda: pieces(u))
