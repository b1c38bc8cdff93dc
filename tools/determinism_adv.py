import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332
for k in (1, 2):
    for nx in (64, 128, 200, 256, 512):
        dt = 0.25 / nx
        ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt)
        e = ts._engine
        rng = np.random.default_rng(1)
        x = rng.standard_normal(e.shape_Q); y = rng.standard_normal(e.shape_Q)
        Px = e.project_bdm_nodal(x)
        r0 = e.apply_advection(Px, y, 0.25 * dt)
        worst = 0.0; nbad = 0
        for _ in range(4):
            r = e.apply_advection(Px, y, 0.25 * dt)
            d = np.abs(r - r0)
            worst = max(worst, d.max()); nbad = max(nbad, int((d > 0).sum()))
        bad = np.argwhere(np.abs(r - r0) > 0)
        cells = np.unique(bad[:, 0] // e.n_u) if len(bad) else []
        print(k, nx, "worst", worst, "entries differing", nbad, "cells", list(cells[:12]), flush=True)
