"""Stand-alone launch timings of the cell-local Schur kernels (hdg_time_kernel ids 3, 10-13) at k, nx:
python tools/time_schur.py K NX   (HDG_NO_MFMA_SCHUR=1 for the per-thread kernels)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from incompressibleeulerhdg_amd._lib import Engine
k, nx = int(sys.argv[1]), int(sys.argv[2])
a = np.array([[0, 0, 0], [0.5, 0, 0], [0.5, 0.5, 0]]); ai = np.array([[0.25, 0, 0], [0, 0.25, 0], [1 / 3, 1 / 3, 1 / 3]])
e = Engine(nx=nx, degree=k, dt=0.25 / nx, nstages=3, a_expl=a, a_impl=ai, b_expl=[1 / 3] * 3, b_impl=[1 / 3] * 3, c_expl=[0, 1, 0.5])
rng = np.random.default_rng(0)
e.set_state(rng.standard_normal(e.shape_Q), rng.standard_normal(e.shape_p))
e.reconstruct_trace()
NQ, NP, NL = e.n_cells * 2 * e.n_u * 8.0, e.n_cells * e.n_p * 8.0, e.n_edges * e.n_l * 8.0
tag = "per-thread" if os.environ.get("HDG_NO_MFMA_SCHUR") else "matrix-core"
for name, kid, nbytes in (("backsub (rw, rp)", 3, NL + 2 * NQ + 2 * NP), ("condense (rp)", 10, NP + NL), ("condense (rw)", 14, NQ + NL), ("pgrad", 11, 3 * NQ + NP + NL),
                          ("weak_div", 12, NQ + NP), ("precon_rhs", 13, 2 * NQ + NP + NL)):
    ms = e.time_kernel(kid, 20)
    print(f"k={k} nx={nx} {tag:12s} {name:18s} {ms * 1e3:9.1f} us  {nbytes / ms / 1e9:8.2f} TB/s algorithmic", flush=True)
