"""Worker of tests/test_gpu_kernels.py::test_two_lane_advection_kernel_all_degrees: advection operator of the
engine vs the oracle's matrix for k = 1..4 and both fluxes.  The parent sets HDG_ADV_SPLIT=1:4 so that every
degree runs through the two-lanes-per-cell kernel (k_adv_apply2), which by default is used for k = 3 only."""
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from incompressibleeulerhdg_amd import _lib
    from oracle import hdg_oracle as orc

    worst = 0.0
    for k, nx in ((1, 6), (2, 5), (3, 4), (4, 3)):
        for flux in ("upwind", "centered"):
            d = orc.HDGDiscretisation(nx, k)
            tb = orc.TABLEAUX["imex_ssp2_332"]
            e = _lib.Engine(nx=nx, degree=k, dt=0.25 / nx, nstages=3, a_expl=tb["a_expl"], a_impl=tb["a_impl"],
                            b_expl=tb["b_expl"], b_impl=tb["b_impl"], c_expl=tb["c_expl"], flux=flux)
            rng = np.random.default_rng(2)
            Qstar = d.project_bdm(rng.standard_normal(e.shape_Q))
            x = rng.standard_normal(e.shape_Q)
            gamma = 0.3 / nx
            F = d.assemble_f_impl(Qstar, flux)
            ref = x.ravel() - gamma * spla.spsolve(d.MQ.tocsc(), F @ x.ravel())
            got = e.apply_advection(Qstar, x, gamma).ravel()
            err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
            print(f"k={k} nx={nx} {flux}: {err:.2e}")
            worst = max(worst, err)
    sys.exit(0 if worst < 5e-11 else 1)


if __name__ == "__main__":
    main()
