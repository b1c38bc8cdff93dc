"""h-convergence of the HIP path on the manufactured Taylor-Green vortex: the one check that is independent of BOTH
home-made checkers (numpy oracle, C++ twin).  SURVEY.md section 8(c): the analytic vortex pins results -- "errors decrease
under h- and dt-refinement"; for [P_{k+1}]^2 x P_k the expected spatial orders of the L2 errors the reference driver prints
(driver.py:371-380) are k+2 (velocity) and k+1 (pressure).  dt is refined with h and kept small enough that the
second-order time error of SSP2(3,3,2) stays below the spatial error on the meshes used (tools/convergence_table.py prints
the long table of DESIGN.md section 3, where the time error takes over on finer meshes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,cfl", [(1, 0.25), (2, 0.0625)])
def test_h_convergence_orders(hip_lib, k, cfl):
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    T = 0.0625
    errs = []
    for nx in (16, 32, 64):
        dt = cfl / nx
        ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt)
        mp = TaylorGreen(ts._V_Q, ts._V_p, "exponential", 0.5)
        Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), T, fused=True)
        Qe, pe = mp.solution(T, ts._engine.integrate_pressure)
        errs.append(ts._engine.l2_norms(Q.dat.data - Qe.dat.data, p.dat.data - pe.dat.data))
        del ts
    errs = np.array(errs)
    orders = np.log2(errs[:-1] / errs[1:])
    print(f"k={k}: errors {errs.tolist()} orders {orders.tolist()}")
    assert np.all(np.abs(orders[:, 0] - (k + 2)) < 0.15), orders  # velocity, P_{k+1}
    assert np.all(np.abs(orders[:, 1] - (k + 1)) < 0.15), orders  # pressure, P_k
