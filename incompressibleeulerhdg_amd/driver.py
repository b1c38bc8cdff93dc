"""Command-line driver with the flags and printed quantities of the reference's src/driver.py:23-385.

    python -m incompressibleeulerhdg_amd.driver --nx 64 --degree 2 --use_projection_method

``--problem taylorgreen`` (unit square), ``--problem shear`` (doubly periodic square, driver.py:182-183) and
``--problem kelvinhelmholtz`` (UnitDiskMesh(refinement), driver.py:184-185: the general-mesh path, projection and monolithic)
are built; the ``conforming`` / ``dg`` discretisations raise.  ``--animation`` (evolution.pvd with
the CG vorticity, callbacks.py:30-85) and ``--tracer_advection`` (driver.py:340-344) work on every mesh.  The final fields are written to ``solution.pvd``
(``--output``) like the reference does (driver.py:356-385).
"""
import argparse
import sys
import time

import numpy as np

from .auxilliary.callbacks import AnimationCallback
from .auxilliary.logging import log_summary
from .mesh import Function, PeriodicSquareMesh, UnitDiskMesh, UnitSquareMesh
from .model_problems import DoubleLayerShearFlow, KelvinHelmholtz, TaylorGreen
from .output import VTKFile
from .timesteppers import (
    IncompressibleEulerHDGIMEXARS2_232,
    IncompressibleEulerHDGIMEXARS3_443,
    IncompressibleEulerHDGIMEXImplicit,
    IncompressibleEulerHDGIMEXSSP2_332,
    IncompressibleEulerHDGIMEXSSP3_433,
    IncompressibleEulerHDGImplicit,
)

TIMESTEPPERS = {
    "imex_implicit": IncompressibleEulerHDGIMEXImplicit,
    "imex_ars2_232": IncompressibleEulerHDGIMEXARS2_232,
    "imex_ars3_443": IncompressibleEulerHDGIMEXARS3_443,
    "imex_ssp2_332": IncompressibleEulerHDGIMEXSSP2_332,
    "imex_ssp3_433": IncompressibleEulerHDGIMEXSSP3_433,
}


def build_parser():
    """Same flags and defaults as driver.py:26-176."""
    parser = argparse.ArgumentParser("Mesh specifications and polynomial degree")
    parser.add_argument("--problem", choices=["taylorgreen", "kelvinhelmholtz", "shear"], type=str, default="taylorgreen", help="model problem to solve")
    parser.add_argument("--nx", metavar="nx", type=int, default=8, help="number of grid cells in x-direction")
    parser.add_argument("--refinement", metavar="refinement", type=int, default=2, help="refinement level for unit disk mesh")
    parser.add_argument("--degree", metavar="degree", type=int, default=1, help="polynomial degree")
    parser.add_argument("--tfinal", metavar="tfinal", type=float, default=1.0, help="final time")
    parser.add_argument("--kappa", type=float, default=0.5, help="exponential decay factor")
    parser.add_argument("--dt", type=float, default=0.04, help="timestep size")
    parser.add_argument("--discretisation", choices=["conforming", "dg", "hdg"], type=str, default="hdg", help="discretisation method")
    parser.add_argument("--use_projection_method", action="store_true", default=False, help="use projection method for timestepping")
    parser.add_argument("--richardson", metavar="richardson", type=int, default=2, help="number of Richardson iterations")
    parser.add_argument("--flux", choices=["upwind", "centered"], type=str, default="upwind", help="numerical flux")
    parser.add_argument("--timestepper", choices=["implicit"] + list(TIMESTEPPERS), type=str, default="imex_ssp2_332", help="timestepper")
    parser.add_argument("--forcing", choices=["exponential", "constant"], type=str, default="exponential", help="forcing")
    parser.add_argument("--test_pressure_solver", action="store_true", default=False, help="carry out a single solve with the pressure solver for testing")
    parser.add_argument("--warmup", action="store_true", default=False, help="only perform one timestep")
    parser.add_argument("--animation", action="store_true", default=False,
                        help="save velocity and pressure fields at the end of each timestep as an animation")
    parser.add_argument("--tracer_advection", action="store_true", default=False, help="advect tracer field")
    # additions of the build
    parser.add_argument("--fused", action="store_true", default=False, help="run each timestep as one device-resident call")
    parser.add_argument("--output", type=str, default="solution.pvd",
                        help="VTK collection written at the end like the reference's solution.pvd ('' = no output)")
    parser.add_argument("--device", type=int, default=0, help="HIP device ordinal")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.discretisation != "hdg":
        raise RuntimeError(f"discretisation '{args.discretisation}' is out of scope of the MI355X hot path")
    callbacks = [AnimationCallback("evolution.pvd")] if args.animation else None  # driver.py:187
    if args.problem == "shear":
        mesh = PeriodicSquareMesh(args.nx, args.nx, L=2 * np.pi, quadrilateral=False)  # driver.py:182-183
    elif args.problem == "kelvinhelmholtz":
        mesh = UnitDiskMesh(refinement_level=args.refinement)  # driver.py:184-185
    else:
        mesh = UnitSquareMesh(args.nx, args.nx, quadrilateral=False)  # driver.py:181
    if args.timestepper == "implicit":
        timestepper = IncompressibleEulerHDGImplicit(  # driver.py:220-228 (passes n_richardson: SURVEY C-1)
            mesh, args.degree, args.dt, flux=args.flux, use_projection_method=args.use_projection_method,
            n_richardson=args.richardson, callbacks=callbacks, device=args.device)
    elif args.timestepper in TIMESTEPPERS:
        timestepper = TIMESTEPPERS[args.timestepper](
            mesh, args.degree, args.dt, flux=args.flux, use_projection_method=args.use_projection_method,
            n_richardson=args.richardson, callbacks=callbacks, device=args.device)
    else:
        raise RuntimeError(f"Invalid timestepping method for HDG discretisation: '{args.timestepper}'")

    print("+-------------------------------------------------+")
    print("! timesteppers for incompressible Euler equations !")
    print("+-------------------------------------------------+")
    print()
    print(f"model problem = {args.problem}")
    if args.problem == "kelvinhelmholtz":
        print(f"refinement level = {args.refinement}")
    else:
        print(f"mesh size = {args.nx} x {args.nx}")
    print(f"forcing = {args.forcing}")
    print(f"kappa = {args.kappa}")
    print(f"polynomial degree = {args.degree}")
    print(f"final time = {args.tfinal}")
    print(f"timestep size = {args.dt}")
    print(f"discretisation = {args.discretisation}")
    print(f"numerical flux = {args.flux}")
    print(f"number of Richardson iterations = {args.richardson}")
    print(f"use projection method = {args.use_projection_method}")
    print(f"advect tracer = {args.tracer_advection}")
    print(f"timestepping method = {timestepper.label}")
    print()

    eng = timestepper._engine
    if args.test_pressure_solver:
        # working equivalent of driver.py:308-324 (the reference's call is stale, SURVEY C-4): random
        # velocity-row right-hand side with seed 123456789, untimed first solve, timed second solve
        if args.timestepper == "implicit":
            raise RuntimeError("--test_pressure_solver needs an IMEX timestepper")
        rng = np.random.default_rng(123456789)
        f_Q = rng.standard_normal(eng.shape_Q)
        print("=== Testing pressure solver")
        print()
        eng.set_field(0, Q=f_Q, p=np.zeros(eng.shape_p), lam=np.zeros(eng.shape_l))
        for i in range(eng.nstages + 1):
            eng.set_forcing_scale(i, 0.0)
        eng.begin_step()
        for i in range(1, eng.nstages):  # stage iterates := the same random field, so r^{n+1} = (f_Q, w)
            eng.set_field(i, Q=f_Q)
        _ = timestepper.pressure_solve("final_stage")
        eng.set_field(0, lam=np.zeros(eng.shape_l))
        t_start = time.perf_counter()
        its = timestepper.pressure_solve("final_stage")
        t_finish = time.perf_counter()
        print(f"    solve time           = {t_finish-t_start:12.4f} s")
        print(f"    number of iterations = {its}")
        return 0

    if args.warmup:
        print("WARNING: performing a single timestep only!")
        print()
    if args.problem == "shear":
        model_problem = DoubleLayerShearFlow(timestepper._V_Q, timestepper._V_p)  # driver.py:334-335
    elif args.problem == "kelvinhelmholtz":
        model_problem = KelvinHelmholtz(timestepper._V_Q, timestepper._V_p)  # driver.py:336-337
    else:
        model_problem = TaylorGreen(timestepper._V_Q, timestepper._V_p, args.forcing, args.kappa)
    Q_0, p_0 = model_problem.initial_condition()
    # driver.py:340-344
    q_0 = (lambda x, y: np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y)) if args.tracer_advection else None
    kw = {"fused": True} if (args.fused and args.timestepper != "implicit") else {}
    Q, p = timestepper.solve(Q_0, p_0, q_0, model_problem.f_rhs(), args.tfinal, warmup=args.warmup, **kw)
    log_summary()
    if args.problem in ("shear", "kelvinhelmholtz"):
        # no exact solution (the reference's driver calls model_problem.solution, which these problems lack: it stops here
        # with an AttributeError); write the final fields
        if args.output:
            Q.rename("velocity")
            p.rename("pressure")
            divQ = Function(timestepper._V_p, eng.apply_weak_divergence(Q.dat.data, broken=True), "divergence")
            VTKFile(args.output).write(Q, p, divQ)
        return 0
    if not args.warmup:
        Q.rename("velocity")
        p.rename("pressure")
        Q_exact, p_exact = model_problem.solution(args.tfinal, eng.integrate_pressure)
        Q_error = Function(timestepper._V_Q, Q.dat.data - Q_exact.dat.data, "velocity_error")
        p_error = Function(timestepper._V_p, p.dat.data - p_exact.dat.data, "pressure_error")
        Q_error_nrm, p_error_nrm = eng.l2_norms(Q_error.dat.data, p_error.dat.data)  # driver.py:376-377
        print()
        print(f"velocity error = {Q_error_nrm}")
        print(f"pressure error = {p_error_nrm}")
        print()
        if args.output:
            # driver.py:356-385: L2 projection of the (broken) divergence onto the pressure space, then
            # velocity, pressure, divergence, exact fields and errors into solution.pvd
            divQ = Function(timestepper._V_p, eng.apply_weak_divergence(Q.dat.data, broken=True), "divergence")
            Q_exact.rename("velocity_exact")
            p_exact.rename("pressure_exact")
            VTKFile(args.output).write(Q, p, divQ, Q_exact, Q_error, p_exact, p_error)
    return 0


if __name__ == "__main__":
    sys.exit(main())
