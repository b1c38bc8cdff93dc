#!/usr/bin/env python
"""Headline benchmark: million DOF-updates/s of the HDG-IMEX timestep (BASELINE.json).

A "step" is one HDG-IMEX SSP2(3,3,2) timestep (2 Richardson iterations, projection method, upwind
flux, Taylor-Green vortex with exponential forcing, dt = 0.25/nx) on the
configuration named in ``config.workload`` -- by default C3: k = 2, 1024 x 1024 triangular mesh.
All state is resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W [--nx 1024 --degree 2]

Prints ONE JSON line (rank 0) with the driver's contract plus ``roofline`` (dominant kernel, HIP
events on the engine's stream) and ``cpu_baseline`` (the numpy/scipy oracle timed on a bounded
sample of the same workload on this box's host cores, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def ssp2_scales(nsteps, dt, kappa, t0=0.0):
    """Forcing scalars g(t_n + c_i dt), i = 0..s-1, and g(t_n + dt) for the separable TG forcing."""
    c = [0.0, 1.0, 0.5]  # hdg_imex.py:949, as written
    g = lambda t: -kappa * np.exp(-kappa * t)
    out = np.zeros((nsteps, 4))
    for n in range(nsteps):
        tn = t0 + n * dt
        out[n, :3] = [g(tn + ci * dt) for ci in c]
        out[n, 3] = g(tn + dt)
    return out


def cpu_baseline(degree, budget_s=20.0):
    """Time the CPU oracle (scipy sparse direct solves: the converged limit of the reference's
    assembled-AIJ + LU/ILU solver stack) on the largest mesh whose single step fits the budget."""
    from oracle import hdg_oracle as orc

    best = None
    for nx in (8, 16, 32, 64):
        d = orc.HDGDiscretisation(nx, degree)
        tg = orc.TaylorGreen(d)
        dt = 0.25 / nx
        o = orc.OracleHDGIMEX(d, dt, "imex_ssp2_332")
        o.set_initial_condition(*tg.initial_condition())
        o.step(tg.f_rhs, 0.0)  # warm-up step (mirrors --warmup, driver.py:157-162)
        t0 = time.perf_counter()
        nsteps = 2
        for n in range(nsteps):
            o.step(tg.f_rhs, (n + 1) * dt)
        el = time.perf_counter() - t0
        best = dict(value=d.N * nsteps / el / 1e6, unit="million DOF-updates/s", cores=1, kind="port",
                    sample=f"oracle (numpy/scipy sparse LU) HDG-IMEX SSP2(3,3,2) k={degree} nx={nx}, "
                           f"1 warm-up + {nsteps} timed steps, {el:.1f} s")
        if el * 5 > budget_s:  # the next mesh is 4x the unknowns and >4x the factorisation time
            break
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tent-precond", type=int, default=2)
    ap.add_argument("--trace-precond", type=int, default=1)
    ap.add_argument("--gmres-restart", type=int, default=8)
    ap.add_argument("--tent-solver", type=int, default=None, help="0 GMRES, 1 GMRES cycle + Chebyshev (default: by degree)")
    ap.add_argument("--comm", choices=["rccl", "shm"], default="rccl", help="inter-rank transport for --gpus > 1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    shared_gpu = False
    if world > 1:
        import torch
        import torch.distributed as dist_

        ndev = torch.cuda.device_count()
        if ndev >= world:
            torch.cuda.set_device(local_rank)
            dist_.init_process_group("nccl")
        else:
            # rehearsal on a box with fewer GPUs than ranks: ranks share devices, rendezvous over gloo,
            # data over the shared-memory transport (RCCL refuses duplicate devices)
            shared_gpu = True
            local_rank = local_rank % max(ndev, 1)
            torch.cuda.set_device(local_rank)
            dist_.init_process_group("gloo")
            args.comm = "shm"
        dist = dist_

    def sync_barrier():
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    def reduce_scalar(x, op):
        if dist is None:
            return x
        import torch

        t = torch.tensor([x], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
        dist.all_reduce(t, op=op)
        return float(t.item())

    from incompressibleeulerhdg_amd._lib import Engine
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    nx, k = args.nx, args.degree
    dt = 0.25 / nx
    kappa = 0.5
    # Multi-GPU: non-overlapping strip partition of the mesh rows, one process per GPU; halo rows,
    # Krylov scalars and the coarse-grid residual travel inside the library (RCCL over xGMI; the
    # shared-memory transport is the fallback when RCCL cannot initialise).  Strong scaling: the
    # global problem is fixed.
    from incompressibleeulerhdg_amd.distributed import comm_kwargs, make_comm_token

    def bcast(obj):
        lst = [obj]
        dist.broadcast_object_list(lst, src=0)
        return lst[0]

    def build(backend):
        token = make_comm_token(backend, rank, bcast) if world > 1 else None
        return IncompressibleEulerHDGIMEXSSP2_332(
            UnitSquareMesh(nx, nx), k, dt, use_projection_method=True, n_richardson=2, device=local_rank,
            tent_precond=args.tent_precond, trace_precond=args.trace_precond, gmres_restart=args.gmres_restart,
            **({} if args.tent_solver is None else {"tent_solver": args.tent_solver}), **comm_kwargs(backend, rank, world, token))

    backend = args.comm if world > 1 else "none"
    if world > 1 and backend == "rccl":
        # rehearse the transport in a child process first: a failure or hang costs a timeout, not the run
        from incompressibleeulerhdg_amd.distributed import probe_transport

        ok_probe = probe_transport("rccl")
        if int(reduce_scalar(1.0 if ok_probe else 0.0, dist.ReduceOp.MIN)) == 0:
            if rank == 0:
                print("[bench] RCCL transport probe failed; using the shared-memory transport", file=sys.stderr)
            backend = "shm"
    try:
        ts = build(backend)
        ok = 1
    except Exception as exc:  # noqa: BLE001
        ok, err = 0, repr(exc)
        print(f"[rank {rank}] transport '{backend}' failed: {err}", file=sys.stderr)
    if world > 1:
        if int(reduce_scalar(float(ok), dist.ReduceOp.MIN)) == 0:
            if backend == "shm":
                raise SystemExit("no working transport")
            backend = "shm"
            ts = build(backend)
    eng = ts._engine
    mp = TaylorGreen(ts._V_Q, ts._V_p, "exponential", kappa)
    eng.set_state(ts._V_Q.interpolate(mp.Q_stationary), ts._V_p.interpolate(mp.p_stationary))
    eng.reconstruct_trace()
    eng.set_forcing_profile(mp.f_rhs().profile)
    if args.warmup > 0:
        eng.run_separable(ssp2_scales(args.warmup, dt, kappa))
    eng.iteration_stats(reset=True)
    sync_barrier()
    t0 = time.perf_counter()
    eng.run_separable(ssp2_scales(args.steps, dt, kappa, t0=args.warmup * dt))  # synchronous on return
    sync_barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = reduce_scalar(elapsed, dist.ReduceOp.MAX)

    if rank == 0:
        sums, cnt = eng.iteration_stats()
        its = {n: (float(s / c) if c else 0.0) for n, s, c in zip(
            ("tentative", "pressure", "final_pressure", "pressure_reconstruction"), sums, cnt)}
        ntot = eng.n_total
        value = ntot * args.steps / elapsed / 1e6
        # --- roofline of the dominant kernel.  A Chebyshev iteration of the tentative-velocity solve is two
        # launches that share ~55 % of the step (profiles/r01_i_kernel_stats_c3.csv: 28.6 % + 26.6 %):
        #   k_adv_apply (residual form): t = b - (I - gamma F(Q*)) x; reads x, Q*, b and writes t: 4 vectors;
        #   k_edge_lift<K,false,2> + Chebyshev step: z = t + sum_e G_e d_e(t) (BDM lift with the element
        #     block-Jacobi folded into the lifting tables), x_{n+1} = x_n + c1 (x_n - x_{n-1}) + c2 z written over
        #     x_{n-1}: reads t, x_n, x_{n-1}, writes x_{n+1}: 4 vectors (the additive preconditioner,
        #     --tent-precond 1, reads one more).
        # 8 B per entry.  Whichever takes longer per launch is reported as the dominant
        # kernel, the other under other_kernels.  Durations from HIP events on the engine's stream.
        NQ = eng.n_cells * 2 * eng.n_u  # this rank's strip: kernel timings below are per-rank launches
        NL = eng.n_edges * eng.n_l
        NP = eng.n_cells * eng.n_p
        hybrid = args.tent_precond == 2
        ms_lift = eng.time_kernel(6 if hybrid else 4, 20)
        ms_adv = eng.time_kernel(7, 20)
        ms_liftT = eng.time_kernel(5, 20)
        ms_tr = eng.time_kernel(1, 50)
        ms_bs = eng.time_kernel(3, 20)
        ms_triad = eng.time_kernel(8, 20)  # y = a x + b y on velocity vectors: the HBM rate this box actually delivers
        lift_bytes = 8.0 * (4 if hybrid else 5) * NQ
        adv_bytes = 8.0 * 4 * NQ
        lift_name = ("k_edge_lift<K,false,2> (BDM lift + block-Jacobi of the remainder + Chebyshev step)" if hybrid
                     else "k_edge_lift<K,false,1> (BDM lift + block-Jacobi + Chebyshev step)")
        adv_name = "k_adv_apply<K> (advection operator, residual form b - (I - gamma F) x)"
        pmc = {}
        try:  # HBM bytes per launch from the committed PMC passes (same workload, single rank only)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if tj["workload"] == {"nx": nx, "degree": k} and world == 1 and hybrid:
                pmc = {"lift": tj["kernels"]["k_edge_lift<K,false,2>+cheb"]["hbm_bytes"],
                       "adv": tj["kernels"]["k_adv_apply(+residual)"]["hbm_bytes"]}
        except Exception:
            pmc = {}
        gbs = lambda b, ms: b / (ms * 1e-3) / 1e9
        cand = {"lift": (lift_name, lift_bytes, ms_lift), "adv": (adv_name, adv_bytes, ms_adv)}
        dom = "adv" if ms_adv >= ms_lift else "lift"
        oth = "lift" if dom == "adv" else "adv"
        dname, dbytes, dms = cand[dom]
        roof = dict(bound="hbm", kernel=dname, achieved=gbs(dbytes, dms), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=gbs(dbytes, dms) / HBM_PEAK_GBS, traffic=pmc.get(dom), algorithmic_bytes=dbytes,
                    ms_per_launch=dms, stream_triad_GBs=gbs(8.0 * 3 * NQ, ms_triad),
                    frac_of_triad=gbs(dbytes, dms) / gbs(8.0 * 3 * NQ, ms_triad),
                    other_kernels={
                        cand[oth][0]: dict(ms=cand[oth][2], GBs=gbs(cand[oth][1], cand[oth][2]),
                                           algorithmic_bytes=cand[oth][1], traffic=pmc.get(oth)),
                        "k_edge_lift<K,true,0>": dict(ms=ms_liftT, GBs=gbs(8.0 * 2 * NQ, ms_liftT)),
                        "k_trace_apply": dict(ms=ms_tr, GBs=gbs(8.0 * 2 * NL, ms_tr)),
                        "k_backsub": dict(ms=ms_bs, GBs=gbs(8.0 * (NL + 2 * NQ + 2 * NP), ms_bs)),
                    })
        line = {
            "metric": "million DOF-updates/sec (HDG-IMEX k=2, 1024^2 tri mesh)" if (nx, k) == (1024, 2)
            else f"million DOF-updates/sec (HDG-IMEX k={k}, {nx}^2 tri mesh)",
            "value": value, "unit": "million DOF-updates/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"HDG-IMEX SSP2(3,3,2) R=2 projection upwind, k={k}, {nx}x{nx} tri mesh, "
                                   f"Taylor-Green kappa=0.5, dt=0.25/nx (BASELINE C3)" if (nx, k) == (1024, 2)
                       else f"HDG-IMEX SSP2(3,3,2) R=2 projection upwind, k={k}, {nx}x{nx} tri mesh",
                       "n_dof": ntot, "krylov_iterations_avg": its,
                       "parallelism": f"strip partition over {world} rank(s), transport {backend}"
                                      + (" (ranks share GPUs: rehearsal)" if shared_gpu else "")},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(k)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
