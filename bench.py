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

``--gpus N`` with N > 1 outside ``torch.distributed.run`` launches the N ranks itself (a child
``python -m torch.distributed.run --nproc-per-node N bench.py ...`` started before anything touches
the GPU) and relays rank 0's line.  ``n_gpus`` is always the number of ranks that actually ran.  With
fewer visible devices than ranks the run is a labelled rehearsal (ranks share devices, shared-memory
transport) and is refused beyond 4 ranks per device; an RCCL failure is an error, never a silent
change of transport.
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


def csrc_sha16():
    """Hash of the kernel sources: PMC traffic figures are only valid for the code they were measured on."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "incompressibleeulerhdg_amd", "csrc")
    for f in sorted(os.listdir(d)):  # every kernel / engine source
        if f.endswith((".hpp", ".hip")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as fresh processes -- before any
    GPU call in this one -- relay rank 0's JSON line, and fail unless exactly N ranks reported."""
    import socket
    import subprocess

    import torch  # device_count() does not initialise the GPU

    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible")
    if args.gpus > ndev and args.gpus > 4 * ndev:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} device(s) visible: refusing "
                         f"(a shared-device rehearsal is limited to 4 ranks per device)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for ln in r.stdout.decode().splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if r.returncode != 0 or line is None:
        raise SystemExit(f"bench.py: the {args.gpus}-rank run failed (exit code {r.returncode})")
    if json.loads(line)["n_gpus"] != args.gpus:
        raise SystemExit(f"bench.py: asked for {args.gpus} ranks, {json.loads(line)['n_gpus']} reported")
    print(line)
    raise SystemExit(0)


FP64_MATRIX_PEAK_TF = 78.6  # MI355X FP64 matrix = FP64 vector peak (AMD datasheet; the guide's MFMA table has no FP64 row)


def roofline_block(eng, args, nx, k, world, ktimers=None, timed_where="the timed steps"):
    """Roofline of the dominant kernel.  Durations of the two kernels of a tentative-velocity iteration: HIP-event
    brackets around every launch, by FORM (`ktimers`, hdg_set_kernel_timing; `timing: "in place"`), for the other kernels a
    stand-alone launch loop on the engine's stream (hdg_time_kernel).

    An iteration of the tentative-velocity solve (55-60 % of the step) is two launches: the advection operator and
    the hybrid preconditioner (BDM lift with the element block-Jacobi folded into the lifting tables).  The forms of a
    kernel are separate instantiations with their own names (last template argument), so profiles tell them apart:
      Chebyshev phase:            k_adv_apply<K, true> / k_adv_mfma<K, true>: residual form t = b - (I - gamma F(Q*)) x
                                  (reads x, Q*, b, writes t: 4 vectors of 8 N_Q bytes);
                                  k <= 2: k_edge_lift_pair<K, false, 2, true> (nx <= 64: k_edge_lift<...>) with the fused Chebyshev
                                  step (reads t, x_n, x_{n-1}, writes x_{n+1}: 4 vectors);  k >= 3: k_edge_lift_mfma<K, true>;
      GMRES / s-step cycles:      k_adv_apply<K, false> / k_adv_mfma<K, false> (reads x, Q*, writes y: 3 vectors) and the plain
                                  lift (2 vectors).  Since round 4 (s-step tail, early hand-over) these are the MORE frequent forms.
    k <= 2: bound HBM.  k >= 3: the advection kernels are bound by FP64 MFMA; `achieved` counts the ALGORITHMIC flops (unpadded
    contraction shapes), `mfma_util` the issued v_mfma_f64_16x16x4 (padding included).  The DOMINANT kernel is the form with the
    largest total time over the bracketed step (launches x average duration); the other three are listed in `other_kernels`."""
    NQ = eng.n_cells * 2 * eng.n_u  # this rank's strip: kernel timings are per-rank launches
    NL = eng.n_edges * eng.n_l
    NP = eng.n_cells * eng.n_p
    NU = eng.n_u
    hybrid = args.tent_precond == 2
    forms = eng.kernel_forms()
    mfma = forms["advection"] == 2 and forms["lift"] == 2
    gbs = lambda b, ms: b / (ms * 1e-3) / 1e9
    tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12
    ms_triad = eng.time_kernel(8, 20)  # y = a x + b y on velocity vectors: the HBM rate this box delivers
    triad = gbs(8.0 * 3 * NQ, ms_triad)
    others = {}
    for name, kid, reps, nbytes in (("k_edge_lift<K,true,0>", 5, 20, 8.0 * 2 * NQ), ("k_trace_apply<K>", 1, 50, 8.0 * 2 * NL),
                                    ("k_backsub<K>", 3, 20, 8.0 * (NL + 2 * NQ + 2 * NP))):
        ms = eng.time_kernel(kid, reps)
        others[name] = dict(ms=ms, GBs=gbs(nbytes, ms), algorithmic_bytes=nbytes)
    pmc, pmc_mfma = {}, {}
    try:  # HBM bytes per launch (and matrix-core busy fractions) from the committed PMC passes -- used only if they
        # were measured on THIS code (hash of the kernel sources) and this workload; otherwise traffic is null
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if tj.get("csrc_sha16") == csrc_sha16() and world == 1:
            for cfg in tj["configs"].values():
                if cfg["workload"] == {"nx": nx, "degree": k}:
                    pmc = {nm: v["hbm_bytes"] for nm, v in cfg["kernels"].items()}
                    pmc_mfma = {nm: v["mfma_busy_frac"] for nm, v in cfg.get("mfma", {}).items()}
    except Exception:
        pmc, pmc_mfma = {}, {}
    lk = lift_kernel_name(eng)
    if mfma:
        nq = {3: 36, 4: 64}[k]
        nqe = (3 * k + 5) // 2
        ksu, mtu, mtq = (NU + 3) // 4, (NU + 15) // 16, (nq + 15) // 16
        n2 = 2 * NU
        ks, mt = (n2 + 3) // 4, (n2 + 15) // 16
        tiles = eng.n_cells / 16.0  # one wave trip per 16 cells (partial tiles of a row ignored)
        adv_issued = (mtq * (6 * ksu + 8 * mtu) + 8 * ksu + 6 * ksu + 12 * mtu) * 2048.0 * tiles
        adv_alg = eng.n_cells * (nq * (16.0 * NU + 8) + 3 * nqe * (14.0 * NU + 20))
        ne = k + 2
        lift_issued = (5 * ks + 5 * mt) * 2048.0 * tiles
        lift_alg = eng.n_cells * (2.0 * 3 * ne * n2 * 2 + 2.0 * n2 * 3 * ne)  # own + neighbour moments, lifting
        cands = {
            "kernel_advection": (f"k_adv_mfma<{k}, true>", " (residual form b - (I - gamma F) x)", 8.0 * 4 * NQ, 7, adv_alg, adv_issued),
            "kernel_lift": (f"k_edge_lift_mfma<{k}, true>", " (lift + fused Chebyshev step)", 8.0 * 4 * NQ, 6, lift_alg, lift_issued),
            "kernel_advection_plain": (f"k_adv_mfma<{k}, false>", " (plain operator)", 8.0 * 3 * NQ, 0, adv_alg, adv_issued),
            "kernel_lift_plain": (f"k_edge_lift_mfma<{k}, false>", " (plain hybrid lift)", 8.0 * 2 * NQ, 9, lift_alg, lift_issued),
        }
    else:
        lt = f"{lk if hybrid else 'k_edge_lift'}<{k}, false, {2 if hybrid else 1}"
        cands = {
            "kernel_advection": (f"k_adv_apply<{k}, true>", " (residual form b - (I - gamma F) x)", 8.0 * 4 * NQ, 7, None, None),
            "kernel_lift": (f"{lt}, true>", " (lift + fused Chebyshev step)", 8.0 * (4 if hybrid else 6) * NQ, 6 if hybrid else 4, None, None),
            "kernel_advection_plain": (f"k_adv_apply<{k}, false>", " (plain operator)", 8.0 * 3 * NQ, 0, None, None),
            "kernel_lift_plain": (f"{lt}, false>", " (plain hybrid lift)", 8.0 * 2 * NQ, 9 if hybrid else None, None, None),
        }
    rows = {}
    for lab, (name, note, nbytes, kid, alg, issued) in cands.items():
        n, tot = (ktimers or {}).get(lab, (0, 0.0))
        alone = eng.time_kernel(kid, 20) if kid is not None else None
        if n:
            ms = tot / n * 1e3
        elif alone is not None:
            ms = alone
        else:
            continue
        row = dict(ms=ms, GBs=gbs(nbytes, ms), algorithmic_bytes=nbytes, traffic=pmc.get(name), launches_timed=n,
                   total_ms_in_bracketed_step=n * ms, ms_stand_alone=alone)
        if alg is not None:
            row.update(TFLOPs=tf(alg, ms), mfma_util=tf(issued, ms) / FP64_MATRIX_PEAK_TF, mfma_busy_pmc=pmc_mfma.get(name),
                       algorithmic_flops=alg, issued_mfma_flops=issued)
        rows[lab] = (name + note, row)
    # dominant: largest total time in the bracketed step; without brackets the longest launch
    dom = max(rows, key=lambda lab: (rows[lab][1]["total_ms_in_bracketed_step"], rows[lab][1]["ms"]))
    dname, drow = rows[dom]
    for lab, (name, row) in rows.items():
        if lab != dom:
            others[name] = row
    timing = ("in place: HIP-event pair around each launch of " + timed_where) if drow["launches_timed"] else "stand-alone launch loop"
    if mfma and dom.startswith("kernel_advection"):
        return dict(bound="mfma", kernel=dname, achieved=drow["TFLOPs"], peak=FP64_MATRIX_PEAK_TF, unit="TFLOP/s",
                    frac=drow["TFLOPs"] / FP64_MATRIX_PEAK_TF, mfma_util=drow["mfma_util"], mfma_busy_pmc=drow["mfma_busy_pmc"],
                    algorithmic_flops=drow["algorithmic_flops"], issued_mfma_flops=drow["issued_mfma_flops"], ms_per_launch=drow["ms"],
                    traffic=drow["traffic"], timing=timing, launches_timed=drow["launches_timed"], ms_stand_alone=drow["ms_stand_alone"],
                    total_ms_in_bracketed_step=drow["total_ms_in_bracketed_step"],
                    hbm_GBs=drow["GBs"], algorithmic_bytes=drow["algorithmic_bytes"], stream_triad_GBs=triad, other_kernels=others)
    return dict(bound="hbm", kernel=dname, achieved=drow["GBs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=drow["GBs"] / HBM_PEAK_GBS,
                traffic=drow["traffic"], algorithmic_bytes=drow["algorithmic_bytes"], ms_per_launch=drow["ms"], timing=timing,
                launches_timed=drow["launches_timed"], ms_stand_alone=drow["ms_stand_alone"],
                total_ms_in_bracketed_step=drow["total_ms_in_bracketed_step"],
                stream_triad_GBs=triad, frac_of_triad=drow["GBs"] / triad, other_kernels=others)


def cpu_baseline(degree, nx_sample=None):
    """The C++/OpenMP CPU twin (oracle/cpu_twin: same discretisation, same solver algorithms and tolerances as the
    engine, written for CPUs -- SIMD lane-blocked advection / lift kernels, one-pass Gram-Schmidt; a stand-in for the
    Firedrake/PETSc path, which cannot be run here) timed on all host cores of this box on a bounded sample of the same
    scheme: same tableau, R, flux, Taylor-Green data and dt = 0.25/nx (1 warm-up + 2 timed steps), at the benchmark
    mesh itself for k <= 2."""
    from oracle.cpu_twin import CpuTwin
    from oracle.hdg_oracle import TABLEAUX

    # k <= 2: the headline mesh itself (C3: about 10 s per step on the 16 cores of a one-GPU box since the SIMD kernels of
    # round 3, so 1 + 2 steps stay near 30 s); higher degrees: a quarter of the benchmark mesh
    nx = nx_sample or (1024 if degree <= 2 else 256)
    dt, kappa = 0.25 / nx, 0.5
    tb = TABLEAUX["imex_ssp2_332"]
    t = CpuTwin(nx=nx, degree=degree, dt=dt, nstages=3, a_expl=tb["a_expl"], a_impl=tb["a_impl"], b_expl=tb["b_expl"],
                b_impl=tb["b_impl"], c_expl=tb["c_expl"])
    xq, xp = t.node_coordinates()
    S = lambda z: np.sin((z - 0.5) * np.pi)
    Cc = lambda z: np.cos((z - 0.5) * np.pi)
    Qs = np.stack([-Cc(xq[:, 0]) * S(xq[:, 1]), S(xq[:, 0]) * Cc(xq[:, 1])], axis=-1)
    ps = (S(xp[:, 0]) ** 2 + S(xp[:, 1]) ** 2) / 2
    t.set_state(Qs, ps)
    t.reconstruct_trace()
    t.set_forcing_profile(Qs)
    t.run_separable(ssp2_scales(1, dt, kappa))  # warm-up step (mirrors --warmup, driver.py:157-162)
    t.iteration_stats(reset=True)
    t.timers(reset=True)
    nsteps = 2
    t0 = time.perf_counter()
    t.run_separable(ssp2_scales(nsteps, dt, kappa, t0=dt))
    el = time.perf_counter() - t0
    sums, cnt = t.iteration_stats()
    its = [float(a / max(b, 1)) for a, b in zip(sums, cnt)]
    # the reference's PerformanceLog labels (logging.py:34-60), as for the GPU run's `timers`
    tm = {lab: {"ncall": n, "total_ms": 1e3 * sec, "avg_ms": 1e3 * sec / max(n, 1)} for lab, (sec, n) in t.timers().items()}
    from oracle.cpu_twin import stream_triad_gbs

    # algorithmic bytes per step of this scheme at this size, from the engine's launch census, are not available on the host;
    # the host's own stream rate is what its kernels can be priced against
    return dict(value=t.n_total * nsteps / el / 1e6, unit="million DOF-updates/s", cores=t.threads, kind="port", timers=tm, nx_sample=nx,
                host_stream_triad_GBs=stream_triad_gbs(),
                sample=f"C++/OpenMP twin (oracle/cpu_twin), HDG-IMEX SSP2(3,3,2) R=2 upwind k={degree} nx={nx} "
                       f"({t.n_total} unknowns), {t.threads} threads, 1 warm-up + {nsteps} timed steps in {el:.1f} s; "
                       f"Krylov iterations tentative/pressure {its[0]:.1f}/{its[1]:.1f} (GMRES(8) / PCG)")


class Watchdog:
    """Bounds every phase of a multi-rank run: `arm(what, seconds)` (re)starts a deadline, `disarm()` clears it; when a
    deadline passes the process prints what overran and ends with exit code 4 without waiting for anybody (a rank blocked
    inside a collective whose peer has died cannot be interrupted from Python)."""

    def __init__(self, rank):
        import threading

        self.rank, self.deadline, self.what = rank, None, ""
        self.lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, what, seconds):
        with self.lock:
            self.what, self.deadline = what, time.monotonic() + seconds

    def disarm(self):
        with self.lock:
            self.deadline = None

    def _run(self):
        while True:
            time.sleep(1.0)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                what = self.what
            if late:
                sys.stderr.write(f"bench.py [rank {self.rank}]: '{what}' overran its deadline: giving up (exit 4)\n")
                sys.stderr.flush()
                os._exit(4)


def gpu_at_size(nx, k, args, kappa):
    """The GPU path on the CPU sample's mesh (same scheme, 1 warm-up + 3 timed steps), so that GPU and CPU numbers exist at
    ONE size as well as the GPU number at the headline size."""
    from incompressibleeulerhdg_amd.mesh import UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    dt = 0.25 / nx
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), k, dt, use_projection_method=True, n_richardson=2,
                                            tent_precond=args.tent_precond, trace_precond=args.trace_precond,
                                            gmres_restart=args.gmres_restart,
                                            **({} if args.tent_solver is None else {"tent_solver": args.tent_solver}))
    eng = ts._engine
    mp = TaylorGreen(ts._V_Q, ts._V_p, "exponential", kappa)
    eng.set_state(ts._V_Q.interpolate(mp.Q_stationary), ts._V_p.interpolate(mp.p_stationary))
    eng.reconstruct_trace()
    eng.set_forcing_profile(mp.f_rhs().profile)
    eng.run_separable(ssp2_scales(1, dt, kappa))
    nsteps = 3
    t0 = time.perf_counter()
    eng.run_separable(ssp2_scales(nsteps, dt, kappa, t0=dt))  # synchronous on return
    el = time.perf_counter() - t0
    out = dict(value=eng.n_total * nsteps / el / 1e6, unit="million DOF-updates/s", nx=nx, ms_per_step=el / nsteps * 1e3)
    eng.close()
    return out


def reassembly_split(eng, k, nx, cb, gpu_s_per_step):
    """BASELINE.md section 4 / SURVEY.md C-11: the reference re-assembles and re-condenses the time-independent mixed-Poisson
    operator on every one of its (s-1) R + 2 solves per step (LinearVariationalProblem without constant_jacobian,
    hdg_imex.py:180-182,190-192,209-213; Slate: per cell LU of the (n_Q+n_p)^2 block, A^-1 G and the Schur complement);
    the engine and the CPU twin build the two shared shape blocks once.  A ONE-LINE ESTIMATE of what that saving is worth,
    so that it is not mistaken for kernel efficiency: flops of the per-cell elimination x cells x solves per step, priced
    at the FP64 rate each side sustains in its densest kernel (GPU: FP64 vector/matrix peak 78.6 TF; CPU: the twin's
    measured rate is unknown, so 16 cores x 16 flop/cycle x 2.5 GHz = 0.64 TF as an optimistic bound)."""
    nxx = 2 * eng.n_u + eng.n_p
    nt = 3 * eng.n_l
    flops_cell = 2.0 / 3.0 * nxx**3 + 2.0 * nxx * nxx * nt + 2.0 * nt * nxx * nt
    solves = (3 - 1) * 2 + 2  # SSP2(3,3,2), R = 2
    per_step = lambda ncell: flops_cell * ncell * solves
    gpu_extra = per_step(2.0 * nx * nx) / (FP64_MATRIX_PEAK_TF * 1e12)
    cpu_nx = cb["gpu_same_size"]["nx"]
    cpu_s_per_step = 1e-3 * cb["timers"]["timestep"]["avg_ms"] if "timestep" in cb.get("timers", {}) else None
    cpu_extra = per_step(2.0 * cpu_nx * cpu_nx) / 0.64e12
    return dict(solves_per_step=solves, flops_per_cell_elimination=flops_cell,
                gpu_extra_ms_per_step_if_reassembled=gpu_extra * 1e3, gpu_share_of_step=gpu_extra / gpu_s_per_step,
                cpu_extra_ms_per_step_if_reassembled=cpu_extra * 1e3,
                cpu_share_of_step=(cpu_extra / cpu_s_per_step) if cpu_s_per_step else None,
                note="lower bounds at peak FP64 rates: not re-assembling is worth at least these shares on either side; "
                     "both sides of the reported GPU/CPU ratio skip it, so the ratio is kernel + solver efficiency only")


def lift_kernel_name(eng):
    """Name of the per-thread lift kernel the engine launches, as the engine itself reports it (hdg_get_kernel_forms: the
    paired form -- both triangles of a square in one workgroup, edge moments through LDS -- or the gather form)."""
    return {0: "k_edge_lift", 1: "k_edge_lift_pair", 2: "k_edge_lift_mfma"}[eng.kernel_forms()["lift"]]


def alt_stop_rule(build, args, dt, kappa, mp, ssp2_scales, tol="1e-15"):
    """The timed steps once more on a fresh engine with HDG_TRACE_BACKWARD_TOL (read when an engine is built): condensed solves
    stop when their preconditioned residual is below tol * |pressure trace| -- ten units in the last place of the field they
    correct -- or at the reference's rtol 1e-12, whichever comes first; update solves start from zero."""
    os.environ["HDG_TRACE_BACKWARD_TOL"] = tol
    try:
        ts2 = build("none")
        e2 = ts2._engine
        e2.set_state(ts2._V_Q.interpolate(mp.Q_stationary), ts2._V_p.interpolate(mp.p_stationary))
        e2.reconstruct_trace()
        e2.set_forcing_profile(mp.f_rhs().profile)
        if args.warmup > 0:
            e2.run_separable(ssp2_scales(args.warmup, dt, kappa))
        e2.iteration_stats(reset=True)
        t0 = time.perf_counter()
        e2.run_separable(ssp2_scales(args.steps, dt, kappa, t0=args.warmup * dt))
        el = time.perf_counter() - t0
        sums, cnt = e2.iteration_stats()
        its = {n: (float(s / c) if c else 0.0) for n, s, c in zip(("tentative", "pressure", "final_pressure", "pressure_reconstruction"), sums, cnt)}
        return dict(value=e2.n_total * args.steps / el / 1e6, unit="million DOF-updates/s", ms_per_step=el / args.steps * 1e3,
                    krylov_iterations_avg=its, backward_error_tol=float(tol),
                    note="secondary number: condensed solves stop at |M r| <= tol * |pressure trace| (or at rtol 1e-12); parity tests "
                         "pass with it (2e-8 against the oracle); not the headline because it is not the reference's stopping rule")
    finally:
        del os.environ["HDG_TRACE_BACKWARD_TOL"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults = the window the round driver uses; the first solves of a run still learn the Chebyshev check schedule
    # (C3: 22.7 tentative-velocity iterations per solve in a 5 + 1 window, 21.8 in this one)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tent-precond", type=int, default=2)
    ap.add_argument("--trace-precond", type=int, default=1)
    ap.add_argument("--gmres-restart", type=int, default=8)
    ap.add_argument("--tent-solver", type=int, default=None, help="0 GMRES, 1 GMRES cycle + Chebyshev (default: by degree)")
    ap.add_argument("--comm", choices=["rccl", "shm"], default="rccl", help="inter-rank transport for --gpus > 1")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}")
    dist = None
    shared_gpu = False
    ndev = 1
    if world > 1:
        import torch
        import torch.distributed as dist_

        ndev = torch.cuda.device_count()
        if ndev >= world:
            torch.cuda.set_device(local_rank)
            dist_.init_process_group("nccl")
        else:
            # rehearsal on a box with fewer GPUs than ranks: ranks share devices, rendezvous over gloo,
            # data over the shared-memory transport (RCCL refuses duplicate devices).  Labelled as such in the
            # output; refused beyond 4 ranks per device (it would say nothing about scaling).
            if ndev < 1 or world > 4 * ndev:
                raise SystemExit(f"bench.py: {world} ranks on {ndev} visible device(s): refusing")
            shared_gpu = True
            local_rank = local_rank % ndev
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
    # A transport that cannot initialise, or a peer that died, must cost a bounded time and a NON-ZERO exit, never a hang:
    # a watchdog thread ends this rank (os._exit: torchrun then tears the job down) when a phase overruns its deadline.
    # No second rendezvous, no child processes: nothing here can wedge or fail a healthy N-rank job.  (The round-2 probe --
    # a child process per rank with its own rendezvous on another port -- remains available as an explicit self-test:
    # BENCH_PROBE_TRANSPORT=1.)  A failure is an ERROR: a number over another transport would be another measurement.
    watchdog = Watchdog(rank)
    if world > 1 and backend == "rccl" and os.environ.get("BENCH_PROBE_TRANSPORT"):
        from incompressibleeulerhdg_amd.distributed import probe_transport

        ok_probe = probe_transport("rccl")
        if int(reduce_scalar(1.0 if ok_probe else 0.0, dist.ReduceOp.MIN)) == 0:
            raise SystemExit("bench.py: the RCCL transport probe failed on at least one rank")
    watchdog.arm("engine construction / transport initialisation", float(os.environ.get("BENCH_INIT_TIMEOUT", "300")))
    try:
        ts = build(backend)
        ok = 1
    except Exception as exc:  # noqa: BLE001
        ok, err = 0, repr(exc)
        print(f"[rank {rank}] transport '{backend}' failed: {err}", file=sys.stderr)
    if int(reduce_scalar(float(ok), dist.ReduceOp.MIN)) == 0 if world > 1 else ok == 0:
        raise SystemExit(f"bench.py: engine construction failed (transport {backend})")
    eng = ts._engine
    if world > 1:
        crank, cn, ct, cname = eng.comm_info()
        if (crank, cn, ct) != (rank, world, world):
            raise SystemExit(f"bench.py: transport reports rank {crank} of {cn} ({ct} in the communicator), launch says {rank} of {world}")
        transport_desc = f"{cname} communicator of {ct} ranks"
    else:
        transport_desc = "none"
    watchdog.arm("state set-up and warm-up steps", float(os.environ.get("BENCH_STEP_TIMEOUT", "600")))
    mp = TaylorGreen(ts._V_Q, ts._V_p, "exponential", kappa)
    eng.set_state(ts._V_Q.interpolate(mp.Q_stationary), ts._V_p.interpolate(mp.p_stationary))
    eng.reconstruct_trace()
    eng.set_forcing_profile(mp.f_rhs().profile)
    if args.warmup > 0:
        eng.run_separable(ssp2_scales(args.warmup, dt, kappa))
    eng.iteration_stats(reset=True)
    eng.solver_events(reset=True)
    eng.timers(reset=True)
    # Every launch of the two kernels of a tentative-velocity iteration can be bracketed by its own HIP-event pair on the
    # engine's stream (hdg_set_kernel_timing: in place, with the operands and cache state of the solve); the roofline block
    # divides by these durations.  The brackets cost two event records per launch (about 1 % of a C3 step, 15 % at C2), so
    # by default they bracket ONE EXTRA step after the timed region: the headline does not pay for its own instrumentation.
    # BENCH_KERNEL_TIMING=timed puts them into the timed steps instead.
    ktiming_extra = os.environ.get("BENCH_KERNEL_TIMING", "extra") == "extra"
    eng.set_kernel_timing(not ktiming_extra)
    eng.launch_stats(reset=True)
    watchdog.arm("timed steps", float(os.environ.get("BENCH_STEP_TIMEOUT", "600")))
    sync_barrier()
    t0 = time.perf_counter()
    eng.run_separable(ssp2_scales(args.steps, dt, kappa, t0=args.warmup * dt))  # synchronous on return
    sync_barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = reduce_scalar(elapsed, dist.ReduceOp.MAX)
    sums, cnt = eng.iteration_stats()
    events = eng.solver_events()  # residual replacements / rounding-floor exits of the condensed solves in the timed steps
    launches = eng.launch_stats(reset=True)  # launch census of exactly the timed steps (this rank)
    timers_raw = eng.timers(reset=ktiming_extra, kernels=True)
    if ktiming_extra:
        eng.set_kernel_timing(True)
        eng.run_separable(ssp2_scales(1, dt, kappa, t0=(args.warmup + args.steps) * dt))
        ktimers = {lab: (n, tot) for lab, (n, tot, _) in eng.timers(reset=True, kernels=True).items() if lab.startswith("kernel_")}
    else:
        ktimers = {lab: (n, tot) for lab, (n, tot, _) in timers_raw.items() if lab.startswith("kernel_")}
    timers_raw = {lab: v for lab, v in timers_raw.items() if not lab.startswith("kernel_")}
    eng.set_kernel_timing(False)
    watchdog.disarm()

    if rank == 0:
        its = {n: (float(s / c) if c else 0.0) for n, s, c in zip(
            ("tentative", "pressure", "final_pressure", "pressure_reconstruction"), sums, cnt)}
        ntot = eng.n_total
        value = ntot * args.steps / elapsed / 1e6
        # device-side section timers of the timed steps (labels of the reference's PerformanceLog)
        timers = {lab: dict(ncall=n, total_ms=tot * 1e3, avg_ms=(tot / n * 1e3 if n else 0.0))
                  for lab, (n, tot, _) in timers_raw.items() if n}
        roof = roofline_block(eng, args, nx, k, world, ktimers, "one extra step after the timed region" if ktiming_extra else "the timed steps")
        # Whole-step roofline (SURVEY.md section 8d / BASELINE.md section 2): sum_k calls_k * algorithmic_bytes_k / elapsed / peak,
        # from the engine's own launch census of the timed steps (every logical vector read or written once per launch,
        # 8 B per owned entry, shared operator tables free; this rank's strip x number of ranks).
        tot_bytes = sum(b for _, b in launches.values()) * world
        tot_calls = sum(c for c, _ in launches.values())
        roof["whole_step"] = dict(
            achieved=tot_bytes / elapsed / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=tot_bytes / elapsed / 1e9 / HBM_PEAK_GBS,
            algorithmic_GB_per_step=tot_bytes / args.steps / 1e9, launches_per_step=tot_calls / args.steps,
            avg_us_per_launch=elapsed / max(tot_calls, 1) * 1e6,
            calls_k={lab: dict(calls_per_step=c / args.steps, GB_per_step=b * world / args.steps / 1e9)
                     for lab, (c, b) in launches.items() if c})
        line = {
            "metric": "million DOF-updates/sec (HDG-IMEX k=2, 1024^2 tri mesh)" if (nx, k) == (1024, 2)
            else f"million DOF-updates/sec (HDG-IMEX k={k}, {nx}^2 tri mesh)",
            "value": value, "unit": "million DOF-updates/s", "n_gpus": world, "n_devices": min(world, ndev), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"HDG-IMEX SSP2(3,3,2) R=2 projection upwind, k={k}, {nx}x{nx} tri mesh, "
                                   f"Taylor-Green kappa=0.5, dt=0.25/nx (BASELINE C3)" if (nx, k) == (1024, 2)
                       else f"HDG-IMEX SSP2(3,3,2) R=2 projection upwind, k={k}, {nx}x{nx} tri mesh",
                       "n_dof": ntot, "krylov_iterations_avg": its, "solver_events": events, "kernel_forms": eng.kernel_forms(),
                       "parallelism": f"strip partition over {world} rank(s), transport {transport_desc}"
                                      + (" (ranks share GPUs: rehearsal)" if shared_gpu else "")},
            "roofline": roof,
            "timers": timers,
        }
        if world == 1 and not os.environ.get("HDG_TRACE_BACKWARD_TOL") and os.environ.get("BENCH_ALT_STOP"):
            # OPT-IN (BENCH_ALT_STOP=1) secondary number, never the headline: the same steps with the normwise backward-error stop
            # of the condensed solves (Engine::pressure_solve) on a second engine.  It is a different
            # stopping rule from the reference's rtol-only KSP (round-3 review): the default run does not spend time on it.
            line["alt_stop_rule"] = alt_stop_rule(build, args, dt, kappa, mp, ssp2_scales)
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(k)
            # the GPU number at the CPU sample's size ("report DOF-updates/s at that size next to the
            # GPU number at the same size and at C3")
            nxs = cb.pop("nx_sample")
            if nxs == nx:  # the CPU sample IS the benchmark mesh: the headline number is the GPU number at the same size
                cb["gpu_same_size"] = dict(value=value, unit="million DOF-updates/s", nx=nx, ms_per_step=elapsed / args.steps * 1e3)
            else:
                cb["gpu_same_size"] = gpu_at_size(nxs, k, args, kappa)
            cb["gpu_over_cpu_same_size"] = cb["gpu_same_size"]["value"] / cb["value"]
            cb["reassembly_split"] = reassembly_split(eng, k, nx, cb, elapsed / args.steps)
            # what the twin moves, priced against ITS memory system: the engine's algorithmic bytes per step (launch census of
            # the timed GPU steps: same discretisation, same stage structure; the twin's GMRES(8) / PCG make at least as many
            # vector passes, so this is a LOWER bound on its traffic) over the twin's seconds per step, next to the host's triad
            if nxs == nx and "timestep" in cb.get("timers", {}):
                gbs_cpu = tot_bytes / args.steps / 1e9 / (cb["timers"]["timestep"]["avg_ms"] * 1e-3)
                cb["algorithmic_GBs_lower_bound"] = gbs_cpu
                cb["fraction_of_host_triad"] = gbs_cpu / cb["host_stream_triad_GBs"] if cb.get("host_stream_triad_GBs") else None
            line["cpu_baseline"] = cb
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
