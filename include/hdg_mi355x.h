/* C-ABI of the MI355X-native HDG / HDG-IMEX timestep engine (libhdg_mi355x.so).
 *
 * The reference (eikehmueller/IncompressibleEulerHDG) has no FFI boundary: its timestepper classes
 * call Firedrake directly.  The seam cut here is the METHOD SURFACE of those classes; every entry
 * point below names the reference code it replaces (paths relative to the reference's src/).
 * Plain pointers and sizes only; every function returns 0 on success or a negative HDG_ERR_* code
 * (message via hdg_last_error).  Nothing throws or aborts across this boundary.
 *
 * Ownership: the caller owns every host buffer (C-contiguous float64); the library owns all device
 * memory and all stage vectors, which PERSIST across steps and solve() calls exactly like the
 * reference's Function objects (timesteppers/hdg_imex.py:72-88,174-175,208).
 * A handle is not thread-safe; all calls are synchronous on return.
 *
 * Field layout at the boundary (what Function.dat.data looks like in the reference):
 *   velocity Q  (N_c * n_u, 2)  cell-major, node within cell, component fastest,  n_u = (k+2)(k+3)/2
 *   pressure p  (N_c * n_p,)    n_p = (k+1)(k+2)/2
 *   trace   lam (N_e * n_l,)    n_l = k+1
 * Cell / edge / node numbering: see oracle/fem.py (documented there once; the product's C++ tables
 * implement the same convention independently).
 */
#ifndef HDG_MI355X_H
#define HDG_MI355X_H

#ifdef __cplusplus
extern "C" {
#endif

#define HDG_MAX_STAGES 5

enum {
  HDG_OK = 0,
  HDG_ERR_ARG = -1,         /* bad argument / unsupported configuration */
  HDG_ERR_HIP = -2,         /* HIP runtime error */
  HDG_ERR_NOT_CONVERGED = -3, /* Krylov hit max iterations or broke down */
  HDG_ERR_SINGULAR = -4,    /* singular local block while building tables */
  HDG_ERR_UNSUPPORTED = -5,
  HDG_ERR_COMM = -6         /* inter-rank transport error (RCCL / shared memory) */
};

/* keys of hdg_pressure_solve (timesteppers/hdg_imex.py:258-272: "stage_i", "final_stage",
 * "pressure_reconstruction").  Stage i is passed as the integer i >= 1. */
#define HDG_KEY_FINAL_STAGE 0
#define HDG_KEY_PRESSURE_RECONSTRUCTION -1

/* `which` of hdg_shift_pressure / hdg_get_field */
#define HDG_STATE_CURRENT 0   /* _current_state          (hdg_imex.py:175) */
#define HDG_STATE_UPDATE -1   /* _update                 (hdg_imex.py:174) */
#define HDG_STATE_RECON -2    /* _pressure_reconstruction (hdg_imex.py:208) */
/* which = i >= 1 : _stage_state[i]; which = 100 + i: (_Q_tentative[i], -, -); 200 + i: (_Qstar[i], -, -) */

typedef struct hdg_config {
  int nx, ny;            /* UnitSquareMesh(nx, ny) triangles (driver.py:181); nx == ny required for GTMG */
  int degree;            /* k = pressure degree (common.py:27), 1..4 */
  double dt;
  int flux_upwind;       /* 1 "upwind", 0 "centered" (hdg_imex.py:325) */
  int use_projection;    /* hdg_imex.py:53; 0 selects the unsplit (monolithic) stage solve */
  int n_richardson;      /* hdg_imex.py:60 */
  double tau;            /* hdg_imex.py:58 */
  double alpha_penalty;  /* hdg_imex.py:56 */
  int nstages;           /* s; tableau arrays are row-major s x s / length s (hdg_imex.py:702-1038) */
  double a_expl[HDG_MAX_STAGES * HDG_MAX_STAGES];
  double a_impl[HDG_MAX_STAGES * HDG_MAX_STAGES];
  double b_expl[HDG_MAX_STAGES];
  double b_impl[HDG_MAX_STAGES + 1]; /* ARS3(4,4,3) carries 6 entries as written (hdg_imex.py:874) */
  double c_expl[HDG_MAX_STAGES];
  int equispaced_nodes;  /* 0: recursive GLL ("spectral") nodes, 1: equispaced lattice */
  double tent_rtol;      /* 1e-10 (hdg_imex.py:226) */
  int tent_maxit;
  int gmres_restart;     /* PETSc default 30 */
  int tent_precond;      /* 0 element block-Jacobi; 1 additive two-level  Dinv + Pi Pi^T;
                          * 2 hybrid two-level  Pi + Dinv (I - Pi)  (Pi = BDM projection), one kernel */
  int tent_solver;       /* 0 restarted GMRES, 1 one GMRES cycle (Ritz bounds) + Chebyshev iteration, GMRES fallback */
  double trace_rtol;     /* 1e-12 (hdg_imex.py:137) */
  int trace_maxit;
  int trace_precond;     /* 0 edge block-Jacobi, 1 GTMG-like two-level (P1 coarse space + geometric MG) */
  double unsplit_rtol;       /* outer FGMRES of the unsplit (monolithic) solves, relative to the initial residual */
  double unsplit_inner_rtol; /* inexact inner tentative / pressure solves of its preconditioner */
  int unsplit_restart;
  int unsplit_maxit;
  int device;            /* HIP device ordinal */
  /* mesh variants: periodic != 0 selects the doubly periodic square
   * PeriodicSquareMesh(nx, nx, L) of driver.py:182-183 (no boundary edges, 3 nx ny edges, single rank); length = side
   * of the square (0 means 1: UnitSquareMesh, driver.py:181) */
  int periodic;
  double length;
} hdg_config;

typedef struct hdg_handle hdg_handle;

/* constructor of IncompressibleEulerHDGIMEX / IncompressibleEulerHDGImplicit
 * (hdg_imex.py:29-255, hdg_implicit.py:17-50, common.py:23-73): spaces, 1/h_F, local operator tables,
 * all persistent stage vectors (zero-initialised), solver workspaces. */
int hdg_create(const hdg_config* cfg, hdg_handle** out);
/* General affine triangulation: `UnitDiskMesh(refinement_level)` of src/driver.py:184-185, the
 * mesh of the Kelvin-Helmholtz set-up (src/model_problems.py:108-131), or any conforming triangulation: coords (n_vertices, 2),
 * cells (n_cells, 3) vertex numbers.  cfg.nx / ny / periodic / length are ignored.  The handle serves the same entry points
 * (state, per-solve calls, hdg_step / hdg_run_separable, norms, node coordinates); per-element geometry replaces the two
 * shared element shapes: assembled solution-independent operators + hand-written advection / reconstruction / tracer kernels,
 * GMRES with the hybrid two-level preconditioner for the tentative velocity (hdg_imex.py:223-255), condensation + CG with
 * Chebyshev / edge block-Jacobi smoothing (ASMStarPC of hdg_imex.py:143-152) and a P1 coarse space solved by a
 * smoothed-aggregation V-cycle (GTMG + GAMG of hdg_imex.py:139-167) for the pressure; projection and monolithic solves,
 * tracer and vorticity (hdg_tracer_*, hdg_vorticity, hdg_cg_*) as on the structured meshes.  Single rank.
 * Numbering at the boundary: cells as given; local edge l joins vertices l and (l+1)%3 of its cell; edges in order of first
 * appearance while walking the cells, directed from the lower to the higher vertex number; velocity / pressure nodes: the
 * lattice of the structured path on x = v0 + (v1 - v0) xi + (v2 - v0) eta; trace nodes along the edge direction.
 * hdg_general_topology: edge_vertices (n_edges, 2) and edge_cells (n_edges, 2; -1 = boundary) in that numbering. */
int hdg_create_general(const hdg_config* cfg, int n_vertices, const double* coords, int n_cells, const int* cells, hdg_handle** out);
int hdg_general_topology(const hdg_handle* h, int* edge_vertices, int* edge_cells);
int hdg_destroy(hdg_handle* h);
const char* hdg_last_error(const hdg_handle* h); /* h may be NULL for create errors */

/* Strip-partitioned engine, one process per GPU: rank r of nranks owns the
 * cell rows r*ny/nranks .. (r+1)*ny/nranks - 1 of the global mesh (cfg->ny must be divisible by
 * nranks); all host arrays of the other entry points then refer to THIS RANK'S STRIP in local
 * numbering (edge rows 0..ny_local; the top row of a non-top rank duplicates its upper neighbour's
 * bottom row).  backend HDG_COMM_RCCL: `token` = the 128 bytes from hdg_rccl_unique_id of rank 0
 * (one GPU per rank); HDG_COMM_SHM: `token` = name of a POSIX shared-memory segment, host-staged
 * transport for several ranks on one node (also on one GPU).  Reductions are over owned entries and
 * identical on every rank, so all ranks take the same Krylov iterations. */
#define HDG_COMM_RCCL 1
#define HDG_COMM_SHM 2
int hdg_create_distributed(const hdg_config* cfg, int rank, int nranks, int backend, const char* token, hdg_handle** out);
int hdg_rccl_unique_id(char* out128);
/* One-GPU self-test of the RCCL transport (no reference counterpart): a 1-rank communicator runs the grouped send / recv
 * pattern of a halo exchange with itself as both neighbours, an all-reduce and an all-gather of n doubles each;
 * *max_err = largest deviation from the expected buffer contents (0 when everything arrived). */
int hdg_rccl_selftest(int device, int n, double* max_err);

/* sizes of this rank's strip: n_cells, n_edges, n_u, n_p, n_l */
int hdg_get_sizes(const hdg_handle* h, long* n_cells, long* n_edges, int* n_u, int* n_p, int* n_l);

/* interpolated initial condition -> _current_state (hdg_imex.py:520-533; hdg_implicit.py:82-84):
 * nodal Q (N_c*n_u,2), nodal p (N_c*n_p); subtracts the pressure mean (hdg_imex.py:522). */
int hdg_set_state(hdg_handle* h, const double* Q, const double* p);
/* nodal copies of a state; any output pointer may be NULL */
int hdg_get_field(hdg_handle* h, int which, double* Q, double* p, double* lam);
/* overwrite a field (test hook; mirrors Function.assign) */
int hdg_set_field(hdg_handle* h, int which, const double* Q, const double* p, const double* lam);

/* forcing b_rhs[slot] <- interpolate f(t) (hdg_imex.py:554-557, slot = stage index; slot = nstages
 * addresses _b_new, hdg_imex.py:629).  Nodal values (N_c*n_u, 2). */
int hdg_set_forcing_nodal(hdg_handle* h, int slot, const double* f);
/* separable forcing f(t) = g(t) * profile (Taylor-Green: model_problems.py:76-79): store the nodal
 * profile once, then set b_rhs[slot] = scale * profile without moving data */
int hdg_set_forcing_profile(hdg_handle* h, const double* profile);
int hdg_set_forcing_scale(hdg_handle* h, int slot, double scale);

/* _reconstruct_trace(_current_state) (hdg_imex.py:450-469) */
int hdg_reconstruct_trace(hdg_handle* h);
/* project_bdm (common.py:91-108): _Qstar[dst] <- BDM(_stage_state[src].Q)  (hdg_imex.py:565-567) */
int hdg_project_bdm(hdg_handle* h, int src_stage, int dst);
/* standalone project_bdm on caller data: nodal in -> nodal out (broken [P_{k+1}]^2 representation) */
int hdg_project_bdm_nodal(hdg_handle* h, const double* Qin, double* Qout);
/* _stage_state[0].assign(_current_state) (hdg_imex.py:558) */
int hdg_begin_step(hdg_handle* h);
/* tentative_velocity_solve("stage_i") (hdg_imex.py:274-281); returns Krylov iterations */
int hdg_tentative_solve(hdg_handle* h, int stage, int* its);
/* the "unsplit_solve" branch (hdg_imex.py:600-620): monolithic (u, phi, lambda) solve of stage i into
 * _stage_state[i]; returns outer Krylov iterations */
int hdg_unsplit_solve(hdg_handle* h, int stage, int* its);
/* pressure_solve(key) (hdg_imex.py:257-272); returns condensed-Krylov iterations */
int hdg_pressure_solve(hdg_handle* h, int key, int* its);
/* _shift_pressure(state) (hdg_imex.py:471-478) */
int hdg_shift_pressure(hdg_handle* h, int which);
/* Richardson update of stage i (hdg_imex.py:580-599) */
int hdg_stage_update(hdg_handle* h, int stage);
/* copy p, lambda from _pressure_reconstruction into _current_state and shift (hdg_imex.py:633-637) */
int hdg_finish_step(hdg_handle* h);
/* the whole loop body hdg_imex.py:551-637, device resident; forcing slots must be set */
int hdg_step(hdg_handle* h);
/* nsteps fused steps with separable forcing: scales[n*(nstages+1) + slot] */
int hdg_run_separable(hdg_handle* h, int nsteps, const double* scales);

/* one step of IncompressibleEulerHDGImplicit.solve (hdg_implicit.py:92-190), projection method or
 * monolithic according to cfg.use_projection; forcing slot 0 holds f(t_k) */
int hdg_implicit_step(hdg_handle* h, int* its_tentative, int* its_pressure);

/* iteration statistics accumulated since the last reset (hdg_imex.py:90-93,648-658):
 * sums[4] / counts[4] for tentative, pressure, final pressure, pressure reconstruction */
int hdg_get_iteration_stats(hdg_handle* h, double* sums, long* counts, int reset);
/* Events of the condensed (trace) solves since the last reset.  The reference's KSP (hdg_imex.py:136-137: rtol 1e-12,
 * PETSc max_it 10000) either converges or raises; SURVEY.md 5.3: "Krylov divergence / max-it is an error code, never
 * silent".  The single-reduction CG of this library replaces its residual by the true one (r = b - T x, recurrences
 * restarted) when the recurrence residual stalls within three decades of the target or p.Ap turns non-positive, and ends
 * a solve only if the TRUE preconditioned residual meets rtol or lies at its rounding floor 32 eps |x|:
 *   events[0] residual replacements, events[1] solves ended at the rounding floor.
 * A third confirmed drift within one solve returns HDG_ERR_NOT_CONVERGED.
 * Tentative-velocity solves (hdg_imex.py:223-228: GMRES rtol 1e-10): events[2] s-step minimal-residual cycles taken by the tail
 * of the Chebyshev iteration, events[3] solves that two weak cycles in a row handed on to GMRES (which errors at its limit). */
#define HDG_N_SOLVER_EVENTS 4
int hdg_get_solver_events(hdg_handle* h, long* events, int reset);

/* Device-side timers with the labels of the reference's PerformanceLog (src/auxilliary/logging.py:11-60, used at
 * hdg_imex.py:257,274,551,564,601): index 0 timestep, 1 bdm_projection, 2 tentative_velocity_solve, 3 pressure_solve,
 * 4 unsplit_solve.  Sections are bracketed by events on the engine's stream (no host synchronisation inside a fused
 * step), so hdg_step / hdg_run_separable report the same per-solve breakdown as the per-solve calls.  For each label
 * total_ms[i], sumsq_ms[i] (sum of squares, for the standard deviation log_summary prints) and ncalls[i] since the
 * last reset. */
#define HDG_N_TIMERS 9
int hdg_get_timers(hdg_handle* h, double* total_ms, double* sumsq_ms, long* ncalls, int reset);
/* Labels 5 .. 8 (no reference counterpart; the measurement SURVEY.md section 8(d) asks for): every launch of the two
 * kernels of a tentative-velocity iteration inside a solver, by FORM -- 5: advection operator in residual form
 * b - (I - gamma F) x (Chebyshev phase, first residual of a cycle), 6: hybrid edge-lift preconditioner with the fused
 * Chebyshev step, 7: plain advection operator, 8: plain hybrid lift (7, 8: GMRES and the s-step cycles) -- bracketed by its
 * own event pair IN PLACE, i.e. with the operands and cache state of the solve.  Recorded only while switched on (two event
 * records per launch: about 1 % of a C3 step). */
int hdg_set_kernel_timing(hdg_handle* h, int on);
/* Transport of a distributed handle as the transport itself reports it: this rank, the number of ranks of the strip
 * partition, the size of the communicator (RCCL: ncclCommCount; must equal nranks) and its name ("self", "rccl", "shm";
 * name16: at least 16 bytes).  No reference counterpart (the reference has no explicit communication). */
int hdg_get_comm_info(const hdg_handle* h, int* rank, int* nranks, int* transport_ranks, char* name16);
/* Launch census since the last reset (no reference counterpart; SURVEY.md section 8(d): "roofline.achieved = sum_k calls_k *
 * bytes_k / elapsed / BW_peak with calls_k printed alongside"): per kernel class the number of launches and the ALGORITHMIC
 * bytes they moved -- every logical vector read or written once per launch, 8 B per owned entry, shared operator tables
 * free.  Classes: 0 advection_apply (K3), 1 edge_lift (K1, K4: BDM projection / two-level
 * preconditioner, incl. the fused Chebyshev step), 2 stage_rhs (K2: pressure-gradient combination, reconstruction rhs),
 * 3 weak_divergence, 4 condense (K5), 5 trace_apply (K6), 6 trace_smooth (K7: smoother steps on the trace space),
 * 7 backsub (K8), 8 vertex_multigrid (K7: P1 V-cycle and transfers), 9 vector_update (K9), 10 dot (K9),
 * 11 copy_fill, 12 other (scalar kernels, trace reconstruction, constraint rows). */
#define HDG_N_LAUNCH_CLASSES 13
int hdg_get_launch_stats(hdg_handle* h, long* calls, double* bytes, int reset);
/* Which form of its kernels this handle launches (no reference counterpart; bench.py labels its roofline block and looks
 * up the PMC traffic by these instead of re-deriving the engine's rules):
 *   forms[0] edge lift (BDM projection / hybrid preconditioner): 0 gather form k_edge_lift, 1 paired form k_edge_lift_pair,
 *            2 matrix-core k_edge_lift_mfma, 3 assembled operator (general meshes);
 *   forms[1] advection operator: 0 per-thread k_adv_apply, 2 matrix-core k_adv_mfma, 3 general-mesh kernel k_g_adv;
 *   forms[2] trace preconditioner: 0 row-stencil kernels k_trace_smooth, 1 LDS-tiled k_trace_pre_tile / k_trace_post_tile (one
 *            thread per grid corner: k <= 3), 2 the same tiles with one thread per edge, k_trace_pre_tile3 / k_trace_post_tile3
 *            (k = 4), 3 general meshes;
 *   forms[3] local Schur kernels (back-substitution, pressure gradient, weak divergence): 0 per-thread, 2 matrix-core,
 *            3 assembled operators (general meshes). */
#define HDG_N_KERNEL_FORMS 4
int hdg_get_kernel_forms(hdg_handle* h, int* forms);

/* ---- passive tracer.  Explicit DG transport of a scalar in DG_k by the L2 projection of
 * the stage velocity onto [CG_{k+1}]^2 (common.py:110-129): q_i = q_0 + dt sum_{j<i} a_expl[i,j] M^-1 T(q_j, P(Q_i))
 * after every stage (hdg_imex.py:415-431,622-623), q^{n+1} = q_0 + dt sum_i b_expl[i] M^-1 T(q_i, P(Q_i)) at the end of
 * the step (:433-448,638-639); hdg_implicit.py:93-96,192-193: q^{n+1} = q^n + dt M^-1 T(q^n, P(Q^n)).
 * hdg_set_tracer(h, q) switches the tracer on (q: nodal DG_k values, pressure layout; NULL switches it off); hdg_step /
 * hdg_run_separable / hdg_implicit_step then carry it along; the three step-level calls serve the per-solve loop. */
int hdg_set_tracer(hdg_handle* h, const double* q);
int hdg_get_tracer(hdg_handle* h, double* q);
int hdg_tracer_begin_step(hdg_handle* h);        /* self._q[0].assign(q_tracer), hdg_imex.py:560 */
int hdg_tracer_stage(hdg_handle* h, int stage);  /* solve(a_tracer == self._tracer_residual(chi, i), self._q[i]), :622-623 */
int hdg_tracer_finish_step(hdg_handle* h);       /* solve(a_tracer == self._tracer_final_residual(chi), q_tracer), :638-639 */
/* test hook: out = M^-1 T(.; q, u) for nodal q (DG_k) and nodal velocity u; project != 0: u is projected onto CG first */
int hdg_apply_tracer_advection(hdg_handle* h, const double* q, const double* u, int project, double* out);

/* ---- continuous space CG_{k+1} (same node family as the broken velocity space)
 * hdg_cg_size: number of dofs; hdg_cg_coordinates: their positions (n_cg, 2);
 * hdg_cg_project_nodal: Function(V_CG).project(u) of common.py:119-122, returned as nodal values of the broken space;
 * hdg_vorticity: the CG_{k+1} vorticity of AnimationCallback.vorticity_solver (callbacks.py:43-69) of the nodal
 * velocity Q (NULL: the current state), n_cg values in dof order */
int hdg_cg_size(hdg_handle* h, long* n_cg);
int hdg_cg_coordinates(hdg_handle* h, double* xy);
int hdg_cg_project_nodal(hdg_handle* h, const double* Qin, double* Qout);
int hdg_vorticity(hdg_handle* h, const double* Q, double* omega);
/* nodal values of a continuous function (n_cg values) on the broken node set of the velocity space, (N_c*n_u,):
 * what the VTK writer needs to write a CG function next to the broken fields (callbacks.py:85) */
int hdg_cg_to_broken(hdg_handle* h, const double* cg_values, double* broken);

/* physical coordinates of the DG nodes, boundary numbering: xq (N_c*n_u, 2), xp (N_c*n_p, 2); what
 * `interpolate` evaluates expressions at (hdg_imex.py:520-521,555; model_problems.py:88-103) */
int hdg_node_coordinates(hdg_handle* h, double* xq, double* xp);
/* L2 norms of nodal fields (driver.py:376-377) and integral of a nodal pressure (model_problems.py:104) */
int hdg_l2_norms(hdg_handle* h, const double* Q, const double* p, double* norm_Q, double* norm_p);
int hdg_integrate_pressure(hdg_handle* h, const double* p, double* integral);

/* kernel-level access for parity tests and micro-benchmarks (nodal in / nodal out) */
int hdg_apply_advection(hdg_handle* h, const double* Qstar, const double* x, double gamma, double* y);
int hdg_apply_trace_operator(hdg_handle* h, const double* lam, double* out);
int hdg_apply_weak_divergence(hdg_handle* h, const double* Q, int broken, double* out_p);
/* device-resident micro-benchmarks for bench.py: run `reps` launches of one kernel on the internal
 * stream, return the average milliseconds per launch measured with HIP events on that stream.
 * kernel: 0 advection apply, 1 trace apply, 2 BDM projection, 3 back-substitution,
 *         4 BDM lift + block-Jacobi + Chebyshev step (fused; additive preconditioner), 5 transposed BDM lift,
 *         6 hybrid preconditioner (BDM lift + block-Jacobi of the remainder) + Chebyshev step (fused),
 *         7 advection apply in residual form  b - (I - gamma F) x,
 *         8 stream triad y = a x + b y on two velocity-sized vectors (the box's achievable HBM rate),
 *         9 hybrid preconditioner without the Chebyshev step (k >= 3: matrix-core lift), 10 condensation (pressure-row
 *         form), 11 pressure-gradient combination, 12 weak divergence, 13 pressure-reconstruction right-hand side */
int hdg_time_kernel(hdg_handle* h, int kernel, int reps, double* ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif
