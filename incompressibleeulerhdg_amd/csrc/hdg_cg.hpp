// Continuous Lagrange space CG_{k+1} on the structured triangulation, the passive-tracer transport operator and the
// vorticity right-hand side (the callers either side of the hot path).
//
// Reference (paths relative to the reference's src/):
//   timesteppers/common.py:110-129   _tracer_advection: u_ = L2 projection of the velocity onto [CG_{k+1}]^2, upwind DG
//                                    transport form T(chi; q, u_)
//   auxilliary/callbacks.py:43-69    vorticity in CG_{k+1} by an L2 projection of the weak curl of the broken velocity
//
// The continuous space shares the node family of the broken velocity space, so a continuous function is a broken one
// whose coincident nodes agree.  Its dofs are attached to the grid corners like the trace space:
//   vertex (i,j) | p-1 interior nodes of H(i,j), V(i,j), D(i,j) | (p-1)(p-2)/2 interior nodes of L(i,j), U(i,j)     p = k+1
// in ONE contiguous vector [vertices | H | V | D | interiors].  A cell reaches its dofs through a small per-shape table
// (CgTabs::fwd, built and cross-checked against node coordinates on the host); a dof gathers from its (at most six)
// cells through the inverse table -- owner computes, no atomics, bitwise reproducible.  On the doubly periodic square
// (driver.py:182-183) the corner indices wrap: nx x ny corners, every one with its three edges and two cells.
// The mass matrix of the continuous space is never formed: apply = per cell (gather nodal values, local mass
// Mloc = Vinv^T Vinv) -> per dof (sum over incident cells), solved by Jacobi-preconditioned CG.
#pragma once
#include "hdg_kernels.hpp"

namespace hdg {

struct CgTabs {
  int p, nint;                 // polynomial degree k+1, interior nodes per cell
  int nx, ny;
  int per, vs;                 // doubly periodic square: corner indices wrap; vs = vertex columns (nx + 1, or nx when periodic)
  long baseH, baseV, baseD, baseI, ncg;
  short fwd[2][21][4];         // [shape][local node] -> {type 0 vertex 1 H 2 V 3 D 4 interior, di, dj, t}
  short vtx[6][4];             // vertex gathers from {shape, ci, cj, node}: cell (shape, i+ci, j+cj)
  short edg[3][4][2][4];       // [H,V,D][t][entry] -> {shape, ci, cj, node}
  short intr[2][6];            // interior dof q of shape s -> local node
};

__device__ __forceinline__ long cg_dof(const CgTabs& C, int type, int I, int J, int t, int s) {
  if (C.per) {  // corner (nx, j) is corner (0, j), corner (i, ny) is corner (i, 0)
    if (I >= C.nx) I -= C.nx;
    if (J >= C.ny) J -= C.ny;
  }
  switch (type) {
    case 0: return (long)J * C.vs + I;
    case 1: return C.baseH + ((long)J * C.nx + I) * (C.p - 1) + t;
    case 2: return C.baseV + ((long)J * C.vs + I) * (C.p - 1) + t;
    case 3: return C.baseD + ((long)J * C.nx + I) * (C.p - 1) + t;
    default: return C.baseI + (((long)J * C.nx + I) * 2 + s) * C.nint + t;
  }
}

// per cell.  MODE 0: y = Mloc * (nodal values gathered from the continuous vector cg)
//            MODE 1: y = Vinv^T x_d          (x: broken velocity, modal; component d)   -> right-hand side of the projection
//            MODE 2: x_d = Vinv * (nodal values gathered from cg)                        -> the projection, modal, component d
//            MODE 3: y = Vinv^T w,  w_m = -int_K (d_x psi_m Q_y - d_y psi_m Q_x) + int_{dK on boundary} psi_m (n x Q)
// y: scalar cell vector with NU planes y[n*Nc + c]
template <int K, int MODE>
__global__ __launch_bounds__(128) void k_cg_cell(Geo g, CgTabs C, DevTables T, const double* __restrict__ Mloc, const double* __restrict__ Vinv,
                                                 const double* __restrict__ Wx0, const double* __restrict__ Wy0,
                                                 const double* __restrict__ Wx1, const double* __restrict__ Wy1,
                                                 const double* __restrict__ Eb, const double* __restrict__ cg,
                                                 double* __restrict__ vel, int d, double* __restrict__ y) {
  constexpr int NU = Dim<K>::NU;
  HDG_CELL_PROLOGUE
  double v[NU], o[NU];
  if (MODE == 0 || MODE == 2) {
#pragma unroll
    for (int n = 0; n < NU; n++) {
      const short* f = C.fwd[s][n];
      v[n] = cg[cg_dof(C, f[0], i + f[1], j + f[2], f[3], s)];
    }
  } else if (MODE == 1) {
#pragma unroll
    for (int m = 0; m < NU; m++) v[m] = vel[vix<NU>(d * NU + m, g.Nc, c)];
  }
  if (MODE == 0) {
#pragma unroll
    for (int n = 0; n < NU; n++) o[n] = 0.0;
    mv_acc<NU, NU>(Mloc, v, o, 1.0);
  } else if (MODE == 1) {
#pragma unroll
    for (int n = 0; n < NU; n++) {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < NU; m++) acc = fma(Vinv[m * NU + n], v[m], acc);
      o[n] = acc;
    }
  } else if (MODE == 2) {
#pragma unroll
    for (int m = 0; m < NU; m++) o[m] = 0.0;
    mv_acc<NU, NU>(Vinv, v, o, 1.0);
#pragma unroll
    for (int m = 0; m < NU; m++) vel[vix<NU>(d * NU + m, g.Nc, c)] = o[m];
    return;
  } else {
    double q[2 * NU], w[NU];
    load_vel<NU>(vel, g.Nc, c, q);
    const double* __restrict__ Wx = s == 0 ? Wx0 : Wx1;
    const double* __restrict__ Wy = s == 0 ? Wy0 : Wy1;
#pragma unroll
    for (int m = 0; m < NU; m++) {
      double acc = 0.0;
#pragma unroll
      for (int l = 0; l < NU; l++) acc += Wy[m * NU + l] * q[l] - Wx[m * NU + l] * q[NU + l];
      w[m] = acc;
    }
#pragma unroll
    for (int e = 0; e < 3; e++) {
      long cn;
      if (nbr(s, e, i, j, g, cn)) continue;
      // boundary edge: + int_e psi_m (n_x Q_y - n_y Q_x), outward normal = sig * n_e
      const double* __restrict__ E = Eb + ((size_t)s * 3 + e) * NU * NU;
      const double nxo = T.sig[s][e] * T.enx[e], nyo = T.sig[s][e] * T.eny[e];
#pragma unroll
      for (int m = 0; m < NU; m++) {
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < NU; l++) acc += E[m * NU + l] * (nxo * q[NU + l] - nyo * q[l]);
        w[m] += acc;
      }
    }
#pragma unroll
    for (int n = 0; n < NU; n++) {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < NU; m++) acc = fma(Vinv[m * NU + n], w[m], acc);
      o[n] = acc;
    }
  }
#pragma unroll
  for (int n = 0; n < NU; n++) y[(long)n * g.Nc + c] = o[n];
}

// nodal values of a continuous function in the broken layout of the library boundary: out[cref*NU + n]
template <int K>
__global__ __launch_bounds__(128) void k_cg_to_broken(Geo g, CgTabs C, const double* __restrict__ cg, double* __restrict__ out) {
  constexpr int NU = Dim<K>::NU;
  HDG_CELL_PROLOGUE
  const long cref = 2 * ((long)j * g.nx + i) + s;
#pragma unroll
  for (int n = 0; n < NU; n++) {
    const short* f = C.fwd[s][n];
    out[cref * NU + n] = cg[cg_dof(C, f[0], i + f[1], j + f[2], f[3], s)];
  }
}

// per grid corner: every continuous dof attached to the corner = sum over its incident cells of y
template <int K>
__global__ __launch_bounds__(128) void k_cg_gather(Geo g, CgTabs C, const double* __restrict__ y, double* __restrict__ out) {
  HDG_CORNER_PROLOGUE
  auto cellval = [&](const short* f) -> double {
    int ci = i + f[1], cj = j + f[2];
    if (C.per) { ci = ci < 0 ? ci + g.nx : ci; cj = cj < 0 ? cj + g.ny : cj; }  // the cells on the other side of the seam
    if (ci < 0 || cj < 0 || ci >= g.nx || cj >= g.ny) return 0.0;
    return y[(long)f[3] * g.Nc + cidx(g, f[0], cj, ci)];
  };
  {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 6; q++) acc += cellval(C.vtx[q]);
    out[cg_dof(C, 0, i, j, 0, 0)] = acc;
  }
  const bool valid[3] = {in_x, in_y, in_x && in_y};
#pragma unroll
  for (int t3 = 0; t3 < 3; t3++) {
    if (!valid[t3]) continue;
    for (int t = 0; t < C.p - 1; t++) out[cg_dof(C, 1 + t3, i, j, t, 0)] = cellval(C.edg[t3][t][0]) + cellval(C.edg[t3][t][1]);
  }
  if (in_x && in_y) {
    for (int s = 0; s < 2; s++)
      for (int q = 0; q < C.nint; q++) out[cg_dof(C, 4, i, j, q, s)] = y[(long)C.intr[s][q] * g.Nc + cidx(g, s, j, i)];
  }
}

// Passive tracer transport (common.py:110-129), tendency in the orthonormal modal basis of DG_k (mass = identity):
//   out_i = int_K q (grad chi_i . u + chi_i div u) - sum_{interior e} int_e chi_i (un_K q_K - un_K' q_K'),
//   un_K = (u.n_K + |u.n_K|)/2 (outflow), un_K' = (|u.n_K| - u.n_K)/2 (inflow); u continuous, so taken from this cell.
// The pressure / tracer modes are the first NP velocity modes (hierarchical Dubiner basis, same scaling), so the
// tabulations of the advection operator serve both.  Cell rule exact to 3k+2, edge rule ceil((3k+4)/2) Gauss points.
template <int K>
__global__ __launch_bounds__(128) void k_tracer_adv(Geo g, DevTables T, const double* __restrict__ q, const double* __restrict__ u,
                                                     double* __restrict__ out) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP;
  HDG_CELL_PROLOGUE
  double qc[NP], uc[2 * NU], F[NP];
  load_cell<NP>(q, g.Nc, c, qc);
  load_vel<NU>(u, g.Nc, c, uc);
#pragma unroll
  for (int r = 0; r < NP; r++) F[r] = 0.0;
  {
    const double* __restrict__ Phi = T.cPhi[s];
    const double* __restrict__ Gx = T.cGx[s];
    const double* __restrict__ Gy = T.cGy[s];
#pragma unroll 1
    for (int p = 0; p < T.nqc; p++) {
      double qq = 0, ux = 0, uy = 0, dv = 0;
#pragma unroll
      for (int m = 0; m < NU; m++) {
        const double ph = Phi[p * NU + m];
        ux = fma(ph, uc[m], ux);
        uy = fma(ph, uc[NU + m], uy);
        dv = fma(Gx[p * NU + m], uc[m], fma(Gy[p * NU + m], uc[NU + m], dv));
        if (m < NP) qq = fma(ph, qc[m], qq);
      }
      const double w = T.cw[p] * qq;
#pragma unroll
      for (int r = 0; r < NP; r++) F[r] = fma(w, fma(Gx[p * NU + r], ux, fma(Gy[p * NU + r], uy, Phi[p * NU + r] * dv)), F[r]);
    }
  }
#pragma unroll
  for (int e = 0; e < 3; e++) {
    long cn;
    if (!nbr(s, e, i, j, g, cn)) continue;
    double qn[NP];
    load_cell<NP>(q, g.Nc, cn, qn);
    const double* __restrict__ Po = T.ePhi[s][e];
    const double* __restrict__ Pn = T.ePhi[1 - s][e];
    const double nxo = T.sig[s][e] * T.enx[e], nyo = T.sig[s][e] * T.eny[e];
#pragma unroll 1
    for (int p = 0; p < T.nqe; p++) {
      double un = 0, qk = 0, qm = 0;
#pragma unroll
      for (int m = 0; m < NU; m++) {
        const double po = Po[p * NU + m];
        un = fma(po, fma(nxo, uc[m], nyo * uc[NU + m]), un);
        if (m < NP) { qk = fma(po, qc[m], qk); qm = fma(Pn[p * NU + m], qn[m], qm); }
      }
      const double a = fabs(un);
      const double flux = T.ew[e][p] * (0.5 * (un + a) * qk - 0.5 * (a - un) * qm);
#pragma unroll
      for (int r = 0; r < NP; r++) F[r] = fma(-Po[p * NU + r], flux, F[r]);
    }
  }
  store_cell<NP>(out, g.Nc, c, F);
}


// ------------------------------------------------------------------------------------------
// Jacobi-preconditioned CG on the continuous-space mass matrix for TWO right-hand sides at once (the two components of the
// velocity projection, common.py:119-122), scalars on the device: per iteration two mass applications, one kernel for the two
// (p, M p), one update kernel (x += alpha p, r -= alpha M p, and (r, r / d) of the new residual in the same pass), one
// direction kernel (p = r / d + beta p), two one-workgroup kernels that finish the reductions and form alpha / beta -- 11
// launches and NO host synchronisation for both components, where the one-vector loop of round 3 (Engine::cg_solve) took
// 20 launches and 4 host round trips.  sc[8 d + ..]: 0 rz, 1 alpha, 2 beta, 3 rz0; sc[16]: both converged (host poll).
// ------------------------------------------------------------------------------------------
#define HDG_CGM_BLOCK 256
__global__ void k_cgm_pap2(long n, const double* __restrict__ p0, const double* __restrict__ a0, const double* __restrict__ p1,
                           const double* __restrict__ a1, double* __restrict__ part) {
  double s0 = 0.0, s1 = 0.0;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    s0 = fma(p0[i], a0[i], s0);
    s1 = fma(p1[i], a1[i], s1);
  }
  __shared__ double sm[HDG_CGM_BLOCK / 64][2];
  const double w0 = wave_sum(s0), w1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = w0; sm[threadIdx.x >> 6][1] = w1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int w = 0; w < HDG_CGM_BLOCK / 64; w++) t += sm[w][threadIdx.x];
    part[(long)threadIdx.x * gridDim.x + blockIdx.x] = t;
  }
}
// finishes a two-value reduction (part[q * nb + b]) and forms the scalars.  mode 0: the values are (p, M p): alpha = rz / pAp;
// mode 1: the values are the new (r, z): beta = new / old, convergence flag; mode 2: first residual: rz = rz0 = value, beta = 0
__global__ void k_cgm_scalars(int nb, int mode, double tol2, const double* __restrict__ part, double* __restrict__ sc, double* __restrict__ hflag) {
  __shared__ double sm[HDG_CGM_BLOCK / 64][2];
  double s0 = 0.0, s1 = 0.0;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) { s0 += part[b]; s1 += part[(long)nb + b]; }
  const double w0 = wave_sum(s0), w1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = w0; sm[threadIdx.x >> 6][1] = w1; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double v[2] = {0.0, 0.0};
  for (int w = 0; w < HDG_CGM_BLOCK / 64; w++) { v[0] += sm[w][0]; v[1] += sm[w][1]; }
  int conv = 0;
  for (int d = 0; d < 2; d++) {
    double* s = sc + 8 * d;
    if (mode == 0) s[1] = (v[d] > 0.0 && s[0] > 0.0) ? s[0] / v[d] : 0.0;  // a converged (or zero) component stands still
    else if (mode == 1) { s[2] = s[0] > 0.0 ? v[d] / s[0] : 0.0; s[0] = v[d]; }
    else { s[0] = s[3] = v[d]; s[2] = 0.0; }
    if (!(s[0] > tol2 * s[3])) conv++;
  }
  sc[16] = conv == 2 ? 1.0 : 0.0;
  if (hflag) { hflag[0] = sc[16]; hflag[1] = sc[0]; hflag[2] = sc[8]; }
}
// x += alpha p ; r -= alpha M p ; partial sums of (r, r / d) of the new residual.  first != 0: only the sums (r = b, x = 0 before)
__global__ void k_cgm_update2(long n, int first, const double* __restrict__ sc, const double* __restrict__ dg, const double* __restrict__ p0,
                              const double* __restrict__ a0, double* __restrict__ x0, double* __restrict__ r0, const double* __restrict__ p1,
                              const double* __restrict__ a1, double* __restrict__ x1, double* __restrict__ r1, double* __restrict__ part) {
  const double al0 = first ? 0.0 : sc[1], al1 = first ? 0.0 : sc[9];
  double s0 = 0.0, s1 = 0.0;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double di = 1.0 / dg[i];
    double ra = r0[i], rb = r1[i];
    if (!first) {
      x0[i] = fma(al0, p0[i], x0[i]);
      x1[i] = fma(al1, p1[i], x1[i]);
      ra = fma(-al0, a0[i], ra);
      rb = fma(-al1, a1[i], rb);
      r0[i] = ra;
      r1[i] = rb;
    }
    s0 = fma(ra * di, ra, s0);
    s1 = fma(rb * di, rb, s1);
  }
  __shared__ double sm[HDG_CGM_BLOCK / 64][2];
  const double w0 = wave_sum(s0), w1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = w0; sm[threadIdx.x >> 6][1] = w1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int w = 0; w < HDG_CGM_BLOCK / 64; w++) t += sm[w][threadIdx.x];
    part[(long)threadIdx.x * gridDim.x + blockIdx.x] = t;
  }
}
// p = r / d + beta p
__global__ void k_cgm_dir2(long n, const double* __restrict__ sc, const double* __restrict__ dg, const double* __restrict__ r0, double* __restrict__ p0,
                           const double* __restrict__ r1, double* __restrict__ p1) {
  const double b0 = sc[2], b1 = sc[10];
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double di = 1.0 / dg[i];
    p0[i] = (b0 == 0.0) ? r0[i] * di : fma(b0, p0[i], r0[i] * di);  // (beta = 0: the buffer may hold anything)
    p1[i] = (b1 == 0.0) ? r1[i] * di : fma(b1, p1[i], r1[i] * di);
  }
}

}  // namespace hdg
