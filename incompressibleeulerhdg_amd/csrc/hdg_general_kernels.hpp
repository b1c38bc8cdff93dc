// Device side of the general-mesh path (hdg_general.hpp): one generic CSR kernel for every solution-independent operator
// and the two solution-dependent forms with per-cell geometry.  Vector layouts: velocity (c * 2NU + d * NU + m), pressure
// (c * NP + r), trace (e * NL + a): cell- / edge-major, modal, physically orthonormal bases.
//
// Bound: the CSR kernel reads 12 B of matrix per 2 flops -- it is matrix-bandwidth bound by construction (the price of
// per-cell geometry without a per-shape table); the advection kernel reads its geometry (12 doubles + 9 ints per cell) and
// shares the reference tabulations through the scalar / L1 path.  This path exists for the reference's unstructured
// set-ups (UnitDiskMesh, a few 10^3 .. 10^5 cells), not for the headline benchmark.
#pragma once
#include <hip/hip_runtime.h>

namespace hdg {

struct DevCsr {
  int nrows = 0, ncols = 0;
  long nnz = 0;
  const int* rowptr = nullptr;
  const int* col = nullptr;
  const double* val = nullptr;
  int tpr = 1;  // threads per row of k_csr_apply (a power of two <= 64, from the average row length)
};

// y = beta * y + alpha * A x.  T threads share a row (rows hold 1 .. 300 entries): consecutive lanes read consecutive
// entries, so a wave's loads of val / col coalesce; the partial sums are combined by shuffles in a fixed order (deterministic).
// (One thread per row made every lane walk its own row: 163 us per launch on average on the 32 768-cell disk, 95 % of a step.)
template <int T>
__global__ __launch_bounds__(256) void k_csr_apply(DevCsr A, const double* __restrict__ x, double alpha, double beta,
                                                   double* __restrict__ y) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long r = gid / T;
  const int lane = (int)(gid % T);
  const bool live = r < A.nrows;
  double acc = 0.0;
  if (live) {
    const int b = A.rowptr[r], e = A.rowptr[r + 1];
    for (int q = b + lane; q < e; q += T) acc = fma(A.val[q], x[A.col[q]], acc);
  }
#pragma unroll
  for (int off = T / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, T);
  if (live && lane == 0) y[r] = beta == 0.0 ? alpha * acc : fma(alpha, acc, beta * y[r]);
}

struct GGeo {
  int nc;
  const double* inv_sdet;  // nc: 1 / sqrt(detJ)  (scale of the orthonormal basis)
  const double* detJ;      // nc
  const double* Jinv;      // nc x 4 row-major: d xi_rho / d x_d = Jinv[rho * 2 + d]
  const int* cnbr;         // nc x 3: neighbour across local edge l, -1 on the boundary
  const int* ctab;         // nc x 3: own edge table  l * 2 + flip
  const int* ntab;         // nc x 3: the neighbour's edge table for the same edge
  const double* csig;      // nc x 3: +1 = the fixed edge normal points out of the cell
  const double* celen;     // nc x 3
  const double* cenx;      // nc x 3
  const double* ceny;      // nc x 3
  const double *cw, *cPhi, *cGxi, *cGeta;  // cell rule: weights (sum 1/2), nqc x NU tabulations on the reference element
  const double *ew, *ePhi, *eGxi, *eGeta;  // edge rule on [0, 1]; [l * 2 + flip][nqe x NU]
  int nqc, nqe;
  double alpha;
};

// y = x - gamma F(Q*) x  (bsub != nullptr: bsub - that), F = f_impl of hdg_imex.py:313-331; the arithmetic of k_adv_apply
// with the two shared shape tables replaced by (reference tabulation, per-cell Jacobian)
template <int K>
__global__ __launch_bounds__(64) void k_g_adv(GGeo G, const double* __restrict__ xin, const double* __restrict__ qstar,
                                              double* __restrict__ out, double gamma, double upwind, const double* __restrict__ bsub) {
  constexpr int NU = Dim<K>::NU, N2 = 2 * NU;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= G.nc) return;
  double x[N2], qs[N2], F[N2];
#pragma unroll
  for (int n = 0; n < N2; n++) { x[n] = xin[(long)c * N2 + n]; qs[n] = qstar[(long)c * N2 + n]; F[n] = 0.0; }
  const double s = G.inv_sdet[c], dj = G.detJ[c];
  const double j00 = G.Jinv[4 * (long)c + 0] * s, j01 = G.Jinv[4 * (long)c + 1] * s, j10 = G.Jinv[4 * (long)c + 2] * s,
               j11 = G.Jinv[4 * (long)c + 3] * s;
  for (int q = 0; q < G.nqc; q++) {
    double qx = 0, qy = 0, dxx = 0, dxy = 0, dyx = 0, dyy = 0;  // dab = d_b x_a
    for (int m = 0; m < NU; m++) {
      const double ph = G.cPhi[q * NU + m] * s, gxi = G.cGxi[q * NU + m], get = G.cGeta[q * NU + m];
      const double gx = j00 * gxi + j10 * get, gy = j01 * gxi + j11 * get;
      qx = fma(ph, qs[m], qx); qy = fma(ph, qs[NU + m], qy);
      dxx = fma(gx, x[m], dxx); dxy = fma(gy, x[m], dxy);
      dyx = fma(gx, x[NU + m], dyx); dyy = fma(gy, x[NU + m], dyy);
    }
    const double w = G.cw[q] * dj;
    const double ax = -w * (qx * dxx + qy * dxy), ay = -w * (qx * dyx + qy * dyy);
    for (int m = 0; m < NU; m++) {
      const double ph = G.cPhi[q * NU + m] * s;
      F[m] = fma(ph, ax, F[m]);
      F[NU + m] = fma(ph, ay, F[NU + m]);
    }
  }
  for (int l = 0; l < 3; l++) {
    const int cn = G.cnbr[3 * (long)c + l];
    const bool has = cn >= 0;
    const double* __restrict__ Po = G.ePhi + (long)G.ctab[3 * (long)c + l] * G.nqe * NU;
    const double* __restrict__ Pn = G.ePhi + (long)(has ? G.ntab[3 * (long)c + l] : 0) * G.nqe * NU;
    const double sn = has ? G.inv_sdet[cn] : 0.0;
    const double nx_ = G.cenx[3 * (long)c + l], ny_ = G.ceny[3 * (long)c + l], sg = G.csig[3 * (long)c + l];
    const double len = G.celen[3 * (long)c + l], pen = G.alpha / len;
    double xn[N2];
#pragma unroll
    for (int n = 0; n < N2; n++) xn[n] = has ? xin[(long)cn * N2 + n] : 0.0;
    for (int q = 0; q < G.nqe; q++) {
      double ox = 0, oy = 0, bx = 0, by = 0, qn = 0;
      for (int m = 0; m < NU; m++) {
        const double po = Po[q * NU + m] * s, pn = Pn[q * NU + m] * sn;
        ox = fma(po, x[m], ox); oy = fma(po, x[NU + m], oy);
        bx = fma(pn, xn[m], bx); by = fma(pn, xn[NU + m], by);
        qn = fma(po, fma(nx_, qs[m], ny_ * qs[NU + m]), qn);
      }
      const double w = G.ew[q] * len;
      const double cf = has ? w * (0.5 * sg * qn - upwind * fabs(qn)) : 0.0;
      const double jx = ox - bx, jy = oy - by;
      const double jn = (jx * nx_ + jy * ny_) * pen * w;
      const double vx = cf * jx - jn * nx_, vy = cf * jy - jn * ny_;
      for (int m = 0; m < NU; m++) {
        const double po = Po[q * NU + m] * s;
        F[m] = fma(po, vx, F[m]);
        F[NU + m] = fma(po, vy, F[NU + m]);
      }
    }
  }
#pragma unroll
  for (int n = 0; n < N2; n++) {
    const double v = fma(-gamma, F[n], x[n]);
    out[(long)c * N2 + n] = bsub ? bsub[(long)c * N2 + n] - v : v;
  }
}

// y = beta * z + alpha * A x  (z: a third vector; the residual b - A x without copying b first)
template <int T>
__global__ __launch_bounds__(256) void k_csr_apply3(DevCsr A, const double* __restrict__ x, double alpha, double beta,
                                                    const double* __restrict__ z, double* __restrict__ y) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long r = gid / T;
  const int lane = (int)(gid % T);
  const bool live = r < A.nrows;
  double acc = 0.0;
  if (live) {
    const int b = A.rowptr[r], e = A.rowptr[r + 1];
    for (int q = b + lane; q < e; q += T) acc = fma(A.val[q], x[A.col[q]], acc);
  }
#pragma unroll
  for (int off = T / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, T);
  if (live && lane == 0) y[r] = fma(alpha, acc, beta * z[r]);
}
// Chebyshev(2) / Jacobi smoother of an algebraic multigrid level in two launches instead of six or seven (round 4: the
// general-mesh path is launch bound, ~5 us per launch on vectors of 10^3 .. 10^4 entries):
//   first  (FIRST = true):   r0 = b - A x  (x_in = nullptr: r0 = b),  d0 = dinv r0 / theta;  stores r0, d0
//   second (FIRST = false):  r1 = r0 - A d0,  d1 = c1 d0 + c2 dinv r1,  x = (x_in ? x_in : 0) + d0 + d1
// (in the second launch a row reads d0 of OTHER rows and writes x of its own row only: no hazard)
template <int T, bool FIRST>
__global__ __launch_bounds__(256) void k_amg_cheb(DevCsr A, const double* __restrict__ dinv, const double* __restrict__ b,
                                                  const double* x_in, double* __restrict__ r0, double* __restrict__ d0,
                                                  double* x_out, double inv_theta, double c1, double c2) {  // x_in may be x_out
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long r = gid / T;
  const int lane = (int)(gid % T);
  const bool live = r < A.nrows;
  const double* __restrict__ v = FIRST ? x_in : d0;
  double acc = 0.0;
  if (live && v) {
    const int bb = A.rowptr[r], e = A.rowptr[r + 1];
    for (int q = bb + lane; q < e; q += T) acc = fma(A.val[q], v[A.col[q]], acc);
  }
#pragma unroll
  for (int off = T / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, T);
  if (live && lane == 0) {
    if (FIRST) {
      const double rr = b[r] - acc;
      r0[r] = rr;
      d0[r] = dinv[r] * rr * inv_theta;
    } else {
      const double r1 = r0[r] - acc, dd = d0[r];
      const double d1 = fma(c1, dd, c2 * dinv[r] * r1);
      x_out[r] = (x_in ? x_in[r] : 0.0) + dd + d1;
    }
  }
}

// BDM projection / hybrid two-level preconditioner on a general triangulation, MATRIX FREE (round 4; the assembled form is one
// CSR product with 80-entry rows, 19 KB of matrix per cell at k = 2: 45 % of a Kelvin-Helmholtz step on the level-6 disk):
//   out_K = x_K + G_K d_K,   d_K[l] = w (N_l^{K'} x_K' - N_l^K x_K)  (interior, w = 1/2),   -N_l^K x_K  (boundary),
// N_l^K = sqrt(len_l / detJ_K) Nref[l, flip] (n_x | n_y) with the FIXED edge normal (hdg_general.hpp: cell_edge_blocks): the
// moment tables are 6 reference tabulations (local edge x orientation) staged in LDS -- a lane picks its own and its
// neighbour's variant -- scaled by per-cell geometry; G_K (hdg_general.hpp: assemble_lift_tables) is the one per-cell matrix
// that remains, n2 x 3 ne doubles read coalesced across the cells of a wave.
template <int K>
__global__ __launch_bounds__(64) void k_g_lift(GGeo G, const double* __restrict__ Nref, const double* __restrict__ Gt,
                                               const double* __restrict__ xin, double* __restrict__ out) {
  constexpr int NU = Dim<K>::NU, N2 = 2 * NU, NE = Dim<K>::NE, NM = 3 * NE;
  __shared__ double Ns[6 * NE * NU];
  for (int q = threadIdx.x; q < 6 * NE * NU; q += 64) Ns[q] = Nref[q];
  __syncthreads();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= G.nc) return;
  double x[N2], d[NM];
#pragma unroll
  for (int n = 0; n < N2; n++) x[n] = xin[(long)c * N2 + n];
  const double s = G.inv_sdet[c];
#pragma unroll
  for (int l = 0; l < 3; l++) {
    const int cn = G.cnbr[3 * (long)c + l];
    const bool has = cn >= 0;
    const double nx_ = G.cenx[3 * (long)c + l], ny_ = G.ceny[3 * (long)c + l], sl = sqrt(G.celen[3 * (long)c + l]);
    const double* __restrict__ No = Ns + G.ctab[3 * (long)c + l] * NE * NU;
    const double* __restrict__ Nn = Ns + (has ? G.ntab[3 * (long)c + l] : 0) * NE * NU;
    const double so = sl * s, sn = has ? sl * G.inv_sdet[cn] : 0.0;
    double xo[NU], xb[NU];  // normal components n . x_m of the own and of the neighbour's coefficients
#pragma unroll
    for (int m = 0; m < NU; m++) {
      xo[m] = fma(nx_, x[m], ny_ * x[NU + m]);
      xb[m] = has ? fma(nx_, xin[(long)cn * N2 + m], ny_ * xin[(long)cn * N2 + NU + m]) : 0.0;
    }
#pragma unroll
    for (int a = 0; a < NE; a++) {
      double own = 0.0, nb = 0.0;
#pragma unroll
      for (int m = 0; m < NU; m++) {
        own = fma(No[a * NU + m], xo[m], own);
        nb = fma(Nn[a * NU + m], xb[m], nb);
      }
      d[l * NE + a] = has ? 0.5 * (sn * nb - so * own) : -so * own;
    }
  }
#pragma unroll
  for (int r = 0; r < N2; r++) {
    double acc = x[r];
#pragma unroll
    for (int q = 0; q < NM; q++) acc = fma(Gt[((long)r * NM + q) * G.nc + c], d[q], acc);
    out[(long)c * N2 + r] = acc;
  }
}

// Passive tracer transport on a general triangulation (common.py:110-129; k_tracer_adv of the structured engine):
//   out_i = int_K q (grad chi_i . u + chi_i div u) - sum_{interior e} int_e chi_i (un_K q_K - un_K' q_K'),
//   un_K = (u.n_K + |u.n_K|)/2, un_K' = (|u.n_K| - u.n_K)/2; u is continuous, so it is taken from this cell.
// Tracer modes = the first NP velocity modes (hierarchical basis, same scaling): the advection tabulations serve both.
template <int K>
__global__ __launch_bounds__(64) void k_g_tracer(GGeo G, const double* __restrict__ qin, const double* __restrict__ u,
                                                 double* __restrict__ out) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, N2 = 2 * NU;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= G.nc) return;
  double qc[NP], uc[N2], F[NP];
#pragma unroll
  for (int r = 0; r < NP; r++) { qc[r] = qin[(long)c * NP + r]; F[r] = 0.0; }
#pragma unroll
  for (int n = 0; n < N2; n++) uc[n] = u[(long)c * N2 + n];
  const double s = G.inv_sdet[c], dj = G.detJ[c];
  const double j00 = G.Jinv[4 * (long)c + 0] * s, j01 = G.Jinv[4 * (long)c + 1] * s, j10 = G.Jinv[4 * (long)c + 2] * s,
               j11 = G.Jinv[4 * (long)c + 3] * s;
  for (int q = 0; q < G.nqc; q++) {
    double qq = 0, ux = 0, uy = 0, dv = 0;
    for (int m = 0; m < NU; m++) {
      const double ph = G.cPhi[q * NU + m] * s, gxi = G.cGxi[q * NU + m], get = G.cGeta[q * NU + m];
      ux = fma(ph, uc[m], ux);
      uy = fma(ph, uc[NU + m], uy);
      dv = fma(j00 * gxi + j10 * get, uc[m], fma(j01 * gxi + j11 * get, uc[NU + m], dv));
      if (m < NP) qq = fma(ph, qc[m], qq);
    }
    const double w = G.cw[q] * dj * qq;
    for (int r = 0; r < NP; r++) {
      const double gxi = G.cGxi[q * NU + r], get = G.cGeta[q * NU + r];
      F[r] = fma(w, fma(j00 * gxi + j10 * get, ux, fma(j01 * gxi + j11 * get, uy, G.cPhi[q * NU + r] * s * dv)), F[r]);
    }
  }
  for (int l = 0; l < 3; l++) {
    const int cn = G.cnbr[3 * (long)c + l];
    if (cn < 0) continue;
    const double* __restrict__ Po = G.ePhi + (long)G.ctab[3 * (long)c + l] * G.nqe * NU;
    const double* __restrict__ Pn = G.ePhi + (long)G.ntab[3 * (long)c + l] * G.nqe * NU;
    const double sn = G.inv_sdet[cn], sg = G.csig[3 * (long)c + l];
    const double nxo = sg * G.cenx[3 * (long)c + l], nyo = sg * G.ceny[3 * (long)c + l], len = G.celen[3 * (long)c + l];
    double qn[NP];
#pragma unroll
    for (int r = 0; r < NP; r++) qn[r] = qin[(long)cn * NP + r];
    for (int q = 0; q < G.nqe; q++) {
      double un = 0, qk = 0, qm = 0;
      for (int m = 0; m < NU; m++) {
        const double po = Po[q * NU + m] * s;
        un = fma(po, fma(nxo, uc[m], nyo * uc[NU + m]), un);
        if (m < NP) { qk = fma(po, qc[m], qk); qm = fma(Pn[q * NU + m] * sn, qn[m], qm); }
      }
      const double a = fabs(un);
      const double flux = G.ew[q] * len * (0.5 * (un + a) * qk - 0.5 * (a - un) * qm);
      for (int r = 0; r < NP; r++) F[r] = fma(-Po[q * NU + r] * s, flux, F[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < NP; r++) out[(long)c * NP + r] = F[r];
}

// v = -b + (Q.grad) Q at one point from tabulated values / reference gradients of one cell
template <int NU>
__device__ __forceinline__ void g_point_v(const double* __restrict__ Ph, const double* __restrict__ Gxi, const double* __restrict__ Geta,
                                          int q, double s, double j00, double j01, double j10, double j11, const double* x,
                                          const double* b, double& vx, double& vy) {
  double qx = 0, qy = 0, bx = 0, by = 0, dxx = 0, dxy = 0, dyx = 0, dyy = 0;
  for (int m = 0; m < NU; m++) {
    const double ph = Ph[q * NU + m] * s, gxi = Gxi[q * NU + m], get = Geta[q * NU + m];
    const double gx = j00 * gxi + j10 * get, gy = j01 * gxi + j11 * get;
    qx = fma(ph, x[m], qx); qy = fma(ph, x[NU + m], qy);
    bx = fma(ph, b[m], bx); by = fma(ph, b[NU + m], by);
    dxx = fma(gx, x[m], dxx); dxy = fma(gy, x[m], dxy);
    dyx = fma(gx, x[NU + m], dyx); dyy = fma(gy, x[NU + m], dyy);
  }
  vx = -bx + qx * dxx + qy * dxy;
  vy = -by + qx * dyx + qy * dyy;
}

// cell part of the pressure-reconstruction right-hand side (hdg_imex.py:201-207):
//   rp = weak_divergence(psi, v),  v = -bscale * b + (Q.grad) Q   (per cell: -(grad psi, v)_K + <psi, {{v}}.n>_int)
template <int K>
__global__ __launch_bounds__(64) void k_g_precon(GGeo G, const double* __restrict__ Q, const double* __restrict__ bnew, double bscale,
                                                 double* __restrict__ rp) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, N2 = 2 * NU;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= G.nc) return;
  double x[N2], b[N2], y[NP];
#pragma unroll
  for (int n = 0; n < N2; n++) { x[n] = Q[(long)c * N2 + n]; b[n] = bscale * bnew[(long)c * N2 + n]; }
#pragma unroll
  for (int r = 0; r < NP; r++) y[r] = 0.0;
  const double s = G.inv_sdet[c], dj = G.detJ[c];
  const double j00 = G.Jinv[4 * (long)c + 0] * s, j01 = G.Jinv[4 * (long)c + 1] * s, j10 = G.Jinv[4 * (long)c + 2] * s,
               j11 = G.Jinv[4 * (long)c + 3] * s;
  for (int q = 0; q < G.nqc; q++) {
    double vx, vy;
    g_point_v<NU>(G.cPhi, G.cGxi, G.cGeta, q, s, j00, j01, j10, j11, x, b, vx, vy);
    const double w = G.cw[q] * dj;
    for (int r = 0; r < NP; r++) {
      const double gxi = G.cGxi[q * NU + r], get = G.cGeta[q * NU + r];
      y[r] -= w * ((j00 * gxi + j10 * get) * vx + (j01 * gxi + j11 * get) * vy);
    }
  }
  for (int l = 0; l < 3; l++) {
    const int cn = G.cnbr[3 * (long)c + l];
    if (cn < 0) continue;
    const long to = (long)G.ctab[3 * (long)c + l] * G.nqe * NU, tn = (long)G.ntab[3 * (long)c + l] * G.nqe * NU;
    const double sn = G.inv_sdet[cn];
    const double n00 = G.Jinv[4 * (long)cn + 0] * sn, n01 = G.Jinv[4 * (long)cn + 1] * sn, n10 = G.Jinv[4 * (long)cn + 2] * sn,
                 n11 = G.Jinv[4 * (long)cn + 3] * sn;
    const double nx_ = G.cenx[3 * (long)c + l], ny_ = G.ceny[3 * (long)c + l], sg = G.csig[3 * (long)c + l], len = G.celen[3 * (long)c + l];
    double xn[N2], bn[N2];
#pragma unroll
    for (int n = 0; n < N2; n++) { xn[n] = Q[(long)cn * N2 + n]; bn[n] = bscale * bnew[(long)cn * N2 + n]; }
    for (int q = 0; q < G.nqe; q++) {
      double ox, oy, px, py;
      g_point_v<NU>(G.ePhi + to, G.eGxi + to, G.eGeta + to, q, s, j00, j01, j10, j11, x, b, ox, oy);
      g_point_v<NU>(G.ePhi + tn, G.eGxi + tn, G.eGeta + tn, q, sn, n00, n01, n10, n11, xn, bn, px, py);
      const double vn = 0.5 * ((ox + px) * nx_ + (oy + py) * ny_);
      const double w = G.ew[q] * len * sg * vn;
      for (int r = 0; r < NP; r++) y[r] = fma(G.ePhi[to + q * NU + r] * s, w, y[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < NP; r++) rp[(long)c * NP + r] = y[r];
}

}  // namespace hdg
