// TEST INFRASTRUCTURE / CPU BASELINE ONLY -- C++/OpenMP twin of the HDG-IMEX timestep on the host cores.
//
// This file is part of oracle/: it is compiled into oracle/cpu_twin/libhdg_cpu.so and loaded only by tests/,
// __graft_entry__.smoke() and the cpu_baseline leg of bench.py (oracle/cpu_twin/__init__.py).  The product
// (incompressibleeulerhdg_amd) never loads it and has no CPU fallback.
//
// What it is: the same discretisation and the same solver algorithms as the MI355X engine -- BDM projection,
// f_impl operator, hybridised mixed Poisson with static condensation, left-preconditioned GMRES with the hybrid
// two-level preconditioner Pi + Dinv (I - Pi) for the tentative velocity (hdg_imex.py:223-255), preconditioned CG with
// Chebyshev(2)/edge-block-Jacobi smoothing and a P1 geometric-multigrid V-cycle for the condensed trace system
// (hdg_imex.py:120-221), same tolerances and stopping rules -- written independently for CPUs: cell-major
// array-of-structures storage, OpenMP over mesh rows, no device code.  It stands in for the Firedrake/PETSc CPU path,
// which cannot be run here, as the timed CPU baseline, and it is the CPU leg of the parity
// tests at sizes the numpy oracle cannot reach.  Restated reference lines (paths relative to the reference's src/):
//   timesteppers/hdg_imex.py:505-660 (loop), :313-413 (forms), :450-478 (trace reconstruction, pressure shift);
//   timesteppers/common.py:91-108 (project_bdm).
// The local operator tables come from the host header csrc/hdg_tables.hpp (long double, shared with the engine);
// everything that loops over the mesh is this file's own.  PARITY UNPINNED (no Firedrake, no reference fixtures):
// the twin is pinned by the numpy oracle (tests/test_cpu_twin.py) and, through it, by the analytic vortex.
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hdg_mi355x.h"
#include "../../incompressibleeulerhdg_amd/csrc/hdg_tables.hpp"

namespace {
using hdg::dvec;
using hdg::Tables;
typedef std::vector<double> vec;

struct Twin {
  hdg_config cfg;
  int K, NU, NP, NL, NE, NX, N2, NT, s, nx, ny;
  long ncell, NH, NV, ND, nedge;
  Tables* T = nullptr;
  std::string err;
  // state (modal): velocity [cell][N2], pressure [cell][NP], trace [edge][NL]
  vec curQ, curP, curL, profile, updU, updP, updL, recP, recL, tr_one;
  std::vector<vec> stQ, stP, stL, Qstar, Qtent, brhs;
  std::vector<double> bscale;
  std::vector<int> bsep;
  std::vector<dvec> hybG[2];  // per stage: (I - Dinv) Lift_e, 3 blocks of N2 x NE per shape
  std::vector<double> hyb_gamma;
  double cheb_lmin = 0, cheb_lmax = 0, tr_nn = 0;
  std::vector<int> mg_n;
  std::vector<vec> mg_x, mg_b, mg_r;
  double it_sum[4] = {0, 0, 0, 0};
  long it_cnt[4] = {0, 0, 0, 0};

  long cid(int sh, int i, int j) const { return 2 * ((long)j * nx + i) + sh; }
  long eH(int i, int j) const { return (long)j * nx + i; }
  long eV(int i, int j) const { return NH + (long)j * (nx + 1) + i; }
  long eD(int i, int j) const { return NH + NV + (long)j * nx + i; }
  // local edge e of cell (sh, i, j): 0 -> H(i, j+sh), 1 -> D(i, j), 2 -> V(i+sh, j)
  long edge_of(int sh, int e, int i, int j) const { return e == 0 ? eH(i, j + sh) : (e == 1 ? eD(i, j) : eV(i + sh, j)); }
  // neighbour across local edge e (-1: boundary)
  long nbr(int sh, int e, int i, int j) const {
    if (sh == 0) {
      if (e == 0) return j > 0 ? cid(1, i, j - 1) : -1;
      if (e == 1) return cid(1, i, j);
      return i > 0 ? cid(1, i - 1, j) : -1;
    }
    if (e == 0) return j < ny - 1 ? cid(0, i, j + 1) : -1;
    if (e == 1) return cid(0, i, j);
    return i < nx - 1 ? cid(0, i + 1, j) : -1;
  }

  explicit Twin(const hdg_config& c) : cfg(c) {
    if (c.degree < 1 || c.degree > 4 || c.nx < 1 || c.nx != c.ny) throw std::string("unsupported configuration");
    K = c.degree; s = c.nstages; nx = c.nx; ny = c.ny;
    NU = hdg::n_scalar(K + 1); NP = hdg::n_scalar(K); NL = K + 1; NE = K + 2; N2 = 2 * NU; NX = N2 + NP; NT = 3 * NL;
    ncell = 2L * nx * ny; NH = (long)nx * (ny + 1); NV = (long)(nx + 1) * ny; ND = (long)nx * ny; nedge = NH + NV + ND;
    T = new Tables(K, 1.0 / nx, c.tau, c.alpha_penalty, c.equispaced_nodes);
    auto q = [&]() { return vec((size_t)ncell * N2, 0.0); };
    auto p = [&]() { return vec((size_t)ncell * NP, 0.0); };
    auto l = [&]() { return vec((size_t)nedge * NL, 0.0); };
    curQ = q(); curP = p(); curL = l(); profile = q(); updU = q(); updP = p(); updL = l(); recP = p(); recL = l();
    for (int i = 0; i < s; i++) { stQ.push_back(q()); stP.push_back(p()); stL.push_back(l()); Qtent.push_back(q()); }
    for (int i = 0; i < std::max(1, s - 1); i++) Qstar.push_back(q());
    for (int i = 0; i <= s; i++) { brhs.push_back(q()); bscale.push_back(1.0); bsep.push_back(0); }
    for (int sh = 0; sh < 2; sh++) hybG[sh].resize(s);
    hyb_gamma.assign(s, -1.0);
    // null-space vector: trace coefficients of the constant 1 (mode 0 = sqrt(len))
    tr_one = l();
    for (long e = 0; e < nedge; e++) tr_one[e * NL] = std::sqrt(edge_len(e));
    tr_nn = dotv(tr_one, tr_one);
    for (int n = nx;; n /= 2) {
      mg_n.push_back(n);
      size_t nv = (size_t)(n + 1) * (n + 1);
      mg_x.emplace_back(nv, 0.0); mg_b.emplace_back(nv, 0.0); mg_r.emplace_back(nv, 0.0);
      if (n % 2 != 0 || n <= 2) break;
    }
    estimate_cheb();
  }
  ~Twin() { delete T; }
  double edge_len(long e) const { return e < NH ? T->elen[0] : (e < NH + NV ? T->elen[2] : T->elen[1]); }

  // ------------------------------------------------------------------ vector helpers
  static double dotv(const vec& a, const vec& b) {
    double acc = 0.0;
    const long n = (long)a.size();
#pragma omp parallel for reduction(+ : acc) schedule(static)
    for (long i = 0; i < n; i++) acc += a[i] * b[i];
    return acc;
  }
  static void axpby(double a, const vec& x, double b, vec& y) {
    const long n = (long)x.size();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) y[i] = a * x[i] + b * y[i];
  }
  static void lincomb(const std::vector<std::pair<const vec*, double>>& t, vec& out) {
    const long n = (long)out.size();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
      double acc = 0.0;
      for (auto& pr : t) acc += pr.second * (*pr.first)[i];
      out[i] = acc;
    }
  }
  static void mv(const double* A, int nr, int nc, int ld, const double* x, double* y, double sc) {  // y += sc A x
    for (int r = 0; r < nr; r++) {
      double acc = 0.0;
      for (int c = 0; c < nc; c++) acc += A[r * ld + c] * x[c];
      y[r] += sc * acc;
    }
  }

  // ------------------------------------------------------------------ K1: edge lift (BDM projection and relatives)
  // out_K = in_K + sum_e Out_e w (N'_e in_K' - N_e in_K),  w = 1/2 interior, boundary: -Out_e N_e in_K  (common.py:91-108)
  void lift(const vec& in, vec& out, const dvec* Out0, const dvec* Out1, bool packed3) const {
    if (use_simd()) {
      switch (K) {
        case 1: lift_simd<1>(in, out, Out0, Out1, packed3); return;
        case 2: lift_simd<2>(in, out, Out0, Out1, packed3); return;
        case 3: lift_simd<3>(in, out, Out0, Out1, packed3); return;
        default: lift_simd<4>(in, out, Out0, Out1, packed3); return;
      }
    }
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
        for (int sh = 0; sh < 2; sh++) {
          const long c = cid(sh, i, j);
          const double* x = &in[c * N2];
          double y[64], d[8];
          for (int n = 0; n < N2; n++) y[n] = x[n];
          for (int e = 0; e < 3; e++) {
            for (int a = 0; a < NE; a++) d[a] = 0.0;
            mv(T->N[sh][e].data(), NE, N2, N2, x, d, -1.0);
            const long cn = nbr(sh, e, i, j);
            if (cn >= 0) {
              mv(T->N[1 - sh][e].data(), NE, N2, N2, &in[cn * N2], d, 1.0);
              for (int a = 0; a < NE; a++) d[a] *= 0.5;
            }
            const dvec* O = sh == 0 ? Out0 : Out1;
            const double* Om = packed3 ? O->data() + (size_t)e * N2 * NE : O[e].data();
            mv(Om, N2, NE, NE, d, y, 1.0);
          }
          for (int n = 0; n < N2; n++) out[c * N2 + n] = y[n];
        }
  }
  // ---- SIMD forms of the two kernels that dominate a step (round 3: "a CPU baseline worth the name").  W consecutive cells
  // of one shape and mesh row are processed together in a lane-blocked structure-of-arrays tile x[n][lane]: the shared
  // operator tables become broadcast scalars and every inner loop runs over the W lanes (#pragma omp simd, AVX-512: one
  // instruction per 8 cells).  Same arithmetic per cell as lift() / adv(), in the same order.
  static constexpr int W = 8;
  template <int KK>
  void lift_simd(const vec& in, vec& out, const dvec* Out0, const dvec* Out1, bool packed3) const {
    constexpr int NU_ = (KK + 2) * (KK + 3) / 2, N2_ = 2 * NU_, NE_ = KK + 2;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; j++)
      for (int sh = 0; sh < 2; sh++)
        for (int i0 = 0; i0 < nx; i0 += W) {
          const int w = std::min(W, nx - i0);
          alignas(64) double x[N2_][W], y[N2_][W], xn[N2_][W], d[NE_][W], has[W];
          for (int n = 0; n < N2_; n++)
            for (int l = 0; l < W; l++) x[n][l] = l < w ? in[cid(sh, i0 + l, j) * N2_ + n] : 0.0;
          for (int n = 0; n < N2_; n++)
#pragma omp simd
            for (int l = 0; l < W; l++) y[n][l] = x[n][l];
          for (int e = 0; e < 3; e++) {
            for (int l = 0; l < W; l++) {
              const long cn = l < w ? nbr(sh, e, i0 + l, j) : -1;
              has[l] = cn >= 0 ? 1.0 : 0.0;
              for (int n = 0; n < N2_; n++) xn[n][l] = cn >= 0 ? in[cn * N2_ + n] : 0.0;
            }
            const double *No = T->N[sh][e].data(), *Nn = T->N[1 - sh][e].data();
            for (int a = 0; a < NE_; a++) {
              alignas(64) double acc[W] = {0};
              for (int n = 0; n < N2_; n++) {
                const double no = No[a * N2_ + n], nn = Nn[a * N2_ + n];
#pragma omp simd
                for (int l = 0; l < W; l++) acc[l] += nn * xn[n][l] - no * x[n][l];
              }
#pragma omp simd
              for (int l = 0; l < W; l++) d[a][l] = acc[l] * (has[l] != 0.0 ? 0.5 : 1.0);
            }
            const dvec* O = sh == 0 ? Out0 : Out1;
            const double* Om = packed3 ? O->data() + (size_t)e * N2_ * NE_ : O[e].data();
            for (int n = 0; n < N2_; n++)
              for (int a = 0; a < NE_; a++) {
                const double o = Om[n * NE_ + a];
#pragma omp simd
                for (int l = 0; l < W; l++) y[n][l] += o * d[a][l];
              }
          }
          for (int l = 0; l < w; l++) {
            double* dst = &out[cid(sh, i0 + l, j) * N2_];
            for (int n = 0; n < N2_; n++) dst[n] = y[n][l];
          }
        }
  }
  template <int KK>
  void adv_simd(const vec& xin, const vec& qstar, vec& out, double gamma, const vec* bsub) const {
    constexpr int NU_ = (KK + 2) * (KK + 3) / 2, N2_ = 2 * NU_;
    const double up = cfg.flux_upwind ? 1.0 : 0.0;
    const int nqc = T->nqc, nqe = T->nqe;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; j++)
      for (int sh = 0; sh < 2; sh++)
        for (int i0 = 0; i0 < nx; i0 += W) {
          const int w = std::min(W, nx - i0);
          alignas(64) double x[N2_][W], qs[N2_][W], F[N2_][W], xn[N2_][W], has[W];
          for (int n = 0; n < N2_; n++)
            for (int l = 0; l < W; l++) {
              const long c = l < w ? cid(sh, i0 + l, j) : cid(sh, i0, j);
              x[n][l] = l < w ? xin[c * N2_ + n] : 0.0;
              qs[n][l] = l < w ? qstar[c * N2_ + n] : 0.0;
              F[n][l] = 0.0;
            }
          const double *Phi = T->cPhi[sh].data(), *Gx = T->cGx[sh].data(), *Gy = T->cGy[sh].data();
          for (int q = 0; q < nqc; q++) {
            alignas(64) double qx[W] = {0}, qy[W] = {0}, dxx[W] = {0}, dxy[W] = {0}, dyx[W] = {0}, dyy[W] = {0}, ax[W], ay[W];
            for (int m = 0; m < NU_; m++) {
              const double ph = Phi[q * NU_ + m], gx = Gx[q * NU_ + m], gy = Gy[q * NU_ + m];
#pragma omp simd
              for (int l = 0; l < W; l++) {
                qx[l] += ph * qs[m][l]; qy[l] += ph * qs[NU_ + m][l];
                dxx[l] += gx * x[m][l]; dxy[l] += gy * x[m][l]; dyx[l] += gx * x[NU_ + m][l]; dyy[l] += gy * x[NU_ + m][l];
              }
            }
            const double wq = T->cw[q];
#pragma omp simd
            for (int l = 0; l < W; l++) { ax[l] = -wq * (qx[l] * dxx[l] + qy[l] * dxy[l]); ay[l] = -wq * (qx[l] * dyx[l] + qy[l] * dyy[l]); }
            for (int m = 0; m < NU_; m++) {
              const double ph = Phi[q * NU_ + m];
#pragma omp simd
              for (int l = 0; l < W; l++) { F[m][l] += ph * ax[l]; F[NU_ + m][l] += ph * ay[l]; }
            }
          }
          for (int e = 0; e < 3; e++) {
            for (int l = 0; l < W; l++) {
              const long cn = l < w ? nbr(sh, e, i0 + l, j) : -1;
              has[l] = cn >= 0 ? 1.0 : 0.0;
              for (int n = 0; n < N2_; n++) xn[n][l] = cn >= 0 ? xin[cn * N2_ + n] : 0.0;
            }
            const double *Po = T->ePhi[sh][e].data(), *Pn = T->ePhi[1 - sh][e].data();
            const double nx_ = T->enx[e], ny_ = T->eny[e], sg = T->sig[sh][e], pen = T->alpha / T->elen[e];
            for (int q = 0; q < nqe; q++) {
              alignas(64) double ox[W] = {0}, oy[W] = {0}, bx[W] = {0}, by[W] = {0}, qn[W] = {0}, vx[W], vy[W];
              for (int m = 0; m < NU_; m++) {
                const double po = Po[q * NU_ + m], pn = Pn[q * NU_ + m];
#pragma omp simd
                for (int l = 0; l < W; l++) {
                  ox[l] += po * x[m][l]; oy[l] += po * x[NU_ + m][l];
                  qn[l] += po * (nx_ * qs[m][l] + ny_ * qs[NU_ + m][l]);
                  bx[l] += pn * xn[m][l]; by[l] += pn * xn[NU_ + m][l];
                }
              }
              const double wq = T->ew[e][q];
#pragma omp simd
              for (int l = 0; l < W; l++) {
                const double cf = has[l] * wq * (0.5 * sg * qn[l] - up * std::fabs(qn[l]));
                const double jx = ox[l] - bx[l], jy = oy[l] - by[l];
                const double jn = (jx * nx_ + jy * ny_) * pen * wq;
                vx[l] = cf * jx - jn * nx_; vy[l] = cf * jy - jn * ny_;
              }
              for (int m = 0; m < NU_; m++) {
                const double po = Po[q * NU_ + m];
#pragma omp simd
                for (int l = 0; l < W; l++) { F[m][l] += po * vx[l]; F[NU_ + m][l] += po * vy[l]; }
              }
            }
          }
          for (int l = 0; l < w; l++) {
            const long c = cid(sh, i0 + l, j);
            for (int n = 0; n < N2_; n++) {
              const double v = x[n][l] - gamma * F[n][l];
              out[c * N2_ + n] = bsub ? (*bsub)[c * N2_ + n] - v : v;
            }
          }
        }
  }
  static bool use_simd() { static const bool off = std::getenv("HDG_CPU_NO_SIMD") != nullptr; return !off; }
  void bdm(const vec& in, vec& out) const { lift(in, out, T->Lift[0], T->Lift[1], false); }

  // ------------------------------------------------------------------ K3: advection operator (hdg_imex.py:313-331)
  //   out = x - gamma F(Q*) x   (bsub: out = bsub - (x - gamma F x))
  void adv(const vec& xin, const vec& qstar, vec& out, double gamma, const vec* bsub) const {
    if (use_simd()) {
      switch (K) {
        case 1: adv_simd<1>(xin, qstar, out, gamma, bsub); return;
        case 2: adv_simd<2>(xin, qstar, out, gamma, bsub); return;
        case 3: adv_simd<3>(xin, qstar, out, gamma, bsub); return;
        default: adv_simd<4>(xin, qstar, out, gamma, bsub); return;
      }
    }
    const double up = cfg.flux_upwind ? 1.0 : 0.0;
    const int nqc = T->nqc, nqe = T->nqe;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
        for (int sh = 0; sh < 2; sh++) {
          const long c = cid(sh, i, j);
          const double *x = &xin[c * N2], *qs = &qstar[c * N2];
          double F[64];
          for (int n = 0; n < N2; n++) F[n] = 0.0;
          const double *Phi = T->cPhi[sh].data(), *Gx = T->cGx[sh].data(), *Gy = T->cGy[sh].data();
          for (int q = 0; q < nqc; q++) {
            double qx = 0, qy = 0, dxx = 0, dxy = 0, dyx = 0, dyy = 0;
            for (int m = 0; m < NU; m++) {
              const double ph = Phi[q * NU + m], gx = Gx[q * NU + m], gy = Gy[q * NU + m];
              qx += ph * qs[m]; qy += ph * qs[NU + m];
              dxx += gx * x[m]; dxy += gy * x[m]; dyx += gx * x[NU + m]; dyy += gy * x[NU + m];
            }
            const double w = T->cw[q];
            const double ax = -w * (qx * dxx + qy * dxy), ay = -w * (qx * dyx + qy * dyy);
            for (int m = 0; m < NU; m++) { F[m] += Phi[q * NU + m] * ax; F[NU + m] += Phi[q * NU + m] * ay; }
          }
          for (int e = 0; e < 3; e++) {
            const long cn = nbr(sh, e, i, j);
            const double* xn = cn >= 0 ? &xin[cn * N2] : nullptr;
            const double *Po = T->ePhi[sh][e].data(), *Pn = T->ePhi[1 - sh][e].data();
            const double nx_ = T->enx[e], ny_ = T->eny[e], sg = T->sig[sh][e], pen = T->alpha / T->elen[e];
            for (int q = 0; q < nqe; q++) {
              double ox = 0, oy = 0, bx = 0, by = 0, qn = 0;
              for (int m = 0; m < NU; m++) {
                const double po = Po[q * NU + m];
                ox += po * x[m]; oy += po * x[NU + m];
                qn += po * (nx_ * qs[m] + ny_ * qs[NU + m]);
                if (xn) { bx += Pn[q * NU + m] * xn[m]; by += Pn[q * NU + m] * xn[NU + m]; }
              }
              const double w = T->ew[e][q];
              const double cf = xn ? w * (0.5 * sg * qn - up * std::fabs(qn)) : 0.0;
              const double jx = ox - bx, jy = oy - by;
              const double jn = (jx * nx_ + jy * ny_) * pen * w;
              const double vx = cf * jx - jn * nx_, vy = cf * jy - jn * ny_;
              for (int m = 0; m < NU; m++) { F[m] += Po[q * NU + m] * vx; F[NU + m] += Po[q * NU + m] * vy; }
            }
          }
          for (int n = 0; n < N2; n++) {
            const double v = x[n] - gamma * F[n];
            out[c * N2 + n] = bsub ? (*bsub)[c * N2 + n] - v : v;
          }
        }
  }

  // ------------------------------------------------------------------ pressure gradient (hdg_imex.py:333-340)
  //   out = ca a + cb b + gamma (B^T p - sum_e sigma N_e^T lambda_e)
  void pgrad(const vec* a, double ca, const vec* b, double cb, const vec& p, const vec& lam, double gamma, vec& out) const {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
        for (int sh = 0; sh < 2; sh++) {
          const long c = cid(sh, i, j);
          double acc[64];
          const double* pp = &p[c * NP];
          const double* Bm = T->B[sh].data();
          for (int n = 0; n < N2; n++) {
            double v = 0.0;
            for (int r = 0; r < NP; r++) v += Bm[r * N2 + n] * pp[r];
            acc[n] = v;
          }
          for (int e = 0; e < 3; e++) {
            const double* l = &lam[edge_of(sh, e, i, j) * NL];
            const double* Nm = T->N[sh][e].data();
            const double sg = T->sig[sh][e];
            for (int n = 0; n < N2; n++) {
              double v = 0.0;
              for (int m = 0; m < NL; m++) v += Nm[m * N2 + n] * l[m];
              acc[n] -= sg * v;
            }
          }
          for (int n = 0; n < N2; n++)
            out[c * N2 + n] = (a ? ca * (*a)[c * N2 + n] : 0.0) + (b ? cb * (*b)[c * N2 + n] : 0.0) + gamma * acc[n];
        }
  }

  // ------------------------------------------------------------------ weak divergence (hdg_imex.py:353-365)
  void weak_div(const vec& q, double sc, vec& out) const {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
        for (int sh = 0; sh < 2; sh++) {
          const long c = cid(sh, i, j);
          double y[16], tr[8];
          for (int r = 0; r < NP; r++) y[r] = 0.0;
          mv(T->D0[sh].data(), NP, N2, N2, &q[c * N2], y, 1.0);
          for (int e = 0; e < 3; e++) {
            const long cn = nbr(sh, e, i, j);
            if (cn < 0) continue;
            for (int m = 0; m < NL; m++) tr[m] = 0.0;
            mv(T->N[sh][e].data(), NL, N2, N2, &q[c * N2], tr, 0.5);
            mv(T->N[1 - sh][e].data(), NL, N2, N2, &q[cn * N2], tr, 0.5);
            const double* Pm = T->Pt[sh][e].data();
            for (int r = 0; r < NP; r++) {
              double v = 0.0;
              for (int m = 0; m < NL; m++) v += Pm[m * NP + r] * tr[m];
              y[r] += T->sig[sh][e] * v;
            }
          }
          for (int r = 0; r < NP; r++) out[c * NP + r] = sc * y[r];
        }
  }

  // ------------------------------------------------------------------ condensed system (SCPC, hdg_imex.py:128-135)
  // visit every edge with its (up to two) cells: fn(edge, type t, var, cellL (shape 0) or -1, local edge, cellU or -1)
  template <class F>
  void for_edges(F fn) const {
#pragma omp parallel for schedule(static)
    for (int j = 0; j <= ny; j++) {
      for (int i = 0; i < nx; i++)  // H(i,j): L(i,j) e0, U(i,j-1) e0
        fn(eH(i, j), 0, j == 0 ? 1 : (j == ny ? 2 : 0), j < ny ? cid(0, i, j) : -1, 0, j > 0 ? cid(1, i, j - 1) : -1);
      if (j < ny) {
        for (int i = 0; i <= nx; i++)  // V(i,j): L(i,j) e2, U(i-1,j) e2
          fn(eV(i, j), 1, i == 0 ? 1 : (i == nx ? 2 : 0), i < nx ? cid(0, i, j) : -1, 2, i > 0 ? cid(1, i - 1, j) : -1);
        for (int i = 0; i < nx; i++) fn(eD(i, j), 2, 0, cid(0, i, j), 1, cid(1, i, j));  // D(i,j): both, e1
      }
    }
  }
  void gather_lam(long c, double* l, const vec& lam) const {  // trace values of the 3 edges of cell c in local order
    const int sh = (int)(c & 1);
    const long sq = c >> 1;
    const int i = (int)(sq % nx), j = (int)(sq / nx);
    for (int e = 0; e < 3; e++) {
      const double* src = &lam[edge_of(sh, e, i, j) * NL];
      for (int m = 0; m < NL; m++) l[e * NL + m] = src[m];
    }
  }
  // out = (-S) lam, lane-blocked (round 4: the condensed solve was the unfused scalar half of the twin, 51 % of its step): a row of
  // squares at a time, W cells per AVX-512 lane block; the products S_K lam_K of both cells of a square go to a row buffer in
  // structure-of-arrays order, the edges of the row are then assembled from the (at most two) cell contributions; the e0
  // contribution of the row below is recomputed (NL of the NT rows of S_U) so that rows stay independent (OpenMP over rows).
  template <int KK>
  void trace_apply_simd(const vec& lam, vec& out) const {
    constexpr int NL_ = KK + 1, NT_ = 3 * NL_;
    const double *SL = T->SK[0].data(), *SU = T->SK[1].data();
    const int nxp = ((nx + W - 1) / W) * W;
#pragma omp parallel
    {
      std::vector<double> yL((size_t)NT_ * nxp), yU((size_t)NT_ * nxp), yB((size_t)NL_ * nxp);
#pragma omp for schedule(static)
      for (int j = 0; j <= ny; j++) {
        // products of the cells of row j (L and U) and the e0 rows of U(., j-1)
        for (int i0 = 0; i0 < nx; i0 += W) {
          const int w = std::min(W, nx - i0);
          alignas(64) double l[NT_][W];
          auto gather = [&](int sh, int jj) {
            for (int e = 0; e < 3; e++)
              for (int q = 0; q < W; q++) {
                const double* src = &lam[edge_of(sh, e, i0 + std::min(q, w - 1), jj) * NL_];
                for (int m = 0; m < NL_; m++) l[e * NL_ + m][q] = src[m];
              }
          };
          auto product = [&](const double* S, int r0, int nr, double* dst) {  // dst[(r - r0) * nxp + i] = -(S l)_r
            for (int r = r0; r < r0 + nr; r++) {
              alignas(64) double acc[W] = {0};
              for (int c = 0; c < NT_; c++) {
                const double sv = S[r * NT_ + c];
#pragma omp simd
                for (int q = 0; q < W; q++) acc[q] -= sv * l[c][q];
              }
#pragma omp simd
              for (int q = 0; q < W; q++) dst[(size_t)(r - r0) * nxp + i0 + q] = acc[q];
            }
          };
          if (j < ny) {
            gather(0, j); product(SL, 0, NT_, yL.data());
            gather(1, j); product(SU, 0, NT_, yU.data());
          }
          if (j > 0) { gather(1, j - 1); product(SU, 0, NL_, yB.data()); }
        }
        // edges of row j: H(i, j) = L(i, j) e0 + U(i, j-1) e0;  D(i, j) = L e1 + U e1;  V(i, j) = L(i, j) e2 + U(i-1, j) e2
        for (int i = 0; i < nx; i++)
          for (int m = 0; m < NL_; m++)
            out[eH(i, j) * NL_ + m] = (j < ny ? yL[(size_t)m * nxp + i] : 0.0) + (j > 0 ? yB[(size_t)m * nxp + i] : 0.0);
        if (j < ny) {
          for (int i = 0; i < nx; i++)
            for (int m = 0; m < NL_; m++)
              out[eD(i, j) * NL_ + m] = yL[(size_t)(NL_ + m) * nxp + i] + yU[(size_t)(NL_ + m) * nxp + i];
          for (int i = 0; i <= nx; i++)
            for (int m = 0; m < NL_; m++)
              out[eV(i, j) * NL_ + m] = (i < nx ? yL[(size_t)(2 * NL_ + m) * nxp + i] : 0.0) + (i > 0 ? yU[(size_t)(2 * NL_ + m) * nxp + i - 1] : 0.0);
        }
      }
    }
  }
  // out = (-S) lam
  void trace_apply(const vec& lam, vec& out) const {
    if (use_simd()) {
      switch (K) {
        case 1: trace_apply_simd<1>(lam, out); return;
        case 2: trace_apply_simd<2>(lam, out); return;
        case 3: trace_apply_simd<3>(lam, out); return;
        default: trace_apply_simd<4>(lam, out); return;
      }
    }
    for_edges([&](long ed, int, int, long cL, int e, long cU) {
      double y[8], l[24];
      for (int m = 0; m < NL; m++) y[m] = 0.0;
      for (int side = 0; side < 2; side++) {
        const long c = side == 0 ? cL : cU;
        if (c < 0) continue;
        gather_lam(c, l, lam);
        mv(T->SK[side].data() + (size_t)e * NL * NT, NL, NT, NT, l, y, -1.0);
      }
      for (int m = 0; m < NL; m++) out[ed * NL + m] = y[m];
    });
  }
  // out_e = sum_K (Y_K r_K)_e - rl_e
  void condense(const vec* rw, const vec* rp, const vec* rl, vec& out) const {
    for_edges([&](long ed, int, int, long cL, int e, long cU) {
      double y[8];
      for (int m = 0; m < NL; m++) y[m] = 0.0;
      for (int side = 0; side < 2; side++) {
        const long c = side == 0 ? cL : cU;
        if (c < 0) continue;
        const double* Y = T->Y[side].data() + (size_t)e * NL * NX;
        if (rw) mv(Y, NL, N2, NX, &(*rw)[c * N2], y, 1.0);
        if (rp) mv(Y + N2, NL, NP, NX, &(*rp)[c * NP], y, 1.0);
      }
      for (int m = 0; m < NL; m++) out[ed * NL + m] = y[m] - (rl ? (*rl)[ed * NL + m] : 0.0);
    });
  }
  void backsub(const vec* rw, const vec* rp, const vec& lam, vec& u, vec& phi) const {
#pragma omp parallel for schedule(static)
    for (long c = 0; c < ncell; c++) {
      const int sh = (int)(c & 1);
      double y[80], l[24];
      for (int n = 0; n < NX; n++) y[n] = 0.0;
      const double* Ai = T->Ainv[sh].data();
      if (rw) mv(Ai, NX, N2, NX, &(*rw)[c * N2], y, 1.0);
      if (rp) mv(Ai + N2, NX, NP, NX, &(*rp)[c * NP], y, 1.0);
      gather_lam(c, l, lam);
      mv(T->W[sh].data(), NX, NT, NT, l, y, -1.0);
      for (int n = 0; n < N2; n++) u[c * N2 + n] = y[n];
      for (int n = 0; n < NP; n++) phi[c * NP + n] = y[N2 + n];
    }
  }
  // edge block-Jacobi: z_e = Dinv_e r_e
  void trace_dinv(const vec& r, vec& z) const {
    for_edges([&](long ed, int t, int var, long, int, long) {
      double y[8];
      for (int m = 0; m < NL; m++) y[m] = 0.0;
      mv(T->trDinv[t][var].data(), NL, NL, NL, &r[ed * NL], y, 1.0);
      for (int m = 0; m < NL; m++) z[ed * NL + m] = y[m];
    });
  }
  // trace reconstruction (hdg_imex.py:450-469)
  void trace_recon(const vec& Q, const vec& p, vec& out) const {
    const double it = 1.0 / T->tau;
    for_edges([&](long ed, int, int, long cL, int e, long cU) {
      double acc[8];
      for (int m = 0; m < NL; m++) acc[m] = 0.0;
      const double w = (cL >= 0 && cU >= 0) ? 0.5 : 1.0;
      for (int side = 0; side < 2; side++) {
        const long c = side == 0 ? cL : cU;
        if (c < 0) continue;
        mv(T->N[side][e].data(), NL, N2, N2, &Q[c * N2], acc, w * it * T->sig[side][e]);
        mv(T->Pt[side][e].data(), NL, NP, NP, &p[c * NP], acc, w);
      }
      for (int m = 0; m < NL; m++) out[ed * NL + m] = acc[m];
    });
  }
  // pressure-reconstruction right-hand side (hdg_imex.py:201-207)
  void precon_rhs(const vec& Q, const vec& bnew, double bsc, vec& rp, vec& rl) const {
    std::fill(rl.begin(), rl.end(), 0.0);
    const int nqc = T->nqc, nqe = T->nqe;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++)
        for (int sh = 0; sh < 2; sh++) {
          const long c = cid(sh, i, j);
          const double* x = &Q[c * N2];
          double b[64], y[16];
          for (int n = 0; n < N2; n++) b[n] = bsc * bnew[c * N2 + n];
          for (int r = 0; r < NP; r++) y[r] = 0.0;
          const double *Phi = T->cPhi[sh].data(), *Gx = T->cGx[sh].data(), *Gy = T->cGy[sh].data();
          auto vfield = [&](const double* ph, const double* gx, const double* gy, const double* xx, const double* bb, double& vx, double& vy) {
            double qx = 0, qy = 0, bx = 0, by = 0, dxx = 0, dxy = 0, dyx = 0, dyy = 0;
            for (int m = 0; m < NU; m++) {
              qx += ph[m] * xx[m]; qy += ph[m] * xx[NU + m]; bx += ph[m] * bb[m]; by += ph[m] * bb[NU + m];
              dxx += gx[m] * xx[m]; dxy += gy[m] * xx[m]; dyx += gx[m] * xx[NU + m]; dyy += gy[m] * xx[NU + m];
            }
            vx = -bx + qx * dxx + qy * dxy;
            vy = -by + qx * dyx + qy * dyy;
          };
          for (int q = 0; q < nqc; q++) {
            double vx, vy;
            vfield(Phi + q * NU, Gx + q * NU, Gy + q * NU, x, b, vx, vy);
            for (int r = 0; r < NP; r++) y[r] -= T->cw[q] * (Gx[q * NU + r] * vx + Gy[q * NU + r] * vy);
          }
          for (int e = 0; e < 3; e++) {
            const long cn = nbr(sh, e, i, j);
            const double nx_ = T->enx[e], ny_ = T->eny[e], sg = T->sig[sh][e];
            if (cn >= 0) {
              double bn[64];
              for (int n = 0; n < N2; n++) bn[n] = bsc * bnew[cn * N2 + n];
              for (int q = 0; q < nqe; q++) {
                double vx, vy, wx, wy;
                vfield(T->ePhi[sh][e].data() + q * NU, T->eGx[sh][e].data() + q * NU, T->eGy[sh][e].data() + q * NU, x, b, vx, vy);
                vfield(T->ePhi[1 - sh][e].data() + q * NU, T->eGx[1 - sh][e].data() + q * NU, T->eGy[1 - sh][e].data() + q * NU,
                       &Q[cn * N2], bn, wx, wy);
                const double vn = 0.5 * (vx * nx_ + vy * ny_) + 0.5 * (wx * nx_ + wy * ny_);
                const double w = T->ew[e][q] * sg * vn;
                for (int r = 0; r < NP; r++) y[r] += T->ePhi[sh][e][q * NU + r] * w;
              }
            } else {
              double tr[8];
              for (int m = 0; m < NL; m++) tr[m] = 0.0;
              mv(T->N[sh][e].data(), NL, N2, N2, b, tr, -sg);
              const long ed = edge_of(sh, e, i, j);
              for (int m = 0; m < NL; m++) rl[ed * NL + m] = tr[m];
            }
          }
          for (int r = 0; r < NP; r++) rp[c * NP + r] = y[r];
        }
  }
  // pressure / trace mean shift (hdg_imex.py:471-478)
  void shift(vec& p, vec* l) const {
    const double c0 = (1.0 / nx) / std::sqrt(2.0);
    double sum = 0.0;
#pragma omp parallel for reduction(+ : sum) schedule(static)
    for (long c = 0; c < ncell; c++) sum += p[c * NP];
    const double pbar = c0 * sum;  // domain volume 1
#pragma omp parallel for schedule(static)
    for (long c = 0; c < ncell; c++) p[c * NP] -= pbar * c0;
    if (l) {
#pragma omp parallel for schedule(static)
      for (long e = 0; e < nedge; e++) (*l)[e * NL] -= pbar * std::sqrt(edge_len(e));
    }
  }

  // ------------------------------------------------------------------ P1 multigrid (coarse space of GTMG, hdg_imex.py:97-118,139-167)
  static void p1_stencil(const vec& x, int n, int i, int j, double& diag, double& off) {
    const int st = n + 1;
    const double wx = (j == 0 || j == n) ? 0.5 : 1.0, wy = (i == 0 || i == n) ? 0.5 : 1.0;
    diag = 0; off = 0;
    if (i > 0) { diag += wx; off += wx * x[(size_t)j * st + i - 1]; }
    if (i < n) { diag += wx; off += wx * x[(size_t)j * st + i + 1]; }
    if (j > 0) { diag += wy; off += wy * x[(size_t)(j - 1) * st + i]; }
    if (j < n) { diag += wy; off += wy * x[(size_t)(j + 1) * st + i]; }
  }
  static void rbgs(int n, vec& x, const vec& b, int colour) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j <= n; j++)
      for (int i = (j + colour) & 1; i <= n; i += 2) {
        double diag, off;
        p1_stencil(x, n, i, j, diag, off);
        x[(size_t)j * (n + 1) + i] = (b[(size_t)j * (n + 1) + i] + off) / diag;
      }
  }
  void smooth(int lev, int sweeps, bool reverse) {
    for (int sw = 0; sw < sweeps; sw++) {
      rbgs(mg_n[lev], mg_x[lev], mg_b[lev], reverse ? 1 : 0);
      rbgs(mg_n[lev], mg_x[lev], mg_b[lev], reverse ? 0 : 1);
    }
  }
  void vcycle(int lev) {
    const int n = mg_n[lev], nsw = 2, ncoarse = 2;
    std::fill(mg_x[lev].begin(), mg_x[lev].end(), 0.0);
    if (lev == (int)mg_n.size() - 1) { smooth(lev, ncoarse, false); smooth(lev, ncoarse, true); return; }
    smooth(lev, nsw, false);
    vec &x = mg_x[lev], &b = mg_b[lev], &r = mg_r[lev];
#pragma omp parallel for schedule(static)
    for (int j = 0; j <= n; j++)
      for (int i = 0; i <= n; i++) {
        double diag, off;
        p1_stencil(x, n, i, j, diag, off);
        r[(size_t)j * (n + 1) + i] = b[(size_t)j * (n + 1) + i] - (diag * x[(size_t)j * (n + 1) + i] - off);
      }
    const int nc = mg_n[lev + 1], st = n + 1;
    vec& bc = mg_b[lev + 1];
#pragma omp parallel for schedule(static)
    for (int J = 0; J <= nc; J++)
      for (int I = 0; I <= nc; I++) {
        const int i = 2 * I, j = 2 * J;
        double acc = r[(size_t)j * st + i];
        if (i > 0) acc += 0.5 * r[(size_t)j * st + i - 1];
        if (i < n) acc += 0.5 * r[(size_t)j * st + i + 1];
        if (j > 0) acc += 0.5 * r[(size_t)(j - 1) * st + i];
        if (j < n) acc += 0.5 * r[(size_t)(j + 1) * st + i];
        if (i > 0 && j < n) acc += 0.5 * r[(size_t)(j + 1) * st + i - 1];
        if (i < n && j > 0) acc += 0.5 * r[(size_t)(j - 1) * st + i + 1];
        bc[(size_t)J * (nc + 1) + I] = acc;
      }
    vcycle(lev + 1);
    const vec& xc = mg_x[lev + 1];
    const int sc = nc + 1;
#pragma omp parallel for schedule(static)
    for (int j = 0; j <= n; j++)
      for (int i = 0; i <= n; i++) {
        const int I = i >> 1, J = j >> 1;
        double v;
        if (!(i & 1) && !(j & 1)) v = xc[(size_t)J * sc + I];
        else if ((i & 1) && !(j & 1)) v = 0.5 * (xc[(size_t)J * sc + I] + xc[(size_t)J * sc + I + 1]);
        else if (!(i & 1) && (j & 1)) v = 0.5 * (xc[(size_t)J * sc + I] + xc[(size_t)(J + 1) * sc + I]);
        else v = 0.5 * (xc[(size_t)J * sc + I + 1] + xc[(size_t)(J + 1) * sc + I]);
        x[(size_t)j * (n + 1) + i] += v;
      }
    smooth(lev, nsw, true);
  }
  // trace <-> P1 transfer: edge-wise L2 projection of the P1 function (hdg_imex.py:491-503) and its transpose
  void p1_to_trace_add(const vec& xc, vec& l) const {
    const double r3 = 0.57735026918962576451;
    const int st = nx + 1;
    for_edges([&](long ed, int t, int, long, int, long) {
      int i, j;
      double va, vb;
      if (t == 0) { j = (int)(ed / nx); i = (int)(ed % nx); va = xc[(size_t)j * st + i]; vb = xc[(size_t)j * st + i + 1]; }
      else if (t == 1) { const long q = ed - NH; j = (int)(q / (nx + 1)); i = (int)(q % (nx + 1)); va = xc[(size_t)j * st + i]; vb = xc[(size_t)(j + 1) * st + i]; }
      else { const long q = ed - NH - NV; j = (int)(q / nx); i = (int)(q % nx); va = xc[(size_t)j * st + i + 1]; vb = xc[(size_t)(j + 1) * st + i]; }
      const double sl = std::sqrt(edge_len(ed));
      l[ed * NL] += sl * 0.5 * (va + vb);
      l[ed * NL + 1] += sl * r3 * 0.5 * (vb - va);
    });
  }
  void trace_to_p1(const vec& l, vec& rc) const {
    const double r3 = 0.57735026918962576451;
    const int st = nx + 1;
    const double sH = 0.5 * std::sqrt(T->elen[0]), sV = 0.5 * std::sqrt(T->elen[2]), sD = 0.5 * std::sqrt(T->elen[1]);
#pragma omp parallel for schedule(static)
    for (int j = 0; j <= ny; j++)
      for (int i = 0; i <= nx; i++) {
        double acc = 0.0;
        auto a_end = [&](long e, double sc) { acc += sc * (l[e * NL] - r3 * l[e * NL + 1]); };
        auto b_end = [&](long e, double sc) { acc += sc * (l[e * NL] + r3 * l[e * NL + 1]); };
        if (i < nx) a_end(eH(i, j), sH);
        if (i > 0) b_end(eH(i - 1, j), sH);
        if (j < ny) a_end(eV(i, j), sV);
        if (j > 0) b_end(eV(i, j - 1), sV);
        if (i > 0 && j < ny) a_end(eD(i - 1, j), sD);   // D(i-1,j) starts at (x_i, y_j)
        if (i < nx && j > 0) b_end(eD(i, j - 1), sD);   // D(i,j-1) ends at (x_i, y_j)
        rc[(size_t)j * st + i] = acc;
      }
  }

  // ------------------------------------------------------------------ trace solver: PCG on -S (hdg_imex.py:128-170)
  vec ch_d, ch_r, wl, cg_r, cg_z, cg_p, cg_Ap;
  void cheb_smooth(const vec& b, vec& x, bool zero_init, int its) {
    const double theta = 0.5 * (cheb_lmax + cheb_lmin), delta = 0.5 * (cheb_lmax - cheb_lmin), sigma1 = theta / delta;
    double rho = 1.0 / sigma1;
    if (zero_init) { ch_r = b; std::fill(x.begin(), x.end(), 0.0); }
    else { trace_apply(x, ch_r); axpby(1.0, b, -1.0, ch_r); }
    trace_dinv(ch_r, ch_d);
    axpby(0.0, ch_d, 1.0 / theta, ch_d);
    axpby(1.0, ch_d, 1.0, x);
    for (int it = 1; it < its; it++) {
      trace_apply(ch_d, wl);
      axpby(-1.0, wl, 1.0, ch_r);
      const double rn = 1.0 / (2.0 * sigma1 - rho);
      trace_dinv(ch_r, wl);
      axpby(2.0 * rn / delta, wl, rn * rho, ch_d);
      axpby(1.0, ch_d, 1.0, x);
      rho = rn;
    }
  }
  void trace_precond(const vec& r, vec& z) {
    cheb_smooth(r, z, true, 2);
    trace_apply(z, wl);
    axpby(1.0, r, -1.0, wl);
    trace_to_p1(wl, mg_b[0]);
    vcycle(0);
    p1_to_trace_add(mg_x[0], z);
    cheb_smooth(r, z, false, 2);
  }
  void estimate_cheb() {
    ch_d.assign((size_t)nedge * NL, 0.0); ch_r = ch_d; wl = ch_d; cg_r = ch_d; cg_z = ch_d; cg_p = ch_d; cg_Ap = ch_d;
    unsigned long long st = 88172645463325252ULL;
    std::vector<double> nodal((size_t)nedge * NL);
    for (auto& v : nodal) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (double)(st % 2000001ULL) / 1.0e6 - 1.0; }
    l_to_modal(nodal.data(), cg_p);
    double lam = 1.0;
    for (int it = 0; it < 20; it++) {
      const double nrm = std::sqrt(dotv(cg_p, cg_p));
      axpby(0.0, cg_p, 1.0 / nrm, cg_p);
      trace_apply(cg_p, cg_Ap);
      trace_dinv(cg_Ap, cg_z);
      lam = std::sqrt(dotv(cg_z, cg_z));
      cg_p = cg_z;
    }
    cheb_lmax = 1.1 * lam;
    cheb_lmin = 0.1 * lam;
  }
  int trace_cg(vec& b, vec& x) {
    const double rtol = cfg.trace_rtol;
    axpby(-dotv(tr_one, b) / tr_nn, tr_one, 1.0, b);  // project the right-hand side onto the range
    trace_apply(x, cg_r);
    axpby(1.0, b, -1.0, cg_r);
    trace_precond(cg_r, cg_z);
    auto project = [&](vec& z) { axpby(-dotv(tr_one, z) / tr_nn, tr_one, 1.0, z); };
    project(cg_z);
    double rz = dotv(cg_r, cg_z);
    const double norm0 = std::sqrt(dotv(cg_z, cg_z));
    if (norm0 == 0.0) return 0;
    cg_p = cg_z;
    for (int its = 1;; its++) {
      trace_apply(cg_p, cg_Ap);
      const double pAp = dotv(cg_p, cg_Ap);
      if (!(pAp > 0)) throw std::string("trace CG: breakdown");
      const double alpha = rz / pAp;
      axpby(alpha, cg_p, 1.0, x);
      axpby(-alpha, cg_Ap, 1.0, cg_r);
      trace_precond(cg_r, cg_z);
      project(cg_z);
      const double rz_new = dotv(cg_r, cg_z), nrm = std::sqrt(dotv(cg_z, cg_z));
      if (nrm <= rtol * norm0) return its;
      if (its >= cfg.trace_maxit) throw std::string("trace CG reached max iterations");
      axpby(1.0, cg_z, rz_new / rz, cg_p);
      rz = rz_new;
    }
  }

  // ------------------------------------------------------------------ tentative velocity: left-preconditioned GMRES(m)
  void ensure_hyb(int idx, double gamma) {
    if (hyb_gamma[idx] == gamma) return;
    for (int sh = 0; sh < 2; sh++) {
      const dvec Di = T->blockJacobiInverse(sh, gamma);
      dvec G((size_t)3 * N2 * NE, 0.0);
      for (int e = 0; e < 3; e++)
        for (int r = 0; r < N2; r++)
          for (int q = 0; q < NE; q++) {
            double acc = T->Lift[sh][e][r * NE + q];
            for (int m = 0; m < N2; m++) acc -= Di[(size_t)r * N2 + m] * T->Lift[sh][e][m * NE + q];
            G[((size_t)e * N2 + r) * NE + q] = acc;
          }
      hybG[sh][idx] = G;
    }
    hyb_gamma[idx] = gamma;
  }
  std::vector<vec> gmV;
  vec gw, gt;
  int gmres(const vec& qstar, double gamma, int idx, const vec& b, vec& x) {
    const int m = std::max(1, cfg.gmres_restart);
    const double rtol = cfg.tent_rtol;
    if ((int)gmV.size() < m + 1) { gmV.assign(m + 1, vec(b.size(), 0.0)); gw.assign(b.size(), 0.0); gt = gw; }
    auto precond = [&](const vec& r, vec& z) { lift(r, z, &hybG[0][idx], &hybG[1][idx], true); };
    std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), gv(m + 1);
    double beta0 = -1.0;
    int its = 0;
    while (true) {
      adv(x, qstar, gt, gamma, &b);  // b - A x
      precond(gt, gw);
      const double beta = std::sqrt(dotv(gw, gw));
      if (beta0 < 0) beta0 = beta;
      if (beta <= rtol * beta0 || beta == 0.0) return its;
      axpby(1.0 / beta, gw, 0.0, gmV[0]);
      std::fill(gv.begin(), gv.end(), 0.0);
      gv[0] = beta;
      int j = 0;
      bool done = false;
      for (; j < m; j++) {
        adv(gmV[j], qstar, gt, gamma, nullptr);
        precond(gt, gw);
        {
          // classical Gram-Schmidt in two passes over memory (PETSc's default: VecMDot, then VecMAXPY), instead of the
          // j + 1 dot / axpy pairs of the modified form: h_l = (w, V_l) in ONE pass, w -= sum h_l V_l in one more
          const long nn = (long)gw.size();
          double hh[64] = {0};
          const int nl_ = j + 1;
#pragma omp parallel
          {
            double loc[64] = {0};
#pragma omp for schedule(static) nowait
            for (long q = 0; q < nn; q++) {
              const double wv = gw[q];
              for (int l = 0; l < nl_; l++) loc[l] += wv * gmV[l][q];
            }
#pragma omp critical
            for (int l = 0; l < nl_; l++) hh[l] += loc[l];
          }
          for (int l = 0; l < nl_; l++) H[(size_t)l * m + j] = hh[l];
#pragma omp parallel for schedule(static)
          for (long q = 0; q < nn; q++) {
            double wv = gw[q];
            for (int l = 0; l < nl_; l++) wv -= hh[l] * gmV[l][q];
            gw[q] = wv;
          }
        }
        const double hn = std::sqrt(dotv(gw, gw));
        H[(size_t)(j + 1) * m + j] = hn;
        if (hn > 0) axpby(1.0 / hn, gw, 0.0, gmV[j + 1]);
        for (int l = 0; l < j; l++) {
          const double a1 = H[(size_t)l * m + j], a2 = H[(size_t)(l + 1) * m + j];
          H[(size_t)l * m + j] = cs[l] * a1 + sn[l] * a2;
          H[(size_t)(l + 1) * m + j] = -sn[l] * a1 + cs[l] * a2;
        }
        const double a1 = H[(size_t)j * m + j], a2 = H[(size_t)(j + 1) * m + j], rr = std::hypot(a1, a2);
        cs[j] = rr == 0 ? 1.0 : a1 / rr;
        sn[j] = rr == 0 ? 0.0 : a2 / rr;
        H[(size_t)j * m + j] = rr;
        gv[j + 1] = -sn[j] * gv[j];
        gv[j] = cs[j] * gv[j];
        its++;
        if (std::fabs(gv[j + 1]) <= rtol * beta0 || hn == 0.0) { j++; done = true; break; }
        if (its >= cfg.tent_maxit) throw std::string("tentative-velocity GMRES reached max iterations");
      }
      std::vector<double> y(j, 0.0);
      for (int l = j - 1; l >= 0; l--) {
        double acc = gv[l];
        for (int q = l + 1; q < j; q++) acc -= H[(size_t)l * m + q] * y[q];
        y[l] = acc / H[(size_t)l * m + l];
      }
      for (int l = 0; l < j; l++) axpby(y[l], gmV[l], 1.0, x);
      if (done) return its;
    }
  }

  // ------------------------------------------------------------------ stage residuals (hdg_imex.py:367-413), mass = identity
  void residual_coeffs(int i, std::vector<double>& cq, std::vector<double>& cb) const {
    cq.assign(s, 0.0); cb.assign(s, 0.0);
    cq[0] = 1.0;
    for (int j = 1; j < i; j++) {  // column 0 never read (hdg_imex.py:377)
      const double aij = cfg.a_impl[i * s + j];
      if (aij != 0.0) {
        const double f = aij / cfg.a_impl[j * s + j];
        std::vector<double> q2, b2;
        residual_coeffs(j, q2, b2);
        cq[j] += f;
        for (int l = 0; l < s; l++) { cq[l] -= f * q2[l]; cb[l] -= f * b2[l]; }
      }
    }
    for (int j = 0; j < i; j++)
      if (cfg.a_expl[i * s + j] != 0.0) cb[j] += cfg.dt * cfg.a_expl[i * s + j];
  }
  void final_residual_coeffs(std::vector<double>& cq, std::vector<double>& cb) const {
    cq.assign(s, 0.0); cb.assign(s, 0.0);
    cq[0] = 1.0;
    for (int i = 1; i < s; i++) {
      if (cfg.b_impl[i] != 0.0) {
        const double f = cfg.b_impl[i] / cfg.a_impl[i * s + i];
        std::vector<double> q2, b2;
        residual_coeffs(i, q2, b2);
        cq[i] += f;
        for (int l = 0; l < s; l++) { cq[l] -= f * q2[l]; cb[l] -= f * b2[l]; }
      }
    }
    for (int i = 0; i < s; i++)
      if (cfg.b_expl[i] != 0.0) cb[i] += cfg.dt * cfg.b_expl[i];
  }
  const vec& bvec(int slot) const { return bsep[slot] ? profile : brhs[slot]; }
  void residual_vector(const std::vector<double>& cq, const std::vector<double>& cb, vec& out) const {
    std::vector<std::pair<const vec*, double>> t;
    for (int j = 0; j < s; j++) if (cq[j] != 0.0) t.push_back({&stQ[j], cq[j]});
    for (int j = 0; j < s; j++) if (cb[j] * bscale[j] != 0.0) t.push_back({&bvec(j), cb[j] * bscale[j]});
    lincomb(t, out);
  }

  // ------------------------------------------------------------------ one step (hdg_imex.py:551-637)
  vec wQ3, wQ4, wP1, wL1, wL2, rhs;
  // wall-clock per PerformanceLog label (logging.py:34-60; the labels of hdg_imex.py:551,564,569,575,625,631):
  // 0 timestep, 1 bdm_projection, 2 tentative_velocity_solve, 3 pressure_solve   [seconds, calls]
  double tm_sec[4] = {0, 0, 0, 0};
  long tm_cnt[4] = {0, 0, 0, 0};
  struct Tm {
    Twin& t; int k; double t0;
    Tm(Twin& t_, int k_) : t(t_), k(k_), t0(omp_get_wtime()) {}
    ~Tm() { t.tm_sec[k] += omp_get_wtime() - t0; t.tm_cnt[k]++; }
  };
  void step() {
    if (!cfg.use_projection) throw std::string("the CPU twin implements the projection method only");
    if (wQ3.empty()) { wQ3 = curQ; wQ4 = curQ; rhs = curQ; wP1 = curP; wL1 = curL; wL2 = curL; }
    Tm tstep(*this, 0);
    stQ[0] = curQ; stP[0] = curP; stL[0] = curL;
    for (int i = 1; i < s; i++) {
      { Tm tb(*this, 1); bdm(stQ[i - 1], Qstar[i - 1]); }
      const double gamma = cfg.a_impl[i * s + i] * cfg.dt;
      ensure_hyb(i, gamma);
      for (int r = 0; r < cfg.n_richardson; r++) {
        int it;
        {
          // the label covers what tentative_velocity_solve() does in the reference: right-hand side and solve
          Tm tt(*this, 2);
          std::vector<double> cq, cb;
          residual_coeffs(i, cq, cb);
          residual_vector(cq, cb, wQ3);
          adv(stQ[i], Qstar[i - 1], wQ4, gamma, nullptr);
          pgrad(&wQ3, 1.0, &wQ4, -1.0, stP[i], stL[i], gamma, rhs);
          it = gmres(Qstar[i - 1], gamma, i, rhs, Qtent[i]);
        }
        it_sum[0] += it; it_cnt[0]++;
        int itp;
        {
          Tm tp(*this, 3);
          weak_div(Qtent[i], -1.0 / gamma, wP1);
          condense(nullptr, &wP1, nullptr, wL1);
          itp = trace_cg(wL1, updL);
          backsub(nullptr, &wP1, updL, updU, updP);
        }
        it_sum[1] += itp; it_cnt[1]++;
        shift(updP, &updL);
        lincomb({{&stQ[i], 1.0}, {&Qtent[i], 1.0}, {&updU, gamma}}, stQ[i]);
        axpby(1.0, updP, 1.0, stP[i]);
        axpby(1.0, updL, 1.0, stL[i]);
      }
      shift(stP[i], &stL[i]);
    }
    {
      std::vector<double> cq, cb;
      final_residual_coeffs(cq, cb);
      residual_vector(cq, cb, wQ3);
      Tm tp(*this, 3);
      condense(&wQ3, nullptr, nullptr, wL1);
      const int it = trace_cg(wL1, curL);
      backsub(&wQ3, nullptr, curL, curQ, curP);
      it_sum[2] += it; it_cnt[2]++;
    }
    {
      Tm tp(*this, 3);
      precon_rhs(curQ, bvec(s), bscale[s], wP1, wL2);
      condense(nullptr, &wP1, &wL2, wL1);
      const int it = trace_cg(wL1, recL);
      backsub(nullptr, &wP1, recL, wQ3, recP);
      it_sum[3] += it; it_cnt[3]++;
    }
    curP = recP; curL = recL;
    shift(curP, &curL);
  }

  // ------------------------------------------------------------------ nodal <-> modal at the boundary (reference layout)
  void q_to_modal(const double* nodal, vec& modal) const {
#pragma omp parallel for schedule(static)
    for (long c = 0; c < ncell; c++)
      for (int d = 0; d < 2; d++)
        for (int m = 0; m < NU; m++) {
          double acc = 0.0;
          for (int n = 0; n < NU; n++) acc += T->Vuinv[m * NU + n] * nodal[(c * NU + n) * 2 + d];
          modal[c * N2 + d * NU + m] = acc;
        }
  }
  void q_to_nodal(const vec& modal, double* nodal) const {
#pragma omp parallel for schedule(static)
    for (long c = 0; c < ncell; c++)
      for (int d = 0; d < 2; d++)
        for (int n = 0; n < NU; n++) {
          double acc = 0.0;
          for (int m = 0; m < NU; m++) acc += T->Vu[n * NU + m] * modal[c * N2 + d * NU + m];
          nodal[(c * NU + n) * 2 + d] = acc;
        }
  }
  void p_to_modal(const double* nodal, vec& modal) const {
#pragma omp parallel for schedule(static)
    for (long c = 0; c < ncell; c++)
      for (int m = 0; m < NP; m++) {
        double acc = 0.0;
        for (int n = 0; n < NP; n++) acc += T->Vpinv[m * NP + n] * nodal[c * NP + n];
        modal[c * NP + m] = acc;
      }
  }
  void p_to_nodal(const vec& modal, double* nodal) const {
#pragma omp parallel for schedule(static)
    for (long c = 0; c < ncell; c++)
      for (int n = 0; n < NP; n++) {
        double acc = 0.0;
        for (int m = 0; m < NP; m++) acc += T->Vp[n * NP + m] * modal[c * NP + m];
        nodal[c * NP + n] = acc;
      }
  }
  void l_to_modal(const double* nodal, vec& modal) const {
#pragma omp parallel for schedule(static)
    for (long e = 0; e < nedge; e++) {
      const double sl = std::sqrt(edge_len(e));
      for (int m = 0; m < NL; m++) {
        double acc = 0.0;
        for (int n = 0; n < NL; n++) acc += T->Vlinv[m * NL + n] * nodal[e * NL + n];
        modal[e * NL + m] = sl * acc;
      }
    }
  }
  void l_to_nodal(const vec& modal, double* nodal) const {
#pragma omp parallel for schedule(static)
    for (long e = 0; e < nedge; e++) {
      const double isl = 1.0 / std::sqrt(edge_len(e));
      for (int n = 0; n < NL; n++) {
        double acc = 0.0;
        for (int m = 0; m < NL; m++) acc += T->Vl[n * NL + m] * modal[e * NL + m];
        nodal[e * NL + n] = isl * acc;
      }
    }
  }
};

struct Handle {
  Twin* t;
  std::string err;
};
std::string g_err;
}  // namespace

#define CPU_BEGIN(h)                       \
  if (!(h) || !(h)->t) return HDG_ERR_ARG; \
  Twin& E = *(h)->t;                       \
  try {
#define CPU_END(h)                                                      \
    return HDG_OK;                                                      \
  } catch (const std::string& e) { (h)->err = e; return HDG_ERR_ARG;    \
  } catch (const std::exception& e) { (h)->err = e.what(); return HDG_ERR_ARG; }

// C entry points: the subset of include/hdg_mi355x.h the CPU leg needs, same argument meaning, prefix hdgcpu_
extern "C" {
int hdgcpu_create(const hdg_config* cfg, Handle** out) {
  if (!cfg || !out) return HDG_ERR_ARG;
  try {
    *out = new Handle{new Twin(*cfg), ""};
    return HDG_OK;
  } catch (const std::string& e) { g_err = e; return HDG_ERR_ARG;
  } catch (const std::exception& e) { g_err = e.what(); return HDG_ERR_SINGULAR; }
}
int hdgcpu_destroy(Handle* h) { if (!h) return HDG_ERR_ARG; delete h->t; delete h; return HDG_OK; }
const char* hdgcpu_last_error(const Handle* h) { return h ? h->err.c_str() : g_err.c_str(); }
int hdgcpu_num_threads() { return omp_get_max_threads(); }
void hdgcpu_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
// host memory bandwidth the twin's threads reach (stream triad a = b + s c on n doubles, best of reps): GB/s, 24 B per entry
double hdgcpu_stream_triad_gbs(long n, int reps) {
  std::vector<double> a((size_t)n), b((size_t)n), c((size_t)n);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; i++) { a[i] = 0.0; b[i] = 1.0 + 1e-9 * (double)i; c[i] = 2.0; }
  double best = 0.0;
  for (int r = 0; r < reps; r++) {
    const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) a[i] = b[i] + 0.5 * c[i];
    const double dt_ = omp_get_wtime() - t0;
    if (dt_ > 0) best = std::max(best, 24.0 * (double)n / dt_ / 1e9);
  }
  return a[(size_t)(n / 2)] > 0 ? best : 0.0;
}
int hdgcpu_get_sizes(const Handle* h, long* n_cells, long* n_edges, int* n_u, int* n_p, int* n_l) {
  if (!h || !h->t) return HDG_ERR_ARG;
  *n_cells = h->t->ncell; *n_edges = h->t->nedge; *n_u = h->t->NU; *n_p = h->t->NP; *n_l = h->t->NL;
  return HDG_OK;
}
int hdgcpu_set_state(Handle* h, const double* Q, const double* p) {
  CPU_BEGIN(h)
  E.q_to_modal(Q, E.curQ);
  E.p_to_modal(p, E.curP);
  E.shift(E.curP, nullptr);  // hdg_imex.py:522
  CPU_END(h)
}
int hdgcpu_get_state(Handle* h, double* Q, double* p, double* lam) {
  CPU_BEGIN(h)
  if (Q) E.q_to_nodal(E.curQ, Q);
  if (p) E.p_to_nodal(E.curP, p);
  if (lam) E.l_to_nodal(E.curL, lam);
  CPU_END(h)
}
int hdgcpu_set_forcing_nodal(Handle* h, int slot, const double* f) {
  CPU_BEGIN(h)
  if (slot < 0 || slot > E.s) throw std::string("bad forcing slot");
  E.q_to_modal(f, E.brhs[slot]);
  E.bscale[slot] = 1.0; E.bsep[slot] = 0;
  CPU_END(h)
}
int hdgcpu_set_forcing_profile(Handle* h, const double* profile) {
  CPU_BEGIN(h)
  E.q_to_modal(profile, E.profile);
  CPU_END(h)
}
int hdgcpu_set_forcing_scale(Handle* h, int slot, double scale) {
  CPU_BEGIN(h)
  if (slot < 0 || slot > E.s) throw std::string("bad forcing slot");
  E.bscale[slot] = scale; E.bsep[slot] = 1;
  CPU_END(h)
}
int hdgcpu_reconstruct_trace(Handle* h) {
  CPU_BEGIN(h)
  E.trace_recon(E.curQ, E.curP, E.curL);
  CPU_END(h)
}
int hdgcpu_step(Handle* h) {
  CPU_BEGIN(h)
  E.step();
  CPU_END(h)
}
int hdgcpu_run_separable(Handle* h, int nsteps, const double* scales) {
  CPU_BEGIN(h)
  for (int n = 0; n < nsteps; n++) {
    for (int sl = 0; sl <= E.s; sl++) { E.bscale[sl] = scales[(long)n * (E.s + 1) + sl]; E.bsep[sl] = 1; }
    E.step();
  }
  CPU_END(h)
}
int hdgcpu_get_iteration_stats(Handle* h, double* sums, long* counts, int reset) {
  CPU_BEGIN(h)
  for (int i = 0; i < 4; i++) {
    if (sums) sums[i] = E.it_sum[i];
    if (counts) counts[i] = E.it_cnt[i];
    if (reset) { E.it_sum[i] = 0; E.it_cnt[i] = 0; }
  }
  CPU_END(h)
}
// physical coordinates of the velocity / pressure nodes in boundary numbering (what `interpolate` evaluates at)
// wall-clock seconds and call counts per label (0 timestep, 1 bdm_projection, 2 tentative_velocity_solve, 3 pressure_solve)
int hdgcpu_get_timers(Handle* h, double* seconds, long* calls, int reset) {
  CPU_BEGIN(h)
  for (int k = 0; k < 4; k++) {
    if (seconds) seconds[k] = E.tm_sec[k];
    if (calls) calls[k] = E.tm_cnt[k];
    if (reset) { E.tm_sec[k] = 0.0; E.tm_cnt[k] = 0; }
  }
  CPU_END(h)
}
int hdgcpu_node_coordinates(Handle* h, double* xq, double* xp) {
  CPU_BEGIN(h)
  for (int which = 0; which < 2; which++) {
    double* out = which == 0 ? xq : xp;
    if (!out) continue;
    std::vector<hdg::real> xi, eta;
    hdg::triangleNodes(which == 0 ? E.K + 1 : E.K, E.cfg.equispaced_nodes, xi, eta);
    const long nn = (long)xi.size();
    const double hh = 1.0 / E.nx;
    for (int j = 0; j < E.ny; j++)
      for (int i = 0; i < E.nx; i++)
        for (int sh = 0; sh < 2; sh++) {
          const long c = E.cid(sh, i, j);
          const double x0 = (sh == 0 ? i : i + 1) * hh, y0 = (sh == 0 ? j : j + 1) * hh, sg = sh == 0 ? 1.0 : -1.0;
          for (long n = 0; n < nn; n++) {
            out[(c * nn + n) * 2 + 0] = x0 + sg * hh * (double)xi[n];
            out[(c * nn + n) * 2 + 1] = y0 + sg * hh * (double)eta[n];
          }
        }
  }
  CPU_END(h)
}
// operator probes (nodal in / nodal out), for the parity tests
int hdgcpu_project_bdm_nodal(Handle* h, const double* Qin, double* Qout) {
  CPU_BEGIN(h)
  vec a(E.curQ.size()), b(E.curQ.size());
  E.q_to_modal(Qin, a);
  E.bdm(a, b);
  E.q_to_nodal(b, Qout);
  CPU_END(h)
}
int hdgcpu_apply_advection(Handle* h, const double* Qstar, const double* x, double gamma, double* y) {
  CPU_BEGIN(h)
  vec a(E.curQ.size()), b(E.curQ.size()), c(E.curQ.size());
  E.q_to_modal(Qstar, a);
  E.q_to_modal(x, b);
  E.adv(b, a, c, gamma, nullptr);
  E.q_to_nodal(c, y);
  CPU_END(h)
}
int hdgcpu_apply_trace_operator(Handle* h, const double* lam, double* out) {
  CPU_BEGIN(h)
  vec a(E.curL.size()), b(E.curL.size());
  E.l_to_modal(lam, a);
  E.trace_apply(a, b);
  E.l_to_nodal(b, out);
  CPU_END(h)
}
int hdgcpu_apply_weak_divergence(Handle* h, const double* Q, double* out_p) {
  CPU_BEGIN(h)
  vec a(E.curQ.size()), b(E.curP.size());
  E.q_to_modal(Q, a);
  E.weak_div(a, 1.0, b);
  E.p_to_nodal(b, out_p);
  CPU_END(h)
}
}  // extern "C"
