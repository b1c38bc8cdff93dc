// Host-side operator tables for the HDG hot path on the structured triangular mesh.
//
// Everything the device kernels need that depends only on (degree k, mesh width h, tau, alpha)
// is computed here once, on the host, in long double where it matters:
//   * orthonormal Dubiner basis on the reference triangle, Legendre basis on edges,
//     scaled so that the PHYSICAL mass matrices are the identity (no mass solves anywhere);
//   * Gauss-Legendre / collapsed Gauss-Jacobi rules;
//   * per element shape (L = lower-left, U = upper-right triangle of a square):
//     normal-trace operators, BDM lifting (reference: src/timesteppers/common.py:91-108),
//     weak divergence / gradient blocks (hdg_imex.py:333-365), the local hybridised
//     mixed-Poisson block, its inverse, local Schur complement and back-substitution maps
//     (what firedrake.SCPC/Slate builds per cell on every solve, hdg_imex.py:128-135),
//     quadrature tabulations for the linearised advection form f_impl (hdg_imex.py:313-331).
//
// Conventions (shared with oracle/fem.py; the build's own documented choice):
//   reference triangle (0,0),(1,0),(0,1);  L: x = (x_i, y_j) + h*xi;  U: x = (x_{i+1}, y_{j+1}) - h*xi
//   local edges e = 0: eta = 0 (H edge), 1: hypotenuse (D edge), 2: xi = 0 (V edge)
//   global edge directions  H: +x,  V: +y,  D: (x_{i+1},y_j) -> (x_i,y_{j+1})
//   fixed edge normals n_e  H: (0,1), D: (1,1)/sqrt2, V: (1,0); outward sign L: (-,+,-), U: (+,-,+)
#pragma once
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <vector>

namespace hdg {

typedef long double real;
typedef std::vector<double> dvec;

inline int n_scalar(int deg) { return (deg + 1) * (deg + 2) / 2; }

// ------------------------------------------------------------------------------------------
// Jacobi polynomials, orthonormal w.r.t. (1-x)^a (1+x)^b on [-1,1]
// ------------------------------------------------------------------------------------------
inline real jacobiP(real x, real a, real b, int N) {
  real g0 = std::pow((real)2, a + b + 1) / (a + b + 1) * std::tgamma(a + 1) * std::tgamma(b + 1) /
            std::tgamma(a + b + 1);
  real p0 = 1 / std::sqrt(g0);
  if (N == 0) return p0;
  real g1 = (a + 1) * (b + 1) / (a + b + 3) * g0;
  real p1 = ((a + b + 2) * x / 2 + (a - b) / 2) / std::sqrt(g1);
  if (N == 1) return p1;
  real aold = 2 / (2 + a + b) * std::sqrt((a + 1) * (b + 1) / (a + b + 3));
  for (int i = 1; i < N; i++) {
    real h1 = 2 * i + a + b;
    real anew = 2 / (h1 + 2) *
                std::sqrt((i + 1) * (i + 1 + a + b) * (i + 1 + a) * (i + 1 + b) / (h1 + 1) / (h1 + 3));
    real bnew = -(a * a - b * b) / h1 / (h1 + 2);
    real p2 = (-aold * p0 + (x - bnew) * p1) / anew;
    p0 = p1;
    p1 = p2;
    aold = anew;
  }
  return p1;
}
inline real gradJacobiP(real x, real a, real b, int N) {
  if (N == 0) return 0;
  return std::sqrt((real)N * (N + a + b + 1)) * jacobiP(x, a + 1, b + 1, N - 1);
}

// Gauss-Jacobi rule on [-1,1] for weight (1-x)^a (1+x)^b : Newton iteration with deflation
inline void gaussJacobi(int n, real a, real b, std::vector<real>& x, std::vector<real>& w) {
  x.assign(n, 0);
  w.assign(n, 0);
  const real pi = std::acos((real)-1);
  for (int k = 0; k < n; k++) {
    real r = -std::cos((2 * k + 1) * pi / (2 * n));
    if (k > 0) r = (r + x[k - 1]) / 2;
    for (int it = 0; it < 200; it++) {
      real s = 0;
      for (int i = 0; i < k; i++) s += 1 / (r - x[i]);
      real p = jacobiP(r, a, b, n), dp = gradJacobiP(r, a, b, n);
      real delta = -p / (dp - s * p);
      r += delta;
      if (std::fabs(delta) < 1e-19L) break;
    }
    x[k] = r;
  }
  for (int k = 0; k < n; k++) {
    real s = 0;
    for (int j = 0; j < n; j++) {
      real p = jacobiP(x[k], a, b, j);
      s += p * p;
    }
    w[k] = 1 / s;
  }
}

// Gauss-Legendre on [0,1]
inline void gaussLegendre01(int n, std::vector<real>& t, std::vector<real>& w) {
  gaussJacobi(n, 0, 0, t, w);
  for (int i = 0; i < n; i++) {
    t[i] = (t[i] + 1) / 2;
    w[i] /= 2;
  }
}

// n+1 Gauss-Lobatto-Legendre points on [0,1]
inline std::vector<real> gllPoints(int n) {
  std::vector<real> p;
  if (n == 0) {
    p.push_back(0.5L);
    return p;
  }
  p.push_back(0);
  if (n >= 2) {
    std::vector<real> x, w;
    gaussJacobi(n - 1, 1, 1, x, w);
    for (int i = 0; i < n - 1; i++) p.push_back((x[i] + 1) / 2);
  }
  p.push_back(1);
  return p;
}

// recursive ("spectral") Lagrange nodes on the triangle, lattice order b outer, a inner
inline void triangleNodes(int n, int equispaced, std::vector<real>& xi, std::vector<real>& eta) {
  xi.clear();
  eta.clear();
  if (n == 0) {
    xi.push_back(1.0L / 3);
    eta.push_back(1.0L / 3);
    return;
  }
  for (int b = 0; b <= n; b++)
    for (int a = 0; a <= n - b; a++) {
      if (equispaced) {
        xi.push_back((real)a / n);
        eta.push_back((real)b / n);
        continue;
      }
      int al[3] = {n - a - b, a, b};
      std::vector<real> xn = gllPoints(n);
      real bary[3] = {0, 0, 0}, wsum = 0;
      for (int i = 0; i < 3; i++) {
        // facet opposite to barycentric index i: 1D node of degree m = n - al[i]
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
        int m = al[i1] + al[i2];
        real w = xn[n - al[i]];
        if (m == 0) continue;  // w == 0 there
        std::vector<real> xm = gllPoints(m);
        bary[i1] += w * xm[al[i1]];
        bary[i2] += w * xm[al[i2]];
        wsum += w;
      }
      xi.push_back(bary[1] / wsum);
      eta.push_back(bary[2] / wsum);
    }
}

// ------------------------------------------------------------------------------------------
// orthonormal basis on the reference triangle [0,1]^2-simplex, ordered by total degree
// (so the degree-k basis is a prefix of the degree-(k+1) basis)
// ------------------------------------------------------------------------------------------
struct Dubiner {
  int deg, n;
  std::vector<int> pi, pj;
  explicit Dubiner(int d) : deg(d), n(n_scalar(d)) {
    for (int t = 0; t <= d; t++)
      for (int i = t; i >= 0; i--) {
        pi.push_back(i);
        pj.push_back(t - i);
      }
  }
  // values and reference gradients (d/dxi, d/deta) at (xi, eta)
  void eval(real xi, real eta, real* val, real* gx, real* gy) const {
    real r = 2 * xi - 1, s = 2 * eta - 1;
    real a = (std::fabs(1 - s) > 1e-30L) ? 2 * (1 + r) / (1 - s) - 1 : -1;
    real b = s;
    for (int m = 0; m < n; m++) {
      int i = pi[m], j = pj[m];
      real fa = jacobiP(a, 0, 0, i), gb = jacobiP(b, 2 * i + 1, 0, j);
      real scale = std::pow((real)2, i + 0.5L);
      real hb = (1 - b) / 2;
      real pw = std::pow(hb, i);
      val[m] = 2 * scale * fa * gb * pw;
      if (gx) {
        real dfa = gradJacobiP(a, 0, 0, i), dgb = gradJacobiP(b, 2 * i + 1, 0, j);
        real pw1 = (i > 0) ? std::pow(hb, i - 1) : 1;
        real dr = dfa * gb;
        if (i > 0) dr *= pw1;
        real ds = dfa * gb * ((1 + a) / 2);
        if (i > 0) ds *= pw1;
        real tmp = dgb * pw;
        if (i > 0) tmp -= 0.5L * i * gb * pw1;
        ds += fa * tmp;
        // phi = 2 P(r,s), r = 2 xi - 1  ->  d/dxi = 4 dP/dr
        gx[m] = 4 * scale * dr;
        gy[m] = 4 * scale * ds;
      }
    }
  }
};

// orthonormal Legendre on [0,1] (unit length): sqrt(2a+1) P_a(2t-1)
inline void legendre01(int n, real t, real* v) {
  real x = 2 * t - 1;
  real p0 = 1, p1 = x;
  for (int a = 0; a < n; a++) {
    real p;
    if (a == 0) p = 1;
    else if (a == 1) p = x;
    else {
      p = ((2 * a - 1) * x * p1 - (a - 1) * p0) / a;
      p0 = p1;
      p1 = p;
    }
    v[a] = std::sqrt((real)(2 * a + 1)) * p;
  }
}

// dense inverse by Gauss-Jordan with partial pivoting (long double)
inline void invert(int n, std::vector<real>& A) {
  std::vector<real> B(n * n, 0);
  for (int i = 0; i < n; i++) B[i * n + i] = 1;
  for (int c = 0; c < n; c++) {
    int p = c;
    for (int r = c + 1; r < n; r++)
      if (std::fabs(A[r * n + c]) > std::fabs(A[p * n + c])) p = r;
    if (std::fabs(A[p * n + c]) < 1e-300L) throw std::runtime_error("singular local block");
    if (p != c)
      for (int k = 0; k < n; k++) {
        std::swap(A[p * n + k], A[c * n + k]);
        std::swap(B[p * n + k], B[c * n + k]);
      }
    real d = 1 / A[c * n + c];
    for (int k = 0; k < n; k++) {
      A[c * n + k] *= d;
      B[c * n + k] *= d;
    }
    for (int r = 0; r < n; r++)
      if (r != c) {
        real f = A[r * n + c];
        if (f != 0)
          for (int k = 0; k < n; k++) {
            A[r * n + k] -= f * A[c * n + k];
            B[r * n + k] -= f * B[c * n + k];
          }
      }
  }
  A = B;
}

// ------------------------------------------------------------------------------------------
// all tables for one (k, h, tau, alpha)
// ------------------------------------------------------------------------------------------
struct Tables {
  int k, nu, np, nl, ne, nx_loc;  // ne = k+2 (edge normal-trace modes), nx_loc = 2nu+np
  int nqc, nqe;                   // cell / edge quadrature sizes for the advection form
  double h, tau, alpha;
  double elen[3];
  double enx[3], eny[3];
  double sig[2][3];

  // nodal <-> modal (shape independent)
  dvec Vu, Vuinv, Vp, Vpinv, Vl, Vlinv;
  // per shape s, per local edge e
  dvec N[2][3];     // ne x 2nu   normal trace (w.r.t. fixed n_e) in edge Legendre modes
  dvec Pt[2][3];    // nl x np    pressure trace in edge Legendre modes
  dvec Lift[2][3];  // 2nu x ne   BDM lifting of a normal-trace defect
  dvec B[2];        // np x 2nu   (psi, div w)_K
  dvec D0[2];       // np x 2nu   -(grad psi, w)_K
  dvec Ainv[2], W[2], Y[2], SK[2];
  // advection quadrature tabulations
  dvec cw;                           // nqc   weights * |K|
  dvec cPhi[2], cGx[2], cGy[2];      // nqc x nu
  dvec ew[3];                        // nqe   weights * edge length
  dvec ePhi[2][3];                   // nqe x nu  (points in GLOBAL edge parametrisation)
  dvec eGx[2][3], eGy[2][3];         // nqe x nu  physical gradients on edges
  // trace block-Jacobi (edge type t = H, V, D; variant 0 both cells, 1 L cell only, 2 U cell only): nl x nl inverse
  dvec trDinv[3][3];

  Tables(int k_, double h_, double tau_, double alpha_, int equispaced_nodes)
      : k(k_), h(h_), tau(tau_), alpha(alpha_) {
    nu = n_scalar(k + 1);
    np = n_scalar(k);
    nl = k + 1;
    ne = k + 2;
    nx_loc = 2 * nu + np;
    const real rh = h;
    const real s2 = std::sqrt((real)2);
    elen[0] = h; elen[1] = (double)(s2 * rh); elen[2] = h;
    enx[0] = 0; eny[0] = 1;
    enx[1] = (double)(1 / s2); eny[1] = (double)(1 / s2);
    enx[2] = 1; eny[2] = 0;
    const double sg[2][3] = {{-1, 1, -1}, {1, -1, 1}};
    for (int s = 0; s < 2; s++) for (int e = 0; e < 3; e++) sig[s][e] = sg[s][e];

    Dubiner U(k + 1);
    std::vector<real> val(nu), gx(nu), gy(nu);

    // ---- nodal <-> modal
    {
      std::vector<real> xi, eta;
      triangleNodes(k + 1, equispaced_nodes, xi, eta);
      std::vector<real> V(nu * nu);
      for (int i = 0; i < nu; i++) {
        U.eval(xi[i], eta[i], val.data(), nullptr, nullptr);
        for (int m = 0; m < nu; m++) V[i * nu + m] = val[m] / rh;
      }
      Vu.assign(V.begin(), V.end());
      invert(nu, V);
      Vuinv.assign(V.begin(), V.end());
      triangleNodes(k, equispaced_nodes, xi, eta);
      std::vector<real> Vq(np * np);
      for (int i = 0; i < np; i++) {
        U.eval(xi[i], eta[i], val.data(), nullptr, nullptr);
        for (int m = 0; m < np; m++) Vq[i * np + m] = val[m] / rh;
      }
      Vp.assign(Vq.begin(), Vq.end());
      invert(np, Vq);
      Vpinv.assign(Vq.begin(), Vq.end());
      // trace: unit-length Legendre; the kernels apply the 1/sqrt(len) scaling per edge type
      std::vector<real> tn = equispaced_nodes ? std::vector<real>() : gllPoints(k);
      if (equispaced_nodes) {
        if (k == 0) tn.push_back(0.5L);
        else for (int i = 0; i <= k; i++) tn.push_back((real)i / k);
      }
      std::vector<real> Vt(nl * nl), lv(nl);
      for (int i = 0; i < nl; i++) {
        legendre01(nl, tn[i], lv.data());
        for (int m = 0; m < nl; m++) Vt[i * nl + m] = lv[m];
      }
      Vl.assign(Vt.begin(), Vt.end());
      invert(nl, Vt);
      Vlinv.assign(Vt.begin(), Vt.end());
    }

    // ---- reference coordinates of a point with global edge parameter t, per shape / local edge
    auto edge_ref = [](int s, int e, real t, real& xi, real& eta) {
      if (s == 0) {
        if (e == 0) { xi = t; eta = 0; }
        else if (e == 1) { xi = 1 - t; eta = t; }
        else { xi = 0; eta = t; }
      } else {
        if (e == 0) { xi = 1 - t; eta = 0; }
        else if (e == 1) { xi = t; eta = 1 - t; }
        else { xi = 0; eta = 1 - t; }
      }
    };

    // ---- exact edge rule for polynomial trace integrals
    std::vector<real> te, we;
    gaussLegendre01(k + 3, te, we);  // exact to degree 2k+5 >= (k+1)+(k+1)
    std::vector<real> lv(ne);
    for (int s = 0; s < 2; s++) {
      const real sgn = (s == 0) ? 1 : -1;  // Jacobian sign: grad_x = sgn/h grad_xi
      for (int e = 0; e < 3; e++) {
        const real len = elen[e];
        std::vector<real> Nm(ne * 2 * nu, 0), Pm(nl * np, 0);
        for (size_t q = 0; q < te.size(); q++) {
          real xi, eta;
          edge_ref(s, e, te[q], xi, eta);
          U.eval(xi, eta, val.data(), nullptr, nullptr);
          legendre01(ne, te[q], lv.data());
          for (int a = 0; a < ne; a++) {
            real chi = lv[a] / std::sqrt(len);
            for (int m = 0; m < nu; m++) {
              real v = we[q] * len * chi * val[m] / rh;
              Nm[a * 2 * nu + m] += v * enx[e];
              Nm[a * 2 * nu + nu + m] += v * eny[e];
              if (a < nl && m < np) Pm[a * np + m] += v;
            }
          }
        }
        N[s][e].assign(Nm.begin(), Nm.end());
        Pt[s][e].assign(Pm.begin(), Pm.end());
      }
      // ---- cell integrals: B, D0
      int mc = k + 2;  // exact to 2k+3 >= 2k+1
      std::vector<real> xa, wa, xb, wb;
      gaussJacobi(mc, 0, 0, xa, wa);
      gaussJacobi(mc, 1, 0, xb, wb);
      std::vector<real> Bm(np * 2 * nu, 0), Dm(np * 2 * nu, 0);
      for (int i = 0; i < mc; i++)
        for (int j = 0; j < mc; j++) {
          real eta = (xb[j] + 1) / 2, xi = (xa[i] + 1) / 2 * (1 - eta);
          real w = wa[i] * wb[j] / 8;
          U.eval(xi, eta, val.data(), gx.data(), gy.data());
          for (int r = 0; r < np; r++)
            for (int m = 0; m < nu; m++) {
              // (psi_r, d_d psi_m) = h^2 w (phi_r/h)(sgn/h^2 dphi_m) = sgn/h * w phi_r dphi_m
              Bm[r * 2 * nu + m] += sgn / rh * w * val[r] * gx[m];
              Bm[r * 2 * nu + nu + m] += sgn / rh * w * val[r] * gy[m];
              Dm[r * 2 * nu + m] -= sgn / rh * w * gx[r] * val[m];
              Dm[r * 2 * nu + nu + m] -= sgn / rh * w * gy[r] * val[m];
            }
        }
      B[s].assign(Bm.begin(), Bm.end());
      D0[s].assign(Dm.begin(), Dm.end());

      // ---- BDM lifting: invert the BDM degree-of-freedom matrix
      {
        int n2 = 2 * nu, nnd = k * (k + 2);
        std::vector<real> Z(n2 * n2, 0);
        for (int e = 0; e < 3; e++)
          for (int a = 0; a < ne; a++)
            for (int c = 0; c < n2; c++) Z[(e * ne + a) * n2 + c] = N[s][e][a * n2 + c];
        if (nnd > 0) {
          int mq = k + 2;
          gaussJacobi(mq, 0, 0, xa, wa);
          gaussJacobi(mq, 1, 0, xb, wb);
          for (int i = 0; i < mq; i++)
            for (int j = 0; j < mq; j++) {
              real eta = (xb[j] + 1) / 2, xi = (xa[i] + 1) / 2 * (1 - eta);
              real w = wa[i] * wb[j] / 8;
              U.eval(xi, eta, val.data(), nullptr, nullptr);
              real xc = xi - 1.0L / 3, yc = eta - 1.0L / 3;
              int row = 3 * ne;
              for (int d = 0; d < k; d++)
                for (int q = 0; q <= d; q++) {
                  real mono = std::pow(xc, d - q) * std::pow(yc, q);
                  for (int m = 0; m < nu; m++) {
                    Z[row * n2 + m] += w * mono * val[m];
                    Z[(row + 1) * n2 + nu + m] += w * mono * val[m];
                  }
                  row += 2;
                }
              for (int q = 0; q < k; q++) {
                real mono = std::pow(xc, k - 1 - q) * std::pow(yc, q);
                for (int m = 0; m < nu; m++) {
                  Z[row * n2 + m] += w * (-yc * mono) * val[m];
                  Z[row * n2 + nu + m] += w * (xc * mono) * val[m];
                }
                row++;
              }
            }
        }
        invert(n2, Z);
        for (int e = 0; e < 3; e++) {
          Lift[s][e].resize(n2 * ne);
          for (int r = 0; r < n2; r++)
            for (int a = 0; a < ne; a++) Lift[s][e][r * ne + a] = (double)Z[r * n2 + e * ne + a];
        }
      }

      poissonBlock(s, tau, Ainv[s], W[s], Y[s], SK[s]);
    }

    // ---- advection quadrature (cell rule exact to 3k+2, edge rule ceil((3k+4)/2) Gauss points:
    //      the rule UFL/FIAT would select for the degree-(3k+3) upwind integrand)
    {
      // cell rule exact to degree 3k+2: symmetric minimal-point rules where known (k = 1: Radon's 7-point
      // degree-5 rule; k = 2: Dunavant's 16-point degree-8 rule, refined here to a 1e-42 moment residual with
      // 40-digit Newton iterations and checked by tests), otherwise the collapsed Gauss-Jacobi rule.
      // The integrand is polynomial, so every exact rule gives the same result up to rounding.
      std::vector<real> qx, qy, qw;
      if (k == 1) {
        const real s15 = std::sqrt((real)15);
        const real a1 = (6 - s15) / 21, a2 = (6 + s15) / 21;
        const real w0 = (real)9 / 80, w1 = (155 - s15) / 2400, w2 = (155 + s15) / 2400;
        qx = {(real)1 / 3, a1, a1, 1 - 2 * a1, a2, a2, 1 - 2 * a2};
        qy = {(real)1 / 3, a1, 1 - 2 * a1, a1, a2, 1 - 2 * a2, a2};
        qw = {w0, w1, w1, w1, w2, w2, w2};
      } else if (k == 2) {
        static const long double R[16][3] = {
            {0.3333333333333333333333333L, 0.3333333333333333333333333L, 0.07215780383889358412554556L},
            {0.4592925882927231560288155L, 0.4592925882927231560288155L, 0.04754581713364231239694805L},
            {0.4592925882927231560288155L, 0.08141482341455368794236897L, 0.04754581713364231239694805L},
            {0.08141482341455368794236897L, 0.4592925882927231560288155L, 0.04754581713364231239694805L},
            {0.1705693077517602066222935L, 0.1705693077517602066222935L, 0.05160868526735912514089578L},
            {0.1705693077517602066222935L, 0.658861384496479586755413L, 0.05160868526735912514089578L},
            {0.658861384496479586755413L, 0.1705693077517602066222935L, 0.05160868526735912514089578L},
            {0.05054722831703097545842355L, 0.05054722831703097545842355L, 0.01622924881159904015546296L},
            {0.05054722831703097545842355L, 0.8989055433659380490831529L, 0.01622924881159904015546296L},
            {0.8989055433659380490831529L, 0.05054722831703097545842355L, 0.01622924881159904015546296L},
            {0.2631128296346381134217858L, 0.7284923929554042812410004L, 0.01361515708721749713242235L},
            {0.7284923929554042812410004L, 0.2631128296346381134217858L, 0.01361515708721749713242235L},
            {0.2631128296346381134217858L, 0.008394777409957605337213835L, 0.01361515708721749713242235L},
            {0.008394777409957605337213835L, 0.2631128296346381134217858L, 0.01361515708721749713242235L},
            {0.7284923929554042812410004L, 0.008394777409957605337213835L, 0.01361515708721749713242235L},
            {0.008394777409957605337213835L, 0.7284923929554042812410004L, 0.01361515708721749713242235L}};
        for (int q = 0; q < 16; q++) { qx.push_back(R[q][0]); qy.push_back(R[q][1]); qw.push_back(R[q][2]); }
      } else {
        int mc = (3 * k + 2 + 2) / 2;
        std::vector<real> xa, wa, xb, wb;
        gaussJacobi(mc, 0, 0, xa, wa);
        gaussJacobi(mc, 1, 0, xb, wb);
        for (int i = 0; i < mc; i++)
          for (int j = 0; j < mc; j++) {
            real eta = (xb[j] + 1) / 2;
            qx.push_back((xa[i] + 1) / 2 * (1 - eta));
            qy.push_back(eta);
            qw.push_back(wa[i] * wb[j] / 8);
          }
      }
      nqc = (int)qw.size();
      cw.resize(nqc);
      for (int s = 0; s < 2; s++) { cPhi[s].resize(nqc * nu); cGx[s].resize(nqc * nu); cGy[s].resize(nqc * nu); }
      for (int q = 0; q < nqc; q++) {
        cw[q] = (double)(qw[q] * rh * rh);
        U.eval(qx[q], qy[q], val.data(), gx.data(), gy.data());
        for (int s = 0; s < 2; s++) {
          real sgn = (s == 0) ? 1 : -1;
          for (int m = 0; m < nu; m++) {
            cPhi[s][q * nu + m] = (double)(val[m] / rh);
            cGx[s][q * nu + m] = (double)(sgn * gx[m] / (rh * rh));
            cGy[s][q * nu + m] = (double)(sgn * gy[m] / (rh * rh));
          }
        }
      }
      nqe = (3 * k + 4 + 1) / 2;
      std::vector<real> tq, wq;
      gaussLegendre01(nqe, tq, wq);
      for (int e = 0; e < 3; e++) {
        ew[e].resize(nqe);
        for (int q = 0; q < nqe; q++) ew[e][q] = (double)(wq[q] * elen[e]);
        for (int s = 0; s < 2; s++) {
          real sgn = (s == 0) ? 1 : -1;
          ePhi[s][e].resize(nqe * nu);
          eGx[s][e].resize(nqe * nu);
          eGy[s][e].resize(nqe * nu);
          for (int q = 0; q < nqe; q++) {
            real xi, eta;
            edge_ref(s, e, tq[q], xi, eta);
            U.eval(xi, eta, val.data(), gx.data(), gy.data());
            for (int m = 0; m < nu; m++) {
              ePhi[s][e][q * nu + m] = (double)(val[m] / rh);
              eGx[s][e][q * nu + m] = (double)(sgn * gx[m] / (rh * rh));
              eGy[s][e][q * nu + m] = (double)(sgn * gy[m] / (rh * rh));
            }
          }
        }
      }
    }

    traceBlockInverses(SK, trDinv);
  }

  // hybridised mixed Poisson local block for stabilisation parameter tau_ (hdg_imex.py:123-127):
  //   A = [[I, -B^T],[B, T]],  G = [C^T; -E^T],  H = [C, E],  S_K = -tau I - H A^{-1} G
  void poissonBlock(int s, double tau_, dvec& Ainv_, dvec& W_, dvec& Y_, dvec& SK_) const {
    int n = nx_loc, nt = 3 * nl, n2 = 2 * nu;
    std::vector<real> A(n * n, 0), G(n * nt, 0), Hm(nt * n, 0);
    for (int i = 0; i < n2; i++) A[i * n + i] = 1;
    for (int r = 0; r < np; r++)
      for (int c = 0; c < n2; c++) {
        A[(n2 + r) * n + c] = B[s][r * n2 + c];
        A[c * n + n2 + r] = -(real)B[s][r * n2 + c];
      }
    for (int e = 0; e < 3; e++)
      for (int a = 0; a < nl; a++) {
        for (int c = 0; c < n2; c++) {
          real Cv = sig[s][e] * N[s][e][a * n2 + c];
          Hm[(e * nl + a) * n + c] = Cv;
          G[c * nt + e * nl + a] = Cv;
        }
        for (int m = 0; m < np; m++) {
          real Ev = (real)tau_ * Pt[s][e][a * np + m];
          Hm[(e * nl + a) * n + n2 + m] = Ev;
          G[(n2 + m) * nt + e * nl + a] = -Ev;
          for (int m2 = 0; m2 < np; m2++)
            A[(n2 + m) * n + n2 + m2] += (real)tau_ * Pt[s][e][a * np + m] * Pt[s][e][a * np + m2];
        }
      }
    invert(n, A);
    std::vector<real> Wm(n * nt, 0), Ym(nt * n, 0), Sm(nt * nt, 0);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < nt; j++) {
        real acc = 0;
        for (int l = 0; l < n; l++) acc += A[i * n + l] * G[l * nt + j];
        Wm[i * nt + j] = acc;
      }
    for (int i = 0; i < nt; i++)
      for (int j = 0; j < n; j++) {
        real acc = 0;
        for (int l = 0; l < n; l++) acc += Hm[i * n + l] * A[l * n + j];
        Ym[i * n + j] = acc;
      }
    for (int i = 0; i < nt; i++)
      for (int j = 0; j < nt; j++) {
        real acc = (i == j) ? -(real)tau_ : 0;
        for (int l = 0; l < n; l++) acc -= Hm[i * n + l] * Wm[l * nt + j];
        Sm[i * nt + j] = acc;
      }
    Ainv_.assign(A.begin(), A.end());
    W_.assign(Wm.begin(), Wm.end());
    Y_.assign(Ym.begin(), Ym.end());
    SK_.assign(Sm.begin(), Sm.end());
  }

  // trace block-Jacobi: inverse diagonal blocks of -S per edge type (H,V,D) <-> local edge (0,2,1);
  // variant 0: both cells, 1: L cell only, 2: U cell only
  void traceBlockInverses(const dvec (&SK_)[2], dvec (&out)[3][3]) const {
    const int loc_of_type[3] = {0, 2, 1};
    int nt = 3 * nl;
    for (int t = 0; t < 3; t++) {
      int e = loc_of_type[t];
      for (int var = 0; var < 3; var++) {
        std::vector<real> Dm(nl * nl, 0);
        for (int a = 0; a < nl; a++)
          for (int b = 0; b < nl; b++) {
            real v = -(real)SK_[0][(e * nl + a) * nt + e * nl + b];
            real v2 = -(real)SK_[1][(e * nl + a) * nt + e * nl + b];
            Dm[a * nl + b] = (var == 0) ? v + v2 : (var == 1 ? v : v2);
          }
        invert(nl, Dm);
        out[t][var].assign(Dm.begin(), Dm.end());
      }
    }
  }

  // inverse of the element diagonal block of (I - gamma f_impl) that depends on geometry only:
  // I + gamma * sum_e alpha/len_e N_e^T N_e      (2nu x 2nu, per shape)
  dvec blockJacobiInverse(int s, double gamma) const {
    int n2 = 2 * nu;
    std::vector<real> D(n2 * n2, 0);
    for (int i = 0; i < n2; i++) D[i * n2 + i] = 1;
    for (int e = 0; e < 3; e++)
      for (int a = 0; a < ne; a++)
        for (int r = 0; r < n2; r++)
          for (int c = 0; c < n2; c++)
            D[r * n2 + c] += (real)gamma * alpha / elen[e] * N[s][e][a * n2 + r] * N[s][e][a * n2 + c];
    invert(n2, D);
    return dvec(D.begin(), D.end());
  }
};

}  // namespace hdg
