// General affine triangulations (UnitDiskMesh / Kelvin-Helmholtz, src/driver.py:184-185,
// src/model_problems.py:108-131): host side.  The structured engine lives on two congruent element shapes whose local
// matrices are shared scalars; on an arbitrary conforming triangulation every cell has its own geometry, so
//   * the topology (edges, the two cells of an edge, local edge numbers, orientation) is built here from (vertices, cells),
//   * the geometry-only local matrices of every cell (normal-trace moments, BDM lifting, divergence blocks, the local
//     hybridised mixed-Poisson block with its inverse, Schur complement and back-substitution maps: what firedrake.SCPC /
//     Slate builds per cell, hdg_imex.py:128-135) are computed once per mesh, in long double where it matters, from
//     reference-element tabulations and the per-cell Jacobian,
//   * every operator that does NOT depend on the solution is assembled into a CSR matrix in the modal (physically
//     orthonormal) bases and applied by one generic device kernel (k_csr_apply): BDM projection (common.py:91-108), weak /
//     broken divergence (hdg_imex.py:353-365, hdg_implicit.py:145), pressure gradient (:333-340), condensation, condensed
//     trace operator, back-substitution (:128-135), trace reconstruction (:450-469), the boundary term of the pressure
//     reconstruction (:201-207), edge / element block-Jacobi, nodal <-> modal conversions;
//   * the two solution-dependent forms -- the linearised advection operator f_impl (hdg_imex.py:313-331) and the cell part
//     of the pressure-reconstruction right-hand side -- are hand-written kernels with per-cell geometry (hdg_general_kernels.hpp).
// Conventions (those of oracle/fem.py TriMesh, restated; the product never imports the oracle): local edge l of a cell joins
// its vertices l and (l+1) % 3; edges are numbered in order of first appearance while walking the cells; the global direction
// of an edge runs from its lower to its higher vertex number; the fixed edge normal is the right-hand normal of that direction.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <exception>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "hdg_tables.hpp"

namespace hdg {

// host-side loop over independent cells on the available cores (the per-cell matrices are inverted in extended precision:
// 9 s on one core for the 32 768-cell disk at k = 2)
template <class F>
inline void parallel_for(int n, F&& f) {
  const unsigned nt = std::min(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if (n < 256 || nt == 1) {
    for (int i = 0; i < n; i++) f(i);
    return;
  }
  std::atomic<int> next{0};
  std::exception_ptr err;
  std::mutex mtx;
  std::vector<std::thread> workers;
  for (unsigned t = 0; t < nt; t++)
    workers.emplace_back([&]() {
      try {
        for (;;) {
          const int b = next.fetch_add(32);
          if (b >= n) break;
          for (int i = b; i < std::min(n, b + 32); i++) f(i);
        }
      } catch (...) {
        std::lock_guard<std::mutex> lk(mtx);
        if (!err) err = std::current_exception();
      }
    });
  for (auto& w : workers) w.join();
  if (err) std::rethrow_exception(err);
}

struct Csr {
  int nrows = 0, ncols = 0;
  std::vector<int> rowptr, col;
  std::vector<double> val;
};
inline std::vector<double> csr_diag_of(const Csr& A) {
  std::vector<double> d((size_t)A.nrows, 0.0);
  for (int r = 0; r < A.nrows; r++)
    for (int q = A.rowptr[(size_t)r]; q < A.rowptr[(size_t)r + 1]; q++)
      if (A.col[(size_t)q] == r) d[(size_t)r] += A.val[(size_t)q];
  return d;
}
struct CsrBuilder {
  int nrows, ncols;
  std::vector<std::vector<std::pair<int, double>>> rows;
  CsrBuilder(int r, int c) : nrows(r), ncols(c), rows((size_t)r) {}
  void add(int r, int c, double v) { if (v != 0.0) rows[(size_t)r].emplace_back(c, v); }
  Csr build() {
    Csr m;
    m.nrows = nrows; m.ncols = ncols;
    m.rowptr.assign((size_t)nrows + 1, 0);
    for (int r = 0; r < nrows; r++) {
      auto& row = rows[(size_t)r];
      std::sort(row.begin(), row.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
      size_t q = 0;
      while (q < row.size()) {
        int c = row[q].first;
        double v = 0.0;
        while (q < row.size() && row[q].first == c) { v += row[q].second; q++; }
        m.col.push_back(c);
        m.val.push_back(v);
      }
      m.rowptr[(size_t)r + 1] = (int)m.col.size();
    }
    return m;
  }
};

struct GMesh {
  int nv = 0, nc = 0, ne = 0;
  std::vector<double> X;      // nv x 2
  std::vector<int> C;         // nc x 3
  std::vector<int> cedge;     // nc x 3: edge of local edge l
  std::vector<int> cflip;     // nc x 3: 1 = the local direction (vertex l -> l+1) runs against the global direction
  std::vector<double> csig;   // nc x 3: +1 = the fixed edge normal points out of this cell
  std::vector<int> ecell;     // ne x 2: the cells of an edge (second = -1 on the boundary)
  std::vector<int> elocal;    // ne x 2: its local number in each of them
  std::vector<int> ev;        // ne x 2: vertices (lower, higher number)
  std::vector<double> elen, enx, eny;
  std::vector<double> detJ;   // |det J| = 2 |K|
  std::vector<double> J;      // nc x 4: x = v0 + J (xi, eta), row-major [[J00, J01], [J10, J11]], column c = v_{c+1} - v_0
  std::vector<double> Jinv;   // nc x 4: inverse, row-major
  double volume = 0.0;

  void build(int nv_, const double* coords, int nc_, const int* cells) {
    nv = nv_; nc = nc_;
    if (nv < 3 || nc < 1) throw std::string("general mesh: need at least one triangle");
    X.assign(coords, coords + (size_t)2 * nv);
    C.assign(cells, cells + (size_t)3 * nc);
    for (int v : C) if (v < 0 || v >= nv) throw std::string("general mesh: vertex number out of range");
    cedge.assign((size_t)3 * nc, -1); cflip.assign((size_t)3 * nc, 0); csig.assign((size_t)3 * nc, 1.0);
    detJ.resize(nc); J.resize((size_t)4 * nc); Jinv.resize((size_t)4 * nc);
    std::map<std::pair<int, int>, int> emap;
    for (int c = 0; c < nc; c++) {
      const int* v = &C[(size_t)3 * c];
      const double x0 = X[2 * v[0]], y0 = X[2 * v[0] + 1];
      const double a = X[2 * v[1]] - x0, b = X[2 * v[2]] - x0, cc = X[2 * v[1] + 1] - y0, d = X[2 * v[2] + 1] - y0;
      const double det = a * d - b * cc;
      if (!(std::fabs(det) > 0.0)) throw std::string("general mesh: degenerate triangle");
      J[4 * (size_t)c + 0] = a; J[4 * (size_t)c + 1] = b; J[4 * (size_t)c + 2] = cc; J[4 * (size_t)c + 3] = d;
      Jinv[4 * (size_t)c + 0] = d / det; Jinv[4 * (size_t)c + 1] = -b / det;
      Jinv[4 * (size_t)c + 2] = -cc / det; Jinv[4 * (size_t)c + 3] = a / det;
      detJ[c] = std::fabs(det);
      volume += 0.5 * detJ[c];
      for (int l = 0; l < 3; l++) {
        const int i0 = v[l], i1 = v[(l + 1) % 3];
        const std::pair<int, int> key(std::min(i0, i1), std::max(i0, i1));
        auto it = emap.find(key);
        int e;
        if (it == emap.end()) {
          e = ne++;
          emap[key] = e;
          ev.push_back(key.first); ev.push_back(key.second);
          ecell.push_back(c); ecell.push_back(-1);
          elocal.push_back(l); elocal.push_back(-1);
        } else {
          e = it->second;
          if (ecell[2 * (size_t)e + 1] >= 0) throw std::string("general mesh: an edge with more than two cells");
          ecell[2 * (size_t)e + 1] = c;
          elocal[2 * (size_t)e + 1] = l;
        }
        cedge[3 * (size_t)c + l] = e;
        cflip[3 * (size_t)c + l] = (i0 == key.first) ? 0 : 1;
      }
    }
    elen.resize(ne); enx.resize(ne); eny.resize(ne);
    for (int e = 0; e < ne; e++) {
      const int a = ev[2 * (size_t)e], b = ev[2 * (size_t)e + 1];
      const double tx = X[2 * b] - X[2 * a], ty = X[2 * b + 1] - X[2 * a + 1];
      const double len = std::sqrt(tx * tx + ty * ty);
      elen[e] = len; enx[e] = ty / len; eny[e] = -tx / len;
      const double mx = 0.5 * (X[2 * a] + X[2 * b]), my = 0.5 * (X[2 * a + 1] + X[2 * b + 1]);
      for (int side = 0; side < 2; side++) {
        const int c = ecell[2 * (size_t)e + side];
        if (c < 0) continue;
        const int* v = &C[(size_t)3 * c];
        const double cx = (X[2 * v[0]] + X[2 * v[1]] + X[2 * v[2]]) / 3.0, cy = (X[2 * v[0] + 1] + X[2 * v[1] + 1] + X[2 * v[2] + 1]) / 3.0;
        csig[3 * (size_t)c + elocal[2 * (size_t)e + side]] = (enx[e] * (mx - cx) + eny[e] * (my - cy)) >= 0.0 ? 1.0 : -1.0;
      }
    }
  }
  // reference coordinates of the point with GLOBAL edge parameter t on local edge l
  static void edge_ref(int l, int flip, real t, real& xi, real& eta) {
    static const real R[3][2] = {{0, 0}, {1, 0}, {0, 1}};
    const real tl = flip ? 1 - t : t;
    const int a = l, b = (l + 1) % 3;
    xi = R[a][0] + tl * (R[b][0] - R[a][0]);
    eta = R[a][1] + tl * (R[b][1] - R[a][1]);
  }
};

// ------------------------------------------------------------------------------------------
// reference tabulations shared by all cells, and the per-cell local matrices built from them
// ------------------------------------------------------------------------------------------
struct GeneralTables {
  int k, nu, np, nl, ne, n2, nx_loc, nt;
  double tau, alpha;
  int nqc, nqe;                      // advection rules (the same rules as on the structured mesh, Tables)
  // device tabulations (reference element; physical values follow from the cell's scale and inverse Jacobian)
  dvec cw, cPhi, cGxi, cGeta;        // nqc; nqc x nu
  dvec ew, ePhi, eGxi, eGeta;        // nqe; [l][flip][nqe x nu]
  // exact moment tables  Nref[l][flip][a][m] = int_0^1 leg_a(t) Dub_m(point(l, flip, t)) dt,  a < ne, m < nu
  std::vector<real> Nref;
  std::vector<real> Gref[2];         // Gref[rho][a][b] = int_ref Dub_a d_rho Dub_b   (nu x nu)
  // interior BDM functionals need the basis at a cell rule: values at the points, weights
  std::vector<real> bq_xi, bq_eta, bq_w, bq_val;  // rule exact for degree 2k+1; values nu per point
  dvec Vu, Vuinv, Vp, Vpinv, Vl, Vlinv;           // unscaled nodal <-> modal (Dubiner / Legendre at the nodes)
  std::vector<real> node_xi[2], node_eta[2];      // velocity / pressure nodes on the reference triangle
  std::vector<real> node_t;                       // trace nodes on [0, 1]

  GeneralTables(int k_, double tau_, double alpha_, int equispaced) : k(k_), tau(tau_), alpha(alpha_) {
    nu = n_scalar(k + 1); np = n_scalar(k); nl = k + 1; ne = k + 2; n2 = 2 * nu; nx_loc = n2 + np; nt = 3 * nl;
    Dubiner U(k + 1);
    std::vector<real> val(nu), gx(nu), gy(nu), lv(ne);
    // nodal <-> modal
    for (int which = 0; which < 2; which++) {
      const int deg = which == 0 ? k + 1 : k, n = which == 0 ? nu : np;
      triangleNodes(deg, equispaced, node_xi[which], node_eta[which]);
      std::vector<real> V((size_t)n * n);
      for (int i = 0; i < n; i++) {
        U.eval(node_xi[which][i], node_eta[which][i], val.data(), nullptr, nullptr);
        for (int m = 0; m < n; m++) V[(size_t)i * n + m] = val[m];
      }
      dvec& Vd = which == 0 ? Vu : Vp;
      dvec& Vi = which == 0 ? Vuinv : Vpinv;
      Vd.assign(V.begin(), V.end());
      invert(n, V);
      Vi.assign(V.begin(), V.end());
    }
    {
      node_t = equispaced ? std::vector<real>() : gllPoints(k);
      if (equispaced) { if (k == 0) node_t.push_back(0.5L); else for (int i = 0; i <= k; i++) node_t.push_back((real)i / k); }
      std::vector<real> Vt((size_t)nl * nl), l1(nl);
      for (int i = 0; i < nl; i++) {
        legendre01(nl, node_t[i], l1.data());
        for (int m = 0; m < nl; m++) Vt[(size_t)i * nl + m] = l1[m];
      }
      Vl.assign(Vt.begin(), Vt.end());
      invert(nl, Vt);
      Vlinv.assign(Vt.begin(), Vt.end());
    }
    // exact edge moments
    std::vector<real> te, we;
    gaussLegendre01(k + 3, te, we);
    Nref.assign((size_t)3 * 2 * ne * nu, 0);
    for (int l = 0; l < 3; l++)
      for (int f = 0; f < 2; f++)
        for (size_t q = 0; q < te.size(); q++) {
          real xi, eta;
          GMesh::edge_ref(l, f, te[q], xi, eta);
          U.eval(xi, eta, val.data(), nullptr, nullptr);
          legendre01(ne, te[q], lv.data());
          for (int a = 0; a < ne; a++)
            for (int m = 0; m < nu; m++) Nref[(((size_t)l * 2 + f) * ne + a) * nu + m] += we[q] * lv[a] * val[m];
        }
    // cell integrals of the reference gradients, and the rule for the interior BDM functionals
    {
      const int mc = k + 2;
      std::vector<real> xa, wa, xb, wb;
      gaussJacobi(mc, 0, 0, xa, wa);
      gaussJacobi(mc, 1, 0, xb, wb);
      Gref[0].assign((size_t)nu * nu, 0); Gref[1].assign((size_t)nu * nu, 0);
      for (int i = 0; i < mc; i++)
        for (int j = 0; j < mc; j++) {
          const real eta = (xb[j] + 1) / 2, xi = (xa[i] + 1) / 2 * (1 - eta), w = wa[i] * wb[j] / 8;
          U.eval(xi, eta, val.data(), gx.data(), gy.data());
          bq_xi.push_back(xi); bq_eta.push_back(eta); bq_w.push_back(w);
          for (int m = 0; m < nu; m++) bq_val.push_back(val[m]);
          for (int a = 0; a < nu; a++)
            for (int b = 0; b < nu; b++) {
              Gref[0][(size_t)a * nu + b] += w * val[a] * gx[b];
              Gref[1][(size_t)a * nu + b] += w * val[a] * gy[b];
            }
        }
    }
    // advection rules: the cell / edge rules of the structured engine (Tables), tabulated on the reference element
    {
      Tables T(k, 1.0, tau, alpha, equispaced);  // h = 1: values = Dubiner values, gradients = reference gradients of shape L
      nqc = T.nqc; nqe = T.nqe;
      cw = T.cw;                    // weights * 1 (sum = 1/2)
      cPhi = T.cPhi[0]; cGxi = T.cGx[0]; cGeta = T.cGy[0];
      std::vector<real> tq, wq;
      gaussLegendre01(nqe, tq, wq);
      ew.resize(nqe);
      for (int q = 0; q < nqe; q++) ew[q] = (double)wq[q];
      ePhi.assign((size_t)6 * nqe * nu, 0.0); eGxi = ePhi; eGeta = ePhi;
      for (int l = 0; l < 3; l++)
        for (int f = 0; f < 2; f++)
          for (int q = 0; q < nqe; q++) {
            real xi, eta;
            GMesh::edge_ref(l, f, tq[q], xi, eta);
            U.eval(xi, eta, val.data(), gx.data(), gy.data());
            for (int m = 0; m < nu; m++) {
              const size_t idx = (((size_t)l * 2 + f) * nqe + q) * nu + m;
              ePhi[idx] = (double)val[m]; eGxi[idx] = (double)gx[m]; eGeta[idx] = (double)gy[m];
            }
          }
    }
  }
};

// per-cell local matrices (row-major, table dof order n = d * nu + m for velocity)
struct CellLocal {
  std::vector<real> N[3];     // ne x n2   normal-trace moments w.r.t. the FIXED edge normal
  std::vector<real> Pt[3];    // nl x np
  std::vector<real> B, D0;    // np x n2
  std::vector<real> Lift[3];  // n2 x ne
  std::vector<real> Ainv, W, Y, SK;
};

inline void cell_edge_blocks(const GeneralTables& T, const GMesh& M, int c, CellLocal& L) {
  const int nu = T.nu, np = T.np, nl = T.nl, ne = T.ne, n2 = T.n2;
  const real sdet = std::sqrt((real)M.detJ[c]);
  for (int l = 0; l < 3; l++) {
    const int e = M.cedge[3 * (size_t)c + l], f = M.cflip[3 * (size_t)c + l];
    const real sc = std::sqrt((real)M.elen[e]) / sdet;
    L.N[l].assign((size_t)ne * n2, 0); L.Pt[l].assign((size_t)nl * np, 0);
    for (int a = 0; a < ne; a++)
      for (int m = 0; m < nu; m++) {
        const real v = sc * T.Nref[(((size_t)l * 2 + f) * ne + a) * nu + m];
        L.N[l][(size_t)a * n2 + m] = v * (real)M.enx[e];
        L.N[l][(size_t)a * n2 + nu + m] = v * (real)M.eny[e];
        if (a < nl && m < np) L.Pt[l][(size_t)a * np + m] = v;
      }
  }
}

// everything of one cell (the edge blocks must have been filled)
inline void cell_matrices(const GeneralTables& T, const GMesh& M, int c, double tau, CellLocal& L) {
  const int nu = T.nu, np = T.np, nl = T.nl, ne = T.ne, n2 = T.n2, k = T.k, n = T.nx_loc, nt = T.nt;
  const double* Ji = &M.Jinv[4 * (size_t)c];  // d xi_rho / d x_d = Ji[rho * 2 + d]
  L.B.assign((size_t)np * n2, 0); L.D0.assign((size_t)np * n2, 0);
  for (int r = 0; r < np; r++)
    for (int m = 0; m < nu; m++)
      for (int d = 0; d < 2; d++) {
        real b = 0, d0 = 0;
        for (int rho = 0; rho < 2; rho++) {
          b += (real)Ji[rho * 2 + d] * T.Gref[rho][(size_t)r * nu + m];
          d0 -= (real)Ji[rho * 2 + d] * T.Gref[rho][(size_t)m * nu + r];
        }
        L.B[(size_t)r * n2 + d * nu + m] = b;
        L.D0[(size_t)r * n2 + d * nu + m] = d0;
      }
  // BDM lifting: inverse of the degree-of-freedom matrix (edge normal moments + interior moments against the Nedelec
  // space of the first kind of degree k in physical coordinates centred at the centroid, [P_{k-1}]^2 + (-y, x) P~_{k-1})
  {
    const int nnd = k * (k + 2);
    std::vector<real> Z((size_t)n2 * n2, 0);
    for (int l = 0; l < 3; l++)
      for (int a = 0; a < ne; a++)
        for (int cc = 0; cc < n2; cc++) Z[(size_t)(l * ne + a) * n2 + cc] = L.N[l][(size_t)a * n2 + cc];
    if (nnd > 0) {
      const double* Jm = &M.J[4 * (size_t)c];
      for (size_t q = 0; q < T.bq_w.size(); q++) {
        const real dxi = T.bq_xi[q] - 1.0L / 3, deta = T.bq_eta[q] - 1.0L / 3;
        const real xc = (real)Jm[0] * dxi + (real)Jm[1] * deta, yc = (real)Jm[2] * dxi + (real)Jm[3] * deta;
        const real w = T.bq_w[q];
        const real* val = &T.bq_val[q * (size_t)nu];
        int row = 3 * ne;
        for (int d = 0; d < k; d++)
          for (int qq = 0; qq <= d; qq++) {
            const real mono = std::pow(xc, d - qq) * std::pow(yc, qq);
            for (int m = 0; m < nu; m++) {
              Z[(size_t)row * n2 + m] += w * mono * val[m];
              Z[(size_t)(row + 1) * n2 + nu + m] += w * mono * val[m];
            }
            row += 2;
          }
        for (int qq = 0; qq < k; qq++) {
          const real mono = std::pow(xc, k - 1 - qq) * std::pow(yc, qq);
          for (int m = 0; m < nu; m++) {
            Z[(size_t)row * n2 + m] += w * (-yc * mono) * val[m];
            Z[(size_t)row * n2 + nu + m] += w * (xc * mono) * val[m];
          }
          row++;
        }
      }
    }
    invert(n2, Z);
    for (int l = 0; l < 3; l++) {
      L.Lift[l].resize((size_t)n2 * ne);
      for (int r = 0; r < n2; r++)
        for (int a = 0; a < ne; a++) L.Lift[l][(size_t)r * ne + a] = Z[(size_t)r * n2 + l * ne + a];
    }
  }
  // hybridised mixed-Poisson block (hdg_imex.py:123-127):  A = [[I, -B^T], [B, T]],  G = [C^T; -E^T],  H = [C, E]
  {
    std::vector<real> A((size_t)n * n, 0), G((size_t)n * nt, 0), Hm((size_t)nt * n, 0);
    for (int i = 0; i < n2; i++) A[(size_t)i * n + i] = 1;
    for (int r = 0; r < np; r++)
      for (int cc = 0; cc < n2; cc++) {
        A[(size_t)(n2 + r) * n + cc] = L.B[(size_t)r * n2 + cc];
        A[(size_t)cc * n + n2 + r] = -L.B[(size_t)r * n2 + cc];
      }
    for (int l = 0; l < 3; l++) {
      const real sg = (real)M.csig[3 * (size_t)c + l];
      for (int a = 0; a < nl; a++) {
        for (int cc = 0; cc < n2; cc++) {
          const real Cv = sg * L.N[l][(size_t)a * n2 + cc];
          Hm[(size_t)(l * nl + a) * n + cc] = Cv;
          G[(size_t)cc * nt + l * nl + a] = Cv;
        }
        for (int m = 0; m < np; m++) {
          const real Ev = (real)tau * L.Pt[l][(size_t)a * np + m];
          Hm[(size_t)(l * nl + a) * n + n2 + m] = Ev;
          G[(size_t)(n2 + m) * nt + l * nl + a] = -Ev;
          for (int m2 = 0; m2 < np; m2++)
            A[(size_t)(n2 + m) * n + n2 + m2] += (real)tau * L.Pt[l][(size_t)a * np + m] * L.Pt[l][(size_t)a * np + m2];
        }
      }
    }
    invert(n, A);
    L.W.assign((size_t)n * nt, 0); L.Y.assign((size_t)nt * n, 0); L.SK.assign((size_t)nt * nt, 0);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < nt; j++) {
        real acc = 0;
        for (int q = 0; q < n; q++) acc += A[(size_t)i * n + q] * G[(size_t)q * nt + j];
        L.W[(size_t)i * nt + j] = acc;
      }
    for (int i = 0; i < nt; i++)
      for (int j = 0; j < n; j++) {
        real acc = 0;
        for (int q = 0; q < n; q++) acc += Hm[(size_t)i * n + q] * A[(size_t)q * n + j];
        L.Y[(size_t)i * n + j] = acc;
      }
    for (int i = 0; i < nt; i++)
      for (int j = 0; j < nt; j++) {
        real acc = (i == j) ? -(real)tau : 0;
        for (int q = 0; q < n; q++) acc -= Hm[(size_t)i * n + q] * L.W[(size_t)q * nt + j];
        L.SK[(size_t)i * nt + j] = acc;
      }
    L.Ainv = A;
  }
}

// ------------------------------------------------------------------------------------------
// the assembled operators.  Vector layouts (cell-major / edge-major, modal):
//   velocity  (c * n2 + d * nu + m),   pressure (c * np + r),   trace (e * nl + a)
// nodal arrays at the C boundary: velocity (c * nu + i) * 2 + d, pressure c * np + i, trace e * nl + i (nodes along the
// global edge direction)
// ------------------------------------------------------------------------------------------
struct GeneralOps {
  Csr Pi, Wdiv, Bdiv, Gp, Gl, Yw, Yp, S, Auu, Aup, Apu, App, Wu, Wp, Dtr, Rq, Rp, Rb;
  // constraint rows Gamma of the monolithic system (hdg_imex.py:342-351; k_gamma_psi / k_gamma_mu of the structured engine):
  //   psi-row  B u + Psi_p phi + Psi_l lambda,  Psi_p = tau sum_e Pt_e^T Pt_e,  Psi_l = -tau Pt_e^T       (B = Bdiv)
  //   mu-row   Mu_u u + Mu_p phi + Mu_l lambda, Mu_u = sigma N_e, Mu_p = tau Pt_e, Mu_l = -tau (cells of the edge) I
  Csr Psi_p, Psi_l, Mu_u, Mu_p, Mu_l;
  Csr Cq, Cqi, Cp, Cpi, Cl, Cli;
  dvec one_p, int_p, one_l;  // coefficients of the constant 1 (pressure, trace), integrals of the pressure basis functions
  dvec xq, xp;               // node coordinates (boundary layout)
};

inline void assemble_general(const GeneralTables& T, const GMesh& M, GeneralOps& O, std::vector<CellLocal>& loc) {
  const int nu = T.nu, np = T.np, nl = T.nl, ne = T.ne, n2 = T.n2, n = T.nx_loc, nt = T.nt;
  const int nc = M.nc, nE = M.ne;
  const long NQ = (long)nc * n2, NP = (long)nc * np, NL = (long)nE * nl;
  if (NQ >= (1L << 31) || (long)nc * n * n >= (1L << 40)) throw std::string("general mesh too large");
  loc.resize((size_t)nc);
  parallel_for(nc, [&](int c) {
    cell_edge_blocks(T, M, c, loc[(size_t)c]);
    cell_matrices(T, M, c, T.tau, loc[(size_t)c]);
  });
  CsrBuilder Pi((int)NQ, (int)NQ), Wd((int)NP, (int)NQ), Bd((int)NP, (int)NQ), Gp((int)NQ, (int)NP), Gl((int)NQ, (int)NL),
      Yw((int)NL, (int)NQ), Yp((int)NL, (int)NP), S((int)NL, (int)NL), Auu((int)NQ, (int)NQ), Aup((int)NQ, (int)NP),
      Apu((int)NP, (int)NQ), App((int)NP, (int)NP), Wu((int)NQ, (int)NL), Wp((int)NP, (int)NL), Rq((int)NL, (int)NQ),
      Rp((int)NL, (int)NP), Rb((int)NL, (int)NQ), Psp((int)NP, (int)NP), Psl((int)NP, (int)NL), Muu((int)NL, (int)NQ),
      Mup((int)NL, (int)NP), Mul((int)NL, (int)NL);
  for (int c = 0; c < nc; c++) {
    const CellLocal& L = loc[(size_t)c];
    const int q0 = c * n2, p0 = c * np;
    for (int r = 0; r < n2; r++) Pi.add(q0 + r, q0 + r, 1.0);
    for (int r = 0; r < np; r++)
      for (int cc = 0; cc < n2; cc++) {
        Wd.add(p0 + r, q0 + cc, (double)L.D0[(size_t)r * n2 + cc]);
        Bd.add(p0 + r, q0 + cc, (double)L.B[(size_t)r * n2 + cc]);
        Gp.add(q0 + cc, p0 + r, (double)L.B[(size_t)r * n2 + cc]);
      }
    for (int l = 0; l < 3; l++) {
      const int e = M.cedge[3 * (size_t)c + l];
      const int side = M.ecell[2 * (size_t)e] == c ? 0 : 1;
      const int cn = M.ecell[2 * (size_t)e + (1 - side)];
      const double sg = M.csig[3 * (size_t)c + l];
      const int l0 = e * nl;
      // BDM projection: Q*_K = Q_K + sum_e Lift_e d_e,  d_e = (N_e^{K'} Q_K' - N_e^K Q_K) / 2,  boundary: -N_e^K Q_K
      const double w = cn >= 0 ? 0.5 : 1.0;
      for (int r = 0; r < n2; r++)
        for (int cc = 0; cc < n2; cc++) {
          real own = 0, nbr = 0;
          for (int a = 0; a < ne; a++) {
            own += L.Lift[l][(size_t)r * ne + a] * L.N[l][(size_t)a * n2 + cc];
            if (cn >= 0) nbr += L.Lift[l][(size_t)r * ne + a] * loc[(size_t)cn].N[M.elocal[2 * (size_t)e + (1 - side)]][(size_t)a * n2 + cc];
          }
          Pi.add(q0 + r, q0 + cc, (double)(-w * own));
          if (cn >= 0) Pi.add(q0 + r, cn * n2 + cc, (double)(w * nbr));
        }
      // weak divergence: interior edges  sigma Pt^T (N_e^K Q_K + N_e^K' Q_K') / 2
      if (cn >= 0) {
        const CellLocal& Ln = loc[(size_t)cn];
        const int ln = M.elocal[2 * (size_t)e + (1 - side)];
        for (int r = 0; r < np; r++)
          for (int cc = 0; cc < n2; cc++) {
            real own = 0, nbr = 0;
            for (int a = 0; a < nl; a++) {
              own += L.Pt[l][(size_t)a * np + r] * L.N[l][(size_t)a * n2 + cc];
              nbr += L.Pt[l][(size_t)a * np + r] * Ln.N[ln][(size_t)a * n2 + cc];
            }
            Wd.add(p0 + r, q0 + cc, (double)(0.5 * sg * own));
            Wd.add(p0 + r, cn * n2 + cc, (double)(0.5 * sg * nbr));
          }
      }
      // pressure gradient: - sigma N_e^T lambda_e
      for (int a = 0; a < nl; a++)
        for (int cc = 0; cc < n2; cc++) Gl.add(q0 + cc, l0 + a, (double)(-sg * L.N[l][(size_t)a * n2 + cc]));
      // constraint rows of the monolithic system
      for (int a = 0; a < nl; a++) {
        for (int cc = 0; cc < n2; cc++) Muu.add(l0 + a, q0 + cc, (double)(sg * L.N[l][(size_t)a * n2 + cc]));
        for (int m = 0; m < np; m++) {
          const double ev = T.tau * (double)L.Pt[l][(size_t)a * np + m];
          Mup.add(l0 + a, p0 + m, ev);
          Psl.add(p0 + m, l0 + a, -ev);
          for (int m2 = 0; m2 < np; m2++) Psp.add(p0 + m, p0 + m2, T.tau * (double)(L.Pt[l][(size_t)a * np + m] * L.Pt[l][(size_t)a * np + m2]));
        }
        Mul.add(l0 + a, l0 + a, -T.tau);
      }
      // condensation, condensed operator, back-substitution
      for (int a = 0; a < nl; a++) {
        for (int cc = 0; cc < n2; cc++) Yw.add(l0 + a, q0 + cc, (double)L.Y[(size_t)(l * nl + a) * n + cc]);
        for (int m = 0; m < np; m++) Yp.add(l0 + a, p0 + m, (double)L.Y[(size_t)(l * nl + a) * n + n2 + m]);
        for (int l2 = 0; l2 < 3; l2++)
          for (int b = 0; b < nl; b++)
            S.add(l0 + a, M.cedge[3 * (size_t)c + l2] * nl + b, (double)(-L.SK[(size_t)(l * nl + a) * nt + l2 * nl + b]));
        for (int r = 0; r < n2; r++) Wu.add(q0 + r, l0 + a, (double)(-L.W[(size_t)r * nt + l * nl + a]));
        for (int r = 0; r < np; r++) Wp.add(p0 + r, l0 + a, (double)(-L.W[(size_t)(n2 + r) * nt + l * nl + a]));
      }
      // trace reconstruction (hdg_imex.py:450-469): interior lambda = {{p}} + [[Q.n]] / (2 tau), boundary p + Q.n / tau
      {
        const double wr = cn >= 0 ? 0.5 : 1.0;
        for (int a = 0; a < nl; a++) {
          for (int cc = 0; cc < n2; cc++) Rq.add(l0 + a, q0 + cc, (double)(wr / T.tau * sg * L.N[l][(size_t)a * n2 + cc]));
          for (int m = 0; m < np; m++) Rp.add(l0 + a, p0 + m, (double)(wr * L.Pt[l][(size_t)a * np + m]));
        }
      }
      // boundary term of the pressure reconstruction:  r_lambda = - int_{dOmega} mu n.b
      if (cn < 0)
        for (int a = 0; a < nl; a++)
          for (int cc = 0; cc < n2; cc++) Rb.add(l0 + a, q0 + cc, (double)(-sg * L.N[l][(size_t)a * n2 + cc]));
    }
    for (int r = 0; r < n2; r++) {
      for (int cc = 0; cc < n2; cc++) Auu.add(q0 + r, q0 + cc, (double)L.Ainv[(size_t)r * n + cc]);
      for (int m = 0; m < np; m++) Aup.add(q0 + r, p0 + m, (double)L.Ainv[(size_t)r * n + n2 + m]);
    }
    for (int r = 0; r < np; r++) {
      for (int cc = 0; cc < n2; cc++) Apu.add(p0 + r, q0 + cc, (double)L.Ainv[(size_t)(n2 + r) * n + cc]);
      for (int m = 0; m < np; m++) App.add(p0 + r, p0 + m, (double)L.Ainv[(size_t)(n2 + r) * n + n2 + m]);
    }
  }
  O.Pi = Pi.build(); O.Wdiv = Wd.build(); O.Bdiv = Bd.build(); O.Gp = Gp.build(); O.Gl = Gl.build();
  O.Yw = Yw.build(); O.Yp = Yp.build(); O.S = S.build(); O.Auu = Auu.build(); O.Aup = Aup.build(); O.Apu = Apu.build();
  O.App = App.build(); O.Wu = Wu.build(); O.Wp = Wp.build(); O.Rq = Rq.build(); O.Rp = Rp.build(); O.Rb = Rb.build();
  O.Psi_p = Psp.build(); O.Psi_l = Psl.build(); O.Mu_u = Muu.build(); O.Mu_p = Mup.build(); O.Mu_l = Mul.build();
  // edge block-Jacobi of -S
  {
    CsrBuilder D((int)NL, (int)NL);
    for (int e = 0; e < nE; e++) {
      std::vector<real> Dm((size_t)nl * nl, 0);
      for (int side = 0; side < 2; side++) {
        const int c = M.ecell[2 * (size_t)e + side];
        if (c < 0) continue;
        const int l = M.elocal[2 * (size_t)e + side];
        for (int a = 0; a < nl; a++)
          for (int b = 0; b < nl; b++) Dm[(size_t)a * nl + b] -= loc[(size_t)c].SK[(size_t)(l * nl + a) * nt + l * nl + b];
      }
      invert(nl, Dm);
      for (int a = 0; a < nl; a++)
        for (int b = 0; b < nl; b++) D.add(e * nl + a, e * nl + b, (double)Dm[(size_t)a * nl + b]);
    }
    O.Dtr = D.build();
  }
  // nodal <-> modal (boundary layout <-> modal layout), node coordinates, constants
  {
    CsrBuilder Cq((int)NQ, (int)NQ), Cqi((int)NQ, (int)NQ), Cp((int)NP, (int)NP), Cpi((int)NP, (int)NP), Cl((int)NL, (int)NL), Cli((int)NL, (int)NL);
    O.xq.assign((size_t)nc * nu * 2, 0.0); O.xp.assign((size_t)nc * np * 2, 0.0);
    O.one_p.assign((size_t)NP, 0.0); O.int_p.assign((size_t)NP, 0.0); O.one_l.assign((size_t)NL, 0.0);
    for (int c = 0; c < nc; c++) {
      const double sd = std::sqrt(M.detJ[c]);
      for (int m = 0; m < nu; m++)
        for (int i = 0; i < nu; i++)
          for (int d = 0; d < 2; d++) {
            Cq.add(c * n2 + d * nu + m, (c * nu + i) * 2 + d, sd * T.Vuinv[(size_t)m * nu + i]);    // modal = sqrt(detJ) V^-1 nodal
            Cqi.add((c * nu + i) * 2 + d, c * n2 + d * nu + m, T.Vu[(size_t)i * nu + m] / sd);
          }
      for (int m = 0; m < np; m++)
        for (int i = 0; i < np; i++) {
          Cp.add(c * np + m, c * np + i, sd * T.Vpinv[(size_t)m * np + i]);
          Cpi.add(c * np + i, c * np + m, T.Vp[(size_t)i * np + m] / sd);
        }
      // the constant 1 = sqrt(|K|) * phi_0 with phi_0 = Dub_0 / sqrt(detJ), Dub_0 = sqrt(2):  coefficient sqrt(detJ / 2)
      O.one_p[(size_t)c * np] = std::sqrt(M.detJ[c] / 2.0);
      O.int_p[(size_t)c * np] = std::sqrt(M.detJ[c] / 2.0);  // int_K phi_0 = |K| sqrt(2 / detJ) = sqrt(detJ / 2)
      const int* v = &M.C[(size_t)3 * c];
      for (int which = 0; which < 2; which++) {
        const int nn = which == 0 ? nu : np;
        dvec& out = which == 0 ? O.xq : O.xp;
        for (int i = 0; i < nn; i++) {
          const double xi = (double)T.node_xi[which][i], eta = (double)T.node_eta[which][i];
          out[((size_t)c * nn + i) * 2 + 0] = M.X[2 * v[0]] + M.J[4 * (size_t)c + 0] * xi + M.J[4 * (size_t)c + 1] * eta;
          out[((size_t)c * nn + i) * 2 + 1] = M.X[2 * v[0] + 1] + M.J[4 * (size_t)c + 2] * xi + M.J[4 * (size_t)c + 3] * eta;
        }
      }
    }
    for (int e = 0; e < nE; e++) {
      const double sl = std::sqrt(M.elen[e]);
      for (int m = 0; m < nl; m++)
        for (int i = 0; i < nl; i++) {
          Cl.add(e * nl + m, e * nl + i, sl * T.Vlinv[(size_t)m * nl + i]);
          Cli.add(e * nl + i, e * nl + m, T.Vl[(size_t)i * nl + m] / sl);
        }
      O.one_l[(size_t)e * nl] = sl;  // 1 = sqrt(len) * (leg_0 / sqrt(len))
    }
    O.Cq = Cq.build(); O.Cqi = Cqi.build(); O.Cp = Cp.build(); O.Cpi = Cpi.build(); O.Cl = Cl.build(); O.Cli = Cli.build();
  }
}

// ------------------------------------------------------------------------------------------
// Continuous space CG_{k+1} on a general triangulation (common.py:110-129 velocity projection for the tracer,
// callbacks.py:43-69 vorticity).  Dofs: vertices, then (p - 1) nodes per edge along the GLOBAL edge direction, then the interior
// nodes cell by cell; the node of lattice index (a, b) of a cell (triangleNodes order: for b, for a) is classified like in the
// structured engine (Engine::cg_setup) and cross-checked against the node coordinates.
//   M      consistent mass matrix  R^T M_K R,  M_K = C_K^T C_K with C_K = sqrt(detJ) V^-1 (nodal -> orthonormal modal)
//   Bp[d]  right-hand side of the L2 projection of velocity component d:  R^T C_K^T (modal coefficients)
//   Ep[d]  the projected component back as a broken modal vector:  C_K R (written into the rows of component d)
//   Vort   right-hand side of the vorticity:  -(d_x tau, Q_y) + (d_y tau, Q_x) + <tau, n_x Q_y - n_y Q_x>_{boundary}
// ------------------------------------------------------------------------------------------
struct GeneralCG {
  int ncg = 0, p = 0;
  std::vector<int> cg_of_dg;  // nc * nu
  dvec xy, diag;              // dof coordinates (ncg x 2), diagonal of M
  Csr M, Bp[2], Ep[2], Vort, R;  // R: continuous dofs -> broken nodal scalar field (nc * nu)
};

inline void assemble_cg(const GeneralTables& T, const GMesh& M, const GeneralOps& O, GeneralCG& C) {
  const int nu = T.nu, n2 = T.n2, p = T.k + 1, nc = M.nc;
  const int nint = (p - 1) * (p - 2) / 2;
  C.p = p;
  C.ncg = M.nv + M.ne * (p - 1) + nc * nint;
  C.cg_of_dg.assign((size_t)nc * nu, -1);
  C.xy.assign((size_t)C.ncg * 2, 0.0);
  std::vector<char> seen((size_t)C.ncg, 0);
  for (int c = 0; c < nc; c++) {
    int n = 0, qi = 0;
    for (int b = 0; b <= p; b++)
      for (int a = 0; a <= p - b; a++, n++) {
        const int cc = p - a - b;
        int dof;
        auto edge_dof = [&](int l, int t_loc) {
          const int e = M.cedge[3 * (size_t)c + l];
          const int t = M.cflip[3 * (size_t)c + l] ? p - t_loc : t_loc;
          return M.nv + e * (p - 1) + (t - 1);
        };
        if (a == 0 && b == 0) dof = M.C[3 * (size_t)c + 0];
        else if (a == p) dof = M.C[3 * (size_t)c + 1];
        else if (b == p) dof = M.C[3 * (size_t)c + 2];
        else if (b == 0) dof = edge_dof(0, a);       // edge v0 -> v1
        else if (cc == 0) dof = edge_dof(1, b);      // edge v1 -> v2
        else if (a == 0) dof = edge_dof(2, p - b);   // edge v2 -> v0
        else dof = M.nv + M.ne * (p - 1) + c * nint + qi++;
        C.cg_of_dg[(size_t)c * nu + n] = dof;
        const double x = O.xq[((size_t)c * nu + n) * 2], y = O.xq[((size_t)c * nu + n) * 2 + 1];
        if (seen[(size_t)dof]) {
          const double tol = 1e-9 * std::sqrt(M.detJ[c]);
          if (std::fabs(C.xy[2 * (size_t)dof] - x) > tol || std::fabs(C.xy[2 * (size_t)dof + 1] - y) > tol)
            throw std::string("continuous space: node classification does not match the node coordinates");
        } else {
          seen[(size_t)dof] = 1;
          C.xy[2 * (size_t)dof] = x; C.xy[2 * (size_t)dof + 1] = y;
        }
      }
    if (n != nu || qi != nint) throw std::string("continuous space: node count mismatch");
  }
  const long NQ = (long)nc * n2;
  CsrBuilder Mb(C.ncg, C.ncg), B0(C.ncg, (int)NQ), B1(C.ncg, (int)NQ), E0((int)NQ, C.ncg), E1((int)NQ, C.ncg), Vb(C.ncg, (int)NQ);
  for (int c = 0; c < nc; c++) {
    const double dj = M.detJ[c], sd = std::sqrt(dj);
    const int* g = &C.cg_of_dg[(size_t)c * nu];
    const int q0 = c * n2;
    for (int i = 0; i < nu; i++)
      for (int j = 0; j < nu; j++) {
        real acc = 0;
        for (int m = 0; m < nu; m++) acc += T.Vuinv[(size_t)m * nu + i] * T.Vuinv[(size_t)m * nu + j];
        Mb.add(g[i], g[j], dj * (double)acc);
      }
    for (int m = 0; m < nu; m++)
      for (int i = 0; i < nu; i++) {
        const double cv = sd * (double)T.Vuinv[(size_t)m * nu + i];
        B0.add(g[i], q0 + m, cv); B1.add(g[i], q0 + nu + m, cv);
        E0.add(q0 + m, g[i], cv); E1.add(q0 + nu + m, g[i], cv);
      }
    // vorticity: modal test functions psi_a first, then tau_i = sum_a C[a][i] psi_a
    std::vector<real> Vm((size_t)nu * n2, 0);  // row a: coefficients of (Q_x modes | Q_y modes)
    const double* Ji = &M.Jinv[4 * (size_t)c];
    for (int a = 0; a < nu; a++)
      for (int m = 0; m < nu; m++) {
        real gx = 0, gy = 0;  // int d_x psi_a psi_m,  int d_y psi_a psi_m
        for (int rho = 0; rho < 2; rho++) {
          gx += (real)Ji[rho * 2 + 0] * T.Gref[rho][(size_t)m * nu + a];  // Gref[rho][u][v] = int Dub_u d_rho Dub_v
          gy += (real)Ji[rho * 2 + 1] * T.Gref[rho][(size_t)m * nu + a];
        }
        Vm[(size_t)a * n2 + nu + m] -= gx;  // -(d_x tau) Q_y
        Vm[(size_t)a * n2 + m] += gy;       // +(d_y tau) Q_x
      }
    for (int l = 0; l < 3; l++) {
      const int e = M.cedge[3 * (size_t)c + l];
      if (M.ecell[2 * (size_t)e + 1] >= 0) continue;  // boundary edges only
      const int tabi = l * 2 + M.cflip[3 * (size_t)c + l];
      const real sg = (real)M.csig[3 * (size_t)c + l], nx = sg * (real)M.enx[e], ny = sg * (real)M.eny[e];
      const real sc = (real)M.elen[e] / (real)dj;  // weights sum to 1 along the edge; both basis functions carry 1 / sqrt(detJ)
      for (int q = 0; q < T.nqe; q++)
        for (int a = 0; a < nu; a++)
          for (int m = 0; m < nu; m++) {
            const real v = sc * (real)T.ew[(size_t)q] * (real)T.ePhi[((size_t)tabi * T.nqe + q) * nu + a] * (real)T.ePhi[((size_t)tabi * T.nqe + q) * nu + m];
            Vm[(size_t)a * n2 + nu + m] += nx * v;  // tau n_x Q_y
            Vm[(size_t)a * n2 + m] -= ny * v;       // -tau n_y Q_x
          }
    }
    for (int i = 0; i < nu; i++)
      for (int cc = 0; cc < n2; cc++) {
        real acc = 0;
        for (int a = 0; a < nu; a++) acc += (real)sd * T.Vuinv[(size_t)a * nu + i] * Vm[(size_t)a * n2 + cc];
        Vb.add(g[i], q0 + cc, (double)acc);
      }
  }
  C.M = Mb.build(); C.Bp[0] = B0.build(); C.Bp[1] = B1.build(); C.Ep[0] = E0.build(); C.Ep[1] = E1.build(); C.Vort = Vb.build();
  C.diag = csr_diag_of(C.M);
  C.R.nrows = nc * nu; C.R.ncols = C.ncg;
  C.R.rowptr.resize((size_t)nc * nu + 1);
  for (int i = 0; i <= nc * nu; i++) C.R.rowptr[(size_t)i] = i;
  C.R.col = C.cg_of_dg;
  C.R.val.assign((size_t)nc * nu, 1.0);
}

// element block-Jacobi of the tentative-velocity operator,  (I + gamma sum_e alpha / len_e N_e^T N_e)^-1  per cell
inline Csr assemble_block_jacobi(const GeneralTables& T, const GMesh& M, const std::vector<CellLocal>& loc, double gamma) {
  const int n2 = T.n2, ne = T.ne;
  CsrBuilder D(M.nc * n2, M.nc * n2);
  parallel_for(M.nc, [&](int c) {  // rows of different cells: disjoint parts of the builder
    std::vector<real> Dm((size_t)n2 * n2, 0);
    for (int i = 0; i < n2; i++) Dm[(size_t)i * n2 + i] = 1;
    for (int l = 0; l < 3; l++) {
      const real f = (real)gamma * (real)T.alpha / (real)M.elen[M.cedge[3 * (size_t)c + l]];
      for (int a = 0; a < ne; a++)
        for (int r = 0; r < n2; r++)
          for (int cc = 0; cc < n2; cc++)
            Dm[(size_t)r * n2 + cc] += f * loc[(size_t)c].N[l][(size_t)a * n2 + r] * loc[(size_t)c].N[l][(size_t)a * n2 + cc];
    }
    invert(n2, Dm);
    for (int r = 0; r < n2; r++)
      for (int cc = 0; cc < n2; cc++) D.add(c * n2 + r, c * n2 + cc, (double)Dm[(size_t)r * n2 + cc]);
  });
  return D.build();
}

// Lifting tables of the MATRIX-FREE BDM projection / hybrid preconditioner on a general triangulation (round 4; k_g_lift):
//   out_K = x_K + G_K d_K,   d_K = the 3 ne edge-moment defects of the cell (w (N^{K'} x_K' - N^K x_K); boundary: -N^K x_K),
//   gamma < 0:  G_K = Lift_K                                  (BDM projection, common.py:91-108)
//   gamma >= 0: G_K = (I - Dinv_K) Lift_K,  Dinv_K = (I + gamma sum_e alpha / len_e N_e^T N_e)^-1   (hybrid preconditioner
//               Pi + Dinv (I - Pi) of the tentative velocity: the conforming part is kept, the rest goes through the element block-Jacobi)
// n2 x 3 ne doubles per cell in cell-fastest order Gt[(r * 3 ne + l * ne + a) * nc + c] (coalesced across the cells of a wave):
// 1.9 KB per cell at k = 2 where the assembled projection holds 19 KB per cell.
inline std::vector<double> assemble_lift_tables(const GeneralTables& T, const GMesh& M, const std::vector<CellLocal>& loc, double gamma) {
  const int n2 = T.n2, ne = T.ne, nm = 3 * ne;
  std::vector<double> Gt((size_t)M.nc * n2 * nm);
  parallel_for(M.nc, [&](int c) {
    std::vector<real> Lm((size_t)n2 * nm), Gm((size_t)n2 * nm);
    for (int l = 0; l < 3; l++)
      for (int r = 0; r < n2; r++)
        for (int a = 0; a < ne; a++) Lm[(size_t)r * nm + l * ne + a] = loc[(size_t)c].Lift[l][(size_t)r * ne + a];
    if (gamma < 0) Gm = Lm;
    else {
      std::vector<real> Dm((size_t)n2 * n2, 0);
      for (int i = 0; i < n2; i++) Dm[(size_t)i * n2 + i] = 1;
      for (int l = 0; l < 3; l++) {
        const real f = (real)gamma * (real)T.alpha / (real)M.elen[M.cedge[3 * (size_t)c + l]];
        for (int a = 0; a < ne; a++)
          for (int r = 0; r < n2; r++)
            for (int cc = 0; cc < n2; cc++)
              Dm[(size_t)r * n2 + cc] += f * loc[(size_t)c].N[l][(size_t)a * n2 + r] * loc[(size_t)c].N[l][(size_t)a * n2 + cc];
      }
      invert(n2, Dm);
      for (int r = 0; r < n2; r++)
        for (int q = 0; q < nm; q++) {
          real acc = Lm[(size_t)r * nm + q];
          for (int cc = 0; cc < n2; cc++) acc -= Dm[(size_t)r * n2 + cc] * Lm[(size_t)cc * nm + q];
          Gm[(size_t)r * nm + q] = acc;
        }
    }
    for (int r = 0; r < n2; r++)
      for (int q = 0; q < nm; q++) Gt[((size_t)r * nm + q) * M.nc + c] = (double)Gm[(size_t)r * nm + q];
  });
  return Gt;
}

}  // namespace hdg
