// Coarse space of the trace preconditioner on general triangulations.
//
// The reference preconditions the condensed trace system with GTMG: Chebyshev(2) / facet-block Jacobi on the trace space,
// a P1 coarse space on the same mesh, and an algebraic multigrid V-cycle (PETSc GAMG) as the coarse solver
// (src/timesteppers/hdg_imex.py:139-167).  The structured engine replaces the coarse solver by a geometric V-cycle on the
// vertex grid; a general triangulation has no such grid, so this file builds, on the host and once per mesh,
//   * the prolongation P1 -> trace space (the trace of a continuous piecewise linear function, edge by edge),
//   * the Galerkin coarse operator  A_0 = P^T S P,
//   * a smoothed-aggregation hierarchy A_1, A_2, ... of A_0 (Vanek / Mandel / Brezina: greedy aggregates of strongly
//     connected vertices, piecewise constant tentative prolongator, one damped Jacobi step on it), and
//   * the dense pseudo-inverse of the coarsest operator (the systems are singular: pure Neumann problem, constants).
// The engine applies the V-cycle with its CSR kernel (hdg_engine.hip: amg_vcycle).  Own design: nothing here follows PETSc code.
#pragma once

#include <algorithm>
#include <cmath>
#include <vector>

#include "hdg_general.hpp"

namespace hdg {

inline Csr csr_transpose(const Csr& A) {
  Csr T;
  T.nrows = A.ncols; T.ncols = A.nrows;
  T.rowptr.assign((size_t)A.ncols + 1, 0);
  for (int c : A.col) T.rowptr[(size_t)c + 1]++;
  for (int i = 0; i < A.ncols; i++) T.rowptr[(size_t)i + 1] += T.rowptr[(size_t)i];
  T.col.resize(A.col.size()); T.val.resize(A.val.size());
  std::vector<int> pos(T.rowptr.begin(), T.rowptr.end() - 1);
  for (int r = 0; r < A.nrows; r++)
    for (int q = A.rowptr[(size_t)r]; q < A.rowptr[(size_t)r + 1]; q++) {
      const int p = pos[(size_t)A.col[(size_t)q]]++;
      T.col[(size_t)p] = r;
      T.val[(size_t)p] = A.val[(size_t)q];
    }
  return T;
}

// C = A B (row by row with a dense accumulator); entries below drop * (largest entry of the row) are left out
inline Csr csr_multiply(const Csr& A, const Csr& B, double drop = 0.0) {
  Csr C;
  C.nrows = A.nrows; C.ncols = B.ncols;
  C.rowptr.assign((size_t)A.nrows + 1, 0);
  std::vector<int> marker((size_t)B.ncols, -1), cols;
  std::vector<double> acc((size_t)B.ncols, 0.0);
  for (int r = 0; r < A.nrows; r++) {
    cols.clear();
    for (int q = A.rowptr[(size_t)r]; q < A.rowptr[(size_t)r + 1]; q++) {
      const double a = A.val[(size_t)q];
      const int k = A.col[(size_t)q];
      for (int p = B.rowptr[(size_t)k]; p < B.rowptr[(size_t)k + 1]; p++) {
        const int c = B.col[(size_t)p];
        if (marker[(size_t)c] != r) { marker[(size_t)c] = r; acc[(size_t)c] = 0.0; cols.push_back(c); }
        acc[(size_t)c] += a * B.val[(size_t)p];
      }
    }
    std::sort(cols.begin(), cols.end());
    double big = 0.0;
    for (int c : cols) big = std::max(big, std::fabs(acc[(size_t)c]));
    for (int c : cols)
      if (std::fabs(acc[(size_t)c]) > drop * big) { C.col.push_back(c); C.val.push_back(acc[(size_t)c]); }
    C.rowptr[(size_t)r + 1] = (int)C.col.size();
  }
  return C;
}

// Prolongation from the vertices to the trace space: row (e, m) = coefficient of Legendre mode m on edge e of the linear
// function with the two vertex values -- sqrt(len) * Vl^{-1} applied to its values at the trace nodes, the conversion the
// engine uses for every trace field (assemble_general: Cl).  The parameter of an edge runs from its first vertex to its second.
inline Csr p1_to_trace_matrix(const GeneralTables& T, const GMesh& M) {
  const int nl = T.nl;
  CsrBuilder P(M.ne * nl, M.nv);
  for (int e = 0; e < M.ne; e++) {
    const double sl = std::sqrt(M.elen[(size_t)e]);
    const int va = M.ev[2 * (size_t)e], vb = M.ev[2 * (size_t)e + 1];
    for (int m = 0; m < nl; m++) {
      double ca = 0.0, cb = 0.0;
      for (int i = 0; i < nl; i++) {
        const double t = (double)T.node_t[(size_t)i];
        ca += (double)T.Vlinv[(size_t)m * nl + i] * (1.0 - t);
        cb += (double)T.Vlinv[(size_t)m * nl + i] * t;
      }
      if (std::fabs(ca) > 1e-13) P.add(e * nl + m, va, sl * ca);
      if (std::fabs(cb) > 1e-13) P.add(e * nl + m, vb, sl * cb);
    }
  }
  return P.build();
}

struct AmgLevel {
  Csr A, P, R;   // P: from the next coarser level to this one; R = P^T
  dvec dinv;     // 1 / diagonal
  double lmax = 1.0;  // largest eigenvalue of D^{-1} A (power iteration, 5 % safety)
};
struct AmgHierarchy {
  std::vector<AmgLevel> lev;
  Csr coarse_pinv;  // dense pseudo-inverse of the coarsest operator (empty: smooth only)
};

inline dvec csr_diagonal(const Csr& A) {
  dvec d((size_t)A.nrows, 0.0);
  for (int r = 0; r < A.nrows; r++)
    for (int q = A.rowptr[(size_t)r]; q < A.rowptr[(size_t)r + 1]; q++)
      if (A.col[(size_t)q] == r) d[(size_t)r] += A.val[(size_t)q];
  return d;
}

inline double jacobi_lambda_max(const Csr& A, const dvec& dinv) {
  const int n = A.nrows;
  dvec x((size_t)n), y((size_t)n);
  unsigned long long st = 0x9E3779B97F4A7C15ULL;
  for (auto& v : x) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (double)(st % 2000001ULL) / 1.0e6 - 1.0; }
  double lam = 1.0;
  for (int it = 0; it < 30; it++) {
    double nrm = 0.0;
    for (double v : x) nrm += v * v;
    nrm = std::sqrt(nrm);
    if (!(nrm > 0.0)) break;
    for (auto& v : x) v /= nrm;
    for (int r = 0; r < n; r++) {
      double acc = 0.0;
      for (int q = A.rowptr[(size_t)r]; q < A.rowptr[(size_t)r + 1]; q++) acc += A.val[(size_t)q] * x[(size_t)A.col[(size_t)q]];
      y[(size_t)r] = dinv[(size_t)r] * acc;
    }
    double l2 = 0.0;
    for (double v : y) l2 += v * v;
    lam = std::sqrt(l2);
    x.swap(y);
  }
  return 1.05 * lam;
}

// greedy aggregation on the strong connections |a_ij| >= theta sqrt(a_ii a_jj)
inline std::vector<int> amg_aggregate(const Csr& A, const dvec& diag, double theta, int& nagg) {
  const int n = A.nrows;
  std::vector<int> agg((size_t)n, -1);
  nagg = 0;
  auto strong = [&](int i, int q) {
    const int j = A.col[(size_t)q];
    return j != i && std::fabs(A.val[(size_t)q]) >= theta * std::sqrt(std::fabs(diag[(size_t)i] * diag[(size_t)j]));
  };
  for (int i = 0; i < n; i++) {  // pass 1: a vertex whose strong neighbourhood is untouched founds an aggregate with it
    if (agg[(size_t)i] != -1) continue;
    bool free_nbrs = true, any = false;
    for (int q = A.rowptr[(size_t)i]; q < A.rowptr[(size_t)i + 1] && free_nbrs; q++)
      if (strong(i, q)) { any = true; if (agg[(size_t)A.col[(size_t)q]] != -1) free_nbrs = false; }
    if (!free_nbrs || !any) continue;
    agg[(size_t)i] = nagg;
    for (int q = A.rowptr[(size_t)i]; q < A.rowptr[(size_t)i + 1]; q++)
      if (strong(i, q)) agg[(size_t)A.col[(size_t)q]] = nagg;
    nagg++;
  }
  std::vector<int> agg2(agg);  // pass 2: the rest joins the aggregate (of pass 1) it is most strongly connected to
  for (int i = 0; i < n; i++) {
    if (agg[(size_t)i] != -1) continue;
    int best = -1;
    double bv = 0.0;
    for (int q = A.rowptr[(size_t)i]; q < A.rowptr[(size_t)i + 1]; q++) {
      const int j = A.col[(size_t)q];
      if (strong(i, q) && agg[(size_t)j] != -1 && std::fabs(A.val[(size_t)q]) > bv) { bv = std::fabs(A.val[(size_t)q]); best = j; }
    }
    if (best >= 0) agg2[(size_t)i] = agg[(size_t)best];
  }
  agg.swap(agg2);
  for (int i = 0; i < n; i++) {  // pass 3: what is left (isolated vertices and their like)
    if (agg[(size_t)i] != -1) continue;
    agg[(size_t)i] = nagg;
    for (int q = A.rowptr[(size_t)i]; q < A.rowptr[(size_t)i + 1]; q++)
      if (strong(i, q) && agg[(size_t)A.col[(size_t)q]] == -1) agg[(size_t)A.col[(size_t)q]] = nagg;
    nagg++;
  }
  return agg;
}

// dense pseudo-inverse of a symmetric positive semi-definite matrix whose kernel is the constants:
// (A + a 1 1^T)^{-1} - 1 1^T / (a n^2), by Gauss-Jordan elimination with partial pivoting
inline Csr dense_pinv_constants(const Csr& A) {
  const int n = A.nrows;
  std::vector<double> B((size_t)n * n, 0.0), I((size_t)n * n, 0.0);
  double tr = 0.0;
  for (int r = 0; r < n; r++)
    for (int q = A.rowptr[(size_t)r]; q < A.rowptr[(size_t)r + 1]; q++) {
      B[(size_t)r * n + A.col[(size_t)q]] += A.val[(size_t)q];
      if (A.col[(size_t)q] == r) tr += A.val[(size_t)q];
    }
  const double a = tr > 0.0 ? tr / ((double)n * n) : 1.0;
  for (auto& v : B) v += a;
  for (int i = 0; i < n; i++) I[(size_t)i * n + i] = 1.0;
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++) if (std::fabs(B[(size_t)r * n + c]) > std::fabs(B[(size_t)piv * n + c])) piv = r;
    if (B[(size_t)piv * n + c] == 0.0) throw std::string("coarsest multigrid operator is singular beyond the constants");
    if (piv != c)
      for (int k = 0; k < n; k++) { std::swap(B[(size_t)c * n + k], B[(size_t)piv * n + k]); std::swap(I[(size_t)c * n + k], I[(size_t)piv * n + k]); }
    const double d = 1.0 / B[(size_t)c * n + c];
    for (int k = 0; k < n; k++) { B[(size_t)c * n + k] *= d; I[(size_t)c * n + k] *= d; }
    for (int r = 0; r < n; r++) {
      if (r == c) continue;
      const double f = B[(size_t)r * n + c];
      if (f == 0.0) continue;
      for (int k = 0; k < n; k++) { B[(size_t)r * n + k] -= f * B[(size_t)c * n + k]; I[(size_t)r * n + k] -= f * I[(size_t)c * n + k]; }
    }
  }
  const double shift = 1.0 / (a * (double)n * n);
  Csr Pinv;
  Pinv.nrows = Pinv.ncols = n;
  Pinv.rowptr.assign((size_t)n + 1, 0);
  Pinv.col.reserve((size_t)n * n); Pinv.val.reserve((size_t)n * n);
  for (int r = 0; r < n; r++) {
    for (int c = 0; c < n; c++) { Pinv.col.push_back(c); Pinv.val.push_back(I[(size_t)r * n + c] - shift); }
    Pinv.rowptr[(size_t)r + 1] = (int)Pinv.col.size();
  }
  return Pinv;
}

// hierarchy of A0 (symmetric, positive semi-definite, kernel = constants)
inline void amg_build(const Csr& A0, AmgHierarchy& H, int max_coarse = 400, int max_levels = 12, double theta = 0.08) {
  H.lev.clear();
  Csr cur = A0;
  while (true) {
    AmgLevel L;
    L.A = cur;
    const dvec diag = csr_diagonal(L.A);
    L.dinv.assign(diag.size(), 0.0);
    for (size_t i = 0; i < diag.size(); i++) L.dinv[i] = diag[i] != 0.0 ? 1.0 / diag[i] : 0.0;
    L.lmax = jacobi_lambda_max(L.A, L.dinv);
    const int n = L.A.nrows;
    if (n <= max_coarse || (int)H.lev.size() + 1 >= max_levels) { H.lev.push_back(std::move(L)); break; }
    int nagg = 0;
    const std::vector<int> agg = amg_aggregate(L.A, diag, theta * std::pow(0.5, (double)H.lev.size()), nagg);
    if (nagg >= n || nagg < 1) { H.lev.push_back(std::move(L)); break; }  // no coarsening possible
    // tentative prolongator (piecewise constant, entries 1: the constants stay the kernel on every level), smoothed by one
    // damped Jacobi step:  P = (I - omega D^{-1} A) P_tent,  omega = 4 / (3 lambda_max)
    Csr Pt;
    Pt.nrows = n; Pt.ncols = nagg;
    Pt.rowptr.resize((size_t)n + 1);
    Pt.col.resize((size_t)n); Pt.val.assign((size_t)n, 1.0);
    for (int i = 0; i <= n; i++) Pt.rowptr[(size_t)i] = i;
    for (int i = 0; i < n; i++) Pt.col[(size_t)i] = agg[(size_t)i];
    const Csr AP = csr_multiply(L.A, Pt);
    const double omega = 4.0 / (3.0 * L.lmax);
    CsrBuilder Pb(n, nagg);
    for (int i = 0; i < n; i++) {
      Pb.add(i, agg[(size_t)i], 1.0);
      for (int q = AP.rowptr[(size_t)i]; q < AP.rowptr[(size_t)i + 1]; q++)
        Pb.add(i, AP.col[(size_t)q], -omega * L.dinv[(size_t)i] * AP.val[(size_t)q]);
    }
    L.P = Pb.build();
    L.R = csr_transpose(L.P);
    cur = csr_multiply(L.R, csr_multiply(L.A, L.P));
    H.lev.push_back(std::move(L));
  }
  const Csr& Ac = H.lev.back().A;
  H.coarse_pinv = Csr();
  if (Ac.nrows <= 2000) H.coarse_pinv = dense_pinv_constants(Ac);
}

}  // namespace hdg
