// CPU check of the host side of the general-mesh path (no GPU, no HIP): topology, assembled operators, P1 coarse space,
// smoothed-aggregation hierarchy, continuous space.  Reads "nv nc / coords / cells" from argv[1], degree from argv[2];
// prints "name value" lines that tests/test_host.py asserts on.  Compiled with g++ by the test.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../incompressibleeulerhdg_amd/csrc/hdg_amg.hpp"

using namespace hdg;

static dvec spmv(const Csr& A, const dvec& x) {
  dvec y((size_t)A.nrows, 0.0);
  for (int r = 0; r < A.nrows; r++)
    for (int q = A.rowptr[(size_t)r]; q < A.rowptr[(size_t)r + 1]; q++) y[(size_t)r] += A.val[(size_t)q] * x[(size_t)A.col[(size_t)q]];
  return y;
}
static double maxabs(const dvec& v) { double m = 0; for (double x : v) m = std::max(m, std::fabs(x)); return m; }
static double asym(const Csr& A) {  // max |A - A^T|
  const Csr T = csr_transpose(A);
  double m = 0;
  for (int r = 0; r < A.nrows; r++) {
    int q = A.rowptr[(size_t)r], p = T.rowptr[(size_t)r];
    const int qe = A.rowptr[(size_t)r + 1], pe = T.rowptr[(size_t)r + 1];
    while (q < qe || p < pe) {
      const int ca = q < qe ? A.col[(size_t)q] : 1 << 30, ct = p < pe ? T.col[(size_t)p] : 1 << 30;
      if (ca == ct) { m = std::max(m, std::fabs(A.val[(size_t)q] - T.val[(size_t)p])); q++; p++; }
      else if (ca < ct) { m = std::max(m, std::fabs(A.val[(size_t)q])); q++; }
      else { m = std::max(m, std::fabs(T.val[(size_t)p])); p++; }
    }
  }
  return m;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = std::fopen(argv[1], "r");
  if (!f) return 2;
  int nv, nc;
  if (std::fscanf(f, "%d %d", &nv, &nc) != 2) return 2;
  std::vector<double> X((size_t)2 * nv);
  std::vector<int> C((size_t)3 * nc);
  for (auto& v : X) if (std::fscanf(f, "%lf", &v) != 1) return 2;
  for (auto& v : C) if (std::fscanf(f, "%d", &v) != 1) return 2;
  std::fclose(f);
  const int k = std::atoi(argv[2]);
  try {
    GMesh M;
    M.build(nv, X.data(), nc, C.data());
    GeneralTables T(k, 1.0, 1.0, 0);
    GeneralOps O;
    std::vector<CellLocal> loc;
    assemble_general(T, M, O, loc);
    std::printf("nv %d\nnc %d\nne %d\nvolume %.15e\n", M.nv, M.nc, M.ne, M.volume);
    // condensed operator: symmetric, the constants are its kernel
    const double sscale = maxabs(O.S.val);
    std::printf("S_asym %.3e\n", asym(O.S) / sscale);
    std::printf("S_null %.3e\n", maxabs(spmv(O.S, O.one_l)) / sscale);
    // BDM projection is a projection
    {
      dvec x((size_t)O.Pi.ncols);
      unsigned long long st = 88172645463325252ULL;
      for (auto& v : x) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (double)(st % 2000001ULL) / 1.0e6 - 1.0; }
      const dvec p1 = spmv(O.Pi, x), p2 = spmv(O.Pi, p1);
      double d = 0;
      for (size_t i = 0; i < p1.size(); i++) d = std::max(d, std::fabs(p1[i] - p2[i]));
      std::printf("Pi_idempotent %.3e\n", d / maxabs(p1));
    }
    // constraint rows of the monolithic system: the mu-row of a constant pressure with its own trace vanishes
    {
      dvec r = spmv(O.Mu_p, O.one_p);
      const dvec r2 = spmv(O.Mu_l, O.one_l);
      for (size_t i = 0; i < r.size(); i++) r[i] += r2[i];
      std::printf("mu_row_constant %.3e\n", maxabs(r));
      dvec s = spmv(O.Psi_p, O.one_p);
      const dvec s2 = spmv(O.Psi_l, O.one_l);
      for (size_t i = 0; i < s.size(); i++) s[i] += s2[i];
      std::printf("psi_row_constant %.3e\n", maxabs(s));
    }
    // P1 coarse space
    const Csr P0 = p1_to_trace_matrix(T, M);
    {
      const dvec one((size_t)M.nv, 1.0);
      const dvec t = spmv(P0, one);
      double d = 0;
      for (size_t i = 0; i < t.size(); i++) d = std::max(d, std::fabs(t[i] - O.one_l[i]));
      std::printf("P0_constants %.3e\n", d);
    }
    const Csr A0 = csr_multiply(csr_transpose(P0), csr_multiply(O.S, P0));
    AmgHierarchy H;
    amg_build(A0, H, 12, 12);  // small coarsest level so that the small test meshes get several levels
    std::printf("amg_levels %d\n", (int)H.lev.size());
    double worst_null = 0, worst_asym = 0, worst_pone = 0;
    int prev = 1 << 30, monotone = 1;
    for (size_t l = 0; l < H.lev.size(); l++) {
      const AmgLevel& L = H.lev[l];
      const dvec one((size_t)L.A.nrows, 1.0);
      const double sc = maxabs(H.lev[0].A.val);  // (a one-vertex level is the 1 x 1 zero matrix)
      worst_null = std::max(worst_null, maxabs(spmv(L.A, one)) / sc);
      worst_asym = std::max(worst_asym, asym(L.A) / sc);
      if (L.A.nrows >= prev) monotone = 0;
      prev = L.A.nrows;
      if (l + 1 < H.lev.size()) {
        const dvec onec((size_t)L.P.ncols, 1.0);
        const dvec t = spmv(L.P, onec);
        for (double v : t) worst_pone = std::max(worst_pone, std::fabs(v - 1.0));
      }
      std::printf("amg_n%d %d\n", (int)l, L.A.nrows);
    }
    std::printf("amg_null %.3e\namg_asym %.3e\namg_P_constants %.3e\namg_monotone %d\n", worst_null, worst_asym, worst_pone, monotone);
    {  // dense pseudo-inverse of the coarsest operator: A A^+ b = b for b orthogonal to the constants
      const Csr& Ac = H.lev.back().A;
      const int n = Ac.nrows;
      dvec b((size_t)n);
      double mean = 0;
      for (int i = 0; i < n; i++) { b[(size_t)i] = std::sin(1.0 + 0.7 * i); mean += b[(size_t)i]; }
      for (auto& v : b) v -= mean / n;
      const dvec x = spmv(H.coarse_pinv, b), r = spmv(Ac, x);
      double d = 0;
      for (int i = 0; i < n; i++) d = std::max(d, std::fabs(r[(size_t)i] - b[(size_t)i]));
      std::printf("coarse_pinv %.3e\n", n > 1 ? d / maxabs(b) : 0.0);  // (one vertex: the range is empty)
    }
    // continuous space
    GeneralCG G;
    assemble_cg(T, M, O, G);
    {
      const dvec one((size_t)G.ncg, 1.0);
      const dvec m1 = spmv(G.M, one);
      double vol = 0;
      for (double v : m1) vol += v;
      std::printf("ncg %d\ncg_volume %.15e\ncg_M_asym %.3e\n", G.ncg, vol, asym(G.M) / maxabs(G.M.val));
      // a constant velocity (1, 2): modal coefficients through the nodal -> modal conversion; projection data Bp = M * value
      dvec nodal((size_t)O.Cq.ncols);
      for (size_t i = 0; i < nodal.size(); i += 2) { nodal[i] = 1.0; nodal[i + 1] = 2.0; }
      const dvec modal = spmv(O.Cq, nodal);
      double d = 0;
      for (int c = 0; c < 2; c++) {
        const dvec rhs = spmv(G.Bp[c], modal);
        for (int i = 0; i < G.ncg; i++) d = std::max(d, std::fabs(rhs[(size_t)i] - (c + 1.0) * m1[(size_t)i]));
      }
      std::printf("cg_projection_rhs_constant %.3e\n", d);
      // vorticity of a rigid rotation (-y, x) is 2: Vort * modal = 2 * M * 1
      for (size_t i = 0; i < nodal.size(); i += 2) { nodal[i] = -O.xq[i + 1]; nodal[i + 1] = O.xq[i]; }
      const dvec rot = spmv(O.Cq, nodal), vr = spmv(G.Vort, rot);
      d = 0;
      for (int i = 0; i < G.ncg; i++) d = std::max(d, std::fabs(vr[(size_t)i] - 2.0 * m1[(size_t)i]));
      std::printf("cg_vorticity_rotation %.3e\n", d);
    }
  } catch (const std::string& e) {
    std::printf("error %s\n", e.c_str());
    return 1;
  }
  return 0;
}
