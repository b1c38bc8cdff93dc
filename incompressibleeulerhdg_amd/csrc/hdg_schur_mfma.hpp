// GEMM-shaped (matrix-core) versions of the cell-local kernels of the hybridised mixed-Poisson solve for k >= 3:
// static condensation, back-substitution (firedrake.SCPC / Slate local elimination, hdg_imex.py:128-135), the
// pressure-gradient combination of the tentative right-hand side (hdg_imex.py:239-247, 333-340) and the weak divergence
// (hdg_imex.py:353-365).
//
// Why: the one-thread-per-cell kernels keep a whole local vector (up to 57 doubles) and stream the local matrix through
// scalar loads; at k = 4 the 57 x 57 block is 3 249 FMAs per cell against 256 VGPRs and a 16 KB scalar cache --
// k_backsub<4> ran at 0.54 TB/s (0.07 of HBM peak), a 3.5x cliff from k = 3.  With shared local matrices the batched
// product is a true GEMM  Y[n_out x cells] = M[n_out x n_in] X[n_in x cells]  (SURVEY.md section 7.2).
//
// Mapping (the one of k_edge_lift_mfma): one wave owns 16 consecutive cells (or grid corners) of one mesh row = the N
// dimension of v_mfma_f64_16x16x4; the coefficient planes of the cell vectors are the B operands, read straight from HBM;
// the local matrices are packed on the host in A-operand lane order (Engine::pack_*_mfma: tile (mt, ks), 64 doubles,
// entry l = M[16 mt + l % 16][4 ks + l / 16]), staged in LDS once per workgroup and read back one conflict-free
// ds_read_b64 per MFMA.  Accumulator layout: lane (lk = l / 16, li = l % 16), register r holds row lk + 4 r of column li.
//
// Velocity vectors move in 16-BYTE accesses (the component-pair layout, hdg_kernels.hpp): the assignment of velocity dofs
// to K slots and result rows is free as long as the host packs the table columns / rows the same way, so
//   K side: K-steps come in pairs (2q, 2q+1); lane (lk, li) loads the pair of mode m = 4q + lk of cell li with ONE
//           buffer_load_dwordx4 and feeds .x (component 0) to K-step 2q, .y to K-step 2q+1:      column(m, d) = 8 (m/4) + 4 d + m%4
//   M side: result tile mt holds the modes 8 mt .. 8 mt + 7; lane (lk, li) finds both components of mode 8 mt + lk in
//           registers 0, 1 and of mode 8 mt + 4 + lk in registers 2, 3: two buffer_store_dwordx4:  row(m, d) = 16 (m/8) + m%4 + 4 (2 ((m%8)/4) + d)
// (1 KiB per wave-instruction instead of 512 B: the 8-byte form of these kernels ran 10-20 % slower).
#pragma once
#include <hip/hip_runtime.h>

namespace hdg {

template <int K>
struct SchurMfma {
  static constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, NL = Dim<K>::NL, N2 = 2 * NU, NT = 3 * NL, NX = N2 + NP;
  static constexpr int KQ = (NU + 3) / 4;                                   // K-step PAIRS of a velocity source
  static constexpr int KSU = 2 * KQ, KSP = (NP + 3) / 4, KST = (NT + 3) / 4;  // K-steps per source block
  static constexpr int MTU = (NU + 7) / 8;                                  // M-tiles of a velocity result (8 modes each)
  static_assert(NP <= 16 && NT <= 16, "pressure / trace results must fit one M-tile");
  // back-substitution: rows = [u (kappa order, MTU tiles) | phi (1 tile)], K = [r_w | r_p | lambda]
  static constexpr int BS_KS = KSU + KSP + KST, BS_MT = MTU + 1, BS_TILES = BS_MT * BS_KS;
  // pressure gradient: rows = velocity (MTU tiles), K = [p | lambda]
  static constexpr int PG_KS = KSP + KST, PG_TILES = MTU * PG_KS;
  // weak divergence: rows = pressure (1 tile); 6 blocks of KSU K-steps (own: base + edge 1, edge 0, edge 2; neighbours 0, 1, 2)
  static constexpr int WD_TILES = 6 * KSU, WDB_TILES = KSU;
  // condensation: rows = (H, D, V) trace modes of a corner (1 tile); 4 cell blocks of K = [r_w | r_p]
  static constexpr int CD_KS = KSU + KSP, CD_TILES = 4 * CD_KS;
};

// waves per workgroup (measured at 512^2, k = 3 / 4, us per launch with 8 | 4 waves): back-substitution 80 / 148 | 77 / 135,
// pressure gradient 96 / 156 | 117 / 163, weak divergence 73 / 103 | 75 / 104, condensation (one workgroup per corner row: few
// workgroups) 32 / 43 | 45 / 63
#define HDG_SCHUR_WAVES 8
#define HDG_BACKSUB_WAVES 4
#define HDG_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
// Scheduling fence at the end of a K-step: without it the compiler hoists the LDS reads of EVERY A tile of a cell tile above
// the first MFMA (k_backsub_mfma<4,true,true>: 76 tiles = 152 VGPRs of operands in flight, 255 VGPRs + scratch, 2 waves per
// SIMD); with it an A tile lives from its ds_read to its MFMA.
#define HDG_KSTEP_FENCE() __builtin_amdgcn_sched_barrier(0)

// (row, shape) of a cell-row workgroup; returns false when the workgroup has nothing to do
__device__ __forceinline__ bool schur_cell_row(const Geo& g, int& j, int& s) {
  const int xcd_ = blockIdx.x & 7, q_ = blockIdx.x >> 3;
  const int jj_ = q_ >> 1;
  s = q_ & 1;
  const int r_ = xcd_ * g.rows_xcd + jj_;
  j = launch_row(g, r_);
  return jj_ < g.rows_xcd && r_ < g.wrows;
}
template <int NTILES, int WAVES = HDG_SCHUR_WAVES>
__device__ __forceinline__ void schur_stage(double* tab, const double* __restrict__ src) {
  for (int p = threadIdx.x; p < NTILES * 64; p += 64 * WAVES) tab[p] = src[p];
  __syncthreads();
}
// index (within the trace vector) of trace row n = e * NL + m of cell (s, i, j); edge types as edge_off()
template <int NL>
__device__ __forceinline__ long schur_trace_index(const Geo& g, int s, int i, int j, int n) {
  const int e = n / NL, m = n - e * NL;
  const int t = e == 0 ? 0 : (e == 1 ? 2 : 1);
  const long off = e == 0 ? (long)(j + s + GH) * g.P + i : (e == 1 ? (long)(j + GH) * g.P + i : (long)(j + GH) * g.P + (s ? i + 1 : i));
  return ((long)t * NL + m) * g.G + off;
}

// ------------------------------------------------------------------------------------------
// K8 on the matrix cores:  (u, phi)_K = Ainv r_{x,K} - W lambda_K
// ------------------------------------------------------------------------------------------
template <int K, bool HASW, bool HASP>
__global__ __launch_bounds__(64 * HDG_BACKSUB_WAVES) void k_backsub_mfma(Geo g, const double* __restrict__ tabs0,
                                                                         const double* __restrict__ tabs1,
                                                                         const double* __restrict__ rw, const double* __restrict__ rp,
                                                                         const double* __restrict__ lam, double* __restrict__ u,
                                                                         double* __restrict__ phi) {
  typedef SchurMfma<K> S;
  constexpr int NU = S::NU, NP = S::NP, NL = S::NL, NT = S::NT, KQ = S::KQ, KSU = S::KSU, KSP = S::KSP, KST = S::KST, MTU = S::MTU, KSA = S::BS_KS;
  __shared__ double tab[S::BS_TILES * 64];
  const VelBuf Brw(rw), Bu(u);
  int j, s;
  if (!schur_cell_row(g, j, s)) return;  // whole workgroup
  schur_stage<S::BS_TILES, HDG_BACKSUB_WAVES>(tab, s == 0 ? tabs0 : tabs1);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const long rowC = rowbase(g, s, j);
  const int ntx = (g.nx + 15) >> 4;
  for (int tx = w; tx < ntx; tx += HDG_BACKSUB_WAVES) {
    const int i = tx * 16 + li;
    const bool col = i < g.nx;
    const int ic = col ? i : g.nx - 1;  // clamped: loads stay in bounds, results of invalid columns are not stored
    const long c = rowC + ic;
    hdg_v4d D[MTU + 1];
#pragma unroll
    for (int mt = 0; mt <= MTU; mt++) D[mt] = hdg_v4d{0, 0, 0, 0};
    // every B operand of the tile is requested first (one latency phase), then the K-steps run behind fences
    double bp[HASP ? KSP : 1], bt[KST];
    hdg_d2 bw[HASW ? KQ : 1];
    if (HASW) {
#pragma unroll
      for (int q = 0; q < KQ; q++) bw[q] = ld_pair_lane<NU>(Brw, g.Nc, c, 4 * q + lk);
    }
    if (HASP) {
#pragma unroll
      for (int ks = 0; ks < KSP; ks++) {
        const int n = 4 * ks + lk;
        bp[ks] = (n < NP) ? rp[(long)n * g.Nc + c] : 0.0;
      }
    }
#pragma unroll
    for (int ks = 0; ks < KST; ks++) {
      const int n = 4 * ks + lk;
      bt[ks] = (n < NT) ? lam[schur_trace_index<NL>(g, s, ic, j, n)] : 0.0;
    }
    HDG_KSTEP_FENCE();
    if (HASW) {
#pragma unroll
      for (int ks = 0; ks < KSU; ks++) {
#pragma unroll
        for (int mt = 0; mt <= MTU; mt++) D[mt] = HDG_MFMA(tab[(mt * KSA + ks) * 64 + l], (ks & 1) ? bw[ks >> 1].y : bw[ks >> 1].x, D[mt]);
        HDG_KSTEP_FENCE();
      }
    }
    if (HASP) {
#pragma unroll
      for (int ks = 0; ks < KSP; ks++) {
#pragma unroll
        for (int mt = 0; mt <= MTU; mt++) D[mt] = HDG_MFMA(tab[(mt * KSA + KSU + ks) * 64 + l], bp[ks], D[mt]);
        HDG_KSTEP_FENCE();
      }
    }
#pragma unroll
    for (int ks = 0; ks < KST; ks++) {
#pragma unroll
      for (int mt = 0; mt <= MTU; mt++) D[mt] = HDG_MFMA(tab[(mt * KSA + KSU + KSP + ks) * 64 + l], bt[ks], D[mt]);
      HDG_KSTEP_FENCE();
    }
    if (col) {
#pragma unroll
      for (int mt = 0; mt < MTU; mt++) {
        const int m0 = 8 * mt + lk, m1 = m0 + 4;
        if (m0 < NU) st_pair_lane(Bu, g.Nc, c, m0, hdg_d2{D[mt][0], D[mt][1]});
        if (m1 < NU) st_pair_lane(Bu, g.Nc, c, m1, hdg_d2{D[mt][2], D[mt][3]});
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = lk + 4 * r;
        if (n < NP) phi[(long)n * g.Nc + c] = D[MTU][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// pressure gradient on the matrix cores:  out = ca*a + cb*b + gamma * ( B^T p - sum_e sigma_e N_e^T lambda_e )
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(64 * HDG_SCHUR_WAVES) void k_pgrad_mfma(Geo g, const double* __restrict__ tabs0,
                                                                       const double* __restrict__ tabs1, const double* __restrict__ a,
                                                                       double ca, const double* __restrict__ b_, double cb,
                                                                       const double* __restrict__ p, const double* __restrict__ lam,
                                                                       double gamma, double* __restrict__ out) {
  typedef SchurMfma<K> S;
  constexpr int NU = S::NU, NP = S::NP, NL = S::NL, NT = S::NT, KSP = S::KSP, KST = S::KST, MTU = S::MTU, KSA = S::PG_KS;
  __shared__ double tab[S::PG_TILES * 64];
  const VelBuf Ba(a), Bb(b_), Bo(out);
  int j, s;
  if (!schur_cell_row(g, j, s)) return;
  schur_stage<S::PG_TILES>(tab, s == 0 ? tabs0 : tabs1);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const long rowC = rowbase(g, s, j);
  const int ntx = (g.nx + 15) >> 4;
  for (int tx = w; tx < ntx; tx += HDG_SCHUR_WAVES) {
    const int i = tx * 16 + li;
    const bool col = i < g.nx;
    const int ic = col ? i : g.nx - 1;
    const long c = rowC + ic;
    hdg_v4d D[MTU];
#pragma unroll
    for (int mt = 0; mt < MTU; mt++) D[mt] = hdg_v4d{0, 0, 0, 0};
    double bp[KSP], bt[KST];
#pragma unroll
    for (int ks = 0; ks < KSP; ks++) {
      const int n = 4 * ks + lk;
      bp[ks] = (n < NP) ? p[(long)n * g.Nc + c] : 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < KST; ks++) {
      const int n = 4 * ks + lk;
      bt[ks] = (n < NT) ? lam[schur_trace_index<NL>(g, s, ic, j, n)] : 0.0;
    }
    HDG_KSTEP_FENCE();
#pragma unroll
    for (int ks = 0; ks < KSP; ks++) {
#pragma unroll
      for (int mt = 0; mt < MTU; mt++) D[mt] = HDG_MFMA(tab[(mt * KSA + ks) * 64 + l], bp[ks], D[mt]);
      HDG_KSTEP_FENCE();
    }
#pragma unroll
    for (int ks = 0; ks < KST; ks++) {
#pragma unroll
      for (int mt = 0; mt < MTU; mt++) D[mt] = HDG_MFMA(tab[(mt * KSA + KSP + ks) * 64 + l], bt[ks], D[mt]);
      HDG_KSTEP_FENCE();
    }
    if (col) {
#pragma unroll
      for (int mt = 0; mt < MTU; mt++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int m = 8 * mt + 4 * h + lk;
          if (m < NU) {
            const hdg_d2 va = a ? ld_pair_lane<NU>(Ba, g.Nc, c, m) : hdg_d2{0.0, 0.0};
            const hdg_d2 vb = b_ ? ld_pair_lane<NU>(Bb, g.Nc, c, m) : hdg_d2{0.0, 0.0};
            hdg_d2 v;
            v.x = fma(gamma, D[mt][2 * h], ca * va.x + cb * vb.x);
            v.y = fma(gamma, D[mt][2 * h + 1], ca * va.y + cb * vb.y);
            st_pair_lane(Bo, g.Nc, c, m, v);
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// weak divergence on the matrix cores (hdg_imex.py:353-365):
//   out = sc * [ D0 x + sum_{e: neighbour exists} sigma_e Pt_e^T (N_e x + N'_e x_nbr(e)) / 2 ]     (BROKEN: out = sc * B x)
// Blocks (each NP x N2): 0 = D0 + E_1 (edge 1 always has its neighbour), 1 = E_0, 2 = E_2 on the own coefficients,
// 3, 4, 5 = E'_0, E'_1, E'_2 on the neighbours'; a missing neighbour zeroes the B operand of its two blocks.
// ------------------------------------------------------------------------------------------
template <int K, bool BROKEN>
__global__ __launch_bounds__(64 * HDG_SCHUR_WAVES) void k_weak_div_mfma(Geo g, const double* __restrict__ tabs0,
                                                                          const double* __restrict__ tabs1,
                                                                          const double* __restrict__ q, double sc,
                                                                          double* __restrict__ out) {
  typedef SchurMfma<K> S;
  constexpr int NU = S::NU, NP = S::NP, KQ = S::KQ, KSU = S::KSU;
  constexpr int NTILES = BROKEN ? S::WDB_TILES : S::WD_TILES;
  __shared__ double tab[NTILES * 64];
  const VelBuf Bq(q);
  int j, s;
  if (!schur_cell_row(g, j, s)) return;
  schur_stage<NTILES>(tab, s == 0 ? tabs0 : tabs1);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const int gj = g.joff + j;
  const bool has0 = s == 0 ? gj > 0 : gj < g.nyg - 1;
  const int jn0 = s == 0 ? j - 1 : j + 1;
  const long rowN0 = rowbase(g, 1 - s, jn0), rowN = rowbase(g, 1 - s, j), rowC = rowbase(g, s, j);
  const int ntx = (g.nx + 15) >> 4;
  for (int tx = w; tx < ntx; tx += HDG_SCHUR_WAVES) {
    const int i = tx * 16 + li;
    const bool col = i < g.nx;
    const int ic = col ? i : g.nx - 1;
    const bool has2 = s == 0 ? ic > 0 : ic < g.nx - 1;
    const int i2 = s == 0 ? ic - 1 : ic + 1;
    const long c = rowC + ic, cn0 = rowN0 + ic, cn1 = rowN + ic, cn2 = rowN + (has2 ? i2 : ic);
    hdg_v4d D = {0, 0, 0, 0};
    hdg_d2 bo[KQ], b0[BROKEN ? 1 : KQ], b1[BROKEN ? 1 : KQ], b2[BROKEN ? 1 : KQ];
    const hdg_d2 zero2 = {0.0, 0.0};
#pragma unroll
    for (int qq = 0; qq < KQ; qq++) {
      const int m = 4 * qq + lk;
      bo[qq] = ld_pair_lane<NU>(Bq, g.Nc, c, m);
      if (!BROKEN) {
        b0[qq] = has0 ? ld_pair_lane<NU>(Bq, g.Nc, cn0, m) : zero2;
        b1[qq] = ld_pair_lane<NU>(Bq, g.Nc, cn1, m);
        b2[qq] = has2 ? ld_pair_lane<NU>(Bq, g.Nc, cn2, m) : zero2;
      }
    }
    HDG_KSTEP_FENCE();
#pragma unroll
    for (int ks = 0; ks < KSU; ks++) {
      const int qq = ks >> 1;
      const double vo = (ks & 1) ? bo[qq].y : bo[qq].x;
      D = HDG_MFMA(tab[(0 * KSU + ks) * 64 + l], vo, D);
      if (!BROKEN) {
        D = HDG_MFMA(tab[(1 * KSU + ks) * 64 + l], has0 ? vo : 0.0, D);
        D = HDG_MFMA(tab[(2 * KSU + ks) * 64 + l], has2 ? vo : 0.0, D);
        D = HDG_MFMA(tab[(3 * KSU + ks) * 64 + l], (ks & 1) ? b0[qq].y : b0[qq].x, D);
        D = HDG_MFMA(tab[(4 * KSU + ks) * 64 + l], (ks & 1) ? b1[qq].y : b1[qq].x, D);
        D = HDG_MFMA(tab[(5 * KSU + ks) * 64 + l], (ks & 1) ? b2[qq].y : b2[qq].x, D);
      }
      HDG_KSTEP_FENCE();
    }
    if (col) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = lk + 4 * r;
        if (n < NP) out[(long)n * g.Nc + c] = sc * D[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K5 on the matrix cores: condensed right-hand side at the three edges (H, D, V) of grid corner (i, j):
//   out_e = sum_{K contains e} (Y_K r_{x,K})_e - r_lambda,e
// One wave owns 16 consecutive corners of a corner row; result rows = (H, D, V) x NL trace modes (one M-tile).
// Blocks: 0 = all of Y_L on L(i,j); 1 = the D rows of Y_U on U(i,j); 2 = the H rows of Y_U on U(i,j-1); 3 = the V rows of
// Y_U on U(i-1,j).  A cell that does not exist zeroes its B operand.
// ------------------------------------------------------------------------------------------
template <int K, bool HASW, bool HASP>
__global__ __launch_bounds__(64 * HDG_SCHUR_WAVES) void k_condense_mfma(Geo g, const double* __restrict__ tabs,
                                                                          const double* __restrict__ rw, const double* __restrict__ rp,
                                                                          const double* __restrict__ rl, double* __restrict__ out) {
  typedef SchurMfma<K> S;
  constexpr int NU = S::NU, NP = S::NP, NL = S::NL, NT = S::NT, KQ = S::KQ, KSU = S::KSU, KSP = S::KSP, KSA = S::CD_KS;
  __shared__ double tab[S::CD_TILES * 64];
  const VelBuf Brw(rw);
  const int xcd_ = blockIdx.x & 7, jj_ = blockIdx.x >> 3;
  const int r_ = xcd_ * g.rows_xcdc + jj_;
  const int j = launch_row(g, r_);
  if (jj_ >= g.rows_xcdc || r_ >= g.wrowsc) return;  // whole workgroup
  schur_stage<S::CD_TILES>(tab, tabs);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const bool in_y = j < g.ny + g.ehi, below = (g.joff + j) > 0;
  const int ntx = (g.nx + 1 + 15) >> 4;
  for (int tx = w; tx < ntx; tx += HDG_SCHUR_WAVES) {
    const int i = tx * 16 + li;
    const bool col = i <= g.nx, in_x = i < g.nx, left = i > 0;
    const int ic = in_x ? i : g.nx - 1, iw = (i > 0 ? (i <= g.nx ? i - 1 : g.nx - 1) : 0);
    // the four cells around the corner (clamped addresses; ghost rows make rows j - 1 and j addressable on every rank)
    const long cq[4] = {cidx(g, 0, j, ic), cidx(g, 1, j, ic), cidx(g, 1, j - 1, ic), cidx(g, 1, j, iw)};
    const bool vq[4] = {in_x && in_y, in_x && in_y, in_x && below, in_y && left && col};
    hdg_v4d D = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (HASW) {
        hdg_d2 bw[KQ];
#pragma unroll
        for (int qq = 0; qq < KQ; qq++) bw[qq] = vq[q] ? ld_pair_lane<NU>(Brw, g.Nc, cq[q], 4 * qq + lk) : hdg_d2{0.0, 0.0};
        HDG_KSTEP_FENCE();
#pragma unroll
        for (int ks = 0; ks < KSU; ks++) {
          D = HDG_MFMA(tab[(q * KSA + ks) * 64 + l], (ks & 1) ? bw[ks >> 1].y : bw[ks >> 1].x, D);
          if ((ks & 3) == 3) HDG_KSTEP_FENCE();
        }
      }
      if (HASP) {
#pragma unroll
        for (int ks = 0; ks < KSP; ks++) {
          const int n = 4 * ks + lk;
          const double b = (n < NP && vq[q]) ? rp[(long)n * g.Nc + cq[q]] : 0.0;
          D = HDG_MFMA(tab[(q * KSA + KSU + ks) * 64 + l], b, D);
        }
      }
    }
    if (col) {
      const long o = (long)(j + GH) * g.P + i;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = lk + 4 * r;
        if (row < NT) {
          const int eb = row / NL, m = row - eb * NL;           // 0 H, 1 D, 2 V
          const int t = eb == 0 ? 0 : (eb == 1 ? 2 : 1);        // plane order of the trace layout: H, V, D
          const bool valid = eb == 0 ? in_x : (eb == 1 ? (in_x && in_y) : in_y);
          const long idx = ((long)t * NL + m) * g.G + o;
          out[idx] = valid ? D[r] - (rl ? rl[idx] : 0.0) : 0.0;
        }
      }
    }
  }
}

}  // namespace hdg
