// The tile kernels of the trace preconditioner (hdg_trace_tile.hpp: same tiles, same stages, same arithmetic per edge) with
// ONE THREAD PER EDGE instead of one per grid corner: a workgroup of 3 x 256 threads, the threads 256 e .. 256 e + 239 take the
// edges of local type e (H, D, V) of the 240 corners of the halo-2 region.
//
// Why: with a corner per thread every stage keeps 3 n_lambda values per array and forms six (n_lambda x 3 n_lambda) products:
// 162-168 VGPRs at k = 2 (3 waves / SIMD), 274 at k = 4 (ONE wave / SIMD: the tile form was slower than the row-stencil
// kernels there and switched off).  The kernels are latency-bound (PMC: 26-28 % of wave cycles waiting, HBM at 0.3 of its
// peak, VALU at a quarter), so what they need is more waves in flight.  An edge's rows of the condensed operator touch two
// cells only -- 2 (n_lambda x 3 n_lambda) products on the corner's own nine values and six of a neighbour corner -- and the
// edge block-Jacobi step is local to the edge: a thread keeps n_lambda values per array.  The edge type is uniform per wave
// (readfirstlane), so the tables stay on the scalar path, a third of them per wave.
#pragma once
#include <hip/hip_runtime.h>

namespace hdg {

template <int K>
struct TraceTile3 {
  static constexpr int NL = Dim<K>::NL, NT = 3 * NL;
  static constexpr int TW = TraceTile<K>::TW, TH = TraceTile<K>::TH, NPER = 256, NTHREADS = 3 * NPER;
  static constexpr int W2 = TW + 4, H2 = TH + 4, N2 = W2 * H2;
  static constexpr int W3 = TW + 6, H3 = TH + 6, N3 = W3 * H3;
  static constexpr int W1 = TW + 2, H1 = TH + 2, N1 = W1 * H1;
  static_assert(N2 <= NPER, "one corner of the halo-2 region per thread and edge type");
};

// local edge type E (0 = H, 1 = D, 2 = V: the row-block order of the local matrices) -> plane group of the trace layout (H, V, D)
template <int E>
struct EdgeOf {
  static constexpr int PL = E == 0 ? 0 : (E == 1 ? 2 : 1);
};
template <int E>
__device__ __forceinline__ bool edge_exists(const CornerInfo& c) { return E == 0 ? c.in_x : (E == 1 ? (c.in_x && c.in_y) : c.in_y); }
template <int NL, int E>
__device__ __forceinline__ void load_edge(const double* __restrict__ v, const Geo& g, const CornerInfo& c, double* e) {
  const bool ok = edge_exists<E>(c);
#pragma unroll
  for (int m = 0; m < NL; m++) e[m] = ok ? v[((long)EdgeOf<E>::PL * NL + m) * g.G + c.o] : 0.0;
}
template <int NL, int E>
__device__ __forceinline__ void store_edge(double* __restrict__ v, const Geo& g, const CornerInfo& c, const double* e) {
  const bool ok = edge_exists<E>(c);
#pragma unroll
  for (int m = 0; m < NL; m++) v[((long)EdgeOf<E>::PL * NL + m) * g.G + c.o] = ok ? e[m] : 0.0;
}
// z = sc * Dinv r on one edge (interior variant through wave-uniform table pointers; boundary lanes redo theirs)
template <int NL, int E>
__device__ __forceinline__ void edge_dinv(const DevTables& T, const CornerInfo& c, double sc, const double* r, double* z) {
#pragma unroll
  for (int q = 0; q < NL; q++) z[q] = 0.0;
  mv_acc_ld<NL, NL>(T.trDinv[EdgeOf<E>::PL][0], NL, r, z, sc);
  const int var = E == 0 ? c.vH : (E == 2 ? c.vV : 0);
  if (E != 1 && var != 0) {
#pragma unroll
    for (int q = 0; q < NL; q++) z[q] = 0.0;
    mv_acc_ld<NL, NL>(T.trDinv[EdgeOf<E>::PL][var], NL, r, z, sc);
  }
}
// the rows of (-S) v that belong to edge E of the corner at (li, lj) of the LDS region A (pitch RW, RH rows; planes as in
// lds_trace_stencil); mine[] = the edge's own values (zero where it does not exist)
template <int K, int E, int RW, int RH>
__device__ __forceinline__ void lds_edge_stencil(const double* __restrict__ A, int li, int lj, const CornerInfo& c, const DevTables& T, double* mine,
                                                 double* y) {
  constexpr int NL = Dim<K>::NL, NT = 3 * NL;
  auto at = [&](int t, int m, int dj, int di) { return A[((t * NL + m) * RH + (lj + dj)) * RW + (li + di)]; };
  const bool vL = c.in_x && c.in_y;
  double own[NT], nb[NT];
#pragma unroll
  for (int m = 0; m < NL; m++) {
    own[m] = c.in_x ? at(0, m, 0, 0) : 0.0;
    own[NL + m] = vL ? at(2, m, 0, 0) : 0.0;
    own[2 * NL + m] = c.in_y ? at(1, m, 0, 0) : 0.0;
    y[m] = 0.0;
  }
  bool v2;
  if (E == 0) {  // H: the cell below, U(i,j-1): edges H(i,j), D(i,j-1), V(i+1,j-1)
    v2 = c.in_x && c.below;
#pragma unroll
    for (int m = 0; m < NL; m++) { nb[m] = own[m]; nb[NL + m] = at(2, m, -1, 0); nb[2 * NL + m] = at(1, m, -1, 1); }
  } else if (E == 1) {  // D: the upper triangle of the same square, U(i,j): edges H(i,j+1), D(i,j), V(i+1,j)
    v2 = vL;
#pragma unroll
    for (int m = 0; m < NL; m++) { nb[m] = at(0, m, 1, 0); nb[NL + m] = own[NL + m]; nb[2 * NL + m] = at(1, m, 0, 1); }
  } else {  // V: the cell to the left, U(i-1,j): edges H(i-1,j+1), D(i-1,j), V(i,j)
    v2 = c.in_y && c.left;
#pragma unroll
    for (int m = 0; m < NL; m++) { nb[m] = at(0, m, 1, -1); nb[NL + m] = at(2, m, 0, -1); nb[2 * NL + m] = own[2 * NL + m]; }
  }
  if (vL) mv_acc_ld<NL, NT>(T.SK[0] + E * NL * NT, NT, own, y, -1.0);
  if (v2) mv_acc_ld<NL, NT>(T.SK[1] + E * NL * NT, NT, nb, y, -1.0);
#pragma unroll
  for (int m = 0; m < NL; m++) mine[m] = own[E * NL + m];
}

// what every stage of a thread needs to know about its corner
struct Tile3Ctx {
  int i0, j0, li, lj, jc;
  bool act;  // the thread has a corner of the halo-2 region
};
#define HDG_EDGE_SWITCH(F, ...)                   \
  if (e == 0) F<K, 0>(__VA_ARGS__);               \
  else if (e == 1) F<K, 1>(__VA_ARGS__);          \
  else F<K, 2>(__VA_ARGS__);

// ---- pre kernel: z = S_pre(r), res = r - T z
template <int K, int E>
__device__ __forceinline__ void pre3_stage1(const Geo& g, const DevTables& T, const CornerInfo& c, const Tile3Ctx& x, const double* __restrict__ r,
                                            double c0, double* rr, double* dd, double* Ds) {
  typedef TraceTile3<K> TT;
  constexpr int NL = TT::NL;
  if (!x.act) return;
  load_edge<NL, E>(r, g, c, rr);
  edge_dinv<NL, E>(T, c, c0, rr, dd);
#pragma unroll
  for (int m = 0; m < NL; m++) Ds[((EdgeOf<E>::PL * NL + m) * TT::H2 + x.lj) * TT::W2 + x.li] = dd[m];
}
template <int K, int E>
__device__ __forceinline__ void pre3_stage2(const Geo& g, const TileRows& tr, const DevTables& T, const CornerInfo& c, const Tile3Ctx& x, double c1,
                                            double c2, const double* rr, const double* dd, const double* Ds, double* Zs,
                                            double* __restrict__ z_out) {
  typedef TraceTile3<K> TT;
  constexpr int NL = TT::NL, W2 = TT::W2, H2 = TT::H2;
  if (!(x.act && x.li >= 1 && x.li < W2 - 1 && x.lj >= 1 && x.lj < H2 - 1)) return;
  double z[NL];
#pragma unroll
  for (int m = 0; m < NL; m++) z[m] = 0.0;
  if (c.exists) {
    double mine[NL], y[NL], r1[NL], zz[NL];
    lds_edge_stencil<K, E, W2, H2>(Ds, x.li, x.lj, c, T, mine, y);
#pragma unroll
    for (int m = 0; m < NL; m++) r1[m] = rr[m] - y[m];
    edge_dinv<NL, E>(T, c, c2, r1, zz);
#pragma unroll
    for (int m = 0; m < NL; m++) z[m] = dd[m] + fma(c1, dd[m], zz[m]);
  }
#pragma unroll
  for (int m = 0; m < NL; m++) Zs[((EdgeOf<E>::PL * NL + m) * TT::H1 + (x.lj - 1)) * TT::W1 + (x.li - 1)] = z[m];
  if (c.exists && c.own_x && x.li >= 2 && x.li < W2 - 2 && x.lj >= 2 && x.lj < H2 - 2 && x.jc < tr.jhi) store_edge<NL, E>(z_out, g, c, z);
}
template <int K, int E>
__device__ __forceinline__ void pre3_stage3(const Geo& g, const TileRows& tr, const DevTables& T, const CornerInfo& c, const Tile3Ctx& x,
                                            const double* rr, const double* Zs, double* __restrict__ res_out) {
  typedef TraceTile3<K> TT;
  constexpr int NL = TT::NL, W2 = TT::W2, H2 = TT::H2;
  if (!(x.act && x.li >= 2 && x.li < W2 - 2 && x.lj >= 2 && x.lj < H2 - 2 && c.exists && c.own_x && x.jc < tr.jhi)) return;
  double mine[NL], y[NL], res[NL];
  lds_edge_stencil<K, E, TT::W1, TT::H1>(Zs, x.li - 1, x.lj - 1, c, T, mine, y);
#pragma unroll
  for (int m = 0; m < NL; m++) res[m] = rr[m] - y[m];
  store_edge<NL, E>(res_out, g, c, res);
}
template <int K>
__global__ __launch_bounds__(TraceTile3<K>::NTHREADS) void k_trace_pre_tile3(int ntx, int nty, Geo g, TileRows tr, DevTables T, const double* __restrict__ r,
                                                                           double c0, double c1, double c2, double* __restrict__ z_out,
                                                                           double* __restrict__ res_out) {
  typedef TraceTile3<K> TT;
  constexpr int NL = TT::NL, NT = TT::NT;
  __shared__ double Ds[NT * TT::N2];  // d0 on the halo-2 region
  __shared__ double Zs[NT * TT::N1];  // z on the halo-1 region
  HDG_TILE_OF_BLOCK
  const int e = __builtin_amdgcn_readfirstlane((int)threadIdx.x / TT::NPER);  // edge type: uniform per wave (scalar branches)
  const int idx = (int)threadIdx.x - e * TT::NPER;
  Tile3Ctx x;
  x.i0 = tile_x * TT::TW; x.j0 = tr.jlo + tile_y * TT::TH;
  x.act = idx < TT::N2;
  x.lj = idx / TT::W2; x.li = idx - x.lj * TT::W2;
  x.jc = x.j0 - 2 + x.lj;
  const CornerInfo c = corner_info(g, tr, x.i0 - 2 + x.li, x.jc);
  double rr[NL], dd[NL];
  HDG_EDGE_SWITCH(pre3_stage1, g, T, c, x, r, c0, rr, dd, Ds)
  __syncthreads();
  HDG_EDGE_SWITCH(pre3_stage2, g, tr, T, c, x, c1, c2, rr, dd, Ds, Zs, z_out)
  __syncthreads();
  HDG_EDGE_SWITCH(pre3_stage3, g, tr, T, c, x, rr, Zs, res_out)
}

// ---- post kernel: z0 = z + P xc, two smoother steps, w = T z2, the five inner products
template <int K, int E>
__device__ __forceinline__ void post3_stage1(const Geo& g, const DevTables& T, const CornerInfo& c, const Tile3Ctx& x, double c0, double* r0,
                                             double* z1, double* d0, const double* Zs, double* Ds) {
  typedef TraceTile3<K> TT;
  constexpr int NL = TT::NL;
  if (!x.act) return;
#pragma unroll
  for (int m = 0; m < NL; m++) z1[m] = d0[m] = 0.0;  // (r0 holds r: zeros where the corner or the edge does not exist)
  if (c.exists) {
    double mine[NL], y[NL];
    lds_edge_stencil<K, E, TT::W3, TT::H3>(Zs, x.li + 1, x.lj + 1, c, T, mine, y);
#pragma unroll
    for (int m = 0; m < NL; m++) r0[m] -= y[m];
    edge_dinv<NL, E>(T, c, c0, r0, d0);
#pragma unroll
    for (int m = 0; m < NL; m++) z1[m] = mine[m] + d0[m];
  }
#pragma unroll
  for (int m = 0; m < NL; m++) Ds[((EdgeOf<E>::PL * NL + m) * TT::H2 + x.lj) * TT::W2 + x.li] = d0[m];
}
template <int K, int E>
__device__ __forceinline__ void post3_stage2(const Geo& g, const TileRows& tr, const DevTables& T, const CornerInfo& c, const Tile3Ctx& x, double c1,
                                             double c2, const double* r0, const double* z1, const double* d0, const double* Ds, double* Zs,
                                             double* __restrict__ z_out) {
  typedef TraceTile3<K> TT;
  constexpr int NL = TT::NL, W2 = TT::W2, H2 = TT::H2;
  if (!(x.act && x.li >= 1 && x.li < W2 - 1 && x.lj >= 1 && x.lj < H2 - 1)) return;
  double z2[NL];
#pragma unroll
  for (int m = 0; m < NL; m++) z2[m] = 0.0;
  if (c.exists) {
    double mine[NL], y[NL], r1[NL], zz[NL];
    lds_edge_stencil<K, E, W2, H2>(Ds, x.li, x.lj, c, T, mine, y);
#pragma unroll
    for (int m = 0; m < NL; m++) r1[m] = r0[m] - y[m];
    edge_dinv<NL, E>(T, c, c2, r1, zz);
#pragma unroll
    for (int m = 0; m < NL; m++) z2[m] = z1[m] + fma(c1, d0[m], zz[m]);
  }
#pragma unroll
  for (int m = 0; m < NL; m++) Zs[((EdgeOf<E>::PL * NL + m) * TT::H3 + (x.lj + 1)) * TT::W3 + (x.li + 1)] = z2[m];
  if (c.exists && c.own_x && x.li >= 2 && x.li < W2 - 2 && x.lj >= 2 && x.lj < H2 - 2 && x.jc < tr.jhi) store_edge<NL, E>(z_out, g, c, z2);
}
template <int K, int E, bool DOTS>
__device__ __forceinline__ void post3_stage3(const Geo& g, const DevTables& T, const CornerInfo& c, const Tile3Ctx& x, bool mine_here, double sE,
                                             const double* rr, const double* Zs, double* __restrict__ w_out, double* dots) {
  typedef TraceTile3<K> TT;
  constexpr int NL = TT::NL;
  if (!mine_here) return;
  double mine[NL], w[NL];
  lds_edge_stencil<K, E, TT::W3, TT::H3>(Zs, x.li + 1, x.lj + 1, c, T, mine, w);
  store_edge<NL, E>(w_out, g, c, w);
  if (DOTS) {
    // (z,n), (z,r), (z,z), (z,w), (n,r): the null vector n has sqrt(edge length) in mode 0 of every edge that exists;
    // mine[] (= z2 of this edge, from LDS) and rr[] are zero where the edge does not exist
    dots[0] = fma(sE, mine[0], dots[0]);
    dots[4] = fma(sE, rr[0], dots[4]);
#pragma unroll
    for (int m = 0; m < NL; m++) {
      dots[1] = fma(mine[m], rr[m], dots[1]);
      dots[2] = fma(mine[m], mine[m], dots[2]);
      dots[3] = fma(mine[m], w[m], dots[3]);
    }
  }
}
template <int K, int E>
__device__ __forceinline__ void post3_load_r(const Geo& g, const CornerInfo& c, bool on, const double* __restrict__ r, double* out) {
  CornerInfo cc = c;
  if (!on) cc.in_x = cc.in_y = false;
  load_edge<TraceTile3<K>::NL, E>(r, g, cc, out);
}
template <int K, bool DOTS>
__global__ __launch_bounds__(TraceTile3<K>::NTHREADS) void k_trace_post_tile3(int ntx, int nty, Geo g, TileRows tr, DevTables T, const double* __restrict__ z_in,
                                                                            const double* __restrict__ r, const double* __restrict__ xc, double sH,
                                                                            double sV, double sD, double c0, double c1, double c2,
                                                                            double* __restrict__ z_out, double* __restrict__ w_out,
                                                                            double* __restrict__ part) {
  typedef TraceTile3<K> TT;
  constexpr int NL = TT::NL, NT = TT::NT, W2 = TT::W2, H2 = TT::H2, W3 = TT::W3, H3 = TT::H3;
  __shared__ double Zs[NT * TT::N3];  // z0 on the halo-3 region; later z2 on its halo-1 part
  __shared__ double Ds[NT * TT::N2];  // d0 on the halo-2 region
  HDG_TILE_OF_BLOCK
  const int e = __builtin_amdgcn_readfirstlane((int)threadIdx.x / TT::NPER);
  const int idx = (int)threadIdx.x - e * TT::NPER;
  Tile3Ctx x;
  x.i0 = tile_x * TT::TW; x.j0 = tr.jlo + tile_y * TT::TH;
  x.act = idx < TT::N2;
  x.lj = idx / W2; x.li = idx - x.lj * W2;
  x.jc = x.j0 - 2 + x.lj;
  const int st = g.nx + 1;
  // stage 0: z0 = z + P xc on the halo-3 region, one (edge plane group, corner) item per thread and trip (pointwise)
  for (int item = threadIdx.x; item < 3 * TT::N3; item += TT::NTHREADS) {
    const int t = item / TT::N3, id3 = item - t * TT::N3;  // t: plane group (0 = H, 1 = V, 2 = D)
    const int lj = id3 / W3, li = id3 - lj * W3;
    const int i = x.i0 - 3 + li, j = x.j0 - 3 + lj;
    const CornerInfo c3 = corner_info(g, tr, i, j);
    const bool ok = t == 0 ? c3.in_x : (t == 1 ? c3.in_y : (c3.in_x && c3.in_y));
    double z[NL];
#pragma unroll
    for (int m = 0; m < NL; m++) z[m] = ok ? z_in[((long)t * NL + m) * g.G + c3.o] : 0.0;
    if (ok) {
      double va, vb;  // the vertex values at the a- and the b-end of the edge
      if (g.px) {
        const int n = g.nx, mm = g.ny;
        const int iw = i < 0 ? i + n : (i >= n ? i - n : i), i1 = iw + 1 == n ? 0 : iw + 1;
        const int jw = j < 0 ? j + mm : (j >= mm ? j - mm : j), j1 = jw + 1 == mm ? 0 : jw + 1;
        const double v00 = xc[(long)jw * n + iw];
        if (t == 0) { va = v00; vb = xc[(long)jw * n + i1]; }
        else if (t == 1) { va = v00; vb = xc[(long)j1 * n + iw]; }
        else { va = xc[(long)jw * n + i1]; vb = xc[(long)j1 * n + iw]; }
      } else {
        const long J = g.joff + j;  // xc is the global (replicated) vertex vector
        if (t == 0) { va = xc[J * st + i]; vb = xc[J * st + i + 1]; }
        else if (t == 1) { va = xc[J * st + i]; vb = xc[(J + 1) * st + i]; }
        else { va = xc[J * st + i + 1]; vb = xc[(J + 1) * st + i]; }
      }
      edge_prolong(va, vb, t == 0 ? sH : (t == 1 ? sV : sD), z);
    }
#pragma unroll
    for (int m = 0; m < NL; m++) Zs[((t * NL + m) * H3 + lj) * W3 + li] = z[m];
  }
  const CornerInfo c = corner_info(g, tr, x.i0 - 2 + x.li, x.jc);
  // r of this thread's edge: requested before the barrier (travels with the loads of stage 0)
  double r0[NL], z1[NL], d0[NL];
  HDG_EDGE_SWITCH(post3_load_r, g, c, x.act, r, r0)
  __syncthreads();
  HDG_EDGE_SWITCH(post3_stage1, g, T, c, x, c0, r0, z1, d0, Zs, Ds)
  __syncthreads();
  HDG_EDGE_SWITCH(post3_stage2, g, tr, T, c, x, c1, c2, r0, z1, d0, Ds, Zs, z_out)
  if (!w_out) return;  // uniform
  const bool mine_here = x.act && x.li >= 2 && x.li < W2 - 2 && x.lj >= 2 && x.lj < H2 - 2 && c.exists && c.own_x && x.jc < tr.jhi;
  double rr[NL];
  if (DOTS) {
    HDG_EDGE_SWITCH(post3_load_r, g, c, mine_here, r, rr)
  }
  __syncthreads();
  double dots[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  if (e == 0) post3_stage3<K, 0, DOTS>(g, T, c, x, mine_here, sH, rr, Zs, w_out, dots);
  else if (e == 1) post3_stage3<K, 1, DOTS>(g, T, c, x, mine_here, sD, rr, Zs, w_out, dots);
  else post3_stage3<K, 2, DOTS>(g, T, c, x, mine_here, sV, rr, Zs, w_out, dots);
  if (DOTS) {  // deterministic two-stage reduction (per workgroup here; the CG's scalar kernel sums over the workgroups)
    __shared__ double sm[TT::NTHREADS / 64][5];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 5; q++) {
      const double sv = wave_sum(dots[q]);
      if (lane == 0) sm[wv][q] = sv;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
      double sv = 0.0;
      for (int w2 = 0; w2 < TT::NTHREADS / 64; w2++) sv += sm[w2][threadIdx.x];
      part[(long)threadIdx.x * (ntx * nty) + tile_v] = sv;  // [inner product][tile]
    }
  }
}

}  // namespace hdg
