// Hand-written HIP kernels (gfx950 / CDNA4) for the HDG / HDG-IMEX timestep hot path.
//
// Data layout in HBM (all float64, modal coefficients in PHYSICALLY orthonormal bases):
//   cell vectors   v[(n * Nc) + c],  c = (s*ny + j)*nx + i   (dof-major, cell fastest;
//                  consecutive lanes <-> consecutive i  => every global access is coalesced)
//   pressure       n = m (Dubiner mode), 8 bytes per lane per access
//   velocity       v[(m * Nc + c) * 2 + d]  (mode-major, cell fastest, COMPONENT PAIR innermost): one lane reads or
//                  writes both components of a mode with ONE 16-byte access (buffer_load/store_dwordx4, 1 KiB per
//                  wave-instruction).  Measured on MI355X: 8-byte-per-lane streams top out at ~4.9-5.1 TB/s (the
//                  rate of a triad with double accesses), 16-byte ones at ~6.3 TB/s.  Inside a kernel the local
//                  arrays keep the order n = d*NU + m of the operator tables.
//   trace vectors  l[((t*NL + m) * G) + j*P + i],  t = 0 (H), 1 (V), 2 (D),  corner-indexed on a
//                  padded (ny+1) x P grid; entries that are not edges stay exactly zero.
// One thread owns one cell (or one grid corner = up to three edges); the element shape s is
// uniform per block (blockIdx.z), so every operator-table read is a wave-uniform scalar load
// (s_load_*) served by the scalar cache - local matrices never occupy vector registers or LDS
// bandwidth, and the FP64 FMA pipe sees one VGPR + one SGPR operand pair per instruction.
// Facet coupling is done by neighbour GATHER (owner computes), never by atomics, so results are
// bitwise reproducible.
#pragma once
#include "hdg_side_rows.hpp"
#include <hip/hip_runtime.h>

namespace hdg {

struct Geo {
  // Strip partition: a rank owns the cell rows joff .. joff+ny-1 of the global
  // nx x nyg mesh.  Every array carries GH = 6 GHOST rows below (j = -6 .. -1) and above (j = ny .. ny+5):
  //   cell index   c = (s*R + (j+GH))*nx + i              j in [-GH, ny+GH-1];  R >= ny+2GH rows per shape plane
  //   trace offset o = (j+GH)*P + i                        corner rows j in [-GH, ny+GH-1]
  // Depth 1 serves one row stencil.  The solvers chain stencils: a Chebyshev / GMRES iteration of the tentative
  // velocity is advection operator + edge lift (2), a preconditioned CG iteration of the trace system operator + two
  // smoother steps before and after the coarse correction (5).  With an input exchanged GH rows deep each stencil of
  // the chain runs on as many ghost rows as its inputs are still valid on (elo / ehi below, Engine::Flow) and its
  // result is valid there: one exchange per two velocity iterations and one per CG iteration instead of 4 and 5.
  // A rank computes the corner rows 0 .. nyc-1 (nyc = ny, or ny+1 on the topmost rank which also
  // owns the edges on the top boundary); row ny of the other ranks is a ghost copy of the upper
  // neighbour's row 0.  With one rank the ghost rows exist but are never referenced.
  int nx, ny, P;
  int nyg, joff, nyc;
  long G;   // (ny+2GH)*P : one trace plane
  long Nc;  // 2*nx*R: stride between dof planes of a cell vector
  int R;    // rows per shape plane of a cell array: ny + 2GH, plus padding rows that keep the plane stride away from
            // large powers of two (C3 with GH = 4: 2 R = 16 * 129 rows -> stride 2^18 * 129 B, advection + lift 25 % slower)
  double h;
  // XCD-aware block mapping (1-D grids): workgroups are dealt round-robin over the 8 XCDs, so
  // blockIdx % 8 labels the XCD.  Each XCD owns a contiguous band of mesh rows and walks it row by
  // row with both element shapes of a row adjacent in its sequence: the neighbour gathers (other
  // shape of the same square, rows j-1 / j+1) then hit that XCD's L2 instead of HBM.
  int nbx;       // blocks per row of cells
  int nbxc;      // blocks per row of corners
  int rows_xcd;  // cell rows per XCD band      = ceil(ny / 8)
  int rows_xcdc; // corner rows per XCD band    = ceil((ny+1) / 8)
  int dbg_nonbr; // experiment switch (HDG_DBG_NONBR, timing only): every edge is treated as a boundary edge
  // Doubly periodic square (PeriodicSquareMesh, driver.py:182-183; SURVEY.md section 8(f) row 2, first step).
  //   x: px != 0 -> column indices wrap (only lanes 0 / nx-1 take another address); corner kernels run nx columns.
  //   y: no kernel knows about it: the engine pretends that the strip lies in the middle of a taller mesh (joff = ny,
  //      nyg = 3 ny: every physical-boundary test in y is false) and fills the ghost rows from the opposite side of
  //      the strip before every stencil operator, exactly as it would from a neighbouring rank.
  int px;
  // THIS launch also computes elo ghost rows below and ehi above the owned rows (at most GH - 1; rows_xcd / rows_xcdc
  // of the copy cover the extended range): the stencil operators of the two solvers (Engine::Flow).
  int elo, ehi;
  // Row WINDOW of this launch (round 3: interior / boundary split, Engine::halo_overlap): the launch visits wrows cell rows
  // (wrowsc corner rows), r = 0 .. wrows-1  ->  j = r - elo + wskip, and rows r >= wgap0 are shifted by a further wgapn.
  // Default (the whole extended strip): wskip = 0, wgapn = 0, wrows = ny + elo + ehi, wrowsc = nyc + elo + ehi.
  // Interior launch: the rows whose stencil reads owned rows only; boundary launch: the rest, the gap spans the interior.
  int wskip, wgap0, wgapn, wrows, wrowsc;
};
__device__ __forceinline__ int launch_row(const Geo& g, int r) { return r - g.elo + g.wskip + ((g.wgapn && r >= g.wgap0) ? g.wgapn : 0); }
constexpr int GH = 6;  // ghost rows on either side of the strip in every cell / pressure / trace array (round 4: 4 -> 6, so that
                       // the LDS-tiled trace preconditioner runs on a strip with ONE exchange of r, 5 rows deep, per CG iteration)
constexpr int DX_DEFAULT = 4;  // depth of the velocity / trace exchanges of the row-stencil solvers (Engine::Flow)
__device__ __forceinline__ long rowbase(const Geo& g, int s, int j) { return ((long)s * g.R + (j + GH)) * g.nx; }
__device__ __forceinline__ int xm1(const Geo& g, int i) { return i > 0 ? i - 1 : g.nx - 1; }       // column to the left
__device__ __forceinline__ int xp1(const Geo& g, int i) { return (g.px && i == g.nx - 1) ? 0 : i + 1; }  // column to the right

template <int K>
struct Dim {
  static constexpr int NU = (K + 2) * (K + 3) / 2;
  static constexpr int NP = (K + 1) * (K + 2) / 2;
  static constexpr int NL = K + 1;
  static constexpr int NE = K + 2;
  static constexpr int NX = 2 * NU + NP;
  static constexpr int NT = 3 * NL;
};

struct DevTables {
  const double *N[2][3], *Nt[2][3], *Lift[2][3], *LiftT[2][3], *Pt[2][3];
  const double *B[2], *D0[2], *Ainv[2], *W[2], *Y[2], *SK[2];
  const double *cw, *cPhi[2], *cGx[2], *cGy[2];
  const double *ew[3], *ePhi[2][3], *eGx[2][3], *eGy[2][3];
  const double* trDinv[3][3];
  const double *Vu, *Vuinv, *Vp, *Vpinv, *Vl, *Vlinv;
  double elen[3], enx[3], eny[3], sig[2][3];
  double h, tau, alpha;
  int nqc, nqe;
};

#define HDG_CELL_PROLOGUE                                          \
  const int xcd_ = blockIdx.x & 7, q_ = blockIdx.x >> 3;           \
  const int jj_ = q_ / (2 * g.nbx), rem_ = q_ - jj_ * 2 * g.nbx;   \
  const int s = rem_ / g.nbx;                                      \
  const int i = (rem_ - s * g.nbx) * blockDim.x + threadIdx.x;     \
  const int r_ = xcd_ * g.rows_xcd + jj_;                          \
  const int j = launch_row(g, r_);                                 \
  if (jj_ >= g.rows_xcd || r_ >= g.wrows || i >= g.nx) return;     \
  const long c = rowbase(g, s, j) + i;

__device__ __forceinline__ bool nbr(int s, int e, int i, int j, const Geo& g, long& cn) {
  int in, jn;
  bool ok;
  if (g.dbg_nonbr) { cn = 0; return false; }  // cn = 0: a valid cell index
  if (s == 0) {
    if (e == 0) { in = i; jn = j - 1; ok = (g.joff + j) > 0; }
    else if (e == 1) { in = i; jn = j; ok = true; }
    else { in = xm1(g, i); jn = j; ok = i > 0 || g.px; }
  } else {
    if (e == 0) { in = i; jn = j + 1; ok = (g.joff + j) < g.nyg - 1; }
    else if (e == 1) { in = i; jn = j; ok = true; }
    else { in = xp1(g, i); jn = j; ok = i < g.nx - 1 || g.px; }
  }
  cn = rowbase(g, 1 - s, jn) + in;
  return ok;
}

// offset (within a trace plane) and type of local edge e of cell (s,i,j)
__device__ __forceinline__ long edge_off(int s, int e, int i, int j, const Geo& g, int& t) {
  if (e == 0) { t = 0; return (long)(j + s + GH) * g.P + i; }
  if (e == 1) { t = 2; return (long)(j + GH) * g.P + i; }
  t = 1;
  return (long)(j + GH) * g.P + (s ? xp1(g, i) : i);
}

__device__ __forceinline__ long cidx(const Geo& g, int s, int j, int i) {
  return rowbase(g, s, j) + i;
}

// Addressing of cell vectors v[n*Nc + c]: buffer loads / stores with the vector's base in a scalar resource
// descriptor, the dof-plane offset n*Nc*8 in a scalar register and ONE 32-bit lane offset c*8 shared by all
// planes.  With plain pointers the compiler materialises a 64-bit address per access in a VGPR pair (two
// registers per outstanding load; `global_load ... v[a:b], off` throughout the ISA), which is what pushed the
// element kernels to their register limits.  Offsets are 32 bit: a vector must stay below 4 GiB (checked on
// the host at engine construction).
typedef unsigned int hdg_u32x2 __attribute__((ext_vector_type(2)));
struct CellBuf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ explicit CellBuf(const double* p)
      : r(__builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0xFFFFFFFF, 0x00020000)) {}
  // plane offset in bytes (wave uniform), lane offset in bytes
  __device__ __forceinline__ double ld(unsigned plane_b, unsigned lane_b) const {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane_b, plane_b, 0));
  }
  __device__ __forceinline__ void st(unsigned plane_b, unsigned lane_b, double x) const {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(hdg_u32x2, x), r, lane_b, plane_b, 0);
  }
};
__device__ __forceinline__ unsigned plane_bytes(int n, long Nc) { return (unsigned)((unsigned long)n * (unsigned long)Nc * 8ul); }
template <int N>
__device__ __forceinline__ void load_cell(const double* __restrict__ v, long Nc, long c, double (&x)[N]) {
  const CellBuf B(v);
  const unsigned lane_b = (unsigned)c * 8u;
#pragma unroll
  for (int n = 0; n < N; n++) x[n] = B.ld(plane_bytes(n, Nc), lane_b);
}
template <int N>
__device__ __forceinline__ void store_cell(double* __restrict__ v, long Nc, long c, const double (&x)[N]) {
  const CellBuf B(v);
  const unsigned lane_b = (unsigned)c * 8u;
#pragma unroll
  for (int n = 0; n < N; n++) B.st(plane_bytes(n, Nc), lane_b, x[n]);
}

// Velocity vectors (component-pair layout, see the header): pair-plane m of cell c at double index (m*Nc + c)*2.
typedef double hdg_d2 __attribute__((ext_vector_type(2)));
typedef unsigned int hdg_u32x4 __attribute__((ext_vector_type(4)));
#ifndef HDG_NT_AUX
#define HDG_NT_AUX 2  // cache-policy bits of the buffer instructions on gfx94x/gfx950: bit 1 = nt
#endif
// cache-policy experiment switches.  k_adv_apply: bit 0 Q* loads, bit 1 b loads, bit 2 result
// stores non-temporal; k_edge_lift Chebyshev epilogue: bit 0 x_n loads, bit 1 x_{n-1} loads, bit 2 x_{n+1} stores
// Measured at C3 (micro-benchmark / whole step): k_adv_apply 313 -> 293 us with 7, k_edge_lift + Chebyshev 284 -> 243 us
// with 7; 147-149 -> 142 ms per step with both.
#ifndef HDG_ADV_NT
#define HDG_ADV_NT 7
#endif
#ifndef HDG_LIFT_NT
#define HDG_LIFT_NT 7
#endif
struct VelBuf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ explicit VelBuf(const double* p)
      : r(__builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0xFFFFFFFF, 0x00020000)) {}
  // pair-plane offset in bytes (wave uniform), lane offset in bytes (cell * 16)
  __device__ __forceinline__ hdg_d2 ld(unsigned plane_b, unsigned lane_b) const {
    return __builtin_bit_cast(hdg_d2, __builtin_amdgcn_raw_buffer_load_b128(r, lane_b, plane_b, 0));
  }
  // streaming (non-temporal) variants for data no other workgroup reads again before it has left the caches
  __device__ __forceinline__ hdg_d2 ld_nt(unsigned plane_b, unsigned lane_b) const {
    return __builtin_bit_cast(hdg_d2, __builtin_amdgcn_raw_buffer_load_b128(r, lane_b, plane_b, HDG_NT_AUX));
  }
  __device__ __forceinline__ void st_nt(unsigned plane_b, unsigned lane_b, hdg_d2 x) const {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(hdg_u32x4, x), r, lane_b + plane_b, 0, HDG_NT_AUX);
  }
  // Stores put the plane offset into the VECTOR offset (soffset = 0).  Measured on gfx950: a 16-byte buffer store
  // whose data registers are overwritten by the next VALU instruction can still read the new value (low dwords of
  // a few lanes: run-to-run differences of 2^-21 relative in k_adv_apply<1>); hipcc pads that hazard with wait
  // states only for stores WITHOUT a scalar offset register (its model: an SGPR offset makes the store safe).
  __device__ __forceinline__ void st(unsigned plane_b, unsigned lane_b, hdg_d2 x) const {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(hdg_u32x4, x), r, lane_b + plane_b, 0, 0);
  }
};
__device__ __forceinline__ unsigned pair_bytes(int m, long Nc) { return (unsigned)((unsigned long)m * (unsigned long)Nc * 16ul); }
// index of local dof n = d*NU + m of cell c (for the few 8-byte accesses that remain)
template <int NU>
__device__ __forceinline__ long vix(int n, long Nc, long c) {
  const int d = n >= NU ? 1 : 0, m = n - d * NU;
  return (((long)m * Nc + c) << 1) + d;
}
// x[m] = x-component, x[NU + m] = y-component of mode m
template <int NU>
__device__ __forceinline__ void load_vel(const double* __restrict__ v, long Nc, long c, double (&x)[2 * NU]) {
  const VelBuf B(v);
  const unsigned lane_b = (unsigned)c * 16u;
#pragma unroll
  for (int m = 0; m < NU; m++) {
    const hdg_d2 t = B.ld(pair_bytes(m, Nc), lane_b);
    x[m] = t.x;
    x[NU + m] = t.y;
  }
}
template <int NU>
__device__ __forceinline__ void load_vel_nt(const double* __restrict__ v, long Nc, long c, double (&x)[2 * NU]) {
  const VelBuf B(v);
  const unsigned lane_b = (unsigned)c * 16u;
#pragma unroll
  for (int m = 0; m < NU; m++) {
    const hdg_d2 t = B.ld_nt(pair_bytes(m, Nc), lane_b);
    x[m] = t.x;
    x[NU + m] = t.y;
  }
}
template <int NU>
__device__ __forceinline__ void store_vel_nt(double* __restrict__ v, long Nc, long c, const double (&x)[2 * NU]) {
  const VelBuf B(v);
  const unsigned lane_b = (unsigned)c * 16u;
#pragma unroll
  for (int m = 0; m < NU; m++) B.st_nt(pair_bytes(m, Nc), lane_b, hdg_d2{x[m], x[NU + m]});
}
template <int NU>
__device__ __forceinline__ void store_vel(double* __restrict__ v, long Nc, long c, const double (&x)[2 * NU]) {
  const VelBuf B(v);
  const unsigned lane_b = (unsigned)c * 16u;
#pragma unroll
  for (int m = 0; m < NU; m++) B.st(pair_bytes(m, Nc), lane_b, hdg_d2{x[m], x[NU + m]});
}

// y[r] += sc * sum_c A[r*NC + c] x[c]   (A wave-uniform -> scalar loads)
template <int NR, int NC>
__device__ __forceinline__ void mv_acc(const double* __restrict__ A, const double (&x)[NC], double (&y)[NR], double sc) {
#pragma unroll
  for (int r = 0; r < NR; r++) {
    double acc = 0.0;
#pragma unroll
    for (int cc = 0; cc < NC; cc++) acc = fma(A[r * NC + cc], x[cc], acc);
    y[r] = fma(sc, acc, y[r]);
  }
}
// same with a row stride LD >= NC (use the first NR rows / a column window of a larger matrix)
template <int NR, int NC>
__device__ __forceinline__ void mv_acc_ld(const double* __restrict__ A, int LD, const double* x, double* y, double sc) {
#pragma unroll
  for (int r = 0; r < NR; r++) {
    double acc = 0.0;
#pragma unroll
    for (int cc = 0; cc < NC; cc++) acc = fma(A[r * LD + cc], x[cc], acc);
    y[r] = fma(sc, acc, y[r]);
  }
}

// ------------------------------------------------------------------------------------------
// K1  edge_lift:  out_K = in_K + sum_e Out_e * w * (In_e^{K'} in_K' - In_e^{K} in_K)
//   w = 1/2 on interior edges; on boundary edges w = 1 and the neighbour term is absent.
//   With (In, Out) = (N, Lift) this is the BDM projection Q -> Q* (common.py:91-108);
//   with (In, Out) = (Lift^T, N^T) it is its transpose (used by the two-level preconditioner).
// ------------------------------------------------------------------------------------------
//   ADD_BJ = 1 (additive two-level preconditioner): result += Dinv_s * r_K  with a separate vector r.
//   ADD_BJ = 2 (hybrid two-level preconditioner M = Pi + Dinv (I - Pi), one kernel): with Pi x = x + sum_e
//              Lift_e d_e(x) this is  M x = x + sum_e G_e d_e(x),  G_e = (I - Dinv_s) Lift_e  (host tables per
//              shape and stage, passed through Dinv0 / Dinv1 as 3 consecutive 2NU x NE blocks): the conforming
//              part of the residual is kept, the non-conforming remainder goes through the element block-Jacobi,
//              at the cost of the plain projection.
//   Optional Chebyshev epilogue (chd != nullptr): with z the kernel's result, x_{n+1} = x_n + c1 (x_n - x_{n-1}) + c2 z
//   (chx = x_n, chd = x_{n-1} -> x_{n+1}); z itself is stored only if out != nullptr; cell_ss != nullptr: per-cell |z_K|^2.
// Register diet: the cell's own normal moments -N_e x are taken first, so that only ONE cell-sized array (y,
// initialised with x) stays live while the neighbours are visited.
// minimum waves/SIMD requested from the compiler; 1 = no constraint.  Measured at C3 (k = 2, hybrid + Chebyshev):
// unconstrained 154 VGPRs / 3 waves / no scratch: 0.329 ms; forced 4 waves (128 VGPRs, 92 B/lane scratch, +0.55 GB
// of HBM traffic per launch): 0.393 ms; forced 5 waves: 0.56 ms
#ifndef HDG_LIFT_WAVES
#define HDG_LIFT_WAVES 1
#endif
#ifndef HDG_LIFT_PIPE
#define HDG_LIFT_PIPE 1
#endif
// CHEB (compile time) = with the Chebyshev epilogue: its own instantiation and kernel NAME (profiles, counters).
template <int K, bool TRANSPOSE, int ADD_BJ, bool CHEB>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(HDG_LIFT_WAVES)))
void k_edge_lift(Geo g, DevTables T, const double* __restrict__ in,
                                                    double* __restrict__ out, const double* __restrict__ r,
                                                    const double* __restrict__ Dinv0, const double* __restrict__ Dinv1,
                                                    double* __restrict__ chd_, double* __restrict__ chx, double c1,
                                                    double c2, double* __restrict__ cell_ss) {
  constexpr int NU = Dim<K>::NU, NE = Dim<K>::NE, N2 = 2 * NU;
  double* __restrict__ chd = CHEB ? chd_ : nullptr;
  HDG_CELL_PROLOGUE
  double y[N2], down[3][NE];
  load_vel<NU>(in, g.Nc, c, y);
#pragma unroll
  for (int e = 0; e < 3; e++) {
#pragma unroll
    for (int a = 0; a < NE; a++) down[e][a] = 0.0;
    mv_acc<NE, N2>(TRANSPOSE ? T.LiftT[s][e] : T.N[s][e], y, down[e], -1.0);
  }
  if (ADD_BJ == 1) {
    double rr[N2];
    load_vel<NU>(r, g.Nc, c, rr);
    mv_acc<N2, N2>(s == 0 ? Dinv0 : Dinv1, rr, y, 1.0);
  }
  // Forms WITHOUT the Chebyshev epilogue (GMRES tail, BDM projection, transposed lift): the coefficients of neighbour e + 1 are
  // requested BEFORE the products of neighbour e (double buffer), so that the four groups of loads of a thread (own cell, three
  // neighbours) are not four dependent round trips: plain hybrid lift 207 -> 171 us at C3 (148 VGPRs, 3 waves / SIMD).  The form
  // with the fused Chebyshev step keeps the serial order: it needs its 4 waves (125 VGPRs) for the epilogue's streams and
  // loses 1 % with the double buffer (HDG_LIFT_PIPE=0: serial order everywhere).
  if constexpr (HDG_LIFT_PIPE && !CHEB && K <= 2) {  // (k >= 3: the matrix-core lift does the work; the per-thread forms keep their registers)
    long cnb[3];
    bool hasn[3];
#pragma unroll
    for (int e = 0; e < 3; e++) hasn[e] = nbr(s, e, i, j, g, cnb[e]);
    double xn[2][N2];
    if (hasn[0]) load_vel<NU>(in, g.Nc, cnb[0], xn[0]);
#pragma unroll
    for (int e = 0; e < 3; e++) {
      const double* __restrict__ Inb = TRANSPOSE ? T.LiftT[1 - s][e] : T.N[1 - s][e];
      const double* __restrict__ Out =
          ADD_BJ == 2 ? (s == 0 ? Dinv0 : Dinv1) + e * N2 * NE : (TRANSPOSE ? T.Nt[s][e] : T.Lift[s][e]);
      if (e + 1 < 3 && hasn[(e + 1) % 3]) load_vel<NU>(in, g.Nc, cnb[(e + 1) % 3], xn[(e + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (hasn[e]) {
        mv_acc<NE, N2>(Inb, xn[e & 1], down[e], 1.0);
#pragma unroll
        for (int a = 0; a < NE; a++) down[e][a] *= 0.5;
      }
      mv_acc<N2, NE>(Out, down[e], y, 1.0);
    }
  } else {
#pragma unroll
  for (int e = 0; e < 3; e++) {
    const double* __restrict__ Inb = TRANSPOSE ? T.LiftT[1 - s][e] : T.N[1 - s][e];
    const double* __restrict__ Out =
        ADD_BJ == 2 ? (s == 0 ? Dinv0 : Dinv1) + e * N2 * NE : (TRANSPOSE ? T.Nt[s][e] : T.Lift[s][e]);
    long cn;
    if (nbr(s, e, i, j, g, cn)) {
      double xn[N2];
      load_vel<NU>(in, g.Nc, cn, xn);
      mv_acc<NE, N2>(Inb, xn, down[e], 1.0);
#pragma unroll
      for (int a = 0; a < NE; a++) down[e][a] *= 0.5;
    }
    mv_acc<N2, NE>(Out, down[e], y, 1.0);
  }
  }
  if (out) store_vel<NU>(out, g.Nc, c, y);
  if (cell_ss) {
    // squared norm of this cell's part of the result (convergence checks: a 1/N2-sized array instead of the
    // whole vector goes through memory; summed deterministically by the multi-dot kernel)
    double ss = 0.0;
#pragma unroll
    for (int n = 0; n < N2; n++) ss = fma(y[n], y[n], ss);
    cell_ss[c] = ss;
  }
  if (chd) {
    // Chebyshev step in three-term form on two iterate buffers (no direction vector):
    //   x_{n+1} = x_n + c1 (x_n - x_{n-1}) + c2 z       chx = x_n (read only), chd = x_{n-1} on entry, x_{n+1} on exit
    // i.e. 2 reads + 1 write per entry instead of the 2 + 2 of  d = c1 d + c2 z, x += d  (d_{n-1} = x_n - x_{n-1}).
    // The c1 == 0 case (first step: x_{n-1} is not read) is decided ONCE, outside the element loop: with the test
    // inside, every element became its own branch -> load -> wait -> store block.  Each half issues its loads back
    // to back, then its stores.
    const bool rd = (c1 != 0.0);
    const VelBuf Bp(chd), Bx(chx);
    const unsigned lane_b = (unsigned)c * 16u;
    constexpr int NCH = NU >= 10 ? 2 : 1;  // mode pairs per chunk: two batches of 16-byte loads keep the register peak low
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      const int m_lo = ch * NU / NCH, m_hi = (ch + 1) * NU / NCH;
      hdg_d2 pp[NU], xx[NU];
#pragma unroll
      for (int m = 0; m < NU; m++)
        if (m >= m_lo && m < m_hi) xx[m] = (HDG_LIFT_NT & 1) ? Bx.ld_nt(pair_bytes(m, g.Nc), lane_b) : Bx.ld(pair_bytes(m, g.Nc), lane_b);
      if (rd) {
#pragma unroll
        for (int m = 0; m < NU; m++)
          if (m >= m_lo && m < m_hi) pp[m] = (HDG_LIFT_NT & 2) ? Bp.ld_nt(pair_bytes(m, g.Nc), lane_b) : Bp.ld(pair_bytes(m, g.Nc), lane_b);
      } else {
#pragma unroll
        for (int m = 0; m < NU; m++)
          if (m >= m_lo && m < m_hi) pp[m] = xx[m];
      }
#pragma unroll
      for (int m = 0; m < NU; m++)
        if (m >= m_lo && m < m_hi) {
          hdg_d2 xn1;
          xn1.x = fma(c1, xx[m].x - pp[m].x, fma(c2, y[m], xx[m].x));
          xn1.y = fma(c1, xx[m].y - pp[m].y, fma(c2, y[NU + m], xx[m].y));
          if (HDG_LIFT_NT & 4) Bp.st_nt(pair_bytes(m, g.Nc), lane_b, xn1); else Bp.st(pair_bytes(m, g.Nc), lane_b, xn1);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K1, paired form (k <= 2, 128-thread workgroups): BOTH triangles of 64 squares of a row in one workgroup -- wave 0 the
// lower-left cells (shape 0), wave 1 the upper-right ones (shape 1; the shape stays wave uniform: scalar table loads) -- and the
// two waves exchange the EDGE MOMENTS  m_e = In_e x  of their cells through LDS instead of gathering each other's coefficients:
//   * a cell's neighbours across edges 1 (same square) and 2 (the square to the left / right) live in the other wave: their
//     moments on the shared edge are what those cells compute for themselves anyway (local edge numbers of the two cells of an
//     edge agree), so 8 doubles per cell go through LDS where 40 came from global memory, and 2 of the 3 neighbour products
//     (In_e x_nbr, 80 of 800 FMAs each) are not repeated;
//   * only the neighbour across edge 0 (the row below / above) is gathered from global memory, and its loads are issued together
//     with the cell's own: ONE memory round trip per thread where the gather form has four dependent ones (ISA of the shipped
//     library: four load groups, each followed by its s_waitcnt chain; 45 % of the wave cycles waiting, profiles/r03h_pmc_probe.txt);
//   * the first / last live lane of a wave, whose edge-2 neighbour lies in the next workgroup, gathers that one cell the old way
//     (predicated: one lane; its loads are issued with the others).  Measured alternative: the wave forms that one moment together
//     (lanes 0 .. NU-1 load one pair each, per-lane table columns, butterfly + broadcast: 152 instead of 164 VGPRs) -- slower,
//     262 / 177 us instead of 237 / 126: the per-lane table loads are one more dependent round trip, and 152 VGPRs are 3 waves too.
// Same results as k_edge_lift up to the order of two additions per moment (m_nbr - m_own instead of -m_own + m_nbr: identical).
// ------------------------------------------------------------------------------------------
template <int K, bool TRANSPOSE, int ADD_BJ, bool CHEB>
__global__ __launch_bounds__(128) void k_edge_lift_pair(Geo g, DevTables T, const double* __restrict__ in, double* __restrict__ out,
                                                        const double* __restrict__ r, const double* __restrict__ Dinv0,
                                                        const double* __restrict__ Dinv1, double* __restrict__ chd_,
                                                        double* __restrict__ chx, double c1, double c2, double* __restrict__ cell_ss) {
  constexpr int NU = Dim<K>::NU, NE = Dim<K>::NE, N2 = 2 * NU;
  __shared__ double ms[2][2][NE][64];  // [shape][edge 1, 2][moment][lane]
  double* __restrict__ chd = CHEB ? chd_ : nullptr;
  const int xcd_ = blockIdx.x & 7, q_ = blockIdx.x >> 3;
  const int jj_ = q_ / (2 * g.nbx), rem_ = q_ - jj_ * 2 * g.nbx;  // rem_: block of 64 squares (2 nbx blocks per row, as before)
  const int s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const int i0 = rem_ * 64, i = i0 + lane;
  const int r_ = xcd_ * g.rows_xcd + jj_;
  const int j = launch_row(g, r_);
  if (jj_ >= g.rows_xcd || r_ >= g.wrows || i0 >= g.nx) return;  // whole workgroup
  const bool live = i < g.nx;
  const int ic = live ? i : g.nx - 1;  // clamped: loads stay in bounds, nothing is stored for dead lanes
  const long c = rowbase(g, s, j) + ic;
  const int ilast = min(i0 + 63, g.nx - 1);
  // neighbours: edge 0 (other row), edge 2 of the lane at the end of the wave's range (next workgroup)
  long cn0 = 0, cnx = 0;
  const bool has0 = !g.dbg_nonbr && nbr(s, 0, ic, j, g, cn0);
  const bool has2 = !g.dbg_nonbr && (s == 0 ? (ic > 0 || g.px) : (ic < g.nx - 1 || g.px));
  const bool outer = live && has2 && (s == 0 ? lane == 0 : ic == ilast);  // its edge-2 neighbour is not in this workgroup
  if (outer) { long t; nbr(s, 2, ic, j, g, t); cnx = t; }
  double y[N2], xn0[N2], xnx[N2], down[3][NE];
  load_vel<NU>(in, g.Nc, c, y);
#pragma unroll
  for (int n = 0; n < N2; n++) { xn0[n] = 0.0; xnx[n] = 0.0; }
  if (has0) load_vel<NU>(in, g.Nc, cn0, xn0);
  if (outer) load_vel<NU>(in, g.Nc, cnx, xnx);
  // own moments; edges 1 and 2 are published for the other wave
#pragma unroll
  for (int e = 0; e < 3; e++) {
#pragma unroll
    for (int a = 0; a < NE; a++) down[e][a] = 0.0;
    mv_acc<NE, N2>(TRANSPOSE ? T.LiftT[s][e] : T.N[s][e], y, down[e], 1.0);  // + m_own (sign applied below)
  }
#pragma unroll
  for (int a = 0; a < NE; a++) { ms[s][0][a][lane] = down[1][a]; ms[s][1][a][lane] = down[2][a]; }
  if (ADD_BJ == 1) {
    double rr[N2];
    load_vel<NU>(r, g.Nc, c, rr);
    mv_acc<N2, N2>(s == 0 ? Dinv0 : Dinv1, rr, y, 1.0);
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 3; e++) {
    const double* __restrict__ Inb = TRANSPOSE ? T.LiftT[1 - s][e] : T.N[1 - s][e];
    const double* __restrict__ Out =
        ADD_BJ == 2 ? (s == 0 ? Dinv0 : Dinv1) + e * N2 * NE : (TRANSPOSE ? T.Nt[s][e] : T.Lift[s][e]);
    double mn[NE];
    bool has;
    if (e == 0) {
      has = has0;
#pragma unroll
      for (int a = 0; a < NE; a++) mn[a] = 0.0;
      mv_acc<NE, N2>(Inb, xn0, mn, 1.0);
    } else if (e == 1) {
      has = !g.dbg_nonbr;
#pragma unroll
      for (int a = 0; a < NE; a++) mn[a] = ms[1 - s][0][a][lane];
    } else {
      has = has2;
      // shape 0: the upper-right cell of the square to the left; shape 1: the lower-left cell of the square to the right
      const int ln = s == 0 ? max(lane - 1, 0) : min(lane + 1, 63);
#pragma unroll
      for (int a = 0; a < NE; a++) mn[a] = ms[1 - s][1][a][ln];
      if (__builtin_amdgcn_ballot_w64(outer)) {  // at most one lane per wave: the neighbour in the next workgroup, gathered
        double mx[NE];
#pragma unroll
        for (int a = 0; a < NE; a++) mx[a] = 0.0;
        mv_acc<NE, N2>(Inb, xnx, mx, 1.0);
#pragma unroll
        for (int a = 0; a < NE; a++) mn[a] = outer ? mx[a] : mn[a];
      }
    }
#pragma unroll
    for (int a = 0; a < NE; a++) down[e][a] = has ? 0.5 * (mn[a] - down[e][a]) : -down[e][a];
    mv_acc<N2, NE>(Out, down[e], y, 1.0);
  }
  if (!live) return;
  if (out) store_vel<NU>(out, g.Nc, c, y);
  if (cell_ss) {
    double ss = 0.0;
#pragma unroll
    for (int n = 0; n < N2; n++) ss = fma(y[n], y[n], ss);
    cell_ss[c] = ss;
  }
  if (chd) {  // Chebyshev step in three-term form (see k_edge_lift)
    const bool rd = (c1 != 0.0);
    const VelBuf Bp(chd), Bx(chx);
    const unsigned lane_b = (unsigned)c * 16u;
    constexpr int NCH = NU >= 10 ? 2 : 1;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
      const int m_lo = ch * NU / NCH, m_hi = (ch + 1) * NU / NCH;
      hdg_d2 pp[NU], xx[NU];
#pragma unroll
      for (int m = 0; m < NU; m++)
        if (m >= m_lo && m < m_hi) xx[m] = (HDG_LIFT_NT & 1) ? Bx.ld_nt(pair_bytes(m, g.Nc), lane_b) : Bx.ld(pair_bytes(m, g.Nc), lane_b);
      if (rd) {
#pragma unroll
        for (int m = 0; m < NU; m++)
          if (m >= m_lo && m < m_hi) pp[m] = (HDG_LIFT_NT & 2) ? Bp.ld_nt(pair_bytes(m, g.Nc), lane_b) : Bp.ld(pair_bytes(m, g.Nc), lane_b);
      } else {
#pragma unroll
        for (int m = 0; m < NU; m++)
          if (m >= m_lo && m < m_hi) pp[m] = xx[m];
      }
#pragma unroll
      for (int m = 0; m < NU; m++)
        if (m >= m_lo && m < m_hi) {
          hdg_d2 xn1;
          xn1.x = fma(c1, xx[m].x - pp[m].x, fma(c2, y[m], xx[m].x));
          xn1.y = fma(c1, xx[m].y - pp[m].y, fma(c2, y[NU + m], xx[m].y));
          if (HDG_LIFT_NT & 4) Bp.st_nt(pair_bytes(m, g.Nc), lane_b, xn1); else Bp.st(pair_bytes(m, g.Nc), lane_b, xn1);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K1 on the matrix cores (k >= 3).  At high order the per-thread formulation above is limited by operand
// delivery (tables through scalar loads, 228-256 VGPRs, 0.8-1.6 TB/s); the same three contractions as MFMA:
//   (1) own normal moments      d   = -W x              W: packed (edge, moment) rows x 2NU
//   (2) neighbour moments       d_e += N'_e x_nbr(e)     then d *= 1/2 where the neighbour exists
//   (3) lifting                 y   = x + G d            G: 2NU x packed moments (Lift_e, or (I - Dinv) Lift_e)
// One wave owns 16 consecutive cells of one shape and row (the N dimension of v_mfma_f64_16x16x4); the
// coefficient planes are the B operands.  The K / M dimension over the 2NU velocity dofs runs in MEMORY order
// kappa = 2m + d (component pair innermost: lanes l/16 = 0,1 read the two halves of one 16-byte pair, so a
// wave-instruction covers two 256-byte segments; the host packs the table columns / rows in the same order); the tables are packed on the host in A-operand lane order (Engine::pack_lift_mfma), staged in LDS
// once per workgroup and read back one tile per MFMA (conflict-free ds_read_b64); d changes from accumulator
// to operand layout through a 4 KB LDS slab per wave.  Packed moment rows: tile 0 = edges 0 and 1 (rows e*NE+a),
// tile 1 = edge 2 (rows a).  A workgroup (8 waves sharing one LDS copy of the tables) handles one (row, shape);
// wave w the tiles w, w+8, ...
// ------------------------------------------------------------------------------------------
typedef double hdg_v4d __attribute__((ext_vector_type(4)));
// 16-byte access of the mode-m pair of cell c with a LANE-dependent mode (the plane offset goes into the vector offset;
// stores keep soffset = 0, the hazard-safe form: VelBuf::st)
template <int NU>
__device__ __forceinline__ hdg_d2 ld_pair_lane(const VelBuf& B, long Nc, long c, int m) {
  if (m >= NU) return hdg_d2{0.0, 0.0};
  return __builtin_bit_cast(hdg_d2, __builtin_amdgcn_raw_buffer_load_b128(B.r, (unsigned)(((unsigned long)m * (unsigned long)Nc + (unsigned long)c) * 16ul), 0, 0));
}
__device__ __forceinline__ void st_pair_lane(const VelBuf& B, long Nc, long c, int m, hdg_d2 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(hdg_u32x4, v), B.r, (unsigned)(((unsigned long)m * (unsigned long)Nc + (unsigned long)c) * 16ul), 0, 0);
}

// double index of velocity dof kappa = 2m + d of cell c (memory order of the component-pair layout)
__device__ __forceinline__ long kix(int kappa, long Nc, long c) { return (((long)(kappa >> 1) * Nc + c) << 1) + (kappa & 1); }
template <int K>
struct LiftMfma {
  static constexpr int NU = Dim<K>::NU, NE = Dim<K>::NE, N2 = 2 * NU;
  static constexpr int KQ = (NU + 3) / 4;    // K-step PAIRS over the coefficient planes (one 16-byte pair load each)
  static constexpr int KS = 2 * KQ;          // K-steps over the coefficient planes
  static constexpr int MT = (NU + 7) / 8;    // M-tiles of the result (8 modes = 16 rows each)
  static constexpr int KD = 5;               // K-steps over the packed moments: 3 (tile 0, rows 0..11) + 2 (tile 1, rows 0..7)
  static constexpr int NTILES = 2 * KS + 3 * KS + MT * KD;  // W (2 M-tiles), N'_e (3), G
  static_assert(2 * NE <= 12 && NE <= 8, "packed moment rows do not fit the K-steps");
  static_assert(2 * MT <= KQ + 1, "result tiles must map onto the loaded pairs");
};
#ifndef HDG_LIFT_MFMA_WAVES
#define HDG_LIFT_MFMA_WAVES 8
#endif
// Velocity dofs <-> K slots / result rows (round 3): 16-BYTE accesses.  K-steps come in pairs (2q, 2q+1): lane (lk, li)
// loads the pair of mode 4q + lk of cell li with one buffer_load_dwordx4 and feeds .x to K-step 2q, .y to 2q+1; result tile
// mt holds modes 8 mt .. 8 mt + 7 with both components of mode 8 mt + lk in accumulator registers 0, 1 and of mode
// 8 mt + 4 + lk in 2, 3 -- the SAME pairs (q = 2 mt, 2 mt + 1), so the loaded own coefficients are B operands and
// accumulator start values at once, and a tile is stored with two buffer_store_dwordx4 (Engine::scol / srow pack the
// tables accordingly).  Before: 8-byte accesses in memory order, 56 loads + 12 stores per cell tile; now 24 + 6.
// CHEB (its own instantiation and kernel name): the Chebyshev step of the tentative-velocity iteration in the store epilogue,
//   x_{n+1} = x_n + c1 (x_n - x_{n-1}) + c2 z   (chx = x_n, chd = x_{n-1} on entry and x_{n+1} on exit, z = this kernel's result,
// stored only when out != nullptr: the check points of the iteration) -- what the per-thread lift does at k <= 2; before, the
// matrix-core lift was followed by a separate vector kernel (k_cheb_update: 4 more passes over velocity vectors).
template <int K, bool CHEB = false>
__global__ __launch_bounds__(64 * HDG_LIFT_MFMA_WAVES) void k_edge_lift_mfma(Geo g, const double* __restrict__ tabs0, const double* __restrict__ tabs1,
                                                         const double* __restrict__ in, double* __restrict__ out,
                                                         double* __restrict__ chd = nullptr, const double* __restrict__ chx = nullptr,
                                                         double c1 = 0.0, double c2 = 0.0) {
  typedef LiftMfma<K> L;
  constexpr int NU = L::NU, NE = L::NE, KQ = L::KQ, KS = L::KS, MT = L::MT, KD = L::KD;
  __shared__ double tab[L::NTILES * 64];
  // per-wave slab for the moments: rows 0..11 = tile 0 (edges 0, 1), rows 12..19 = tile 1 rows 0..7 (edge 2)
  __shared__ double dst[HDG_LIFT_MFMA_WAVES][20][16];
  // (xcd band, row, shape) of this workgroup
  const int xcd_ = blockIdx.x & 7, q_ = blockIdx.x >> 3;
  const int jj_ = q_ >> 1, s = q_ & 1;
  const int r_ = xcd_ * g.rows_xcd + jj_;
  const int j = launch_row(g, r_);
  if (jj_ >= g.rows_xcd || r_ >= g.wrows) return;  // whole workgroup
  const double* __restrict__ tsrc = s == 0 ? tabs0 : tabs1;
  for (int p = threadIdx.x; p < L::NTILES * 64; p += 64 * HDG_LIFT_MFMA_WAVES) tab[p] = tsrc[p];
  __syncthreads();
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const double* __restrict__ tW = tab;                     // [2][KS][64]
  const double* __restrict__ tN = tab + 2 * KS * 64;       // [3][KS][64]
  const double* __restrict__ tG = tab + 5 * KS * 64;       // [MT][KD][64]
  const VelBuf Bin(in), Bout(out), Bp(CHEB ? chd : out), Bx(CHEB ? chx : in);
  const int gj = g.joff + j;
  const bool has0 = s == 0 ? gj > 0 : gj < g.nyg - 1;
  const int jn0 = s == 0 ? j - 1 : j + 1;
  const long rowN0 = rowbase(g, 1 - s, jn0), rowN = rowbase(g, 1 - s, j);
  const long rowC = rowbase(g, s, j);
  const int ntx = (g.nx + 15) >> 4;
  const hdg_d2 zero2 = {0.0, 0.0};
  for (int tx = w; tx < ntx; tx += HDG_LIFT_MFMA_WAVES) {
    const int i = tx * 16 + li;
    const bool col = i < g.nx;
    const int ic = col ? i : g.nx - 1;  // clamped: loads stay in bounds, results of invalid columns are not stored
    const bool has2 = s == 0 ? i > 0 : i < g.nx - 1;
    const int i2 = s == 0 ? ic - 1 : ic + 1;
    const long c = rowC + ic, cn0 = rowN0 + ic, cn1 = rowN + ic, cn2 = rowN + (has2 ? i2 : ic);
    // all coefficient pairs of the tile: own cell and the three neighbours (a missing neighbour contributes zeros)
    hdg_d2 xo[KQ], x0[KQ], x1[KQ], x2[KQ];
#pragma unroll
    for (int q = 0; q < KQ; q++) {
      const int m = 4 * q + lk;
      xo[q] = ld_pair_lane<NU>(Bin, g.Nc, c, m);
      x0[q] = has0 ? ld_pair_lane<NU>(Bin, g.Nc, cn0, m) : zero2;
      x1[q] = ld_pair_lane<NU>(Bin, g.Nc, cn1, m);
      x2[q] = has2 ? ld_pair_lane<NU>(Bin, g.Nc, cn2, m) : zero2;
    }
    // (1) own moments, (2) neighbours: edges 0, 1 -> tile 0, edge 2 -> tile 1
    hdg_v4d D0 = {0, 0, 0, 0}, D1 = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int q = ks >> 1;
      const double b = (ks & 1) ? xo[q].y : xo[q].x;
      D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(tW[(0 * KS + ks) * 64 + l], b, D0, 0, 0, 0);
      D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tW[(1 * KS + ks) * 64 + l], b, D1, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const int q = ks >> 1;
      D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(tN[(0 * KS + ks) * 64 + l], (ks & 1) ? x0[q].y : x0[q].x, D0, 0, 0, 0);
      D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(tN[(1 * KS + ks) * 64 + l], (ks & 1) ? x1[q].y : x1[q].x, D0, 0, 0, 0);
      D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tN[(2 * KS + ks) * 64 + l], (ks & 1) ? x2[q].y : x2[q].x, D1, 0, 0, 0);
    }
    // weights: 1/2 where the neighbour exists.  Accumulator layout (measured): lane (lk, li) register r holds
    // row lk + 4 r of column li
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = lk + 4 * r;
      D0[r] *= (row < NE) ? (has0 ? 0.5 : 1.0) : 0.5;  // rows NE .. 2NE-1: edge 1 always has its neighbour
      D1[r] *= has2 ? 0.5 : 1.0;
      if (row < 12) dst[w][row][li] = D0[r];
      if (row < 8) dst[w][12 + row][li] = D1[r];
    }
    // the slab is private to the wave: LDS operations of one wave execute in order, a wave-level barrier keeps the
    // compiler from moving the reads above the writes
    __builtin_amdgcn_wave_barrier();
    // (3) lifting: y = x + G d, d as B operand: K-steps 0..2 = tile 0 rows 0..11, 3..4 = tile 1 rows 0..7
    double bd[KD];
#pragma unroll
    for (int kd = 0; kd < KD; kd++) bd[kd] = dst[w][4 * kd + lk][li];  // K index q: 0..11 tile 0, 12..19 tile 1
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const hdg_d2 lo = xo[2 * mt], hi = (2 * mt + 1 < KQ) ? xo[(2 * mt + 1 < KQ) ? 2 * mt + 1 : 0] : zero2;
      hdg_v4d Y = {lo.x, lo.y, hi.x, hi.y};
      const int m0 = 8 * mt + lk, m1 = m0 + 4;
      hdg_d2 xa = zero2, xb = zero2, pa = zero2, pb = zero2;
      if (CHEB) {  // requested before the lifting products (clamped column: in bounds)
        xa = ld_pair_lane<NU>(Bx, g.Nc, c, m0); xb = ld_pair_lane<NU>(Bx, g.Nc, c, m1);
        if (c1 != 0.0) { pa = ld_pair_lane<NU>(Bp, g.Nc, c, m0); pb = ld_pair_lane<NU>(Bp, g.Nc, c, m1); }
        else { pa = xa; pb = xb; }
      }
#pragma unroll
      for (int kd = 0; kd < KD; kd++) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(tG[(mt * KD + kd) * 64 + l], bd[kd], Y, 0, 0, 0);
      if (col) {
        if (!CHEB || out) {
          if (m0 < NU) st_pair_lane(Bout, g.Nc, c, m0, hdg_d2{Y[0], Y[1]});
          if (m1 < NU) st_pair_lane(Bout, g.Nc, c, m1, hdg_d2{Y[2], Y[3]});
        }
        if (CHEB) {
          const hdg_d2 na = {fma(c1, xa.x - pa.x, fma(c2, Y[0], xa.x)), fma(c1, xa.y - pa.y, fma(c2, Y[1], xa.y))};
          const hdg_d2 nb = {fma(c1, xb.x - pb.x, fma(c2, Y[2], xb.x)), fma(c1, xb.y - pb.y, fma(c2, Y[3], xb.y))};
          if (m0 < NU) st_pair_lane(Bp, g.Nc, c, m0, na);
          if (m1 < NU) st_pair_lane(Bp, g.Nc, c, m1, nb);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K3  advection apply:  y = x - gamma * F(Q*) x     with F = f_impl (hdg_imex.py:313-331):
//   F_K(x)[w] = sum_{e int} int_e [ (1/2)(Q*.n_K) - up |Q*.n_K| ] (x_K - x_K').w
//             - int_K w.((Q*.grad) x)  -  sum_e alpha/h_F int_e ((x_K - x_K').n)(w.n)
//   (x_K' := 0 on boundary edges; Q*.n = 0 there).  Quadrature: cell rule exact to 3k+2, edge rule
//   ceil((3k+4)/2) Gauss points (the |Q*.n| integrand is not polynomial: SURVEY.md App. D.3).
// ------------------------------------------------------------------------------------------
//   Optional residual epilogue: with bsub != nullptr the kernel writes  bsub - (x - gamma F x).
#ifndef HDG_ADV_WAVES
#define HDG_ADV_WAVES 1
#endif
// facet terms of one edge: F += sum_q Po[q] * flux(q)   (xn: coefficients of the neighbour across edge e, zero if none)
template <int K>
__device__ __forceinline__ void adv_facet(const DevTables& T, int s, int e, bool has, double upwind, const double (&x)[2 * Dim<K>::NU],
                                          const double (&qs)[2 * Dim<K>::NU], const double (&xn)[2 * Dim<K>::NU],
                                          double (&F)[2 * Dim<K>::NU]) {
  constexpr int NU = Dim<K>::NU;
  const double* __restrict__ Po = T.ePhi[s][e];
  const double* __restrict__ Pn = T.ePhi[1 - s][e];
  const double nx_ = T.enx[e], ny_ = T.eny[e], sg = T.sig[s][e];
  const double pen = T.alpha / T.elen[e];
  const int nq = T.nqe;
#pragma unroll 1
  for (int q = 0; q < nq; q++) {
    double ox = 0, oy = 0, bx = 0, by = 0, qn = 0;
#pragma unroll
    for (int m = 0; m < NU; m++) {
      const double po = Po[q * NU + m], pn = Pn[q * NU + m];
      ox = fma(po, x[m], ox);
      oy = fma(po, x[NU + m], oy);
      bx = fma(pn, xn[m], bx);
      by = fma(pn, xn[NU + m], by);
      qn = fma(po, fma(nx_, qs[m], ny_ * qs[NU + m]), qn);
    }
    if (!has) bx = by = 0.0;  // boundary edge: xn holds whatever the clamped address delivered
    const double w = T.ew[e][q];
    const double cf = has ? w * (0.5 * sg * qn - upwind * fabs(qn)) : 0.0;
    const double jx = ox - bx, jy = oy - by;
    const double jn = (jx * nx_ + jy * ny_) * pen * w;
    const double vx = cf * jx - jn * nx_;
    const double vy = cf * jy - jn * ny_;
#pragma unroll
    for (int m = 0; m < NU; m++) {
      const double po = Po[q * NU + m];
      F[m] = fma(po, vx, F[m]);
      F[NU + m] = fma(po, vy, F[NU + m]);
    }
  }
}
// Software-pipelined neighbour loads (PIPE): the kernel sits at 2 waves/SIMD whatever it does (x, Q*, F and one
// neighbour array are 160 VGPRs), so the ~50 registers below the 256 limit buy a SECOND neighbour buffer: the
// coefficients of the next edge's neighbour (and finally b) are requested before the arithmetic of the current
// edge starts, instead of each edge exposing its own memory latency.  The neighbour loads are UNCONDITIONAL (the
// neighbour index of a boundary edge points into a ghost row or an adjacent cell, always inside the vector; the
// trace is zeroed instead): with a load inside `if (has)` the compiler must assume the shorter queue on the other
// path, and its s_waitcnt vmcnt(N) for the own coefficients then also waits for the prefetched ones.
#ifndef HDG_ADV_PIPE
#define HDG_ADV_PIPE 1
#endif
// RESID (compile time) = the residual form b - A x: a separate instantiation, so that profiles and counters tell the two
// forms of the kernel apart by NAME (k_adv_apply<2, true> reads 3 vectors and writes 1, <2, false> reads 2).
template <int K, bool RESID>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(HDG_ADV_WAVES)))
void k_adv_apply(Geo g, DevTables T, const double* __restrict__ xin,
                                                    const double* __restrict__ qstar, double* __restrict__ out,
                                                    double gamma, double upwind, const double* __restrict__ bsub_) {
  constexpr int NU = Dim<K>::NU, N2 = 2 * NU;
  const double* __restrict__ bsub = RESID ? bsub_ : nullptr;
  HDG_CELL_PROLOGUE
  double x[N2], qs[N2], F[N2];
  load_vel<NU>(xin, g.Nc, c, x);
#if HDG_ADV_NT & 1
  load_vel_nt<NU>(qstar, g.Nc, c, qs);
#else
  load_vel<NU>(qstar, g.Nc, c, qs);
#endif
  long cn0, cn1, cn2;
  const bool has0 = nbr(s, 0, i, j, g, cn0), has1 = nbr(s, 1, i, j, g, cn1), has2 = nbr(s, 2, i, j, g, cn2);
  double xa[N2], xb[N2];
  // K = 2 only (measured at nx = 1024, residual form: 290.5 -> 286.6 us; K = 1 loses a wave per SIMD, 152 instead of 126
  // VGPRs: 137.7 -> 140.6 us; K >= 3 is at the register cap and runs on the matrix cores by default)
  constexpr bool PIPE = HDG_ADV_PIPE != 0 && K == 2;
  if (PIPE) load_vel<NU>(xin, g.Nc, cn0, xa);
#pragma unroll
  for (int n = 0; n < N2; n++) F[n] = 0.0;
  // ---- cell term
  {
    const double* __restrict__ Phi = T.cPhi[s];
    const double* __restrict__ Gx = T.cGx[s];
    const double* __restrict__ Gy = T.cGy[s];
    const int nq = T.nqc;
#pragma unroll 1
    for (int q = 0; q < nq; q++) {
      double qx = 0, qy = 0, dxx = 0, dxy = 0, dyx = 0, dyy = 0;  // dab = d_b x_a
#pragma unroll
      for (int m = 0; m < NU; m++) {
        const double ph = Phi[q * NU + m], gx = Gx[q * NU + m], gy = Gy[q * NU + m];
        qx = fma(ph, qs[m], qx);
        qy = fma(ph, qs[NU + m], qy);
        dxx = fma(gx, x[m], dxx);
        dxy = fma(gy, x[m], dxy);
        dyx = fma(gx, x[NU + m], dyx);
        dyy = fma(gy, x[NU + m], dyy);
      }
      const double w = T.cw[q];
      const double ax = -w * (qx * dxx + qy * dxy);
      const double ay = -w * (qx * dyx + qy * dyy);
#pragma unroll
      for (int m = 0; m < NU; m++) {
        const double ph = Phi[q * NU + m];
        F[m] = fma(ph, ax, F[m]);
        F[NU + m] = fma(ph, ay, F[NU + m]);
      }
    }
  }
  // ---- facet terms
  if (PIPE) {
    load_vel<NU>(xin, g.Nc, cn1, xb);
    adv_facet<K>(T, s, 0, has0, upwind, x, qs, xa, F);
    load_vel<NU>(xin, g.Nc, cn2, xa);
    adv_facet<K>(T, s, 1, has1, upwind, x, qs, xb, F);
    if (bsub) {
#if HDG_ADV_NT & 2
      load_vel_nt<NU>(bsub, g.Nc, c, xb);
#else
      load_vel<NU>(bsub, g.Nc, c, xb);
#endif
    }
    adv_facet<K>(T, s, 2, has2, upwind, x, qs, xa, F);
  } else {
    load_vel<NU>(xin, g.Nc, cn0, xa);
    adv_facet<K>(T, s, 0, has0, upwind, x, qs, xa, F);
    load_vel<NU>(xin, g.Nc, cn1, xa);
    adv_facet<K>(T, s, 1, has1, upwind, x, qs, xa, F);
    load_vel<NU>(xin, g.Nc, cn2, xa);
    adv_facet<K>(T, s, 2, has2, upwind, x, qs, xa, F);
    if (bsub) {
#if HDG_ADV_NT & 2
      load_vel_nt<NU>(bsub, g.Nc, c, xb);
#else
      load_vel<NU>(bsub, g.Nc, c, xb);
#endif
    }
  }
  if (bsub) {
#pragma unroll
    for (int n = 0; n < N2; n++) F[n] = xb[n] - fma(-gamma, F[n], x[n]);
  } else {
#pragma unroll
    for (int n = 0; n < N2; n++) F[n] = fma(-gamma, F[n], x[n]);
  }
#if HDG_ADV_NT & 4
  store_vel_nt<NU>(out, g.Nc, c, F);
#else
  store_vel<NU>(out, g.Nc, c, F);
#endif
}

// ------------------------------------------------------------------------------------------
// K3 in PAIRED form (round 4; k <= 2, 128-thread workgroups; the mapping of k_edge_lift_pair): a workgroup holds BOTH
// triangles of 64 squares of a row -- wave 0 the lower-left cells, wave 1 the upper-right ones; the shape stays wave uniform,
// so every table is still a scalar load.  The facet terms need the NEIGHBOUR'S trace at the edge quadrature points; across
// edges 1 (same square) and 2 (the square to the left / right) the neighbour lives in the other wave, and its trace at those
// points is what that cell evaluates for itself anyway (the local edge numbers of the two cells of an edge agree, and both
// tabulate the edge at the same physical points in the same order: k_adv_apply evaluates the neighbour with ePhi[1-s][e], the
// neighbour's own table).  So the two waves exchange 2 edges x NQE points x 2 components through LDS instead of gathering each
// other's 2 NU coefficients and repeating the 2 NU NQE products: per cell 40 of the 100 sixteen-byte loads and 200 of ~2 800
// FMAs (k = 2) go away.  Only the neighbour across edge 0 (the row below / above) is gathered, its loads issued with the cell's
// own; the first / last live lane of a wave, whose edge-2 neighbour lies in the next workgroup, gathers that one cell into the
// SAME registers once edge 0 is done (the cell term covers the latency).  Same arithmetic per output entry as k_adv_apply (the
// neighbour trace is the same sum, formed by the other thread); order of the facet terms: 0, 2, 1.
// ------------------------------------------------------------------------------------------
template <int K, bool RESID>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(HDG_ADV_WAVES)))
void k_adv_pair(Geo g, DevTables T, const double* __restrict__ xin, const double* __restrict__ qstar, double* __restrict__ out,
                double gamma, double upwind, const double* __restrict__ bsub_) {
  constexpr int NU = Dim<K>::NU, N2 = 2 * NU, NQE = (3 * K + 5) / 2;
  __shared__ double tr[2][2][2 * NQE][64];  // [shape][edge 1, 2][x-trace at the NQE points, y-trace][lane]
  const double* __restrict__ bsub = RESID ? bsub_ : nullptr;
  const int xcd_ = blockIdx.x & 7, q_ = blockIdx.x >> 3;
  const int jj_ = q_ / (2 * g.nbx), rem_ = q_ - jj_ * 2 * g.nbx;
  const int s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  const int i0 = rem_ * 64, i = i0 + lane;
  const int r_ = xcd_ * g.rows_xcd + jj_;
  const int j = launch_row(g, r_);
  if (jj_ >= g.rows_xcd || r_ >= g.wrows || i0 >= g.nx) return;  // whole workgroup
  const bool live = i < g.nx;
  const int ic = live ? i : g.nx - 1;  // clamped: loads stay in bounds, nothing is stored for dead lanes
  const long c = rowbase(g, s, j) + ic;
  const int ilast = min(i0 + 63, g.nx - 1);
  long cn0 = 0, cnx = 0;
  const bool has0 = nbr(s, 0, ic, j, g, cn0);
  const bool has2 = !g.dbg_nonbr && (s == 0 ? (ic > 0 || g.px) : (ic < g.nx - 1 || g.px));
  const bool outer = live && has2 && (s == 0 ? lane == 0 : ic == ilast);  // its edge-2 neighbour is not in this workgroup
  if (outer) { long t; nbr(s, 2, ic, j, g, t); cnx = t; }
  double x[N2], qs[N2], F[N2], xa[N2];
  load_vel<NU>(xin, g.Nc, c, x);
#if HDG_ADV_NT & 1
  load_vel_nt<NU>(qstar, g.Nc, c, qs);
#else
  load_vel<NU>(qstar, g.Nc, c, qs);
#endif
  load_vel<NU>(xin, g.Nc, cn0, xa);  // unconditional (a boundary edge points into a ghost row: inside the vector)
  // own traces at the quadrature points of edges 1 and 2: published for the other wave
#pragma unroll
  for (int e = 1; e < 3; e++) {
    const double* __restrict__ Po = T.ePhi[s][e];
#pragma unroll 1
    for (int q = 0; q < NQE; q++) {
      double ox = 0, oy = 0;
#pragma unroll
      for (int m = 0; m < NU; m++) {
        const double po = Po[q * NU + m];
        ox = fma(po, x[m], ox);
        oy = fma(po, x[NU + m], oy);
      }
      tr[s][e - 1][q][lane] = ox;
      tr[s][e - 1][NQE + q][lane] = oy;
    }
  }
#pragma unroll
  for (int n = 0; n < N2; n++) F[n] = 0.0;
  // ---- facet term of edge 0 (gathered neighbour)
  adv_facet<K>(T, s, 0, has0, upwind, x, qs, xa, F);
  // the one lane whose edge-2 neighbour lies in the next workgroup gathers it now (xa is free; consumed after the cell term)
  if (outer) load_vel<NU>(xin, g.Nc, cnx, xa);
  // ---- cell term
  {
    const double* __restrict__ Phi = T.cPhi[s];
    const double* __restrict__ Gx = T.cGx[s];
    const double* __restrict__ Gy = T.cGy[s];
    const int nq = T.nqc;
#pragma unroll 1
    for (int q = 0; q < nq; q++) {
      double qx = 0, qy = 0, dxx = 0, dxy = 0, dyx = 0, dyy = 0;  // dab = d_b x_a
#pragma unroll
      for (int m = 0; m < NU; m++) {
        const double ph = Phi[q * NU + m], gx = Gx[q * NU + m], gy = Gy[q * NU + m];
        qx = fma(ph, qs[m], qx);
        qy = fma(ph, qs[NU + m], qy);
        dxx = fma(gx, x[m], dxx);
        dxy = fma(gy, x[m], dxy);
        dyx = fma(gx, x[NU + m], dyx);
        dyy = fma(gy, x[NU + m], dyy);
      }
      const double w = T.cw[q];
      const double ax = -w * (qx * dxx + qy * dxy);
      const double ay = -w * (qx * dyx + qy * dyy);
#pragma unroll
      for (int m = 0; m < NU; m++) {
        const double ph = Phi[q * NU + m];
        F[m] = fma(ph, ax, F[m]);
        F[NU + m] = fma(ph, ay, F[NU + m]);
      }
    }
  }
  __syncthreads();
  // ---- facet terms of edges 2 and 1: the neighbour's trace from LDS
#pragma unroll
  for (int ee = 0; ee < 2; ee++) {
    const int e = 2 - ee;
    const double* __restrict__ Po = T.ePhi[s][e];
    const double* __restrict__ Pn = T.ePhi[1 - s][e];
    const double nx_ = T.enx[e], ny_ = T.eny[e], sg = T.sig[s][e];
    const double pen = T.alpha / T.elen[e];
    const bool has = e == 1 ? !g.dbg_nonbr : has2;
    // edge 2: shape 0 faces the upper-right cell of the square to the left, shape 1 the lower-left cell of the square to the right
    const int ln = e == 1 ? lane : (s == 0 ? max(lane - 1, 0) : min(lane + 1, 63));
    const bool any_outer = e == 2 && __builtin_amdgcn_ballot_w64(outer) != 0;  // wave uniform
#pragma unroll 1
    for (int q = 0; q < NQE; q++) {
      const double ox = tr[s][e - 1][q][lane], oy = tr[s][e - 1][NQE + q][lane];
      double bx = tr[1 - s][e - 1][q][ln], by = tr[1 - s][e - 1][NQE + q][ln];
      double qn = 0;
#pragma unroll
      for (int m = 0; m < NU; m++) qn = fma(Po[q * NU + m], fma(nx_, qs[m], ny_ * qs[NU + m]), qn);
      if (any_outer) {  // at most one lane per wave: its neighbour's trace from the gathered coefficients
        double gx_ = 0, gy_ = 0;
#pragma unroll
        for (int m = 0; m < NU; m++) {
          const double pn = Pn[q * NU + m];
          gx_ = fma(pn, xa[m], gx_);
          gy_ = fma(pn, xa[NU + m], gy_);
        }
        bx = outer ? gx_ : bx;
        by = outer ? gy_ : by;
      }
      if (!has) bx = by = 0.0;
      const double w = T.ew[e][q];
      const double cf = has ? w * (0.5 * sg * qn - upwind * fabs(qn)) : 0.0;
      const double jx = ox - bx, jy = oy - by;
      const double jn = (jx * nx_ + jy * ny_) * pen * w;
      const double vx = cf * jx - jn * nx_;
      const double vy = cf * jy - jn * ny_;
#pragma unroll
      for (int m = 0; m < NU; m++) {
        const double po = Po[q * NU + m];
        F[m] = fma(po, vx, F[m]);
        F[NU + m] = fma(po, vy, F[NU + m]);
      }
    }
    if (e == 2 && bsub) {  // xa is free now: the right-hand side of the residual form, requested before the last facet term
#if HDG_ADV_NT & 2
      load_vel_nt<NU>(bsub, g.Nc, c, xa);
#else
      load_vel<NU>(bsub, g.Nc, c, xa);
#endif
    }
  }
  if (!live) return;
  if (bsub) {
#pragma unroll
    for (int n = 0; n < N2; n++) F[n] = xa[n] - fma(-gamma, F[n], x[n]);
  } else {
#pragma unroll
    for (int n = 0; n < N2; n++) F[n] = fma(-gamma, F[n], x[n]);
  }
#if HDG_ADV_NT & 4
  store_vel_nt<NU>(out, g.Nc, c, F);
#else
  store_vel<NU>(out, g.Nc, c, F);
#endif
}

// ------------------------------------------------------------------------------------------
// K3, two lanes per cell (used for k >= 3): lane parity = velocity component a.  Every table of this kernel
// (cell / edge basis tabulations) is the same for both components, so the table operands stay wave uniform
// (SGPR); only two quantities couple the components and cross the lane pair with one DPP swap each:
// Q* at the quadrature points (each lane evaluates its own component) and the normal jump of the penalty.
// Per lane: x_a, Q*_a, F_a, neighbour x_a = 4 NU doubles instead of 8 NU: 120 instead of 240+ VGPRs at
// k = 3 (the one-lane kernel sits at the 256-VGPR cap with AGPR spills, 1 wave/SIMD).  Same arithmetic
// per output entry as k_adv_apply, in the same order, except for the two pairwise sums.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double pair_swap(double v) { return __shfl_xor(v, 1, 64); }
template <int K>
__global__ __launch_bounds__(128) void k_adv_apply2(Geo g, DevTables T, const double* __restrict__ xin,
                                                     const double* __restrict__ qstar, double* __restrict__ out,
                                                     double gamma, double upwind, const double* __restrict__ bsub) {
  constexpr int NU = Dim<K>::NU;
  // cell prologue with blockDim.x / 2 cells per workgroup (same XCD-aware row walk as HDG_CELL_PROLOGUE)
  const int cpb = blockDim.x >> 1, nbx2 = (g.nx + cpb - 1) / cpb;
  const int xcd_ = blockIdx.x & 7, q_ = blockIdx.x >> 3;
  const int jj_ = q_ / (2 * nbx2), rem_ = q_ - jj_ * 2 * nbx2;
  const int s = rem_ / nbx2;
  const int i = (rem_ - s * nbx2) * cpb + (threadIdx.x >> 1);
  const int a = threadIdx.x & 1;  // velocity component of this lane
  const int r_ = xcd_ * g.rows_xcd + jj_;
  const int j = launch_row(g, r_);
  if (jj_ >= g.rows_xcd || r_ >= g.wrows || i >= g.nx) return;  // both lanes of a pair leave together
  const long c = rowbase(g, s, j) + i;
  // lane parity = component = the slot inside a mode's 16-byte pair: a lane pair reads one pair per mode
  const long cbase = (c << 1) + a;
  const long pstride = g.Nc << 1;  // doubles between consecutive modes
  double x[NU], qs[NU], F[NU];
#pragma unroll
  for (int m = 0; m < NU; m++) { x[m] = xin[cbase + (long)m * pstride]; qs[m] = qstar[cbase + (long)m * pstride]; F[m] = 0.0; }
  // ---- cell term:  F_a[m] -= w Phi[q,m] (Q*.grad) x_a
  {
    const double* __restrict__ Phi = T.cPhi[s];
    const double* __restrict__ Gx = T.cGx[s];
    const double* __restrict__ Gy = T.cGy[s];
    const int nq = T.nqc;
#pragma unroll 1
    for (int q = 0; q < nq; q++) {
      double qa = 0, dx_ = 0, dy_ = 0;
#pragma unroll
      for (int m = 0; m < NU; m++) {
        qa = fma(Phi[q * NU + m], qs[m], qa);
        dx_ = fma(Gx[q * NU + m], x[m], dx_);
        dy_ = fma(Gy[q * NU + m], x[m], dy_);
      }
      const double qb = pair_swap(qa);
      const double qx = a == 0 ? qa : qb, qy = a == 0 ? qb : qa;
      const double av = -T.cw[q] * (qx * dx_ + qy * dy_);
#pragma unroll
      for (int m = 0; m < NU; m++) F[m] = fma(Phi[q * NU + m], av, F[m]);
    }
  }
  // ---- facet terms
#pragma unroll
  for (int e = 0; e < 3; e++) {
    long cn;
    const bool has = nbr(s, e, i, j, g, cn);
    double xn[NU];
    if (has) {
#pragma unroll
      for (int m = 0; m < NU; m++) xn[m] = xin[(((long)m * g.Nc + cn) << 1) + a];
    } else {
#pragma unroll
      for (int m = 0; m < NU; m++) xn[m] = 0.0;
    }
    const double* __restrict__ Po = T.ePhi[s][e];
    const double* __restrict__ Pn = T.ePhi[1 - s][e];
    const double na = a == 0 ? T.enx[e] : T.eny[e], sg = T.sig[s][e];
    const double pen = T.alpha / T.elen[e];
    const int nq = T.nqe;
#pragma unroll 1
    for (int q = 0; q < nq; q++) {
      double oa = 0, ba = 0, qna = 0;
#pragma unroll
      for (int m = 0; m < NU; m++) {
        const double po = Po[q * NU + m];
        oa = fma(po, x[m], oa);
        ba = fma(Pn[q * NU + m], xn[m], ba);
        qna = fma(po, na * qs[m], qna);
      }
      const double qn = qna + pair_swap(qna);  // Q*.n: both components
      const double w = T.ew[e][q];
      const double cf = has ? w * (0.5 * sg * qn - upwind * fabs(qn)) : 0.0;
      const double ja = oa - ba;
      const double jna = ja * na;
      const double jn = (jna + pair_swap(jna)) * pen * w;  // normal jump: both components
      const double va = cf * ja - jn * na;
#pragma unroll
      for (int m = 0; m < NU; m++) F[m] = fma(Po[q * NU + m], va, F[m]);
    }
  }
  if (bsub) {
#pragma unroll
    for (int m = 0; m < NU; m++) out[cbase + (long)m * pstride] = bsub[cbase + (long)m * pstride] - fma(-gamma, F[m], x[m]);
  } else {
#pragma unroll
    for (int m = 0; m < NU; m++) out[cbase + (long)m * pstride] = fma(-gamma, F[m], x[m]);
  }
}

// ------------------------------------------------------------------------------------------
// K3 on the matrix cores (k >= 3): k_adv_mfma.  Mapping of k_edge_lift_mfma (one wave per 16 cells, tables packed on
// the host in A-operand lane order -- Engine::pack_adv_mfma -- and staged in LDS once per workgroup).
// Cell term, per block of 16 quadrature points: six contractions over the NU basis functions (Q*_x, Q*_y = Phi q*,
// d_x x_a, d_y x_a = Gx / Gy x_a) as MFMA chains with the coefficient planes as B operands, the pointwise product on
// the accumulators, an LDS slab per wave to turn the result into a B operand, and the test contraction with
// A2[m][q] = -w_q Phi[q][m] accumulated over the blocks.  Facet terms: see the kernel's header comment.
// ------------------------------------------------------------------------------------------
template <int K>
struct AdvMfma {
  static constexpr int NU = Dim<K>::NU;
  static constexpr int NQ = (K == 2) ? 16 : ((K == 3) ? 36 : 64);  // cell quadrature points (Tables::nqc; k = 2: experiment only)
  static constexpr int MTQ = (NQ + 15) / 16, KSU = (NU + 3) / 4, MTU = (NU + 15) / 16;
  static constexpr int NT1 = MTQ * KSU;            // tiles per stage-1 table
  static constexpr int NTILES = 3 * NT1 + MTU * 4 * MTQ;
};
// The whole operator in ONE kernel: cell term as described above, then the facet terms in the same mapping.  Edge-point rows are packed 8 per edge: tile 0 = edges 0 and 1 (rows 8 e + q), tile 1 = edge 2 (rows q).
//   own traces / Q*.n   OX, OY = EO x_a ;  QN = EQX q*_x + EQY q*_y        (EQX/EQY = n_x / n_y * Po)
//   neighbour traces    NBX, NBY += EN_e x_a(neighbour e)                   (EN_e: Pn rows of edge e, zero elsewhere)
//   flux (pointwise)    v = cf (own - nbr) - pen w ((own - nbr).n) n        -> LDS slab -> B operand
//   test                F += ET v                                            (ET[m][(e,q)] = Po_e[q][m])
//   result              out = x - gamma F   or   b - (x - gamma F)
// Workgroup = 8 waves sharing one LDS copy of the tables; a wave owns 16 consecutive cells per trip.
template <int K>
struct AdvMfmaFull {
  typedef AdvMfma<K> C;
  static constexpr int NU = C::NU, MTQ = C::MTQ, KSU = C::KSU, MTU = C::MTU, NT1 = C::NT1;
  static constexpr int NQE = (3 * K + 5) / 2;
  static constexpr int OFF_EO = C::NTILES, OFF_EQX = OFF_EO + 2 * KSU, OFF_EQY = OFF_EQX + 2 * KSU;
  static constexpr int OFF_EN = OFF_EQY + 2 * KSU, OFF_ET = OFF_EN + 3 * KSU, NTILES = OFF_ET + MTU * 6;
  static_assert(NQE <= 8, "edge rule does not fit the 8-row packing");
};
template <int K, bool RESID>  // RESID: residual form b - A x (own kernel name, see k_adv_apply)
__global__ __launch_bounds__(512) void k_adv_mfma(Geo g, DevTables T, const double* __restrict__ tabs0,
                                                   const double* __restrict__ tabs1, const double* __restrict__ xin,
                                                   const double* __restrict__ qstar, double* __restrict__ out, double gamma,
                                                   double upwind, const double* __restrict__ bsub_) {
  typedef AdvMfmaFull<K> A;
  const double* __restrict__ bsub = RESID ? bsub_ : nullptr;
  constexpr int NU = A::NU, MTQ = A::MTQ, KSU = A::KSU, MTU = A::MTU, NT1 = A::NT1, NQE = A::NQE;
  __shared__ double tab[A::NTILES * 64];
  __shared__ double slab[8][2][24][16];
  const int xcd_ = blockIdx.x & 7, q_ = blockIdx.x >> 3;
  const int jj_ = q_ >> 1, s = q_ & 1;
  const int r_ = xcd_ * g.rows_xcd + jj_;
  const int j = launch_row(g, r_);
  if (jj_ >= g.rows_xcd || r_ >= g.wrows) return;  // whole workgroup
  const double* __restrict__ tsrc = s == 0 ? tabs0 : tabs1;
  for (int p = threadIdx.x; p < A::NTILES * 64; p += 512) tab[p] = tsrc[p];
  __syncthreads();
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, lk = l >> 4;
  const double* __restrict__ tPhi = tab;
  const double* __restrict__ tGx = tab + NT1 * 64;
  const double* __restrict__ tGy = tab + 2 * NT1 * 64;
  const double* __restrict__ tA2 = tab + 3 * NT1 * 64;
  const double* __restrict__ tEO = tab + A::OFF_EO * 64;
  const double* __restrict__ tEQX = tab + A::OFF_EQX * 64;
  const double* __restrict__ tEQY = tab + A::OFF_EQY * 64;
  const double* __restrict__ tEN = tab + A::OFF_EN * 64;
  const double* __restrict__ tET = tab + A::OFF_ET * 64;
  // per-lane constants of the edge-point rows this lane holds in the accumulator layout (row = lk + 4 r)
  double cw[2][4], cnx[2][4], cny[2][4], csg[2][4], cpen[2][4];
  int ce[2][4];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = lk + 4 * r, e = t == 0 ? (row >> 3) : 2, q = row & 7;
      const bool valid = q < NQE && (t == 0 || row < 8);
      ce[t][r] = valid ? e : -1;
      cw[t][r] = valid ? T.ew[e][q] : 0.0;
      cnx[t][r] = T.enx[e];
      cny[t][r] = T.eny[e];
      csg[t][r] = T.sig[s][e];
      cpen[t][r] = T.alpha / T.elen[e];
    }
  const int gj = g.joff + j;
  const bool has0 = s == 0 ? gj > 0 : gj < g.nyg - 1;
  const int jn0 = s == 0 ? j - 1 : j + 1;
  const long rowN0 = rowbase(g, 1 - s, jn0), rowN = rowbase(g, 1 - s, j);
  const long rowC = rowbase(g, s, j);
  const int ntx = (g.nx + 15) >> 4;
  for (int tx = w; tx < ntx; tx += 8) {
    const int i = tx * 16 + li;
    const bool col = i < g.nx;
    const int ic = col ? i : g.nx - 1;
    const bool has2 = s == 0 ? i > 0 : i < g.nx - 1;
    const int i2 = s == 0 ? ic - 1 : ic + 1;
    const long c = rowC + ic, cn0 = rowN0 + ic, cn1 = rowN + ic, cn2 = rowN + (has2 ? i2 : ic);
    double bq0[KSU], bq1[KSU], bx0[KSU], bx1[KSU];
#pragma unroll
    for (int ks = 0; ks < KSU; ks++) {
      const int m = 4 * ks + lk;
      const bool mv = m < NU;
      // one 16-byte access per mode: both components (4 pair-planes x 16 cells = four 256-byte segments per instruction)
      const hdg_d2 tq = mv ? *reinterpret_cast<const hdg_d2*>(qstar + (((long)m * g.Nc + c) << 1)) : hdg_d2{0.0, 0.0};
      const hdg_d2 tx = mv ? *reinterpret_cast<const hdg_d2*>(xin + (((long)m * g.Nc + c) << 1)) : hdg_d2{0.0, 0.0};
      bq0[ks] = tq.x; bq1[ks] = tq.y;
      bx0[ks] = tx.x; bx1[ks] = tx.y;
    }
    hdg_v4d F[2][MTU];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int mu = 0; mu < MTU; mu++) F[a][mu] = hdg_v4d{0, 0, 0, 0};
    // ---- cell term
#pragma unroll
    for (int mt = 0; mt < MTQ; mt++) {
      hdg_v4d QX = {0, 0, 0, 0}, QY = QX, DXX = QX, DXY = QX, DYX = QX, DYY = QX;
#pragma unroll
      for (int ks = 0; ks < KSU; ks++) {
        const double ap = tPhi[(mt * KSU + ks) * 64 + l], ax_ = tGx[(mt * KSU + ks) * 64 + l], ay_ = tGy[(mt * KSU + ks) * 64 + l];
        QX = __builtin_amdgcn_mfma_f64_16x16x4f64(ap, bq0[ks], QX, 0, 0, 0);
        QY = __builtin_amdgcn_mfma_f64_16x16x4f64(ap, bq1[ks], QY, 0, 0, 0);
        DXX = __builtin_amdgcn_mfma_f64_16x16x4f64(ax_, bx0[ks], DXX, 0, 0, 0);
        DXY = __builtin_amdgcn_mfma_f64_16x16x4f64(ay_, bx0[ks], DXY, 0, 0, 0);
        DYX = __builtin_amdgcn_mfma_f64_16x16x4f64(ax_, bx1[ks], DYX, 0, 0, 0);
        DYY = __builtin_amdgcn_mfma_f64_16x16x4f64(ay_, bx1[ks], DYY, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        slab[w][0][lk + 4 * r][li] = QX[r] * DXX[r] + QY[r] * DXY[r];
        slab[w][1][lk + 4 * r][li] = QX[r] * DYX[r] + QY[r] * DYY[r];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const double b0 = slab[w][0][4 * kk + lk][li], b1 = slab[w][1][4 * kk + lk][li];
#pragma unroll
        for (int mu = 0; mu < MTU; mu++) {
          const double a2 = tA2[(mu * 4 * MTQ + 4 * mt + kk) * 64 + l];
          F[0][mu] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b0, F[0][mu], 0, 0, 0);
          F[1][mu] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b1, F[1][mu], 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    // ---- facet terms: traces at the edge points
    hdg_v4d OX[2], OY[2], QN[2], NBX[2], NBY[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      OX[t] = OY[t] = QN[t] = NBX[t] = NBY[t] = hdg_v4d{0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < KSU; ks++) {
        const double eo = tEO[(t * KSU + ks) * 64 + l];
        OX[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(eo, bx0[ks], OX[t], 0, 0, 0);
        OY[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(eo, bx1[ks], OY[t], 0, 0, 0);
        QN[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(tEQX[(t * KSU + ks) * 64 + l], bq0[ks], QN[t], 0, 0, 0);
        QN[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(tEQY[(t * KSU + ks) * 64 + l], bq1[ks], QN[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KSU; ks++) {
      const int m = 4 * ks + lk;
      const bool mv = m < NU;
      const hdg_d2 zz = {0.0, 0.0};
      const hdg_d2 t0 = (mv && has0) ? *reinterpret_cast<const hdg_d2*>(xin + (((long)m * g.Nc + cn0) << 1)) : zz;
      const hdg_d2 t1 = mv ? *reinterpret_cast<const hdg_d2*>(xin + (((long)m * g.Nc + cn1) << 1)) : zz;
      const hdg_d2 t2 = (mv && has2) ? *reinterpret_cast<const hdg_d2*>(xin + (((long)m * g.Nc + cn2) << 1)) : zz;
      const double n00 = t0.x, n01 = t0.y, n10 = t1.x, n11 = t1.y, n20 = t2.x, n21 = t2.y;
      const double e0 = tEN[(0 * KSU + ks) * 64 + l], e1 = tEN[(1 * KSU + ks) * 64 + l], e2 = tEN[(2 * KSU + ks) * 64 + l];
      NBX[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(e0, n00, NBX[0], 0, 0, 0);
      NBY[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(e0, n01, NBY[0], 0, 0, 0);
      NBX[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(e1, n10, NBX[0], 0, 0, 0);
      NBY[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(e1, n11, NBY[0], 0, 0, 0);
      NBX[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(e2, n20, NBX[1], 0, 0, 0);
      NBY[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(e2, n21, NBY[1], 0, 0, 0);
    }
    // pointwise flux on the accumulators -> slab rows 0..15 (tile 0), 16..23 (tile 1 rows 0..7)
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = lk + 4 * r, e = ce[t][r];
        const bool has = e == 0 ? has0 : (e == 2 ? has2 : true);
        const double wq = cw[t][r], qn = QN[t][r];
        const double cf = has ? wq * (0.5 * csg[t][r] * qn - upwind * fabs(qn)) : 0.0;
        const double jx = OX[t][r] - NBX[t][r], jy = OY[t][r] - NBY[t][r];
        const double jn = (jx * cnx[t][r] + jy * cny[t][r]) * cpen[t][r] * wq;
        const double vx = e >= 0 ? cf * jx - jn * cnx[t][r] : 0.0, vy = e >= 0 ? cf * jy - jn * cny[t][r] : 0.0;
        if (t == 0 || row < 8) {
          slab[w][0][16 * t + row][li] = vx;
          slab[w][1][16 * t + row][li] = vy;
        }
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kd = 0; kd < 6; kd++) {
      const double b0 = slab[w][0][4 * kd + lk][li], b1 = slab[w][1][4 * kd + lk][li];
#pragma unroll
      for (int mu = 0; mu < MTU; mu++) {
        const double et = tET[(mu * 6 + kd) * 64 + l];
        F[0][mu] = __builtin_amdgcn_mfma_f64_16x16x4f64(et, b0, F[0][mu], 0, 0, 0);
        F[1][mu] = __builtin_amdgcn_mfma_f64_16x16x4f64(et, b1, F[1][mu], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- result: both components of a mode in one 16-byte access
#pragma unroll
    for (int mu = 0; mu < MTU; mu++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = 16 * mu + lk + 4 * r;
        if (m < NU && col) {
          const long idx = ((long)m * g.Nc + c) << 1;
          const hdg_d2 xo = *reinterpret_cast<const hdg_d2*>(xin + idx);
          hdg_d2 v;
          v.x = fma(-gamma, F[0][mu][r], xo.x);
          v.y = fma(-gamma, F[1][mu][r], xo.y);
          if (bsub) {
            const hdg_d2 bb = *reinterpret_cast<const hdg_d2*>(bsub + idx);
            v.x = bb.x - v.x;
            v.y = bb.y - v.y;
          }
          *reinterpret_cast<hdg_d2*>(out + idx) = v;
        }
      }
  }
}
// ------------------------------------------------------------------------------------------
// K4  element block-Jacobi:  out = cz * zin + Dinv_s * r      (Dinv: 2NU x 2NU per shape)
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(128) void k_blockdiag(Geo g, const double* __restrict__ Dinv0,
                                                    const double* __restrict__ Dinv1, const double* __restrict__ r,
                                                    const double* __restrict__ zin, double cz, double* __restrict__ out) {
  constexpr int NU = Dim<K>::NU, N2 = 2 * NU;
  HDG_CELL_PROLOGUE
  double x[N2], y[N2];
  load_vel<NU>(r, g.Nc, c, x);
  if (zin) {
    load_vel<NU>(zin, g.Nc, c, y);
#pragma unroll
    for (int n = 0; n < N2; n++) y[n] *= cz;
  } else {
#pragma unroll
    for (int n = 0; n < N2; n++) y[n] = 0.0;
  }
  mv_acc<N2, N2>(s == 0 ? Dinv0 : Dinv1, x, y, 1.0);
  store_vel<NU>(out, g.Nc, c, y);
}

// ------------------------------------------------------------------------------------------
// pressure gradient:  out = ca*a + cb*b + gamma * ( B^T p - sum_e C_e^T lambda_e )
//   g(w,p,lambda) = (p, div w)_K - <lambda, w.n_K>_{dK}   (hdg_imex.py:333-340)
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(128) void k_pgrad(Geo g, DevTables T, const double* __restrict__ a, double ca,
                                                const double* __restrict__ b, double cb, const double* __restrict__ p,
                                                const double* __restrict__ lam, double gamma, double* __restrict__ out) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, NL = Dim<K>::NL, N2 = 2 * NU;
  HDG_CELL_PROLOGUE
  double y[N2], pp[NP], acc[N2];
  {
    double ta[N2], tb[N2];
    if (a) load_vel<NU>(a, g.Nc, c, ta);
    if (b) load_vel<NU>(b, g.Nc, c, tb);
#pragma unroll
    for (int n = 0; n < N2; n++) {
      y[n] = (a ? ca * ta[n] : 0.0) + (b ? cb * tb[n] : 0.0);
      acc[n] = 0.0;
    }
  }
  load_cell<NP>(p, g.Nc, c, pp);
  const double* __restrict__ Bm = T.B[s];
#pragma unroll
  for (int n = 0; n < N2; n++) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < NP; r++) v = fma(Bm[r * N2 + n], pp[r], v);
    acc[n] = v;
  }
#pragma unroll
  for (int e = 0; e < 3; e++) {
    int t;
    const long off = edge_off(s, e, i, j, g, t);
    double l[NL];
#pragma unroll
    for (int m = 0; m < NL; m++) l[m] = lam[((long)t * NL + m) * g.G + off];
    const double* __restrict__ Nm = T.N[s][e];
    const double sg = T.sig[s][e];
#pragma unroll
    for (int n = 0; n < N2; n++) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < NL; m++) v = fma(Nm[m * N2 + n], l[m], v);
      acc[n] = fma(-sg, v, acc[n]);
    }
  }
#pragma unroll
  for (int n = 0; n < N2; n++) y[n] = fma(gamma, acc[n], y[n]);
  store_vel<NU>(out, g.Nc, c, y);
}

// ------------------------------------------------------------------------------------------
// weak divergence (hdg_imex.py:353-365):  out = sc * [ -(grad psi, Q)_K + <psi, {{Q}}.n>_{dK, int} ]
//   BROKEN = true gives sc * (psi, div Q)_K instead (hdg_implicit.py:145)
// ------------------------------------------------------------------------------------------
template <int K, bool BROKEN>
__global__ __launch_bounds__(128) void k_weak_div(Geo g, DevTables T, const double* __restrict__ q, double sc,
                                                   double* __restrict__ out) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, NL = Dim<K>::NL, N2 = 2 * NU;
  HDG_CELL_PROLOGUE
  double x[N2], y[NP];
  load_vel<NU>(q, g.Nc, c, x);
#pragma unroll
  for (int r = 0; r < NP; r++) y[r] = 0.0;
  if (BROKEN) {
    mv_acc<NP, N2>(T.B[s], x, y, 1.0);
  } else {
    mv_acc<NP, N2>(T.D0[s], x, y, 1.0);
#pragma unroll
    for (int e = 0; e < 3; e++) {
      long cn;
      if (nbr(s, e, i, j, g, cn)) {
        double xn[N2], tr[NL];
        load_vel<NU>(q, g.Nc, cn, xn);
#pragma unroll
        for (int m = 0; m < NL; m++) tr[m] = 0.0;
        mv_acc_ld<NL, N2>(T.N[s][e], N2, x, tr, 0.5);
        mv_acc_ld<NL, N2>(T.N[1 - s][e], N2, xn, tr, 0.5);
        const double* __restrict__ Pm = T.Pt[s][e];
        const double sg = T.sig[s][e];
#pragma unroll
        for (int r = 0; r < NP; r++) {
          double v = 0.0;
#pragma unroll
          for (int m = 0; m < NL; m++) v = fma(Pm[m * NP + r], tr[m], v);
          y[r] = fma(sg, v, y[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NP; r++) out[(long)r * g.Nc + c] = sc * y[r];
}

// ------------------------------------------------------------------------------------------
// corner-thread helpers for the trace space
// ------------------------------------------------------------------------------------------
#define HDG_CORNER_PROLOGUE                                        \
  const int xcd_ = blockIdx.x & 7, q_ = blockIdx.x >> 3;           \
  const int jj_ = q_ / g.nbxc;                                     \
  const int i = (q_ - jj_ * g.nbxc) * blockDim.x + threadIdx.x;    \
  const int r_ = xcd_ * g.rows_xcdc + jj_;                         \
  const int j = launch_row(g, r_);                                 \
  if (jj_ >= g.rows_xcdc || r_ >= g.wrowsc || i > g.nx - g.px) return; \
  const long o = (long)(j + GH) * g.P + i;                        \
  const bool in_x = i < g.nx, in_y = j < g.ny + g.ehi;  /* an extended launch only reaches rows that exist globally */ \
  const bool below = (g.joff + j) > 0;                             \
  const bool left = i > 0 || g.px;          /* a cell column to the left exists */ \
  const long oL = (long)(j + GH) * g.P + xm1(g, i), oR = (long)(j + GH) * g.P + xp1(g, i); /* corners (i-1, j), (i+1, j) */ \
  (void)below; (void)left; (void)oL; (void)oR;

template <int NL>
__device__ __forceinline__ void load_tr(const double* __restrict__ l, const Geo& g, int t, long off, double* v) {
#pragma unroll
  for (int m = 0; m < NL; m++) v[m] = l[((long)t * NL + m) * g.G + off];
}

// (-S) lam at the three edges (H, V, D) attached to grid corner (i, j): sums over the <= 2 cells per edge.
// own[] receives lam at those edges in local-edge order (H0, D0, V0) of cell L(i,j).
// MODE (round 3: two launches of the trace preconditioner folded into their consumers):
//   0  the input vector as stored
//   1  PRE:  v = c0 * Dinv * lam at every edge the stencil reads (edge block-Jacobi on the fly): the first step of the
//      zero-start Chebyshev pre-smoother, d0 = Dinv b / theta, never goes through memory (was k_trace_cheb + its vector)
//   2  POST: v = lam + P xc at every edge the stencil reads (prolongation of the vertex-grid correction on the fly: was
//      k_p1_to_trace, a read-modify-write pass over z)
struct StencilAux {
  int j;                     // corner row of this thread
  double c0;                 // MODE 1
  const double* xc;          // MODE 2: global (replicated) vertex vector, row pitch nx + 1
  double sH, sV, sD;         // MODE 2: sqrt(edge length) per edge type
};
template <int NL>
__device__ __forceinline__ void edge_dinv(const double* __restrict__ Dm, double c0, double* v) {
  double z[NL];
#pragma unroll
  for (int a = 0; a < NL; a++) {
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < NL; b++) acc = fma(Dm[a * NL + b], v[b], acc);
    z[a] = c0 * acc;
  }
#pragma unroll
  for (int a = 0; a < NL; a++) v[a] = z[a];
}
__device__ __forceinline__ void edge_prolong(double va, double vb, double sl, double* v) {
  v[0] += sl * 0.5 * (va + vb);
  v[1] += sl * 0.57735026918962576451 * 0.5 * (vb - va);
}
template <int K, int MODE = 0>
__device__ __forceinline__ void trace_stencil(const Geo& g, const DevTables& T, const double* __restrict__ lam, long o, int i,
                                              bool in_x, bool in_y, bool below, double* own, double* yH, double* yV,
                                              double* yD, const StencilAux aux = StencilAux{0, 0.0, nullptr, 0.0, 0.0, 0.0}) {
  const bool left = i > 0 || g.px;
  const long oL = o - i + xm1(g, i), oR = o - i + xp1(g, i);
  constexpr int NL = Dim<K>::NL, NT = 3 * NL;
#pragma unroll
  for (int m = 0; m < NL; m++) yH[m] = yV[m] = yD[m] = 0.0;
  const double* __restrict__ SL = T.SK[0];
  const double* __restrict__ SU = T.SK[1];
  // All loads first, unconditionally (ghost rows and the row padding make every address below a valid one; what a
  // missing cell would have contributed is simply not used): inside the branches each load waited for its own
  // predicate and the four cell blocks ran one after the other at memory latency.
  const bool vL = in_x && in_y, vB = in_x && below, vW = in_y && left;
  double uA[NT], uB[NT], uC[NT];
  load_tr<NL>(lam, g, 0, o, own);
  load_tr<NL>(lam, g, 2, o, own + NL);
  load_tr<NL>(lam, g, 1, o, own + 2 * NL);
  load_tr<NL>(lam, g, 0, o + g.P, uA);            // U(i,j):   H(i,j+1)
  load_tr<NL>(lam, g, 1, oR, uA + 2 * NL);        //           V(i+1,j)
  load_tr<NL>(lam, g, 2, o - g.P, uB + NL);       // U(i,j-1): D(i,j-1)
  load_tr<NL>(lam, g, 1, oR - g.P, uB + 2 * NL);  //           V(i+1,j-1)
  load_tr<NL>(lam, g, 0, oL + g.P, uC);           // U(i-1,j): H(i-1,j+1)
  load_tr<NL>(lam, g, 2, oL, uC + NL);            //           D(i-1,j)
  if (MODE == 1) {
    // block-Jacobi variant of an edge: H edges on the bottom / top boundary and V edges on the left / right boundary have
    // one cell only (variants 1 / 2); the row-dependent choice is wave uniform, the column-dependent one is not, so the
    // interior table is applied to every lane and the two boundary columns redo their V edges
    const int J = g.joff + aux.j;
    const int vH0 = J == 0 ? 1 : (J == g.nyg ? 2 : 0), vH1 = (J + 1) == g.nyg ? 2 : 0;
    const int vV0 = g.px ? 0 : (i == 0 ? 1 : (i == g.nx ? 2 : 0)), vV1 = g.px ? 0 : ((i + 1) == g.nx ? 2 : 0);
    edge_dinv<NL>(T.trDinv[0][vH0], aux.c0, own);
    edge_dinv<NL>(T.trDinv[2][0], aux.c0, own + NL);
    edge_dinv<NL>(T.trDinv[0][vH1], aux.c0, uA);
    edge_dinv<NL>(T.trDinv[2][0], aux.c0, uB + NL);
    edge_dinv<NL>(T.trDinv[0][vH1], aux.c0, uC);
    edge_dinv<NL>(T.trDinv[2][0], aux.c0, uC + NL);
    if (vV0 == 0) edge_dinv<NL>(T.trDinv[1][0], aux.c0, own + 2 * NL); else edge_dinv<NL>(T.trDinv[1][vV0], aux.c0, own + 2 * NL);
    if (vV1 == 0) {
      edge_dinv<NL>(T.trDinv[1][0], aux.c0, uA + 2 * NL);
      edge_dinv<NL>(T.trDinv[1][0], aux.c0, uB + 2 * NL);
    } else {
      edge_dinv<NL>(T.trDinv[1][vV1], aux.c0, uA + 2 * NL);
      edge_dinv<NL>(T.trDinv[1][vV1], aux.c0, uB + 2 * NL);
    }
  }
  if (MODE == 2) {
    // vertices of the 3 x 3 patch that the nine edges touch (clamped: what a clamped index delivers belongs to an edge
    // that does not exist and is masked below or multiplied by a cell that does not exist)
    const int st = g.nx + 1;
    const long J = g.joff + aux.j;
    const long Jm = J > 0 ? J - 1 : 0, Jp = J < g.nyg ? J + 1 : g.nyg;
    const int im = i > 0 ? i - 1 : 0, ip = i < g.nx ? i + 1 : g.nx;
    const double v00 = aux.xc[J * st + i], v10 = aux.xc[J * st + ip], v01 = aux.xc[Jp * st + i], v11 = aux.xc[Jp * st + ip];
    const double v1m = aux.xc[Jm * st + ip], vm1 = aux.xc[Jp * st + im];
    edge_prolong(v00, v10, aux.sH, own);           // H(i,j)
    edge_prolong(v10, v01, aux.sD, own + NL);      // D(i,j)
    edge_prolong(v00, v01, aux.sV, own + 2 * NL);  // V(i,j)
    edge_prolong(v01, v11, aux.sH, uA);            // H(i,j+1)
    edge_prolong(v10, v11, aux.sV, uA + 2 * NL);   // V(i+1,j)
    edge_prolong(v1m, v00, aux.sD, uB + NL);       // D(i,j-1)
    edge_prolong(v1m, v10, aux.sV, uB + 2 * NL);   // V(i+1,j-1)
    edge_prolong(vm1, v01, aux.sH, uC);            // H(i-1,j+1)
    edge_prolong(v00, vm1, aux.sD, uC + NL);       // D(i-1,j)
  }
#pragma unroll
  for (int m = 0; m < NL; m++) {
    if (!in_x) own[m] = 0.0;
    if (!vL) own[NL + m] = 0.0;
    if (!in_y) own[2 * NL + m] = 0.0;
    uA[NL + m] = own[NL + m];
    uB[m] = own[m];
    uC[2 * NL + m] = own[2 * NL + m];
  }
  if (vL) {  // L(i,j): all three rows;  U(i,j): edges (H(i,j+1), D(i,j), V(i+1,j)), row block e1 -> D
    mv_acc_ld<NL, NT>(SL + 0 * NL * NT, NT, own, yH, -1.0);
    mv_acc_ld<NL, NT>(SL + 1 * NL * NT, NT, own, yD, -1.0);
    mv_acc_ld<NL, NT>(SL + 2 * NL * NT, NT, own, yV, -1.0);
    mv_acc_ld<NL, NT>(SU + 1 * NL * NT, NT, uA, yD, -1.0);
  }
  if (vB) mv_acc_ld<NL, NT>(SU + 0 * NL * NT, NT, uB, yH, -1.0);  // U(i,j-1): edges (H(i,j), D(i,j-1), V(i+1,j-1)), row block e0 -> H
  if (vW) mv_acc_ld<NL, NT>(SU + 2 * NL * NT, NT, uC, yV, -1.0);  // U(i-1,j): edges (H(i-1,j+1), D(i-1,j), V(i,j)), row block e2 -> V
}

// K6  trace apply (condensed Schur operator of firedrake.SCPC, hdg_imex.py:128-135), SPD form:
//   out = cb*base + ct * (-S) lam
template <int K>
__global__ __launch_bounds__(128) void k_trace_apply(Geo g, DevTables T, const double* __restrict__ lam,
                                                      const double* __restrict__ base, double cb, double ct,
                                                      double* __restrict__ out) {
  constexpr int NL = Dim<K>::NL, NT = 3 * NL;
  HDG_CORNER_PROLOGUE
  double yH[NL], yV[NL], yD[NL], own[NT];
  trace_stencil<K>(g, T, lam, o, i, in_x, in_y, below, own, yH, yV, yD);
  double bH[NL], bV[NL], bD[NL];
  if (base) {
#pragma unroll
    for (int m = 0; m < NL; m++) {
      bH[m] = cb * base[((long)0 * NL + m) * g.G + o];
      bV[m] = cb * base[((long)1 * NL + m) * g.G + o];
      bD[m] = cb * base[((long)2 * NL + m) * g.G + o];
    }
  } else {
#pragma unroll
    for (int m = 0; m < NL; m++) bH[m] = bV[m] = bD[m] = 0.0;
  }
#pragma unroll
  for (int m = 0; m < NL; m++) {
    const long iH = ((long)0 * NL + m) * g.G + o, iV = ((long)1 * NL + m) * g.G + o, iD = ((long)2 * NL + m) * g.G + o;
    out[iH] = in_x ? fma(ct, yH[m], bH[m]) : 0.0;
    out[iV] = in_y ? fma(ct, yV[m], bV[m]) : 0.0;
    out[iD] = (in_x && in_y) ? fma(ct, yD[m], bD[m]) : 0.0;
  }
}

// Fused smoother step on the trace space (operator + edge block-Jacobi + Chebyshev update in one pass):
//   r = cb*base + ct*(-S) v ;  z = Dinv r ;  dn = c1*v + c2*z     (v is the previous direction when c1 != 0)
//   r_out <- r, d_out <- dn (each only if the pointer is given);
//   x given:  x <- (xadd ? x : 0) + xv*v + dn
// v is read through the stencil by neighbouring threads, so dn goes to a DIFFERENT buffer (d_out != v).
// The two-step Chebyshev smoother of the trace preconditioner is then 2 launches and 5-8 vector passes
// instead of 3-4 launches and 11-15 passes (Engine::cheb_smooth).
// MODE 1 (v = c0 Dinv v_stored on the fly) / MODE 2 (v = v_stored + P xc on the fly; xadd == 2: x receives that v and must
// be a vector OTHER than v, which the neighbouring threads are still reading): see trace_stencil
template <int K, int MODE>
__global__ __launch_bounds__(128) void k_trace_smooth(Geo g, DevTables T, const double* __restrict__ v,
                                                       const double* __restrict__ base, double cb, double ct, double c1,
                                                       double c2, double* __restrict__ r_out, double* __restrict__ d_out,
                                                       double* x, int xadd, double xv, StencilAux aux,
                                                       const double* xin /* may alias x */) {
  constexpr int NL = Dim<K>::NL, NT = 3 * NL;
  HDG_CORNER_PROLOGUE
  double y[3][NL], own[NT];  // y[t]: t = 0 H, 1 V, 2 D  (plane order of the trace layout)
  aux.j = j;
  trace_stencil<K, MODE>(g, T, v, o, i, in_x, in_y, below, own, y[0], y[1], y[2], aux);
  const bool valid[3] = {in_x, in_y, in_x && in_y};
  const int ownoff[3] = {0, 2 * NL, NL};  // own[] is in local-edge order (H, D, V)
  const int var[3] = {(g.joff + j == 0) ? 1 : (g.joff + j == g.nyg ? 2 : 0), g.px ? 0 : ((i == 0) ? 1 : (i == g.nx ? 2 : 0)), 0};
  double bb[3][NL], xo[3][NL];
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int m = 0; m < NL; m++) {
      const long idx = ((long)t * NL + m) * g.G + o;
      bb[t][m] = (base && valid[t]) ? cb * base[idx] : 0.0;
      xo[t][m] = (x && xadd == 1 && valid[t]) ? xin[idx] : 0.0;  // xin == x unless the old value lives in another vector
    }
#pragma unroll
  for (int t = 0; t < 3; t++) {
    if (!valid[t]) continue;
    double r[NL], z[NL];
#pragma unroll
    for (int m = 0; m < NL; m++) { r[m] = fma(ct, y[t][m], bb[t][m]); z[m] = 0.0; }
    mv_acc_ld<NL, NL>(T.trDinv[t][var[t]], NL, r, z, 1.0);
#pragma unroll
    for (int m = 0; m < NL; m++) {
      const long idx = ((long)t * NL + m) * g.G + o;
      const double vo = own[ownoff[t] + m];
      const double dn = fma(c1, vo, c2 * z[m]);
      if (r_out) r_out[idx] = r[m];
      if (d_out) d_out[idx] = dn;
      if (x) x[idx] = xadd == 2 ? vo : xo[t][m] + fma(xv, vo, dn);
    }
  }
}

// edge-block Jacobi / Chebyshev update on the trace space:
//   d = c1*d + c2 * Dinv r ;  x += d        (ASMStarPC with construct_dim=1: one block per facet)
//   assign != 0:  x = d  (first step from a zero initial guess: x is not read)
template <int K>
__global__ __launch_bounds__(128) void k_trace_cheb(Geo g, DevTables T, const double* __restrict__ r, double* __restrict__ d,
                                                     double* __restrict__ x, double c1, double c2, int assign) {
  constexpr int NL = Dim<K>::NL;
  HDG_CORNER_PROLOGUE
#pragma unroll
  for (int t = 0; t < 3; t++) {
    bool valid;
    int var;
    if (t == 0) { valid = in_x; var = (g.joff + j == 0) ? 1 : (g.joff + j == g.nyg ? 2 : 0); }
    else if (t == 1) { valid = in_y; var = g.px ? 0 : ((i == 0) ? 1 : (i == g.nx ? 2 : 0)); }
    else { valid = in_x && in_y; var = 0; }
    if (!valid) continue;
    double rr[NL], z[NL];
    load_tr<NL>(r, g, t, o, rr);
#pragma unroll
    for (int m = 0; m < NL; m++) z[m] = 0.0;
    const double* __restrict__ Dm = T.trDinv[t][var];
    mv_acc_ld<NL, NL>(Dm, NL, rr, z, 1.0);
    // loads first (the c1 == 0 / assign decisions are wave uniform and taken outside the element loop)
    double dold[NL], xold[NL];
    if (c1 != 0.0) {
#pragma unroll
      for (int m = 0; m < NL; m++) dold[m] = d[((long)t * NL + m) * g.G + o];
    } else {
#pragma unroll
      for (int m = 0; m < NL; m++) dold[m] = 0.0;
    }
    if (x && !assign) {
#pragma unroll
      for (int m = 0; m < NL; m++) xold[m] = x[((long)t * NL + m) * g.G + o];
    } else {
#pragma unroll
      for (int m = 0; m < NL; m++) xold[m] = 0.0;
    }
#pragma unroll
    for (int m = 0; m < NL; m++) {
      const long idx = ((long)t * NL + m) * g.G + o;
      const double dn = fma(c1, dold[m], c2 * z[m]);
      d[idx] = dn;
      if (x) x[idx] = assign ? dn : xold[m] + dn;
    }
  }
}

// K5  condensed right-hand side (SCPC forward elimination), SPD sign convention:
//   out_e = sum_{K contains e} (Y_K r_{x,K})_e  -  r_lambda,e       with r_x = (rw, rp)
template <int K, bool HASW, bool HASP>
__device__ __forceinline__ void y_rows(const double* __restrict__ Y, int erow, const double* __restrict__ rw,
                                       const double* __restrict__ rp, long Nc, long c, double* acc) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, NL = Dim<K>::NL, NX = Dim<K>::NX, N2 = 2 * NU;
  if (HASW) {
    double x[N2];
    load_vel<NU>(rw, Nc, c, x);
    mv_acc_ld<NL, N2>(Y + (long)erow * NL * NX, NX, x, acc, 1.0);
  }
  if (HASP) {
    double x[NP];
    load_cell<NP>(rp, Nc, c, x);
    mv_acc_ld<NL, NP>(Y + (long)erow * NL * NX + N2, NX, x, acc, 1.0);
  }
}

template <int K, bool HASW, bool HASP>
__global__ __launch_bounds__(128) void k_condense(Geo g, DevTables T, const double* __restrict__ rw,
                                                   const double* __restrict__ rp, const double* __restrict__ rl,
                                                   double* __restrict__ out) {
  constexpr int NL = Dim<K>::NL;
  HDG_CORNER_PROLOGUE
  double yH[NL], yV[NL], yD[NL];
#pragma unroll
  for (int m = 0; m < NL; m++) yH[m] = yV[m] = yD[m] = 0.0;
  if (in_x && in_y) {
    const long cL = cidx(g, 0, j, i), cU = cidx(g, 1, j, i);
    y_rows<K, HASW, HASP>(T.Y[0], 0, rw, rp, g.Nc, cL, yH);
    y_rows<K, HASW, HASP>(T.Y[0], 1, rw, rp, g.Nc, cL, yD);
    y_rows<K, HASW, HASP>(T.Y[0], 2, rw, rp, g.Nc, cL, yV);
    y_rows<K, HASW, HASP>(T.Y[1], 1, rw, rp, g.Nc, cU, yD);
  }
  if (in_x && below) y_rows<K, HASW, HASP>(T.Y[1], 0, rw, rp, g.Nc, cidx(g, 1, j - 1, i), yH);
  if (in_y && left) y_rows<K, HASW, HASP>(T.Y[1], 2, rw, rp, g.Nc, cidx(g, 1, j, xm1(g, i)), yV);
#pragma unroll
  for (int m = 0; m < NL; m++) {
    const long iH = ((long)0 * NL + m) * g.G + o, iV = ((long)1 * NL + m) * g.G + o, iD = ((long)2 * NL + m) * g.G + o;
    out[iH] = in_x ? yH[m] - (rl ? rl[iH] : 0.0) : 0.0;
    out[iV] = in_y ? yV[m] - (rl ? rl[iV] : 0.0) : 0.0;
    out[iD] = (in_x && in_y) ? yD[m] - (rl ? rl[iD] : 0.0) : 0.0;
  }
}

// K8  local back-substitution:  (u, phi)_K = Ainv r_{x,K} - W lambda_K
template <int K, bool HASW, bool HASP>
__global__ __launch_bounds__(128) void k_backsub(Geo g, DevTables T, const double* __restrict__ rw,
                                                  const double* __restrict__ rp, const double* __restrict__ lam,
                                                  double* __restrict__ u, double* __restrict__ phi) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, NL = Dim<K>::NL, NX = Dim<K>::NX, NT = 3 * NL, N2 = 2 * NU;
  HDG_CELL_PROLOGUE
  double y[NX];
#pragma unroll
  for (int n = 0; n < NX; n++) y[n] = 0.0;
  const double* __restrict__ Ai = T.Ainv[s];
  if (HASW) {
    double x[N2];
    load_vel<NU>(rw, g.Nc, c, x);
    mv_acc_ld<NX, N2>(Ai, NX, x, y, 1.0);
  }
  if (HASP) {
    double x[NP];
    load_cell<NP>(rp, g.Nc, c, x);
    mv_acc_ld<NX, NP>(Ai + N2, NX, x, y, 1.0);
  }
  double l[NT];
#pragma unroll
  for (int e = 0; e < 3; e++) {
    int t;
    const long off = edge_off(s, e, i, j, g, t);
    load_tr<NL>(lam, g, t, off, l + e * NL);
  }
  mv_acc_ld<NX, NT>(T.W[s], NT, l, y, -1.0);
  {
    const VelBuf Bu(u);
#pragma unroll
    for (int m = 0; m < NU; m++) Bu.st(pair_bytes(m, g.Nc), (unsigned)c * 16u, hdg_d2{y[m], y[NU + m]});
  }
#pragma unroll
  for (int n = 0; n < NP; n++) phi[(long)n * g.Nc + c] = y[N2 + n];
}

// trace reconstruction (hdg_imex.py:450-469):
//   interior: lambda = {{p}} + (Q+.n+ + Q-.n-)/(2 tau);  boundary: lambda = p + Q.n/tau   (L2 on P_k(e))
template <int K>
__device__ __forceinline__ void trace_side(const DevTables& T, int s, int e, const double* __restrict__ Q,
                                           const double* __restrict__ p, long Nc, long c, double wq, double wp,
                                           double* acc) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, NL = Dim<K>::NL, N2 = 2 * NU;
  double x[N2], pp[NP];
  load_vel<NU>(Q, Nc, c, x);
  load_cell<NP>(p, Nc, c, pp);
  mv_acc_ld<NL, N2>(T.N[s][e], N2, x, acc, wq * T.sig[s][e]);
  mv_acc_ld<NL, NP>(T.Pt[s][e], NP, pp, acc, wp);
}

template <int K>
__global__ __launch_bounds__(128) void k_trace_recon(Geo g, DevTables T, const double* __restrict__ Q,
                                                      const double* __restrict__ p, double* __restrict__ out) {
  constexpr int NL = Dim<K>::NL;
  HDG_CORNER_PROLOGUE
  const double it = 1.0 / T.tau;
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const int e = (t == 0) ? 0 : (t == 1 ? 2 : 1);
    bool valid, hasL, hasU;
    long cL, cU;
    if (t == 0) { valid = in_x; hasL = in_y; hasU = below; cL = cidx(g, 0, j, i); cU = cidx(g, 1, j - 1, i); }
    else if (t == 1) { valid = in_y; hasL = in_x; hasU = left; cL = cidx(g, 0, j, i); cU = cidx(g, 1, j, xm1(g, i)); }
    else { valid = in_x && in_y; hasL = hasU = true; cL = cidx(g, 0, j, i); cU = cidx(g, 1, j, i); }
    double acc[NL];
#pragma unroll
    for (int m = 0; m < NL; m++) acc[m] = 0.0;
    if (valid) {
      const double w = (hasL && hasU) ? 0.5 : 1.0;
      if (hasL) trace_side<K>(T, 0, e, Q, p, g.Nc, cL, w * it, w, acc);
      if (hasU) trace_side<K>(T, 1, e, Q, p, g.Nc, cU, w * it, w, acc);
    }
#pragma unroll
    for (int m = 0; m < NL; m++) out[((long)t * NL + m) * g.G + o] = acc[m];
  }
}

// ------------------------------------------------------------------------------------------
// pressure-reconstruction right-hand side (hdg_imex.py:201-207):
//   rp = weak_divergence(psi, v),  v = -b + (Q.grad)Q  (per cell: -(grad psi, v)_K + <psi,{{v}}.n>_int)
//   rl = - int_{dOmega} mu n.b    (boundary edges only; interior entries zero)
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(128) void k_precon_rhs(Geo g, DevTables T, const double* __restrict__ Q,
                                                     const double* __restrict__ bnew, double bscale,
                                                     double* __restrict__ rp, double* __restrict__ rl) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, NL = Dim<K>::NL, N2 = 2 * NU;
  HDG_CELL_PROLOGUE
  double x[N2], b[N2], y[NP];
  load_vel<NU>(Q, g.Nc, c, x);
  load_vel<NU>(bnew, g.Nc, c, b);
#pragma unroll
  for (int n = 0; n < N2; n++) b[n] *= bscale;
#pragma unroll
  for (int r = 0; r < NP; r++) y[r] = 0.0;
  {
    const double* __restrict__ Phi = T.cPhi[s];
    const double* __restrict__ Gx = T.cGx[s];
    const double* __restrict__ Gy = T.cGy[s];
#pragma unroll 1
    for (int q = 0; q < T.nqc; q++) {
      double qx = 0, qy = 0, bx = 0, by = 0, dxx = 0, dxy = 0, dyx = 0, dyy = 0;
#pragma unroll
      for (int m = 0; m < NU; m++) {
        const double ph = Phi[q * NU + m], gx = Gx[q * NU + m], gy = Gy[q * NU + m];
        qx = fma(ph, x[m], qx); qy = fma(ph, x[NU + m], qy);
        bx = fma(ph, b[m], bx); by = fma(ph, b[NU + m], by);
        dxx = fma(gx, x[m], dxx); dxy = fma(gy, x[m], dxy);
        dyx = fma(gx, x[NU + m], dyx); dyy = fma(gy, x[NU + m], dyy);
      }
      const double w = T.cw[q];
      const double vx = -bx + qx * dxx + qy * dxy, vy = -by + qx * dyx + qy * dyy;
#pragma unroll
      for (int r = 0; r < NP; r++) y[r] -= w * (Gx[q * NU + r] * vx + Gy[q * NU + r] * vy);
    }
  }
#pragma unroll
  for (int e = 0; e < 3; e++) {
    long cn;
    const bool has = nbr(s, e, i, j, g, cn);
    const double nx_ = T.enx[e], ny_ = T.eny[e], sg = T.sig[s][e];
    if (has) {
      double xn[N2], bn[N2];
      load_vel<NU>(Q, g.Nc, cn, xn);
      load_vel<NU>(bnew, g.Nc, cn, bn);
      const double* __restrict__ Po = T.ePhi[s][e];
      const double* __restrict__ Gxo = T.eGx[s][e];
      const double* __restrict__ Gyo = T.eGy[s][e];
      const double* __restrict__ Pn = T.ePhi[1 - s][e];
      const double* __restrict__ Gxn = T.eGx[1 - s][e];
      const double* __restrict__ Gyn = T.eGy[1 - s][e];
#pragma unroll 1
      for (int q = 0; q < T.nqe; q++) {
        double vn = 0.0;
        {
          double qx = 0, qy = 0, bx = 0, by = 0, dxx = 0, dxy = 0, dyx = 0, dyy = 0;
#pragma unroll
          for (int m = 0; m < NU; m++) {
            const double ph = Po[q * NU + m], gx = Gxo[q * NU + m], gy = Gyo[q * NU + m];
            qx = fma(ph, x[m], qx); qy = fma(ph, x[NU + m], qy);
            bx = fma(ph, b[m], bx); by = fma(ph, b[NU + m], by);
            dxx = fma(gx, x[m], dxx); dxy = fma(gy, x[m], dxy);
            dyx = fma(gx, x[NU + m], dyx); dyy = fma(gy, x[NU + m], dyy);
          }
          vn += 0.5 * ((-bx + qx * dxx + qy * dxy) * nx_ + (-by + qx * dyx + qy * dyy) * ny_);
        }
        {
          double qx = 0, qy = 0, bx = 0, by = 0, dxx = 0, dxy = 0, dyx = 0, dyy = 0;
#pragma unroll
          for (int m = 0; m < NU; m++) {
            const double ph = Pn[q * NU + m], gx = Gxn[q * NU + m], gy = Gyn[q * NU + m];
            qx = fma(ph, xn[m], qx); qy = fma(ph, xn[NU + m], qy);
            bx = fma(ph, bscale * bn[m], bx); by = fma(ph, bscale * bn[NU + m], by);
            dxx = fma(gx, xn[m], dxx); dxy = fma(gy, xn[m], dxy);
            dyx = fma(gx, xn[NU + m], dyx); dyy = fma(gy, xn[NU + m], dyy);
          }
          vn += 0.5 * ((-bx + qx * dxx + qy * dxy) * nx_ + (-by + qx * dyx + qy * dyy) * ny_);
        }
        const double w = T.ew[e][q] * sg * vn;
#pragma unroll
        for (int r = 0; r < NP; r++) y[r] = fma(Po[q * NU + r], w, y[r]);
      }
    } else {
      // boundary edge: rl = - <mu, n_K . b> = - sigma * (N_e b)[0:NL]
      double tr[NL];
#pragma unroll
      for (int m = 0; m < NL; m++) tr[m] = 0.0;
      mv_acc_ld<NL, N2>(T.N[s][e], N2, b, tr, -sg);
      int t;
      const long off = edge_off(s, e, i, j, g, t);
#pragma unroll
      for (int m = 0; m < NL; m++) rl[((long)t * NL + m) * g.G + off] = tr[m];
    }
  }
  store_cell<NP>(rp, g.Nc, c, y);
}

// ------------------------------------------------------------------------------------------
// constraint rows of the monolithic (unsplit) system, Gamma(psi, mu; u, phi, lambda) of hdg_imex.py:342-351
//   psi-row (per cell):   out = B u + tau * sum_e Pt_e^T (Pt_e phi - lambda_e)
//   mu-row  (per edge):   out_e = sum_{K contains e} [ sigma N_e u_K + tau Pt_e phi_K - tau lambda_e ]
// any of u / phi / lam may be null (treated as zero)
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(128) void k_gamma_psi(Geo g, DevTables T, const double* __restrict__ u,
                                                    const double* __restrict__ phi, const double* __restrict__ lam,
                                                    double* __restrict__ out) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, NL = Dim<K>::NL, N2 = 2 * NU;
  HDG_CELL_PROLOGUE
  double y[NP];
#pragma unroll
  for (int r = 0; r < NP; r++) y[r] = 0.0;
  if (u) {
    double x[N2];
    load_vel<NU>(u, g.Nc, c, x);
    mv_acc<NP, N2>(T.B[s], x, y, 1.0);
  }
  if (phi || lam) {
    double pp[NP];
    if (phi) load_cell<NP>(phi, g.Nc, c, pp);
#pragma unroll
    for (int e = 0; e < 3; e++) {
      double tr[NL];
#pragma unroll
      for (int m = 0; m < NL; m++) tr[m] = 0.0;
      if (phi) mv_acc_ld<NL, NP>(T.Pt[s][e], NP, pp, tr, 1.0);
      if (lam) {
        int t;
        const long off = edge_off(s, e, i, j, g, t);
#pragma unroll
        for (int m = 0; m < NL; m++) tr[m] -= lam[((long)t * NL + m) * g.G + off];
      }
      const double* __restrict__ Pm = T.Pt[s][e];
#pragma unroll
      for (int r = 0; r < NP; r++) {
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < NL; m++) v = fma(Pm[m * NP + r], tr[m], v);
        y[r] = fma(T.tau, v, y[r]);
      }
    }
  }
  store_cell<NP>(out, g.Nc, c, y);
}

template <int K>
__device__ __forceinline__ void gamma_mu_side(const DevTables& T, int s, int e, const double* __restrict__ u,
                                              const double* __restrict__ phi, long Nc, long c, double* acc) {
  constexpr int NU = Dim<K>::NU, NP = Dim<K>::NP, NL = Dim<K>::NL, N2 = 2 * NU;
  if (u) {
    double x[N2];
    load_vel<NU>(u, Nc, c, x);
    mv_acc_ld<NL, N2>(T.N[s][e], N2, x, acc, T.sig[s][e]);
  }
  if (phi) {
    double pp[NP];
    load_cell<NP>(phi, Nc, c, pp);
    mv_acc_ld<NL, NP>(T.Pt[s][e], NP, pp, acc, T.tau);
  }
}

template <int K>
__global__ __launch_bounds__(128) void k_gamma_mu(Geo g, DevTables T, const double* __restrict__ u,
                                                   const double* __restrict__ phi, const double* __restrict__ lam,
                                                   double* __restrict__ out) {
  constexpr int NL = Dim<K>::NL;
  HDG_CORNER_PROLOGUE
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const int e = (t == 0) ? 0 : (t == 1 ? 2 : 1);
    bool valid, hasL, hasU;
    long cL, cU;
    if (t == 0) { valid = in_x; hasL = in_y; hasU = below; cL = cidx(g, 0, j, i); cU = cidx(g, 1, j - 1, i); }
    else if (t == 1) { valid = in_y; hasL = in_x; hasU = left; cL = cidx(g, 0, j, i); cU = cidx(g, 1, j, xm1(g, i)); }
    else { valid = in_x && in_y; hasL = hasU = true; cL = cidx(g, 0, j, i); cU = cidx(g, 1, j, i); }
    double acc[NL];
#pragma unroll
    for (int m = 0; m < NL; m++) acc[m] = 0.0;
    if (valid) {
      if (hasL) gamma_mu_side<K>(T, 0, e, u, phi, g.Nc, cL, acc);
      if (hasU) gamma_mu_side<K>(T, 1, e, u, phi, g.Nc, cU, acc);
      if (lam) {
        const double ncell = (hasL ? 1.0 : 0.0) + (hasU ? 1.0 : 0.0);
#pragma unroll
        for (int m = 0; m < NL; m++) acc[m] -= T.tau * ncell * lam[((long)t * NL + m) * g.G + o];
      }
    }
#pragma unroll
    for (int m = 0; m < NL; m++) out[((long)t * NL + m) * g.G + o] = acc[m];
  }
}

// ------------------------------------------------------------------------------------------
// nodal <-> modal conversion at the library boundary (array-of-structures, reference layout)
// ------------------------------------------------------------------------------------------
// velocity: nodal[(cref*NU + node)*2 + d], cref = 2*(j*nx+i)+s
template <int K>
__global__ void k_q_nodal_to_modal(Geo g, DevTables T, const double* __restrict__ nodal, double* __restrict__ modal) {
  constexpr int NU = Dim<K>::NU;
  HDG_CELL_PROLOGUE
  const long cref = 2 * ((long)j * g.nx + i) + s;
#pragma unroll 1
  for (int d = 0; d < 2; d++) {
    double v[NU], m[NU];
#pragma unroll
    for (int n = 0; n < NU; n++) { v[n] = nodal[(cref * NU + n) * 2 + d]; m[n] = 0.0; }
    mv_acc<NU, NU>(T.Vuinv, v, m, 1.0);
#pragma unroll
    for (int n = 0; n < NU; n++) modal[(((long)n * g.Nc + c) << 1) + d] = m[n];
  }
}
template <int K>
__global__ void k_q_modal_to_nodal(Geo g, DevTables T, const double* __restrict__ modal, double* __restrict__ nodal) {
  constexpr int NU = Dim<K>::NU;
  HDG_CELL_PROLOGUE
  const long cref = 2 * ((long)j * g.nx + i) + s;
#pragma unroll 1
  for (int d = 0; d < 2; d++) {
    double v[NU], m[NU];
#pragma unroll
    for (int n = 0; n < NU; n++) { m[n] = modal[(((long)n * g.Nc + c) << 1) + d]; v[n] = 0.0; }
    mv_acc<NU, NU>(T.Vu, m, v, 1.0);
#pragma unroll
    for (int n = 0; n < NU; n++) nodal[(cref * NU + n) * 2 + d] = v[n];
  }
}
template <int K>
__global__ void k_p_nodal_to_modal(Geo g, DevTables T, const double* __restrict__ nodal, double* __restrict__ modal) {
  constexpr int NP = Dim<K>::NP;
  HDG_CELL_PROLOGUE
  const long cref = 2 * ((long)j * g.nx + i) + s;
  double v[NP], m[NP];
#pragma unroll
  for (int n = 0; n < NP; n++) { v[n] = nodal[cref * NP + n]; m[n] = 0.0; }
  mv_acc<NP, NP>(T.Vpinv, v, m, 1.0);
  store_cell<NP>(modal, g.Nc, c, m);
}
template <int K>
__global__ void k_p_modal_to_nodal(Geo g, DevTables T, const double* __restrict__ modal, double* __restrict__ nodal) {
  constexpr int NP = Dim<K>::NP;
  HDG_CELL_PROLOGUE
  const long cref = 2 * ((long)j * g.nx + i) + s;
  double v[NP], m[NP];
  load_cell<NP>(modal, g.Nc, c, m);
#pragma unroll
  for (int n = 0; n < NP; n++) v[n] = 0.0;
  mv_acc<NP, NP>(T.Vp, m, v, 1.0);
#pragma unroll
  for (int n = 0; n < NP; n++) nodal[cref * NP + n] = v[n];
}
// trace: nodal[e*NL + node], edge numbering of oracle/fem.py; modal basis chi_a = Leg_a / sqrt(len)
template <int K, bool TO_MODAL>
__global__ void k_l_convert(Geo g, DevTables T, double* __restrict__ nodal, double* __restrict__ modal) {
  constexpr int NL = Dim<K>::NL;
  HDG_CORNER_PROLOGUE
  // periodic square: the edges of the top row / right column are those of the bottom row / left column
  const int nvx = g.nx + 1 - g.px;
  const long NH = (long)g.nx * (g.ny + 1 - g.px), NV = (long)nvx * g.ny;
#pragma unroll
  for (int t = 0; t < 3; t++) {
    bool valid;
    long eidx;
    double len;
    if (t == 0) { valid = in_x; eidx = (long)j * g.nx + i; len = T.elen[0]; }
    else if (t == 1) { valid = in_y; eidx = NH + (long)j * nvx + i; len = T.elen[2]; }
    else { valid = in_x && in_y; eidx = NH + NV + (long)j * g.nx + i; len = T.elen[1]; }
    if (!valid) continue;
    double a[NL], b[NL];
    if (TO_MODAL) {
#pragma unroll
      for (int m = 0; m < NL; m++) { a[m] = nodal[eidx * NL + m]; b[m] = 0.0; }
      mv_acc<NL, NL>(T.Vlinv, a, b, sqrt(len));
#pragma unroll
      for (int m = 0; m < NL; m++) modal[((long)t * NL + m) * g.G + o] = b[m];
    } else {
#pragma unroll
      for (int m = 0; m < NL; m++) { a[m] = modal[((long)t * NL + m) * g.G + o]; b[m] = 0.0; }
      mv_acc<NL, NL>(T.Vl, a, b, 1.0 / sqrt(len));
#pragma unroll
      for (int m = 0; m < NL; m++) nodal[eidx * NL + m] = b[m];
    }
  }
}

// ------------------------------------------------------------------------------------------
// vector kernels: grid-stride over PAIRS of doubles, one 16-byte access per lane (global_load/store_dwordx4;
// 8-byte-per-lane streams stop at ~4.9 TB/s on MI355X, 16-byte ones reach ~6.3 TB/s).  Every vector is a
// hipMalloc'ed array (256-byte aligned); an odd length leaves one tail element to thread 0.
// ------------------------------------------------------------------------------------------
#define HDG_VEC_PROLOGUE                                                     \
  const long tid_ = (long)blockIdx.x * blockDim.x + threadIdx.x;            \
  const long stride_ = (long)gridDim.x * blockDim.x;                        \
  const long NP2_ = N >> 1;                                                 \
  const bool tail_ = (N & 1) && tid_ == 0;                                  \
  const long it_ = N - 1;                                                   \
  (void)it_;
__device__ __forceinline__ const hdg_d2* as2(const double* p) { return reinterpret_cast<const hdg_d2*>(p); }
__device__ __forceinline__ hdg_d2* as2(double* p) { return reinterpret_cast<hdg_d2*>(p); }
// NT = true: streaming (non-temporal) accesses, for vectors that are larger than the caches anyway (velocity vectors at
// the benchmark sizes); measured with tools/probes/stream_probe: 3 reads + 1 write 5.85 -> 6.13 TB/s
template <bool NT>
__device__ __forceinline__ hdg_d2 ldv(const double* p, long i) { return NT ? __builtin_nontemporal_load(as2(p) + i) : as2(p)[i]; }
template <bool NT>
__device__ __forceinline__ void stv(double* p, long i, hdg_d2 v) { if (NT) __builtin_nontemporal_store(v, as2(p) + i); else as2(p)[i] = v; }
__device__ __forceinline__ hdg_d2 fma2(double a, hdg_d2 x, hdg_d2 y) { return hdg_d2{fma(a, x.x, y.x), fma(a, x.y, y.y)}; }
// single-precision STORAGE of a Krylov basis (arithmetic stays FP64): a pair of floats widened on load
typedef float hdg_f2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ hdg_d2 ldv(const float* p, long i) {
  const hdg_f2* q = reinterpret_cast<const hdg_f2*>(p) + i;
  const hdg_f2 v = NT ? __builtin_nontemporal_load(q) : *q;
  return hdg_d2{(double)v.x, (double)v.y};
}

struct LinComb {
  const double* v[8];
  double c[8];
  int n;
};
template <bool NT>
__global__ void k_lincomb(long N, LinComb lc, double* __restrict__ out) {
  HDG_VEC_PROLOGUE
  for (long i = tid_; i < NP2_; i += stride_) {
    hdg_d2 acc = {0.0, 0.0};
    for (int k = 0; k < lc.n; k++) acc = fma2(lc.c[k], ldv<NT>(lc.v[k], i), acc);
    stv<NT>(out, i, acc);
  }
  if (tail_) {
    double acc = 0.0;
    for (int k = 0; k < lc.n; k++) acc = fma(lc.c[k], lc.v[k][it_], acc);
    out[it_] = acc;
  }
}
// conjugate-gradient updates of the trace solver in one pass each:
//   x += alpha p ; r -= alpha Ap
__device__ __forceinline__ void cg_xr_body(long N, double alpha, const double* __restrict__ p, const double* __restrict__ Ap,
                                           double* __restrict__ x, double* __restrict__ r) {
  HDG_VEC_PROLOGUE
  for (long i = tid_; i < NP2_; i += stride_) {
    const hdg_d2 pv = as2(p)[i], av = as2(Ap)[i], xv = as2(x)[i], rv = as2(r)[i];
    as2(x)[i] = fma2(alpha, pv, xv);
    as2(r)[i] = fma2(-alpha, av, rv);
  }
  if (tail_) {
    x[it_] = fma(alpha, p[it_], x[it_]);
    r[it_] = fma(-alpha, Ap[it_], r[it_]);
  }
}
//   p = (z - c n) + beta p      (n: null-space vector; z - c n is the projected preconditioned residual)
__device__ __forceinline__ void cg_p_body(long N, const double* __restrict__ z, const double* __restrict__ nvec, double c, double beta,
                                          double* __restrict__ p) {
  HDG_VEC_PROLOGUE
  for (long i = tid_; i < NP2_; i += stride_) {
    const hdg_d2 zp = fma2(-c, as2(nvec)[i], as2(z)[i]);
    as2(p)[i] = (beta == 0.0) ? zp : fma2(beta, as2(p)[i], zp);
  }
  if (tail_) {
    const double zp = fma(-c, nvec[it_], z[it_]);
    p[it_] = (beta == 0.0) ? zp : fma(beta, p[it_], zp);
  }
}
__global__ void k_cg_xr(long N, double alpha, const double* __restrict__ p, const double* __restrict__ Ap,
                        double* __restrict__ x, double* __restrict__ r) {
  cg_xr_body(N, alpha, p, Ap, x, r);
}
__global__ void k_cg_p(long N, const double* __restrict__ z, const double* __restrict__ nvec, double c, double beta,
                       double* __restrict__ p) {
  cg_p_body(N, z, nvec, c, beta, p);
}
// Device-resident scalars of the trace CG (no host round trip for alpha / beta / projection coefficient):
//   sc[0] rz, sc[1] alpha, sc[2] beta, sc[3] c, sc[4] (z',z'), sc[5] p.Ap, sc[6] flag (1 breakdown, 2 cancellation),
//   sc[7] (z,z) before the projection
__global__ void k_cg_alpha(const double* __restrict__ res, double* __restrict__ sc) {
  const double pAp = res[0];
  sc[5] = pAp;
  if (!(pAp > 0.0)) sc[6] = 1.0;
  sc[1] = sc[0] / pAp;
}
// res = (z,n), (z,r), (z,z), (n,r);  nn = (n,n)
__global__ void k_cg_beta(const double* __restrict__ res, double* __restrict__ sc, double nn, int first) {
  const double c = res[0] / nn;
  const double rzn = res[1] - c * res[3];
  const double zz = res[2] - c * res[0];
  if (!(zz > 1e-6 * res[2])) sc[6] = 2.0;
  sc[3] = c;
  sc[4] = zz;
  sc[7] = res[2];
  sc[2] = first ? 0.0 : rzn / sc[0];
  sc[0] = rzn;
}
__global__ void k_cg_xr_dev(long N, const double* __restrict__ sc, const double* __restrict__ p, const double* __restrict__ Ap,
                            double* __restrict__ x, double* __restrict__ r) {
  cg_xr_body(N, sc[1], p, Ap, x, r);
}
__global__ void k_cg_p_dev(long N, const double* __restrict__ z, const double* __restrict__ nvec, const double* __restrict__ sc,
                           double* __restrict__ p) {
  cg_p_body(N, z, nvec, sc[3], sc[2], p);
}
// Single-reduction form of the same preconditioned CG (Chronopoulos & Gear): with w = A z the step length follows
// from (z',r) and (w,z) of ONE reduction, p.Ap = (w,z) - beta (z',r) / alpha_old, and s = A p is carried by the
// recurrence s = w + beta s.  A n = 0 and A symmetric: (w, z') = (w, z) up to rounding.
//   res = (z,n), (z,r), (z,z), (z,w), (n,r);  nn = (n,n);  sc[] as above
// Zero residual (gamma = 0): alpha = beta = 0, the update is the identity and the host returns after the snapshot.
__global__ void k_cg_sr_scalars(const double* __restrict__ res, double* __restrict__ sc, double nn, int first,
                                double* __restrict__ hsc = nullptr) {
  const double c = res[0] / nn;
  const double rzn = res[1] - c * res[4];
  const double zz = res[2] - c * res[0];
  if (!(zz > 1e-6 * res[2])) sc[6] = 2.0;
  const double beta = first ? 0.0 : rzn / sc[0];
  const double pAp = first ? res[3] : res[3] - beta * rzn / sc[1];
  double alpha = rzn / pAp;
  if (rzn == 0.0) alpha = 0.0;
  else if (!(pAp > 0.0)) { sc[6] = 1.0; alpha = 0.0; }
  sc[3] = c;
  sc[4] = zz;
  sc[7] = res[2];
  sc[5] = pAp;
  sc[2] = (rzn == 0.0) ? 0.0 : beta;
  sc[1] = alpha;
  sc[0] = rzn;
  if (hsc) {  // snapshot for the host's lagged convergence check (pinned memory; read after the event behind this kernel)
#pragma unroll
    for (int q = 0; q < 8; q++) hsc[q] = sc[q];
  }
}
//   p = (z - c n) + beta p ;  s = w + beta s ;  x += alpha p ;  r -= alpha s       (one pass; whole arrays, ghost rows too)
__global__ void k_cg_sr_update(long N, const double* __restrict__ sc, const double* __restrict__ z, const double* __restrict__ nvec,
                               const double* __restrict__ w, double* __restrict__ p, double* __restrict__ s,
                               double* __restrict__ x, double* __restrict__ r) {
  const double alpha = sc[1], beta = sc[2], c = sc[3];
  HDG_VEC_PROLOGUE
  for (long i = tid_; i < NP2_; i += stride_) {
    const hdg_d2 zp = fma2(-c, as2(nvec)[i], as2(z)[i]);
    const hdg_d2 wv = as2(w)[i], xv = as2(x)[i], rv = as2(r)[i];
    const hdg_d2 pv = (beta == 0.0) ? zp : fma2(beta, as2(p)[i], zp);
    const hdg_d2 sv = (beta == 0.0) ? wv : fma2(beta, as2(s)[i], wv);
    as2(p)[i] = pv;
    as2(s)[i] = sv;
    as2(x)[i] = fma2(alpha, pv, xv);
    as2(r)[i] = fma2(-alpha, sv, rv);
  }
  if (tail_) {
    const double zp = fma(-c, nvec[it_], z[it_]);
    const double pv = (beta == 0.0) ? zp : fma(beta, p[it_], zp);
    const double sv = (beta == 0.0) ? w[it_] : fma(beta, s[it_], w[it_]);
    p[it_] = pv;
    s[it_] = sv;
    x[it_] = fma(alpha, pv, x[it_]);
    r[it_] = fma(-alpha, sv, r[it_]);
  }
}
// The same update in two launches (one rank, tile preconditioner): the half the next preconditioner application waits
// for -- s = w + beta s ; r -= alpha s -- and the half nothing reads before the next update -- p = (z - c n) + beta p ;
// x += alpha p --, which Engine::trace_cg_sr queues on a second stream underneath the latency-bound vertex-grid V-cycle.
__global__ void k_cg_sr_update_r(long N, const double* __restrict__ sc, const double* __restrict__ w, double* __restrict__ s,
                                 double* __restrict__ r) {
  const double alpha = sc[1], beta = sc[2];
  HDG_VEC_PROLOGUE
  for (long i = tid_; i < NP2_; i += stride_) {
    const hdg_d2 wv = as2(w)[i], rv = as2(r)[i];
    const hdg_d2 sv = (beta == 0.0) ? wv : fma2(beta, as2(s)[i], wv);
    as2(s)[i] = sv;
    as2(r)[i] = fma2(-alpha, sv, rv);
  }
  if (tail_) {
    const double sv = (beta == 0.0) ? w[it_] : fma(beta, s[it_], w[it_]);
    s[it_] = sv;
    r[it_] = fma(-alpha, sv, r[it_]);
  }
}
__global__ void k_cg_sr_update_xp(long N, const double* __restrict__ sc, const double* __restrict__ z, const double* __restrict__ nvec,
                                  double* __restrict__ p, double* __restrict__ x, long ibeg = 0) {
  const double alpha = sc[1], beta = sc[2], c = sc[3];
  HDG_VEC_PROLOGUE
  for (long i = ibeg + tid_; i < NP2_; i += stride_) {  // pairs before ibeg: done by the side jobs of the V-cycle legs
    const hdg_d2 zp = fma2(-c, as2(nvec)[i], as2(z)[i]);
    const hdg_d2 xv = as2(x)[i];
    const hdg_d2 pv = (beta == 0.0) ? zp : fma2(beta, as2(p)[i], zp);
    as2(p)[i] = pv;
    as2(x)[i] = fma2(alpha, pv, xv);
  }
  if (tail_) {
    const double zp = fma(-c, nvec[it_], z[it_]);
    const double pv = (beta == 0.0) ? zp : fma(beta, p[it_], zp);
    p[it_] = pv;
    x[it_] = fma(alpha, pv, x[it_]);
  }
}
// y = a*x + b*y
template <bool NT>
__global__ void k_axpby(long N, double a, const double* __restrict__ x, double b, double* __restrict__ y) {
  HDG_VEC_PROLOGUE
  for (long i = tid_; i < NP2_; i += stride_) {
    const hdg_d2 xv = ldv<NT>(x, i);
    if (b == 0.0) stv<NT>(y, i, hdg_d2{a * xv.x, a * xv.y});
    else {
      const hdg_d2 yv = ldv<NT>(y, i);
      stv<NT>(y, i, fma2(a, xv, hdg_d2{b * yv.x, b * yv.y}));
    }
  }
  if (tail_) y[it_] = (b == 0.0) ? a * x[it_] : fma(a, x[it_], b * y[it_]);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// partial dots of w against nv vectors V[k] (k < nv <= MAXV): part[block*nvo + k]; deterministic.
// self >= 0: V[self] is w itself and is not loaded a second time
// cross != 0 (nv >= 2, nv < MAXV): one more result, (V[0], V[1]), from the values already loaded; nvo = nv + cross
#define HDG_DOT_BLOCK 256
// entries outside the rows [lo, hi] of the (ghosted) row structure are skipped: every dot product
// counts each OWNED entry exactly once across ranks (w_ == 0: no mask)
struct RowMask {
  int w_, nrows, lo, hi;
};
__device__ __forceinline__ double row_factor(const RowMask& mk, long N, long idx) {
  // 32-bit arithmetic whenever the index fits (always, up to 2^31 entries): 64-bit division is ~4x dearer
  const int row = (N <= 0x7fffffffL) ? (int)(((unsigned)idx / (unsigned)mk.w_) % (unsigned)mk.nrows)
                                     : (int)((idx / mk.w_) % mk.nrows);
  return (row < mk.lo || row > mk.hi) ? 0.0 : 1.0;
}
// vector list passed by value (kernel arguments): no host->device pointer upload per call
template <int MAXV, typename TV = double>
struct VecList {
  const TV* p[MAXV];
};
// TV = float: the vectors of the list are a single-precision Krylov basis (V[self], if any, is still w itself)
template <int MAXV, bool NT, typename TV = double>
__global__ __launch_bounds__(HDG_DOT_BLOCK) void k_multidot(long N, const double* __restrict__ w,
                                                             const VecList<MAXV, TV> V, int nv,
                                                             double* __restrict__ part, RowMask mk, int cross, int self = -1) {
  __shared__ double sm[HDG_DOT_BLOCK / 64][MAXV];
  double acc[MAXV];
  double accx = 0.0;  // cross product (own register: a runtime index into acc[] would demote it to scratch)
#pragma unroll
  for (int k = 0; k < MAXV; k++) acc[k] = 0.0;
  // U independent PAIRS per trip (lean instantiation only): all their loads are issued before the first FMA.
  // Masked-out rows and the tail are handled by a 0/1 factor on w instead of a branch (every index read is a
  // valid one: ghost rows exist, the tail is clamped), so the loads of a trip never wait for a predicate.
  constexpr int U = MAXV <= 4 ? 2 : 1;
  HDG_VEC_PROLOGUE
  const bool same_row = (mk.w_ & 1) == 0;  // even row length: both entries of a pair lie in the same row
  for (long base = tid_; base < NP2_; base += U * stride_) {
    hdg_d2 wv[U], vv[U][MAXV];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i0 = base + u * stride_;
      const bool inb = i0 < NP2_;
      const long i = inb ? i0 : tid_;
      double m0 = inb ? 1.0 : 0.0, m1 = m0;
      if (mk.w_ > 0) {
        m0 *= row_factor(mk, N, 2 * i);
        m1 = same_row ? m0 : m1 * row_factor(mk, N, 2 * i + 1);
      }
      const hdg_d2 t = ldv<NT>(w, i);
      wv[u] = hdg_d2{t.x * m0, t.y * m1};
      // self: V[self] IS w (norms, the (w, w) entry of a Gram-Schmidt pass): its values are already here
#pragma unroll
      for (int k = 0; k < MAXV; k++) vv[u][k] = (k < nv) ? (k == self ? t : ldv<NT>(V.p[k], i)) : hdg_d2{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
#pragma unroll
      for (int k = 0; k < MAXV; k++)
        if (k < nv) acc[k] = fma(wv[u].y, vv[u][k].y, fma(wv[u].x, vv[u][k].x, acc[k]));
    }
    if (cross) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long i0 = base + u * stride_;
        double m0 = i0 < NP2_ ? 1.0 : 0.0, m1 = m0;
        if (mk.w_ > 0 && i0 < NP2_) {
          m0 = row_factor(mk, N, 2 * i0);
          m1 = same_row ? m0 : row_factor(mk, N, 2 * i0 + 1);
        }
        accx = fma(m1 * vv[u][0].y, vv[u][1].y, fma(m0 * vv[u][0].x, vv[u][1].x, accx));
      }
    }
  }
  if (tail_) {
    const double mf = mk.w_ > 0 ? row_factor(mk, N, it_) : 1.0;
    for (int k = 0; k < nv; k++) acc[k] = fma(w[it_] * mf, k == self ? w[it_] : (double)V.p[k][it_], acc[k]);
    if (cross) accx = fma(mf * (double)V.p[0][it_], (double)V.p[1][it_], accx);
  }
  const int nvo = nv + (cross ? 1 : 0);
  const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < MAXV; k++) {
    if (k < nv) {
      const double s = wave_sum(acc[k]);
      if (lane == 0) sm[wv_][k] = s;
    }
  }
  if (cross) {
    const double s = wave_sum(accx);
    if (lane == 0) sm[wv_][nv] = s;
  }
  __syncthreads();
  if (threadIdx.x < nvo) {
    double s = 0.0;
    for (int w2 = 0; w2 < HDG_DOT_BLOCK / 64; w2++) s += sm[w2][threadIdx.x];
    part[(long)blockIdx.x * nvo + threadIdx.x] = s;
  }
}
// hres != nullptr: the result goes to pinned host memory as well (the host reads it after a stream / event synchronisation:
// no copy kernel, which costs 8 us on the stream)
// transposed != 0: part[k * nblocks + b] (the tile kernels of the trace preconditioner: coalesced for the second stage)
__global__ void k_reduce_parts(int nblocks, int nv, const double* __restrict__ part, double* __restrict__ res,
                               double* __restrict__ hres = nullptr, int transposed = 0) {
  const int k = blockIdx.x;
  double acc = 0.0;
  if (transposed) for (int b = threadIdx.x; b < nblocks; b += blockDim.x) acc += part[(long)k * nblocks + b];
  else for (int b = threadIdx.x; b < nblocks; b += blockDim.x) acc += part[(long)b * nv + k];
  __shared__ double sm[4];
  const double s = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) tot += sm[w];
    res[k] = tot;
    if (hres) hres[k] = tot;
  }
}
// Second reduction stage of the tile preconditioner's five inner products AND the scalars of the single-reduction CG in
// one launch (one rank: no all-reduce in between): one workgroup of 1024 threads, fixed summation order (deterministic).
__global__ __launch_bounds__(1024) void k_cg_sr_reduce_scalars(int nblocks, const double* __restrict__ part, double* __restrict__ res,
                                                               double* __restrict__ sc, double nn, int first, double* __restrict__ hsc) {
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int b = threadIdx.x; b < nblocks; b += 1024) {  // part[q * nblocks + tile]: as the tile kernels leave it
#pragma unroll
    for (int q = 0; q < 5; q++) acc[q] += part[(long)q * nblocks + b];
  }
  __shared__ double sm[16][5];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 5; q++) {
    const double sv = wave_sum(acc[q]);
    if (lane == 0) sm[wv][q] = sv;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double r5[5];
#pragma unroll
  for (int q = 0; q < 5; q++) {
    double t = 0.0;
    for (int w = 0; w < 16; w++) t += sm[w][q];
    r5[q] = t;
    res[q] = t;
  }
  // == k_cg_sr_scalars
  const double c = r5[0] / nn;
  const double rzn = r5[1] - c * r5[4];
  const double zz = r5[2] - c * r5[0];
  if (!(zz > 1e-6 * r5[2])) sc[6] = 2.0;
  const double beta = first ? 0.0 : rzn / sc[0];
  const double pAp = first ? r5[3] : r5[3] - beta * rzn / sc[1];
  double alpha = rzn / pAp;
  if (rzn == 0.0) alpha = 0.0;
  else if (!(pAp > 0.0)) { sc[6] = 1.0; alpha = 0.0; }
  sc[3] = c;
  sc[4] = zz;
  sc[7] = r5[2];
  sc[5] = pAp;
  sc[2] = (rzn == 0.0) ? 0.0 : beta;
  sc[1] = alpha;
  sc[0] = rzn;
  if (hsc) {
#pragma unroll
    for (int q = 0; q < 8; q++) hsc[q] = sc[q];
  }
}
// coefficient list passed by value (Gram-Schmidt / basis updates)
struct Coefs {
  double c[32];
};
// ------------------------------------------------------------------------------------------
// s-step minimal-residual cycle of the tentative-velocity solver (round 4; Engine::sstep_mr).  A GMRES iteration with
// classical Gram-Schmidt reads the basis twice (inner products, update): 2j + 5 velocity-vector passes at step j, 60 for a
// cycle of six -- more than the operator and the preconditioner of those six iterations.  Here the cycle first builds the
// power basis K_0 = M(b - A x), K_i = (M A) K_{i-1} with no inner product at all, then ONE pass over the s + 1 vectors gives
// their Gram matrix (k_gram), the host solves the (s x s) least-squares problem min |K_0 - sum y_i K_i|, and ONE more pass
// forms x += sum y_i K_{i-1} and the new residual K_0 - sum y_i K_i together with its norm (k_sstep_update): 2 s + 5 passes
// per cycle, 17 instead of 60 for s = 6.  A cycle whose predicted residual misses the target is EXTENDED (more basis vectors, the
// Gram matrix once more) instead of restarted, up to s = 8: the iteration counts of GMRES(8) at a third of its vector traffic.
// ------------------------------------------------------------------------------------------
// Gram matrix of the first nv <= NV vectors over the owned entries: part[block * npair + p], npair = nv (nv + 1) / 2, pairs
// (a <= b) in row-major order of the nv x nv upper triangle
#define HDG_SSTEP_MAXV 9
template <int NV, bool NT>
__global__ __launch_bounds__(HDG_DOT_BLOCK) void k_gram(long N, const VecList<NV> V, int nv, double* __restrict__ part, RowMask mk) {
  constexpr int NPAIR = NV * (NV + 1) / 2;
  __shared__ double sm[HDG_DOT_BLOCK / 64][NPAIR];
  double acc[NPAIR];
#pragma unroll
  for (int p = 0; p < NPAIR; p++) acc[p] = 0.0;
  HDG_VEC_PROLOGUE
  const bool same_row = (mk.w_ & 1) == 0;
  for (long i = tid_; i < NP2_; i += stride_) {
    double m0 = 1.0, m1 = 1.0;
    if (mk.w_ > 0) {
      m0 = row_factor(mk, N, 2 * i);
      m1 = same_row ? m0 : row_factor(mk, N, 2 * i + 1);
    }
    hdg_d2 v[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = (k < nv) ? ldv<NT>(V.p[k], i) : hdg_d2{0.0, 0.0};
    int p = 0;  // index in the FULL NV x NV triangle (compile-time after unrolling)
#pragma unroll
    for (int a = 0; a < NV; a++) {
      const double ax = v[a].x * m0, ay = v[a].y * m1;
#pragma unroll
      for (int b = a; b < NV; b++, p++)
        if (b < nv) acc[p] = fma(ay, v[b].y, fma(ax, v[b].x, acc[p]));
    }
  }
  if (tail_) {
    const double mf = mk.w_ > 0 ? row_factor(mk, N, it_) : 1.0;
    int p = 0;
#pragma unroll
    for (int a = 0; a < NV; a++) {
#pragma unroll
      for (int b = a; b < NV; b++, p++)
        if (b < nv) acc[p] = fma(mf * V.p[a][it_], V.p[b][it_], acc[p]);
    }
  }
  const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
#pragma unroll
  for (int p = 0; p < NPAIR; p++) {
    const double s = wave_sum(acc[p]);
    if (lane == 0) sm[wv_][p] = s;
  }
  __syncthreads();
  // compact the full triangle to the nv x nv one
  if (threadIdx.x < NPAIR) {
    int a = 0, rem = threadIdx.x;
    while (rem >= NV - a) { rem -= NV - a; a++; }
    const int b = a + rem;
    if (b < nv) {
      double s = 0.0;
      for (int w2 = 0; w2 < HDG_DOT_BLOCK / 64; w2++) s += sm[w2][threadIdx.x];
      const int q = a * nv - a * (a - 1) / 2 + (b - a);
      part[(long)blockIdx.x * (nv * (nv + 1) / 2) + q] = s;
    }
  }
}
// x += sum_{k<nv} cx[k] V_k;  V_0 <- sum_{k<nv} cr[k] V_k (the new preconditioned residual; whole arrays, ghost rows follow);
// part[block] = the block's share of |new V_0|^2 over the owned entries.  u_out / c_out (may be null, may alias a V_k): the
// correction just applied, x_new - x_old, and its image B (x_new - x_old) = V_0_old - V_0_new -- the pair the next cycle is
// augmented with (LGMRES: the error approximation of a cycle carries the information a restart throws away).
#define HDG_SSTEP_MAXU 11
template <int NV, bool NT>
__global__ __launch_bounds__(HDG_DOT_BLOCK) void k_sstep_update(long N, double* __restrict__ x, double* __restrict__ k0, const VecList<NV> K, int nv,
                                                                Coefs cx, Coefs cr, double* __restrict__ part, RowMask mk,
                                                                double* u_out = nullptr, double* c_out = nullptr) {
  __shared__ double sm[HDG_DOT_BLOCK / 64];
  double acc = 0.0;
  HDG_VEC_PROLOGUE
  const bool same_row = (mk.w_ & 1) == 0;
  for (long i = tid_; i < NP2_; i += stride_) {
    double m0 = 1.0, m1 = 1.0;
    if (mk.w_ > 0) {
      m0 = row_factor(mk, N, 2 * i);
      m1 = same_row ? m0 : row_factor(mk, N, 2 * i + 1);
    }
    hdg_d2 v[NV];
    v[0] = ldv<NT>(k0, i);
#pragma unroll
    for (int k = 1; k < NV; k++) v[k] = (k < nv) ? ldv<NT>(K.p[k], i) : hdg_d2{0.0, 0.0};
    hdg_d2 r = hdg_d2{0.0, 0.0}, du = hdg_d2{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NV; k++) {
      if (k < nv) {
        r = fma2(cr.c[k], v[k], r);
        du = fma2(cx.c[k], v[k], du);
      }
    }
    const hdg_d2 xo = ldv<NT>(x, i);
    stv<NT>(x, i, hdg_d2{xo.x + du.x, xo.y + du.y});
    stv<NT>(k0, i, r);
    if (u_out) {
      stv<NT>(u_out, i, du);
      stv<NT>(c_out, i, hdg_d2{v[0].x - r.x, v[0].y - r.y});
    }
    acc = fma(m1 * r.y, r.y, fma(m0 * r.x, r.x, acc));
  }
  if (tail_) {
    const double mf = mk.w_ > 0 ? row_factor(mk, N, it_) : 1.0;
    double du = 0.0, r = 0.0;
    const double v0 = k0[it_];
    for (int k = 0; k < nv; k++) {
      const double vk = k == 0 ? v0 : K.p[k][it_];
      r = fma(cr.c[k], vk, r);
      du = fma(cx.c[k], vk, du);
    }
    x[it_] += du;
    k0[it_] = r;
    if (u_out) { u_out[it_] = du; c_out[it_] = v0 - r; }
    acc = fma(mf * r, r, acc);
  }
  const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
  const double s = wave_sum(acc);
  if (lane == 0) sm[wv_] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w2 = 0; w2 < HDG_DOT_BLOCK / 64; w2++) t += sm[w2];
    part[blockIdx.x] = t;
  }
}

// Chebyshev step on velocity vectors, three-term form:  pn = x + c1 (x - pn) + c2 z   (pn: x_{n-1} -> x_{n+1})
template <bool NT>
__global__ void k_cheb_update(long N, double* __restrict__ pn, const double* __restrict__ z, const double* __restrict__ x,
                              double c1, double c2) {
  HDG_VEC_PROLOGUE
  for (long i = tid_; i < NP2_; i += stride_) {
    const hdg_d2 xv = ldv<NT>(x, i), zv = ldv<NT>(z, i);
    const hdg_d2 pv = (c1 != 0.0) ? ldv<NT>(pn, i) : xv;
    stv<NT>(pn, i, hdg_d2{fma(c1, xv.x - pv.x, fma(c2, zv.x, xv.x)), fma(c1, xv.y - pv.y, fma(c2, zv.y, xv.y))});
  }
  if (tail_) {
    const double xv = x[it_];
    const double pv = (c1 != 0.0) ? pn[it_] : xv;
    pn[it_] = fma(c1, xv - pv, fma(c2, z[it_], xv));
  }
}

// coefficients passed by value (kernel arguments): no host->device copy, no extra sync
// out = scale * (w - sum_k h[k] V[k])     (classical Gram-Schmidt update fused with the normalisation)
// TV = float (single-precision basis storage): the new vector is ROUNDED to single precision, stored in outf, and out
// receives the same rounded values as doubles (the operator kernels read doubles): the Arnoldi relation then holds for
// one and the same set of vectors, only their orthonormality is accurate to single precision
template <int MAXV, bool NT, typename TV = double>
__global__ void k_gs_update(long N, const double* __restrict__ w, const TV* const* __restrict__ V, Coefs h, int nv,
                            double scale, double* __restrict__ out, float* __restrict__ outf = nullptr) {
  constexpr bool F32 = sizeof(TV) == 4;
  HDG_VEC_PROLOGUE
  for (long i = tid_; i < NP2_; i += stride_) {
    hdg_d2 acc = ldv<NT>(w, i);
#pragma unroll
    for (int k = 0; k < MAXV; k++)
      if (k < nv) acc = fma2(-h.c[k], ldv<NT>(V[k], i), acc);
    if (F32) {
      const hdg_f2 f{(float)(scale * acc.x), (float)(scale * acc.y)};
      reinterpret_cast<hdg_f2*>(outf)[i] = f;
      stv<NT>(out, i, hdg_d2{(double)f.x, (double)f.y});
    } else {
      stv<NT>(out, i, hdg_d2{scale * acc.x, scale * acc.y});
    }
  }
  if (tail_) {
    double acc = w[it_];
    for (int k = 0; k < nv; k++) acc = fma(-h.c[k], (double)V[k][it_], acc);
    if (F32) {
      const float f = (float)(scale * acc);
      outf[it_] = f;
      out[it_] = (double)f;
    } else {
      out[it_] = scale * acc;
    }
  }
}
// x += sum_k y[k] V[k]
template <int MAXV, bool NT, typename TV = double>
__global__ void k_basis_axpy(long N, double* __restrict__ x, const TV* const* __restrict__ V, Coefs y, int nv) {
  HDG_VEC_PROLOGUE
  for (long i = tid_; i < NP2_; i += stride_) {
    hdg_d2 acc = ldv<NT>(x, i);
#pragma unroll
    for (int k = 0; k < MAXV; k++)
      if (k < nv) acc = fma2(y.c[k], ldv<NT>(V[k], i), acc);
    stv<NT>(x, i, acc);
  }
  if (tail_) {
    double acc = x[it_];
    for (int k = 0; k < nv; k++) acc = fma(y.c[k], (double)V[k][it_], acc);
    x[it_] = acc;
  }
}

// pressure / trace mean shift (hdg_imex.py:471-478): p -= pbar, lambda -= pbar (modal mode 0 only)
__global__ void k_shift_p(long Nc, double* __restrict__ p, const double* __restrict__ sum0, double factor, double c0) {
  // pbar = factor * sum0[0];  p_{K,0} -= pbar * c0
  const double pbar = factor * sum0[0];
  const long stride = (long)gridDim.x * blockDim.x;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < Nc; idx += stride) p[idx] -= pbar * c0;
}
__global__ void k_shift_l(Geo g, int NL, double* __restrict__ l, const double* __restrict__ sum0, double factor,
                          double sH, double sV, double sD) {
  HDG_CORNER_PROLOGUE
  const double pbar = factor * sum0[0];
  if (in_x) l[((long)0 * NL) * g.G + o] -= pbar * sH;
  if (in_y) l[((long)1 * NL) * g.G + o] -= pbar * sV;
  if (in_x && in_y) l[((long)2 * NL) * g.G + o] -= pbar * sD;
}

// ------------------------------------------------------------------------------------------
// geometric multigrid on the P1 vertex grid (coarse space of the GTMG preconditioner,
// hdg_imex.py:97-118,139-167).  Vertex (i,j) of an (n+1)x(n+1) grid at index j*(n+1)+i.
// Operator: P1 stiffness matrix of the right-triangle mesh = 5-point Laplacian with
// half weights along boundary edges (homogeneous Neumann).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void p1_stencil(const double* __restrict__ x, int n, int i, int j, double& diag, double& off) {
  const int st = n + 1;
  const double wx = (j == 0 || j == n) ? 0.5 : 1.0;  // weight of horizontal edges in this row
  const double wy = (i == 0 || i == n) ? 0.5 : 1.0;
  diag = 0.0;
  off = 0.0;
  if (i > 0) { diag += wx; off += wx * x[j * st + i - 1]; }
  if (i < n) { diag += wx; off += wx * x[j * st + i + 1]; }
  if (j > 0) { diag += wy; off += wy * x[(j - 1) * st + i]; }
  if (j < n) { diag += wy; off += wy * x[(j + 1) * st + i]; }
}
// red-black Gauss-Seidel half sweep on A x = b
__global__ void k_p1_rbgs(int n, double* __restrict__ x, const double* __restrict__ b, int colour) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i > n || ((i + j) & 1) != colour) return;
  double diag, off;
  p1_stencil(x, n, i, j, diag, off);
  x[j * (n + 1) + i] = (b[j * (n + 1) + i] + off) / diag;
}
__global__ void k_p1_residual(int n, const double* __restrict__ x, const double* __restrict__ b, double* __restrict__ r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i > n) return;
  double diag, off;
  p1_stencil(x, n, i, j, diag, off);
  r[j * (n + 1) + i] = b[j * (n + 1) + i] - (diag * x[j * (n + 1) + i] - off);
}
// restriction = transpose of nested P1 interpolation (coarse diagonal joins (I+1,J) and (I,J+1))
__global__ void k_p1_restrict(int nc, const double* __restrict__ rf, double* __restrict__ rc) {
  const int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y;
  if (I > nc) return;
  const int nf = 2 * nc, st = nf + 1, i = 2 * I, j = 2 * J;
  double acc = rf[j * st + i];
  if (i > 0) acc += 0.5 * rf[j * st + i - 1];
  if (i < nf) acc += 0.5 * rf[j * st + i + 1];
  if (j > 0) acc += 0.5 * rf[(j - 1) * st + i];
  if (j < nf) acc += 0.5 * rf[(j + 1) * st + i];
  if (i > 0 && j < nf) acc += 0.5 * rf[(j + 1) * st + i - 1];
  if (i < nf && j > 0) acc += 0.5 * rf[(j - 1) * st + i + 1];
  rc[J * (nc + 1) + I] = acc;
}
__global__ void k_p1_prolong_add(int nc, const double* __restrict__ xc, double* __restrict__ xf) {
  const int nf = 2 * nc;
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i > nf) return;
  const int I = i >> 1, J = j >> 1, sc = nc + 1;
  double v;
  if (!(i & 1) && !(j & 1)) v = xc[J * sc + I];
  else if ((i & 1) && !(j & 1)) v = 0.5 * (xc[J * sc + I] + xc[J * sc + I + 1]);
  else if (!(i & 1) && (j & 1)) v = 0.5 * (xc[J * sc + I] + xc[(J + 1) * sc + I]);
  else v = 0.5 * (xc[J * sc + I + 1] + xc[(J + 1) * sc + I]);
  xf[j * (nf + 1) + i] += v;
}
// z = r / d (Jacobi preconditioner of the continuous-space mass matrix)
__global__ void k_pointwise_div(long N, const double* __restrict__ r, const double* __restrict__ d, double* __restrict__ z) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < N; idx += stride) z[idx] = r[idx] / d[idx];
}
__global__ void k_fill(long N, double* __restrict__ x, double v) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < N; idx += stride) x[idx] = v;
}

// ------------------------------------------------------------------------------------------
// Fused legs of one V-cycle level (levels too large for the LDS tail below).  Per level the unfused path
// is 12 launches of ~6 us latency-bound kernels (fill, 2 nsw half sweeps, residual, restriction | prolongation,
// 2 nsw half sweeps); here each leg is ONE kernel: a workgroup holds a TS x TS tile of vertices plus a halo in
// LDS and recomputes the halo (temporal blocking; a half sweep has dependency radius 1, so after m half
// sweeps everything further than m from the edge of the loaded region is exact).  Every vertex value is
// produced by the same expression in the same order as in k_p1_rbgs / k_p1_residual / k_p1_restrict /
// k_p1_prolong_add, so the results are bit-identical to the unfused path.
// ------------------------------------------------------------------------------------------
#define HDG_P1_TS 32
#define HDG_P1_MAXSW 3
#define HDG_P1_THREADS 1024
// Side job of a leg launch (Engine::trace_cg_sr, one rank): the legs are latency-bound (8 barrier-separated LDS phases on a
// grid of at most 33 x 33 tiles) and leave HBM idle, so the half of the CG update that nothing reads before the next
// update -- p = (z - c n) + beta p ; x += alpha p (k_cg_sr_update_xp) -- rides along: the workgroups of the tile rows
// >= nrows of a leg launch each take 1024 sixteen-byte pairs of the slice [i0, i1) instead of a tile.  (A second stream for
// the same purpose gained 14 us per CG iteration of the 64 possible: two cross-stream dependencies cost 17 us.)
struct SideXP {
  const double* sc;
  const double* z;
  const double* nvec;
  double* p;
  double* x;
  long i0, i1;
};
__device__ __forceinline__ void side_xp(const SideXP& sj, long blk, long nblk) {
  const double alpha = sj.sc[1], beta = sj.sc[2], c = sj.sc[3];
  for (long i = sj.i0 + blk * HDG_P1_THREADS + threadIdx.x; i < sj.i1; i += nblk * HDG_P1_THREADS) {
    const hdg_d2 zp = fma2(-c, as2(sj.nvec)[i], as2(sj.z)[i]);
    const hdg_d2 xv = as2(sj.x)[i];
    const hdg_d2 pv = (beta == 0.0) ? zp : fma2(beta, as2(sj.p)[i], zp);
    as2(sj.p)[i] = pv;
    as2(sj.x)[i] = fma2(alpha, pv, xv);
  }
}
// by = the tile row of a workgroup that has one; side rows do their share of the side job and leave (hdg_side_rows.hpp)
#define HDG_P1_SIDE_JOB                                                                                              \
  const SideRow srow_ = side_row_of((int)blockIdx.y, (int)gridDim.y, extra, period);                                 \
  if (srow_.side) {                                                                                                  \
    side_xp(sj, (long)srow_.idx * gridDim.x + blockIdx.x, (long)extra * gridDim.x);                                  \
    return;                                                                                                          \
  }                                                                                                                  \
  const int by = srow_.idx;
__device__ __forceinline__ int pw2(int a, int n) { return a < 0 ? a + n : (a >= n ? a - n : a); }  // periodic index (|shift| < n)
// region-local stencil: (gi, gj) global vertex, (li, lj) local; false if a neighbour inside the domain lies
// outside the loaded region (the point is then part of the garbage ring and is skipped).
// The diagonal is 4 (interior), 2 (boundary edge) or 1 (corner): its reciprocal is exact, so multiplying by
// `inv` is bit-identical to the division in k_p1_rbgs.
template <int W, bool PER = false>
__device__ __forceinline__ bool p1_stencil_tile(const double* X, int n, int gi, int gj, int li, int lj, double& diag,
                                                double& off) {
  if (PER) {  // periodic n x n grid: every vertex is interior (same order of the four terms as k_p1p_rbgs / k_p1p_residual)
    diag = 4.0;
    off = 0.0;
    if (li == 0 || li == W - 1 || lj == 0 || lj == W - 1) return false;
    off = X[lj * W + li - 1] + X[lj * W + li + 1] + X[(lj - 1) * W + li] + X[(lj + 1) * W + li];
    return true;
  }
  const double wx = (gj == 0 || gj == n) ? 0.5 : 1.0;
  const double wy = (gi == 0 || gi == n) ? 0.5 : 1.0;
  diag = 0.0;
  off = 0.0;
  if ((gi > 0 && li == 0) || (gi < n && li == W - 1) || (gj > 0 && lj == 0) || (gj < n && lj == W - 1)) return false;
  if (gi > 0) { diag += wx; off += wx * X[lj * W + li - 1]; }
  if (gi < n) { diag += wx; off += wx * X[lj * W + li + 1]; }
  if (gj > 0) { diag += wy; off += wy * X[(lj - 1) * W + li]; }
  if (gj < n) { diag += wy; off += wy * X[(lj + 1) * W + li]; }
  return true;
}
// 2 nsw half sweeps; each thread owns the q-th vertex of the active colour (W even: W/2 per row and colour)
template <int W, bool PER = false>
__device__ __forceinline__ void p1_tile_sweeps(double* X, const double* B, int n, int gi0, int gj0, int nsw, bool reverse) {
  for (int hs = 0; hs < 2 * nsw; hs++) {
    const int colour = ((hs & 1) == 0) ? (reverse ? 1 : 0) : (reverse ? 0 : 1);
    for (int q = threadIdx.x; q < W * W / 2; q += HDG_P1_THREADS) {
      const int lj = q / (W / 2), li = 2 * (q - lj * (W / 2)) + ((colour + gi0 + gj0 + lj) & 1);
      const int gi = gi0 + li, gj = gj0 + lj;
      if (!PER && (gi < 0 || gj < 0 || gi > n || gj > n)) continue;
      double diag, off;
      if (p1_stencil_tile<W, PER>(X, n, gi, gj, li, lj, diag, off)) {
        const double inv = diag == 4.0 ? 0.25 : (diag == 2.0 ? 0.5 : 1.0 / diag);
        X[lj * W + li] = (B[lj * W + li] + off) * inv;
      }
    }
    __syncthreads();
  }
}
// down leg: x = 0, nsw sweeps on A x = b, r = b - A x, bc = R r (coarse right-hand side); x is stored to xpre
// PER: the doubly periodic n x n vertex grid (indices wrap; the coarse grid is n/2 x n/2; same expressions as k_p1p_*)
template <int NSW, bool PER = false>
__global__ __launch_bounds__(HDG_P1_THREADS) void k_p1_down(int n, const double* __restrict__ b, double* __restrict__ xpre,
                                                            double* __restrict__ bc, int jt0 = 0, int extra = 0, int period = 2, SideXP sj = SideXP{}) {
  constexpr int H = 2 * NSW + 2, W = HDG_P1_TS + 2 * H;
  __shared__ double X[W * W];
  __shared__ double B[W * W];
  HDG_P1_SIDE_JOB
  const int st = PER ? n : n + 1, last = PER ? n - 1 : n;  // row pitch, last vertex index
  const int i0 = blockIdx.x * HDG_P1_TS, j0 = (by + jt0) * HDG_P1_TS, gi0 = i0 - H, gj0 = j0 - H;  // jt0: first tile row of this launch
  for (int p = threadIdx.x; p < W * W; p += HDG_P1_THREADS) {
    const int lj = p / W, li = p - lj * W, gi = gi0 + li, gj = gj0 + lj;
    if (PER) B[p] = b[pw2(gj, n) * st + pw2(gi, n)];
    else {
      const bool in = gi >= 0 && gj >= 0 && gi <= n && gj <= n;
      B[p] = in ? b[gj * st + gi] : 0.0;
    }
    X[p] = 0.0;
  }
  __syncthreads();
  p1_tile_sweeps<W, PER>(X, B, n, gi0, gj0, NSW, false);
  // store the tile's x, then overwrite B by the residual (r_p needs b_p and x only)
  {
    const int p = threadIdx.x;  // HDG_P1_TS^2 == HDG_P1_THREADS
    const int tj = p / HDG_P1_TS, ti = p - tj * HDG_P1_TS, gi = i0 + ti, gj = j0 + tj;
    if (gi <= last && gj <= last) xpre[gj * st + gi] = X[(tj + H) * W + ti + H];
  }
  for (int p = threadIdx.x; p < W * W; p += HDG_P1_THREADS) {
    const int lj = p / W, li = p - lj * W, gi = gi0 + li, gj = gj0 + lj;
    if (!PER && (gi < 0 || gj < 0 || gi > n || gj > n)) continue;
    double diag, off;
    if (p1_stencil_tile<W, PER>(X, n, gi, gj, li, lj, diag, off)) B[p] = B[p] - (diag * X[p] - off);
  }
  __syncthreads();
  constexpr int HT = HDG_P1_TS / 2;
  const int nc = n >> 1;
  if (threadIdx.x < HT * HT) {
    const int p = threadIdx.x;
    const int tJ = p / HT, tI = p - tJ * HT, i = i0 + 2 * tI, j = j0 + 2 * tJ;
    if (i <= last && j <= last) {
      const int li = i - gi0, lj = j - gj0;
      if (PER) {  // (the order of k_p1p_restrict)
        bc[(j >> 1) * nc + (i >> 1)] = B[lj * W + li] + 0.5 * (B[lj * W + li - 1] + B[lj * W + li + 1] + B[(lj - 1) * W + li] + B[(lj + 1) * W + li] +
                                                                 B[(lj + 1) * W + li - 1] + B[(lj - 1) * W + li + 1]);
      } else {
        double acc = B[lj * W + li];
        if (i > 0) acc += 0.5 * B[lj * W + li - 1];
        if (i < n) acc += 0.5 * B[lj * W + li + 1];
        if (j > 0) acc += 0.5 * B[(lj - 1) * W + li];
        if (j < n) acc += 0.5 * B[(lj + 1) * W + li];
        if (i > 0 && j < n) acc += 0.5 * B[(lj + 1) * W + li - 1];
        if (i < n && j > 0) acc += 0.5 * B[(lj - 1) * W + li + 1];
        bc[(j >> 1) * (nc + 1) + (i >> 1)] = acc;
      }
    }
  }
}
// up leg: x = xpre + P xc, nsw sweeps with the colours reversed.  xpre (the down leg's result) and x are
// DIFFERENT buffers: a workgroup reads the halo of its tile while its neighbours store theirs.
template <int NSW, bool PER = false>
__global__ __launch_bounds__(HDG_P1_THREADS) void k_p1_up(int n, const double* __restrict__ xc, const double* __restrict__ b,
                                                          const double* __restrict__ xpre, double* __restrict__ x, int jt0 = 0, int extra = 0, int period = 2,
                                                          SideXP sj = SideXP{}) {
  constexpr int H = 2 * NSW, W = HDG_P1_TS + 2 * H;
  __shared__ double X[W * W];
  __shared__ double B[W * W];
  HDG_P1_SIDE_JOB
  const int st = PER ? n : n + 1, last = PER ? n - 1 : n, nc = n >> 1, sc = PER ? nc : nc + 1;
  const int i0 = blockIdx.x * HDG_P1_TS, j0 = (by + jt0) * HDG_P1_TS, gi0 = i0 - H, gj0 = j0 - H;  // jt0: first tile row of this launch
  for (int p = threadIdx.x; p < W * W; p += HDG_P1_THREADS) {
    const int lj = p / W, li = p - lj * W;
    int i = gi0 + li, j = gj0 + lj;
    if (PER) { i = pw2(i, n); j = pw2(j, n); }
    else if (i < 0 || j < 0 || i > n || j > n) { X[p] = 0.0; B[p] = 0.0; continue; }
    const int I = i >> 1, J = j >> 1;
    const int I1 = PER ? (I + 1 == nc ? 0 : I + 1) : I + 1, J1 = PER ? (J + 1 == nc ? 0 : J + 1) : J + 1;
    double v;
    if (!(i & 1) && !(j & 1)) v = xc[J * sc + I];
    else if ((i & 1) && !(j & 1)) v = 0.5 * (xc[J * sc + I] + xc[J * sc + I1]);
    else if (!(i & 1) && (j & 1)) v = 0.5 * (xc[J * sc + I] + xc[J1 * sc + I]);
    else v = 0.5 * (xc[J * sc + I1] + xc[J1 * sc + I]);
    X[p] = xpre[j * st + i] + v;
    B[p] = b[j * st + i];
  }
  __syncthreads();
  p1_tile_sweeps<W, PER>(X, B, n, gi0, gj0, NSW, true);
  {
    const int p = threadIdx.x;
    const int tj = p / HDG_P1_TS, ti = p - tj * HDG_P1_TS, gi = i0 + ti, gj = j0 + tj;
    if (gi <= last && gj <= last) x[gj * st + gi] = X[(tj + H) * W + ti + H];
  }
}

// ------------------------------------------------------------------------------------------
// The coarse tail of the V-cycle (all levels with n <= 32, i.e. <= 33^2 vertices) in ONE workgroup with
// every level resident in LDS: replaces ~70 launches of 1-4 us kernels per V-cycle by one.
// Same algorithm as the per-level kernels: V(nsw,nsw) with red-black Gauss-Seidel (colours swapped on
// the way up), residual restriction by the transpose of the nested P1 interpolation, ncoarse+ncoarse
// sweeps on the coarsest level.
// ------------------------------------------------------------------------------------------
struct P1Tail {
  int nlev;
  int n[8];
};
__device__ __forceinline__ void p1_stencil_lds(const double* x, int n, int i, int j, double& diag, double& off) {
  const int st = n + 1;
  const double wx = (j == 0 || j == n) ? 0.5 : 1.0;
  const double wy = (i == 0 || i == n) ? 0.5 : 1.0;
  diag = 0.0;
  off = 0.0;
  if (i > 0) { diag += wx; off += wx * x[j * st + i - 1]; }
  if (i < n) { diag += wx; off += wx * x[j * st + i + 1]; }
  if (j > 0) { diag += wy; off += wy * x[(j - 1) * st + i]; }
  if (j < n) { diag += wy; off += wy * x[(j + 1) * st + i]; }
}
__device__ __forceinline__ void p1_sweeps_lds(double* x, const double* b, int n, int sweeps, bool reverse) {
  const int npts = (n + 1) * (n + 1);
  // n even (every level of the tail): the row length is odd, so the points of colour c are exactly the odd / even linear
  // indices: every thread of a trip has work (n = 32: one trip of 545 points instead of two over 1089).  The diagonal is
  // 4, 2 or 1, so the multiplication by its inverse is exact (bitwise equal to the division of k_p1_rbgs).
  const int step = (n & 1) ? 1 : 2;
  for (int sw = 0; sw < 2 * sweeps; sw++) {
    const int colour = ((sw & 1) == 0) ? (reverse ? 1 : 0) : (reverse ? 0 : 1);
    for (int p = step * threadIdx.x + (step == 2 ? colour : 0); p < npts; p += step * blockDim.x) {
      const int j = p / (n + 1), i = p - j * (n + 1);
      if (step == 2 || ((i + j) & 1) == colour) {
        double diag, off;
        p1_stencil_lds(x, n, i, j, diag, off);
        const double inv = diag == 4.0 ? 0.25 : (diag == 2.0 ? 0.5 : 1.0 / diag);
        x[p] = (b[p] + off) * inv;
      }
    }
    __syncthreads();
  }
}
__device__ __forceinline__ double p1_res_lds(const double* x, const double* b, int n, int i, int j) {
  double diag, off;
  p1_stencil_lds(x, n, i, j, diag, off);
  return b[j * (n + 1) + i] - (diag * x[j * (n + 1) + i] - off);
}
#define HDG_P1_TAIL_MAX 1600
// unit_pitch > 0 (set-up of the dense form, k_p1_dense_tail): workgroup c solves for the c-th UNIT right-hand side and writes its
// result to x_out + c * unit_pitch -- every column of the tail's matrix in ONE launch
__global__ __launch_bounds__(1024) void k_p1_vcycle_tail(P1Tail tl, const double* __restrict__ b_in,
                                                          double* __restrict__ x_out, int nsw, int ncoarse, int unit_pitch = 0) {
  __shared__ double X[HDG_P1_TAIL_MAX];
  __shared__ double B[HDG_P1_TAIL_MAX];
  int offs[9];
  offs[0] = 0;
  for (int l = 0; l < tl.nlev; l++) offs[l + 1] = offs[l] + (tl.n[l] + 1) * (tl.n[l] + 1);
  if (unit_pitch > 0) {
    for (int p = threadIdx.x; p < offs[1]; p += blockDim.x) B[p] = p == (int)blockIdx.x ? 1.0 : 0.0;
    x_out += (long)blockIdx.x * unit_pitch;
  } else {
    for (int p = threadIdx.x; p < offs[1]; p += blockDim.x) B[p] = b_in[p];
  }
  for (int p = threadIdx.x; p < offs[tl.nlev]; p += blockDim.x) X[p] = 0.0;
  __syncthreads();
  for (int l = 0; l < tl.nlev - 1; l++) {
    const int n = tl.n[l], nc = tl.n[l + 1];
    double* x = X + offs[l];
    const double* b = B + offs[l];
    p1_sweeps_lds(x, b, n, nsw, false);
    double* bc = B + offs[l + 1];
    for (int p = threadIdx.x; p < (nc + 1) * (nc + 1); p += blockDim.x) {
      const int J = p / (nc + 1), I = p - J * (nc + 1), i = 2 * I, j = 2 * J;
      double acc = p1_res_lds(x, b, n, i, j);
      if (i > 0) acc += 0.5 * p1_res_lds(x, b, n, i - 1, j);
      if (i < n) acc += 0.5 * p1_res_lds(x, b, n, i + 1, j);
      if (j > 0) acc += 0.5 * p1_res_lds(x, b, n, i, j - 1);
      if (j < n) acc += 0.5 * p1_res_lds(x, b, n, i, j + 1);
      if (i > 0 && j < n) acc += 0.5 * p1_res_lds(x, b, n, i - 1, j + 1);
      if (i < n && j > 0) acc += 0.5 * p1_res_lds(x, b, n, i + 1, j - 1);
      bc[p] = acc;
    }
    __syncthreads();
  }
  {
    const int l = tl.nlev - 1;
    p1_sweeps_lds(X + offs[l], B + offs[l], tl.n[l], ncoarse, false);
    p1_sweeps_lds(X + offs[l], B + offs[l], tl.n[l], ncoarse, true);
  }
  for (int l = tl.nlev - 2; l >= 0; l--) {
    const int n = tl.n[l], nc = tl.n[l + 1], sc = nc + 1;
    double* x = X + offs[l];
    const double* xc = X + offs[l + 1];
    for (int p = threadIdx.x; p < (n + 1) * (n + 1); p += blockDim.x) {
      const int j = p / (n + 1), i = p - j * (n + 1);
      const int I = i >> 1, J = j >> 1;
      double v;
      if (!(i & 1) && !(j & 1)) v = xc[J * sc + I];
      else if ((i & 1) && !(j & 1)) v = 0.5 * (xc[J * sc + I] + xc[J * sc + I + 1]);
      else if (!(i & 1) && (j & 1)) v = 0.5 * (xc[J * sc + I] + xc[(J + 1) * sc + I]);
      else v = 0.5 * (xc[J * sc + I + 1] + xc[(J + 1) * sc + I]);
      x[p] += v;
    }
    __syncthreads();
    p1_sweeps_lds(x, B + offs[l], n, nsw, true);
  }
  for (int p = threadIdx.x; p < offs[1]; p += blockDim.x) x_out[p] = X[p];
}

// Dense form of the tail (round 4).  The tail is a fixed LINEAR map of its right-hand side (zero start, fixed sweeps), so
// it is one matrix M of (n+1)^2 <= 1089 rows: k_p1_vcycle_tail walks through ~48 barrier-separated phases in ONE workgroup
// (33.7 us at every mesh size, a quarter of a V-cycle at C3); M b is 9.5 MB of matrix from the L2 / Infinity Cache spread
// over 273 workgroups.  M is built once per engine by applying the tail kernel to the unit vectors (the same operator up
// to the rounding of the summation order); one wave per row, 16-byte loads, deterministic wave reduction.
__global__ __launch_bounds__(256) void k_p1_dense_tail(int N, int pitch, const double* __restrict__ M, const double* __restrict__ b,
                                                       double* __restrict__ x) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= N) return;
  const double* __restrict__ mr = M + (long)row * pitch;
  double acc0 = 0.0, acc1 = 0.0;
  for (int c = 2 * lane; c < N; c += 128) {
    const hdg_d2 m = *reinterpret_cast<const hdg_d2*>(mr + c);
    acc0 = fma(m.x, b[c], acc0);
    if (c + 1 < N) acc1 = fma(m.y, b[c + 1], acc1);
  }
  const double acc = wave_sum(acc0 + acc1);
  if (lane == 0) x[row] = acc;
}
__global__ void k_transpose_sq(int N, int pitch, const double* __restrict__ in, double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < N) out[(long)r * pitch + c] = in[(long)c * pitch + r];
}

// trace <-> P1 transfer: P = edge-wise L2 projection of the P1 function (hdg_imex.py:491-503,
// without the reference's 1/2 on interior edges - a preconditioner detail)
__global__ void k_p1_to_trace(Geo g, int NL, const double* __restrict__ xc, double* __restrict__ l, double accumulate,
                              double lH, double lV, double lD) {
  HDG_CORNER_PROLOGUE
  const int st = g.nx + 1;
  const long J = g.joff + j;  // global vertex row; xc is the global (replicated) vertex vector
  const double r3 = 0.57735026918962576451;
  const double v00 = xc[J * st + i];
  if (in_x) {
    const double vb = xc[J * st + i + 1], sl = sqrt(lH);
    double* p0 = l + ((long)0 * NL) * g.G + o;
    double* p1 = l + ((long)0 * NL + 1) * g.G + o;
    *p0 = accumulate * (*p0) + sl * 0.5 * (v00 + vb);
    *p1 = accumulate * (*p1) + sl * r3 * 0.5 * (vb - v00);
  }
  if (in_y) {
    const double vb = xc[(J + 1) * st + i], sl = sqrt(lV);
    double* p0 = l + ((long)1 * NL) * g.G + o;
    double* p1 = l + ((long)1 * NL + 1) * g.G + o;
    *p0 = accumulate * (*p0) + sl * 0.5 * (v00 + vb);
    *p1 = accumulate * (*p1) + sl * r3 * 0.5 * (vb - v00);
  }
  if (in_x && in_y) {
    const double va = xc[J * st + i + 1], vb = xc[(J + 1) * st + i], sl = sqrt(lD);
    double* p0 = l + ((long)2 * NL) * g.G + o;
    double* p1 = l + ((long)2 * NL + 1) * g.G + o;
    *p0 = accumulate * (*p0) + sl * 0.5 * (va + vb);
    *p1 = accumulate * (*p1) + sl * r3 * 0.5 * (vb - va);
  }
}
// transpose: a vertex row gathers from its (up to six) incident edges (rows j and j-1); writes the rank's rows
// joff .. joff+ny of the global vertex vector.  Strip partition (partial != 0, launched over rows 0..ny): every
// rank sums its OWNED edges only, so no halo of the residual is needed -- the vertex row on a cut receives the
// edges of the row below from the lower rank (its extra row ny: only those terms) and everything else from the
// upper rank (its row 0 without them); k_p1_assemble adds the two parts.
__global__ void k_trace_to_p1(Geo g, int NL, const double* __restrict__ l, double* __restrict__ rc, double lH, double lV,
                              double lD, int partial) {
  HDG_CORNER_PROLOGUE
  const int st = g.nx + 1;
  const double r3 = 0.57735026918962576451;
  const double* H0 = l + ((long)0 * NL) * g.G;
  const double* H1 = l + ((long)0 * NL + 1) * g.G;
  const double* V0 = l + ((long)1 * NL) * g.G;
  const double* V1 = l + ((long)1 * NL + 1) * g.G;
  const double* D0 = l + ((long)2 * NL) * g.G;
  const double* D1 = l + ((long)2 * NL + 1) * g.G;
  const double sH = 0.5 * sqrt(lH), sV = 0.5 * sqrt(lV), sD = 0.5 * sqrt(lD);
  const bool cut_lo = partial && g.joff > 0 && j == 0;                      // edges of row -1 belong to the lower rank
  const bool extra = partial && (g.joff + g.ny < g.nyg) && j == g.ny;       // lower side of the cut above: row ny-1 only
  const bool from_below = below && !cut_lo;
  double acc = 0.0;
  if (!extra) {
    if (i < g.nx) acc += sH * (H0[o] - r3 * H1[o]);                            // H(i,j): a-end
    if (i > 0) acc += sH * (H0[o - 1] + r3 * H1[o - 1]);                       // H(i-1,j): b-end
    if (in_y) acc += sV * (V0[o] - r3 * V1[o]);                                // V(i,j): a-end
  }
  if (from_below) acc += sV * (V0[o - g.P] + r3 * V1[o - g.P]);                // V(i,j-1): b-end
  if (!extra && i > 0 && in_y) acc += sD * (D0[o - 1] - r3 * D1[o - 1]);       // D(i-1,j): a-end (x_i,y_j)
  if (i < g.nx && from_below) acc += sD * (D0[o - g.P] + r3 * D1[o - g.P]);    // D(i,j-1): b-end (x_i,y_j)
  rc[(long)(g.joff + j) * st + i] = acc;
}
// Distributed finest level of the vertex-grid V-cycle (strip partition): after the restriction from the rank's own edges
// the neighbours swap `depth` vertex rows next to the cut plus their PARTIAL sums of the cut row itself.  A message holds
// depth + 1 rows of st values; from_lo = [rows J0-depth .. J0-1, partial row J0], from_hi = [partial row J1, rows J1+1 ..
// J1+depth] (J0 / J1: the rank's lowest / highest vertex row).  The partial rows are added, the others stored.
__global__ void k_p1_merge_halo(int st, int depth, int J0, int J1, int has_lo, int has_hi, const double* __restrict__ from_lo,
                                const double* __restrict__ from_hi, double* __restrict__ b) {
  const long n = (long)(depth + 1) * st;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const int row = (int)(idx / st), i = (int)(idx - (long)row * st);
    if (has_lo) {
      if (row < depth) b[(long)(J0 - depth + row) * st + i] = from_lo[idx];
      else b[(long)J0 * st + i] += from_lo[idx];
    }
    if (has_hi) {
      if (row == 0) b[(long)J1 * st + i] += from_hi[idx];
      else b[(long)(J1 + row) * st + i] = from_hi[idx];
    }
  }
}
// global vertex vector from the all-gathered blocks of (ny+1) rows per rank, in ONE launch: rank r owns the rows
// r*ny .. (r+1)*ny-1 (the last rank also the top row); on a cut the lower rank's extra row is added (partial sums)
__global__ void k_p1_assemble(int P, int ny, int st, const double* __restrict__ gathered, double* __restrict__ out,
                              int partial) {
  const long n = ((long)P * ny + 1) * st, blk = (long)(ny + 1) * st;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const int R = (int)(idx / st), i = (int)(idx - (long)R * st);
    int r = R / ny;
    if (r > P - 1) r = P - 1;
    const int lr = R - r * ny;
    double v = gathered[r * blk + (long)lr * st + i];
    if (partial && lr == 0 && r > 0) v += gathered[(r - 1) * blk + (long)ny * st + i];
    out[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------
// P1 coarse space on the doubly periodic square: n x n vertices (vertex (i,j) at j*n + i, indices wrap), operator = the
// full 5-point stencil (diagonal 4; singular: constants).  Same V-cycle as above, per-level kernels.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int pw(int a, int n) { return a < 0 ? a + n : (a >= n ? a - n : a); }
__global__ void k_p1p_rbgs(int n, double* __restrict__ x, const double* __restrict__ b, int colour) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= n || ((i + j) & 1) != colour) return;
  const double off = x[j * n + pw(i - 1, n)] + x[j * n + pw(i + 1, n)] + x[pw(j - 1, n) * n + i] + x[pw(j + 1, n) * n + i];
  x[j * n + i] = (b[j * n + i] + off) * 0.25;
}
__global__ void k_p1p_residual(int n, const double* __restrict__ x, const double* __restrict__ b, double* __restrict__ r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= n) return;
  const double off = x[j * n + pw(i - 1, n)] + x[j * n + pw(i + 1, n)] + x[pw(j - 1, n) * n + i] + x[pw(j + 1, n) * n + i];
  r[j * n + i] = b[j * n + i] - (4.0 * x[j * n + i] - off);
}
__global__ void k_p1p_restrict(int nc, const double* __restrict__ rf, double* __restrict__ rc) {
  const int I = blockIdx.x * blockDim.x + threadIdx.x, J = blockIdx.y;
  if (I >= nc) return;
  const int nf = 2 * nc, i = 2 * I, j = 2 * J;
  const int im = pw(i - 1, nf), ip = pw(i + 1, nf), jm = pw(j - 1, nf), jp = pw(j + 1, nf);
  rc[J * nc + I] = rf[j * nf + i] + 0.5 * (rf[j * nf + im] + rf[j * nf + ip] + rf[jm * nf + i] + rf[jp * nf + i] + rf[jp * nf + im] + rf[jm * nf + ip]);
}
__global__ void k_p1p_prolong_add(int nc, const double* __restrict__ xc, double* __restrict__ xf) {
  const int nf = 2 * nc;
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= nf) return;
  const int I = i >> 1, J = j >> 1, I1 = pw(I + 1, nc), J1 = pw(J + 1, nc);
  double v;
  if (!(i & 1) && !(j & 1)) v = xc[J * nc + I];
  else if ((i & 1) && !(j & 1)) v = 0.5 * (xc[J * nc + I] + xc[J * nc + I1]);
  else if (!(i & 1) && (j & 1)) v = 0.5 * (xc[J * nc + I] + xc[J1 * nc + I]);
  else v = 0.5 * (xc[J * nc + I1] + xc[J1 * nc + I]);
  xf[j * nf + i] += v;
}
// trace <-> periodic P1 grid (ghost trace rows must be current for the restriction)
__global__ void k_p1p_to_trace(Geo g, int NL, const double* __restrict__ xc, double* __restrict__ l, double accumulate,
                               double lH, double lV, double lD) {
  HDG_CORNER_PROLOGUE
  const int n = g.nx, i1 = pw(i + 1, n), j1 = pw(j + 1, n);
  const double r3 = 0.57735026918962576451;
  const double v00 = xc[j * n + i], v10 = xc[j * n + i1], v01 = xc[j1 * n + i];
  const double ea[3] = {v00, v00, v10}, eb[3] = {v10, v01, v01}, len[3] = {lH, lV, lD};
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const double sl = sqrt(len[t]);
    double* p0 = l + ((long)t * NL) * g.G + o;
    double* p1 = l + ((long)t * NL + 1) * g.G + o;
    *p0 = accumulate * (*p0) + sl * 0.5 * (ea[t] + eb[t]);
    *p1 = accumulate * (*p1) + sl * r3 * 0.5 * (eb[t] - ea[t]);
  }
}
__global__ void k_trace_to_p1p(Geo g, int NL, const double* __restrict__ l, double* __restrict__ rc, double lH, double lV, double lD) {
  HDG_CORNER_PROLOGUE
  const double r3 = 0.57735026918962576451;
  const double* H0 = l + ((long)0 * NL) * g.G;
  const double* H1 = l + ((long)0 * NL + 1) * g.G;
  const double* V0 = l + ((long)1 * NL) * g.G;
  const double* V1 = l + ((long)1 * NL + 1) * g.G;
  const double* D0 = l + ((long)2 * NL) * g.G;
  const double* D1 = l + ((long)2 * NL + 1) * g.G;
  const double sH = 0.5 * sqrt(lH), sV = 0.5 * sqrt(lV), sD = 0.5 * sqrt(lD);
  double acc = sH * (H0[o] - r3 * H1[o]) + sH * (H0[oL] + r3 * H1[oL])          // H(i,j) a-end, H(i-1,j) b-end
             + sV * (V0[o] - r3 * V1[o]) + sV * (V0[o - g.P] + r3 * V1[o - g.P])  // V(i,j) a-end, V(i,j-1) b-end (ghost row for j = 0)
             + sD * (D0[oL] - r3 * D1[oL]) + sD * (D0[o - g.P] + r3 * D1[o - g.P]);  // D(i-1,j) a-end, D(i,j-1) b-end
  rc[j * g.nx + i] = acc;
}
// periodic strip on one rank: the ghost rows are the owned rows of the opposite side (array rows: 0 ghost below,
// 1..ny owned, ny+1 ghost above)
__global__ void k_wrap_rows(double* __restrict__ v, long plane_stride, int row_len, int nplanes, int ny, int gh) {
  // array rows: gh ghost rows, ny owned rows (gh .. gh+ny-1), gh ghost rows; every ghost depth is filled
  const long n = (long)nplanes * row_len * gh;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const long pl = idx / (row_len * gh);
    const int rem = (int)(idx - pl * row_len * gh);
    const int d = rem / row_len, i = rem - d * row_len;
    double* p = v + pl * plane_stride;
    // ghost row j = d - gh below / j = ny + d above <- owned row j mod ny (ny may be smaller than gh: GH = 6 since round 4)
    const int jb = ((d - gh) % ny + ny) % ny, ja = d % ny;
    p[(long)d * row_len + i] = p[(long)(gh + jb) * row_len + i];
    p[(long)(gh + ny + d) * row_len + i] = p[(long)(gh + ja) * row_len + i];
  }
}

// ------------------------------------------------------------------------------------------
// halo rows of the strip partition: pack the lowest / highest `depth` OWNED rows of every plane into a
// contiguous buffer, unpack received rows into the ghost rows next to the strip.
//   cell vectors:  planes = ndof * 2 shapes, row length 2 nx (pairs), plane stride (ny+2GH)*2nx
//   trace vectors: planes = 3 * NL,           row length P,           plane stride G
// buf layout [plane][d][i]; rows are array row indices (gh-1 = first ghost below, gh..gh+ny-1 owned, gh+ny = first above);
// row_lo / row_hi = first row of the `depth` consecutive rows of the lower / upper message
// ------------------------------------------------------------------------------------------
// blockIdx.y = 0 / 1: lower / upper message, so one launch packs (unpacks) both halos
__global__ void k_pack_rows(const double* __restrict__ v, long plane_stride, int row_len, int nplanes, int depth, int row_lo,
                            int row_hi, double* __restrict__ buf_lo, double* __restrict__ buf_hi) {
  const int row = blockIdx.y == 0 ? row_lo : row_hi;
  double* __restrict__ buf = blockIdx.y == 0 ? buf_lo : buf_hi;
  const long chunk = (long)depth * row_len;  // consecutive rows of a plane are contiguous
  const long n = (long)nplanes * chunk;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const long pl = idx / chunk;
    buf[idx] = v[pl * plane_stride + (long)row * row_len + (idx - pl * chunk)];
  }
}
// a negative row = that neighbour does not exist
__global__ void k_unpack_rows(double* __restrict__ v, long plane_stride, int row_len, int nplanes, int depth, int row_lo,
                              int row_hi, const double* __restrict__ buf_lo, const double* __restrict__ buf_hi) {
  const int row = blockIdx.y == 0 ? row_lo : row_hi;
  if (row < 0) return;
  const double* __restrict__ buf = blockIdx.y == 0 ? buf_lo : buf_hi;
  const long chunk = (long)depth * row_len;
  const long n = (long)nplanes * chunk;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const long pl = idx / chunk;
    v[pl * plane_stride + (long)row * row_len + (idx - pl * chunk)] = buf[idx];
  }
}

// Self-check of the ghost-row bookkeeping (HDG_FLOW_CHECK): compare received neighbour rows with the ghost rows in place;
// out[0] = max |ghost - received|, out[1] = max |received| (non-negative doubles order like their bit patterns)
__global__ void k_compare_rows(const double* __restrict__ v, long plane_stride, int row_len, int nplanes, int depth, int row_lo,
                               int row_hi, const double* __restrict__ buf_lo, const double* __restrict__ buf_hi,
                               unsigned long long* __restrict__ out) {
  const int row = blockIdx.y == 0 ? row_lo : row_hi;
  if (row < 0) return;
  const double* __restrict__ buf = blockIdx.y == 0 ? buf_lo : buf_hi;
  const long chunk = (long)depth * row_len;
  const long n = (long)nplanes * chunk;
  const long stride = (long)gridDim.x * blockDim.x;
  double dmax = 0.0, amax = 0.0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const long pl = idx / chunk;
    const double a = v[pl * plane_stride + (long)row * row_len + (idx - pl * chunk)], b = buf[idx];
    const double d = fabs(a - b);
    dmax = (d == d) ? fmax(dmax, d) : 1e300;  // a NaN on either side counts as a mismatch
    amax = fmax(amax, fabs(b));
  }
  for (int off = 32; off > 0; off >>= 1) {
    dmax = fmax(dmax, __shfl_down(dmax, off, 64));
    amax = fmax(amax, __shfl_down(amax, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out, (unsigned long long)__double_as_longlong(dmax));
    atomicMax(out + 1, (unsigned long long)__double_as_longlong(amax));
  }
}

}  // namespace hdg
