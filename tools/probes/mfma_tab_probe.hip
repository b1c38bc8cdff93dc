// Research probe (not part of libhdg_mi355x.so): y[q, c] = sum_m T[q, m] x[m, c] for a tabulation matrix T
// (NQ x NK, k = 4: 64 quadrature points x 21 basis functions, K padded to 24) applied to coefficient planes in
// the engine's layout x[m * Nc + c] (cell fastest), consumed as s[c] = sum_q y[q, c]^2 so that nothing but the
// contraction is measured.  Two formulations:
//   scalar : one thread per cell, T through scalar loads (what k_adv_apply does today)
//   mfma   : one wave per 16 cells, T as the A operand of v_mfma_f64_16x16x4 (held in VGPRs for the whole
//            kernel), the coefficient planes streamed as the B operand
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_tab_probe mfma_tab_probe.hip ; run: ./mfma_tab_probe [ncells]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int NQ = 64, NK = 24;  // NK = 21 padded to a multiple of 4
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(128) void k_scalar(const double* __restrict__ T, const double* __restrict__ x, double* __restrict__ s,
                                                long Nc) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Nc) return;
  double xv[NK];
#pragma unroll
  for (int m = 0; m < NK; m++) xv[m] = x[(long)m * Nc + c];
  double acc = 0.0;
#pragma unroll 1
  for (int q = 0; q < NQ; q++) {
    double y = 0.0;
#pragma unroll
    for (int m = 0; m < NK; m++) y = fma(T[q * NK + m], xv[m], y);
    acc = fma(y, y, acc);
  }
  s[c] = acc;
}

// one wave: 16 cells per trip.  A tile (mt, ks): lane l holds T[16 mt + l % 16][4 ks + l / 16];
// B tile (ks): lane l holds x[4 ks + l / 16][c0 + l % 16]; D tile (mt): register r of lane l holds
// y[16 mt + (l / 16) + 4 r][c0 + l % 16]  (only the sum over rows is used here)
__global__ __launch_bounds__(64) void k_mfma(const double* __restrict__ T, const double* __restrict__ x, double* __restrict__ s,
                                             long Nc, int trips) {
  const int l = threadIdx.x, li = l & 15, lk = l >> 4;
  double A[NQ / 16][NK / 4];
#pragma unroll
  for (int mt = 0; mt < NQ / 16; mt++)
#pragma unroll
    for (int ks = 0; ks < NK / 4; ks++) A[mt][ks] = T[(16 * mt + li) * NK + 4 * ks + lk];
  for (int t = 0; t < trips; t++) {
    const long c0 = ((long)blockIdx.x * trips + t) * 16;
    if (c0 >= Nc) return;
    double B[NK / 4];
#pragma unroll
    for (int ks = 0; ks < NK / 4; ks++) B[ks] = x[(long)(4 * ks + lk) * Nc + c0 + li];
    double part = 0.0;
#pragma unroll
    for (int mt = 0; mt < NQ / 16; mt++) {
      v4d D = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < NK / 4; ks++) D = __builtin_amdgcn_mfma_f64_16x16x4f64(A[mt][ks], B[ks], D, 0, 0, 0);
      part = fma(D.x, D.x, part);
      part = fma(D.y, D.y, part);
      part = fma(D.z, D.z, part);
      part = fma(D.w, D.w, part);
    }
    // rows of a column are spread over the 4 lane groups: sum over lk
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (lk == 0) s[c0 + li] = part;
  }
}

// ---- the shape of the advection cell term at k = 4: three tables (Phi, Gx, Gy), four coefficient vectors
// (Q*_x, Q*_y, x_x, x_y), six contractions per quadrature point, pointwise product, consumed as a sum of squares
__global__ __launch_bounds__(128) void k_scalar3(const double* __restrict__ T3, const double* __restrict__ x4, double* __restrict__ s,
                                                 long Nc) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Nc) return;
  const double* __restrict__ Phi = T3;
  const double* __restrict__ Gx = T3 + NQ * NK;
  const double* __restrict__ Gy = T3 + 2 * NQ * NK;
  double q0[NK], q1[NK], x0[NK], x1[NK];
#pragma unroll
  for (int m = 0; m < NK; m++) {
    q0[m] = x4[((long)0 * NK + m) * Nc + c];
    q1[m] = x4[((long)1 * NK + m) * Nc + c];
    x0[m] = x4[((long)2 * NK + m) * Nc + c];
    x1[m] = x4[((long)3 * NK + m) * Nc + c];
  }
  double acc = 0.0;
#pragma unroll 1
  for (int q = 0; q < NQ; q++) {
    double qx = 0, qy = 0, dxx = 0, dxy = 0, dyx = 0, dyy = 0;
#pragma unroll
    for (int m = 0; m < NK; m++) {
      const double ph = Phi[q * NK + m], gx = Gx[q * NK + m], gy = Gy[q * NK + m];
      qx = fma(ph, q0[m], qx);
      qy = fma(ph, q1[m], qy);
      dxx = fma(gx, x0[m], dxx);
      dxy = fma(gy, x0[m], dxy);
      dyx = fma(gx, x1[m], dyx);
      dyy = fma(gy, x1[m], dyy);
    }
    const double ax = qx * dxx + qy * dxy, ay = qx * dyx + qy * dyy;
    acc = fma(ax, ax, fma(ay, ay, acc));
  }
  s[c] = acc;
}
__global__ __launch_bounds__(64) void k_mfma3(const double* __restrict__ T3, const double* __restrict__ x4, double* __restrict__ s,
                                              long Nc, int trips) {
  const int l = threadIdx.x, li = l & 15, lk = l >> 4;
  double A[3][NQ / 16][NK / 4];  // 3 x 4 x 6 = 72 doubles per lane: the tables live in VGPRs
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int mt = 0; mt < NQ / 16; mt++)
#pragma unroll
      for (int ks = 0; ks < NK / 4; ks++) A[t][mt][ks] = T3[t * NQ * NK + (16 * mt + li) * NK + 4 * ks + lk];
  for (int t = 0; t < trips; t++) {
    const long c0 = ((long)blockIdx.x * trips + t) * 16;
    if (c0 >= Nc) return;
    double B[4][NK / 4];
#pragma unroll
    for (int v = 0; v < 4; v++)
#pragma unroll
      for (int ks = 0; ks < NK / 4; ks++) B[v][ks] = x4[((long)v * NK + 4 * ks + lk) * Nc + c0 + li];
    double part = 0.0;
#pragma unroll
    for (int mt = 0; mt < NQ / 16; mt++) {
      v4d QX = {0, 0, 0, 0}, QY = QX, DXX = QX, DXY = QX, DYX = QX, DYY = QX;
#pragma unroll
      for (int ks = 0; ks < NK / 4; ks++) {
        QX = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0][mt][ks], B[0][ks], QX, 0, 0, 0);
        QY = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0][mt][ks], B[1][ks], QY, 0, 0, 0);
        DXX = __builtin_amdgcn_mfma_f64_16x16x4f64(A[1][mt][ks], B[2][ks], DXX, 0, 0, 0);
        DXY = __builtin_amdgcn_mfma_f64_16x16x4f64(A[2][mt][ks], B[2][ks], DXY, 0, 0, 0);
        DYX = __builtin_amdgcn_mfma_f64_16x16x4f64(A[1][mt][ks], B[3][ks], DYX, 0, 0, 0);
        DYY = __builtin_amdgcn_mfma_f64_16x16x4f64(A[2][mt][ks], B[3][ks], DYY, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const double ax = QX[r] * DXX[r] + QY[r] * DXY[r], ay = QX[r] * DYX[r] + QY[r] * DYY[r];
        part = fma(ax, ax, fma(ay, ay, part));
      }
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (lk == 0) s[c0 + li] = part;
  }
}

int main(int argc, char** argv) {
  const long Nc = argc > 1 ? atol(argv[1]) : (1L << 21);  // multiple of 16
  std::vector<double> hT(NQ * NK), hx((size_t)NK * Nc);
  unsigned long long st = 88172645463325252ULL;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st % 2000001ULL) / 1.0e6 - 1.0; };
  for (auto& v : hT) v = rnd();
  for (int q = 0; q < NQ; q++) for (int m = 21; m < NK; m++) hT[q * NK + m] = 0.0;  // padding columns
  for (auto& v : hx) v = rnd();
  double *dT, *dx, *ds1, *ds2;
  CK(hipMalloc(&dT, sizeof(double) * hT.size()));
  CK(hipMalloc(&dx, sizeof(double) * hx.size()));
  CK(hipMalloc(&ds1, sizeof(double) * Nc));
  CK(hipMalloc(&ds2, sizeof(double) * Nc));
  CK(hipMemcpy(dT, hT.data(), sizeof(double) * hT.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int trips = 8;
  const long nwaves = (Nc / 16 + trips - 1) / trips;
  auto time = [&](auto launch) {
    for (int i = 0; i < 3; i++) launch();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 20.0;
  };
  const double t_s = time([&]() { k_scalar<<<(Nc + 127) / 128, 128>>>(dT, dx, ds1, Nc); });
  const double t_m = time([&]() { k_mfma<<<nwaves, 64>>>(dT, dx, ds2, Nc, trips); });
  CK(hipDeviceSynchronize());
  std::vector<double> h1(Nc), h2(Nc);
  CK(hipMemcpy(h1.data(), ds1, sizeof(double) * Nc, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), ds2, sizeof(double) * Nc, hipMemcpyDeviceToHost));
  double err = 0.0, ref = 0.0;
  for (long c = 0; c < Nc; c++) { err = fmax(err, fabs(h1[c] - h2[c])); ref = fmax(ref, fabs(h1[c])); }
  // host check of a few cells
  double herr = 0.0;
  for (long c = 0; c < 64; c++) {
    double acc = 0.0;
    for (int q = 0; q < NQ; q++) { double y = 0.0; for (int m = 0; m < NK; m++) y += hT[q * NK + m] * hx[(size_t)m * Nc + c]; acc += y * y; }
    herr = fmax(herr, fabs(acc - h1[c]) / fabs(acc));
  }
  const double flop = 2.0 * NQ * NK * (double)Nc, bytes = 8.0 * (NK + 1) * (double)Nc;
  printf("cells %ld: scalar-operand FMA %.3f ms = %.1f TFLOP/s (%.2f TB/s) | MFMA f64 16x16x4 %.3f ms = %.1f TFLOP/s (%.2f TB/s) | "
         "max |diff| / max = %.2e, host check %.2e\n",
         Nc, t_s, flop / t_s / 1e9, bytes / t_s / 1e9, t_m, flop / t_m / 1e9, bytes / t_m / 1e9, err / ref, herr);
  // ---- three tables, four vectors
  std::vector<double> hT3(3 * NQ * NK), hx4((size_t)4 * NK * Nc);
  for (auto& v : hT3) v = rnd();
  for (int t = 0; t < 3; t++) for (int q = 0; q < NQ; q++) for (int m = 21; m < NK; m++) hT3[t * NQ * NK + q * NK + m] = 0.0;
  for (auto& v : hx4) v = rnd();
  double *dT3, *dx4;
  CK(hipMalloc(&dT3, sizeof(double) * hT3.size()));
  CK(hipMalloc(&dx4, sizeof(double) * hx4.size()));
  CK(hipMemcpy(dT3, hT3.data(), sizeof(double) * hT3.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dx4, hx4.data(), sizeof(double) * hx4.size(), hipMemcpyHostToDevice));
  const double t_s3 = time([&]() { k_scalar3<<<(Nc + 127) / 128, 128>>>(dT3, dx4, ds1, Nc); });
  const double t_m3 = time([&]() { k_mfma3<<<nwaves, 64>>>(dT3, dx4, ds2, Nc, trips); });
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h1.data(), ds1, sizeof(double) * Nc, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), ds2, sizeof(double) * Nc, hipMemcpyDeviceToHost));
  double err3 = 0.0, ref3 = 0.0;
  for (long c = 0; c < Nc; c++) { err3 = fmax(err3, fabs(h1[c] - h2[c])); ref3 = fmax(ref3, fabs(h1[c])); }
  const double flop3 = 2.0 * 6 * NQ * NK * (double)Nc, bytes3 = 8.0 * (4 * NK + 1) * (double)Nc;
  printf("cell-term shape (3 tables, 4 vectors, 6 contractions): scalar-operand FMA %.3f ms = %.1f TFLOP/s (%.2f TB/s) | "
         "MFMA %.3f ms = %.1f TFLOP/s (%.2f TB/s) | max |diff| / max = %.2e\n",
         t_s3, flop3 / t_s3 / 1e9, bytes3 / t_s3 / 1e9, t_m3, flop3 / t_m3 / 1e9, bytes3 / t_m3 / 1e9, err3 / ref3);
  return (err / ref < 1e-12 && herr < 1e-12 && err3 / ref3 < 1e-11) ? 0 : 2;
}
