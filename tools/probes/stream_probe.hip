// Streaming-rate probe for MI355X: what a 16-byte-per-lane kernel with R read streams and 1 write stream delivers,
// by grid shape and cache policy.  Build: hipcc -O3 --offload-arch=gfx950 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int R, bool NT, int UNR>
__global__ void k_stream(long n2, const d2* __restrict__ a, const d2* __restrict__ b, const d2* __restrict__ c, d2* __restrict__ out) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride * UNR) {
    d2 va[UNR], vb[UNR], vc[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const long j = i + u * stride < n2 ? i + u * stride : i;
      va[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
      if (R > 1) vb[u] = NT ? __builtin_nontemporal_load(b + j) : b[j];
      if (R > 2) vc[u] = NT ? __builtin_nontemporal_load(c + j) : c[j];
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      const long j = i + u * stride;
      if (j < n2) {
        d2 r = va[u];
        if (R > 1) r = r * 0.5 + vb[u];
        if (R > 2) r = r + 0.25 * vc[u];
        if (NT) __builtin_nontemporal_store(r, out + j); else out[j] = r;
      }
    }
  }
}
template <int R, bool NT, int UNR>
double run(long n2, int blocks, int threads, d2* a, d2* b, d2* c, d2* o) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) k_stream<R, NT, UNR><<<blocks, threads>>>(n2, a, b, c, o);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; i++) k_stream<R, NT, UNR><<<blocks, threads>>>(n2, a, b, c, o);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)(R + 1) * n2 * 16.0 / (ms / reps * 1e-3) / 1e12;
}
int main() {
  const long n2 = 41943040L / 2;  // one C3 velocity vector: 335 MB
  d2 *a, *b, *c, *o;
  hipMalloc(&a, n2 * 16); hipMalloc(&b, n2 * 16); hipMalloc(&c, n2 * 16); hipMalloc(&o, n2 * 16);
  hipMemset(a, 0, n2 * 16); hipMemset(b, 0, n2 * 16); hipMemset(c, 0, n2 * 16);
  for (int blocks : {2048, 4096, 8192, 16384, 81920}) for (int threads : {256, 512}) {
    printf("blocks %6d threads %3d | copy %.2f nt %.2f | 2r1w %.2f nt %.2f unr2 %.2f | 3r1w %.2f nt %.2f unr2 %.2f nt-unr2 %.2f TB/s\n", blocks, threads,
           run<1, false, 1>(n2, blocks, threads, a, b, c, o), run<1, true, 1>(n2, blocks, threads, a, b, c, o),
           run<2, false, 1>(n2, blocks, threads, a, b, c, o), run<2, true, 1>(n2, blocks, threads, a, b, c, o), run<2, false, 2>(n2, blocks, threads, a, b, c, o),
           run<3, false, 1>(n2, blocks, threads, a, b, c, o), run<3, true, 1>(n2, blocks, threads, a, b, c, o), run<3, false, 2>(n2, blocks, threads, a, b, c, o),
           run<3, true, 2>(n2, blocks, threads, a, b, c, o));
    fflush(stdout);
  }
  return 0;
}
