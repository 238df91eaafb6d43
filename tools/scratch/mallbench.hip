// Infinity-Cache (MALL) probe: how fast is a read of a buffer that the PREVIOUS kernel wrote, as a function of its size,
// of the store kind (plain | nontemporal) and of what else that kernel streamed through the cache?
//   producer: reads src (size S, plain or nt loads) and writes dst (size S, plain or nt stores)   [a GEMM-like pass]
//   consumer: reads dst (plain or nt loads) and reduces                                          [a statistics pass]
// hipcc --offload-arch=gfx950 -O3 mallbench.hip -o mallbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NTL, int NTS>
__global__ __launch_bounds__(256) void prod_k(const f4* __restrict__ a, f4* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    f4 v = NTL ? __builtin_nontemporal_load(a + i) : a[i];
    v = v * 1.0001f;
    if (NTS) __builtin_nontemporal_store(v, b + i); else b[i] = v;
  }
}
template <int NTL>
__global__ __launch_bounds__(256) void cons_k(const f4* __restrict__ b, long n, float* out) {
  f4 acc = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const f4 v = NTL ? __builtin_nontemporal_load(b + i) : b[i];
    acc += v;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
template <int PL, int PS, int CL>
void run(const char* name, f4* a, f4* b, f4* flush, long nflush, long n, float* out) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  const int blocks = 8192;
  float tot = 0, totp = 0;
  const int reps = 10;
  for (int r = 0; r < reps + 2; ++r) {
    hipLaunchKernelGGL((cons_k<1>), dim3(blocks), dim3(256), 0, 0, flush, nflush, out);       // flush the cache
    hipEventRecord(s);
    hipLaunchKernelGGL((prod_k<PL, PS>), dim3(blocks), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e);
    hipEvent_t s2, e2; hipEventCreate(&s2); hipEventCreate(&e2);
    hipEventRecord(s2);
    hipLaunchKernelGGL((cons_k<CL>), dim3(blocks), dim3(256), 0, 0, b, n, out);
    hipEventRecord(e2); hipEventSynchronize(e2);
    float ms, msp; hipEventElapsedTime(&ms, s2, e2); hipEventElapsedTime(&msp, s, e);
    if (r >= 2) { tot += ms; totp += msp; }
    hipEventDestroy(s2); hipEventDestroy(e2);
  }
  printf("%-34s %4ld MB  producer %7.1f us (%5.0f GB/s)  consumer %7.1f us  %6.0f GB/s\n", name, n * 16 >> 20,
         totp / reps * 1e3, 2.0 * n * 16 / (totp / reps) / 1e6, tot / reps * 1e3, (double)n * 16 / (tot / reps) / 1e6);
}
int main() {
  const long nmax = 48L * 1024 * 1024;       // 768 MB
  f4 *a, *b, *fl; float* out;
  hipMalloc(&a, nmax * 16); hipMalloc(&b, nmax * 16); hipMalloc(&fl, 64L * 1024 * 1024 * 16); hipMalloc(&out, 4);
  hipMemset(a, 0, nmax * 16); hipMemset(b, 0, nmax * 16); hipMemset(fl, 0, 64L * 1024 * 1024 * 16);
  const long nfl = 64L * 1024 * 1024;
  for (long mb : {40L, 80L, 120L, 165L, 200L, 321L, 642L}) {
    const long n = mb * 1024 * 1024 / 16;
    run<0, 0, 0>("ld plain, st plain | rd plain", a, b, fl, nfl, n, out);
    run<1, 0, 0>("ld nt,    st plain | rd plain", a, b, fl, nfl, n, out);
    run<1, 0, 1>("ld nt,    st plain | rd nt", a, b, fl, nfl, n, out);
    run<0, 1, 0>("ld plain, st nt    | rd plain", a, b, fl, nfl, n, out);
    run<1, 1, 1>("ld nt,    st nt    | rd nt", a, b, fl, nfl, n, out);
    printf("\n");
  }
  return 0;
}
