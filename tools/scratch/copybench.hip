// Streaming-copy probe: what 1:1 read:write rate can this box sustain, and with which access pattern?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NTL, int NTS, int U>
__global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ a, f4* __restrict__ b, long n) {
  long i = (long)blockIdx.x * 256 * U + threadIdx.x;
  const long stride = (long)gridDim.x * 256 * U;
  for (; i < n; i += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { long j = i + u * 256; if (j < n) v[u] = NTL ? __builtin_nontemporal_load(a + j) : a[j]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { long j = i + u * 256; if (j < n) { if (NTS) __builtin_nontemporal_store(v[u], b + j); else b[j] = v[u]; } }
  }
}
// R read streams, W write streams of n f4 each (R + W buffers), grid-stride: the read : write mix of the products
template <int R, int W>
__global__ __launch_bounds__(256) void mix_k(const f4* __restrict__ a, f4* __restrict__ b, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i < n; i += stride) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; ++r) v += a[i + r * n];
#pragma unroll
    for (int w = 0; w < W; ++w) __builtin_nontemporal_store(v, b + i + w * n);
    if (W == 0) acc += v;
  }
  if (W == 0 && acc.x == 123.456f) b[0] = acc;
}
template <int R, int W>
void run_mix(f4* a, f4* b, long n, int blocks) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix_k<R, W>), dim3(blocks), dim3(256), 0, 0, a, b, n);
  hipEventRecord(s);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((mix_k<R, W>), dim3(blocks), dim3(256), 0, 0, a, b, n);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= 10;
  printf("reads : writes = %d : %d  (%4.0f MB each)  blocks=%6d  %.3f ms  %.0f GB/s\n", R, W, n * 16 / 1e6, blocks, ms, (double)(R + W) * n * 16 / ms / 1e6);
}
template <int NTL, int NTS, int U>
void run(const char* name, f4* a, f4* b, long n, int blocks) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((copy_k<NTL, NTS, U>), dim3(blocks), dim3(256), 0, 0, a, b, n);
  hipEventRecord(s);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((copy_k<NTL, NTS, U>), dim3(blocks), dim3(256), 0, 0, a, b, n);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= 20;
  printf("%-28s blocks=%7d  %.3f ms  %.0f GB/s\n", name, blocks, ms, 2.0 * n * 16 / ms / 1e6);
}
int main() {
  const long n = 64L * 1024 * 1024;  // f4 elements: 1 GiB per buffer
  f4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16);
  for (int blocks : {2048, 4096, 8192, 65536, (int)(n / 256)}) {
    run<0, 0, 1>("plain/plain U1", a, b, n, blocks);
    run<0, 1, 1>("plain/nt U1", a, b, n, blocks);
    run<1, 1, 1>("nt/nt U1", a, b, n, blocks);
    run<0, 0, 4>("plain/plain U4", a, b, n, blocks / 4 > 0 ? blocks / 4 : 1);
    run<0, 1, 4>("plain/nt U4", a, b, n, blocks / 4 > 0 ? blocks / 4 : 1);
    run<1, 1, 4>("nt/nt U4", a, b, n, blocks / 4 > 0 ? blocks / 4 : 1);
  }
  {
    const long m = 16L * 1024 * 1024;      // 256 MB per stream, up to 4 streams per side of the 1 GiB buffers
    for (int blocks : {4096, 16384}) {
      run_mix<1, 0>(a, b, m, blocks); run_mix<2, 0>(a, b, m, blocks); run_mix<3, 1>(a, b, m, blocks); run_mix<2, 1>(a, b, m, blocks);
      run_mix<1, 1>(a, b, m, blocks); run_mix<1, 2>(a, b, m, blocks); run_mix<0, 1>(a, b, m, blocks);
    }
  }
  hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0); hipDeviceSynchronize();
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e); hipEventRecord(s);
  for (int i = 0; i < 20; ++i) hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0);
  hipEventRecord(e); hipEventSynchronize(e); float ms; hipEventElapsedTime(&ms, s, e);
  printf("hipMemcpyAsync D2D  %.3f ms  %.0f GB/s\n", ms / 20, 2.0 * n * 16 / (ms / 20) / 1e6);
  return 0;
}
