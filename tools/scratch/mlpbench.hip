// How many bytes must a PERSISTENT workgroup keep in flight to stream at the chip's rate?  One 256-thread workgroup per CU
// (grid = CUs x WPC), interleaved tiles (workgroup b takes tiles b, b + grid, ...), a tile = 256 threads x U float4 of input
// (contiguous), the loads of tile t+1 issued before tile t is reduced 2 : 1 and stored (nt) — the shape of the Linear kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, int DEPTH>
__global__ __launch_bounds__(256) void stream_k(const f4* __restrict__ a, f4* __restrict__ o, long ntiles) {
  // tile: U*256 f4 in, U*128 f4 out.  DEPTH tiles of loads in flight.
  f4 buf[DEPTH][U];
  long t = blockIdx.x;
  const long G = gridDim.x;
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) {
    const long tt = t + d * G;
#pragma unroll
    for (int u = 0; u < U; ++u) buf[d][u] = tt < ntiles ? a[(tt * U + u) * 256 + threadIdx.x] : f4{0, 0, 0, 0};
  }
  for (long base = t; base < ntiles; base += G * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const long cur = base + d * G;
      const long nxt = cur + (DEPTH - 1) * G;
      constexpr int slot_cur = 0;  // placeholder to keep indices static below
      (void)slot_cur;
      // issue loads of tile nxt into slot (d + DEPTH - 1) % DEPTH
#pragma unroll
      for (int u = 0; u < U; ++u) buf[(d + DEPTH - 1) % DEPTH][u] = nxt < ntiles ? a[(nxt * U + u) * 256 + threadIdx.x] : f4{0, 0, 0, 0};
      if (cur < ntiles) {
#pragma unroll
        for (int u = 0; u < U; u += 2) __builtin_nontemporal_store(buf[d][u] + buf[d][u + 1], o + (cur * (U / 2) + u / 2) * 256 + threadIdx.x);
      }
    }
  }
}
template <int U, int DEPTH>
void run(const f4* a, f4* o, long nf4, int wpc) {
  const long ntiles = nf4 / (256L * U);
  const int grid = 256 * wpc;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_k<U, DEPTH>), dim3(grid), dim3(256), 0, 0, a, o, ntiles);
  hipEventRecord(s);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stream_k<U, DEPTH>), dim3(grid), dim3(256), 0, 0, a, o, ntiles);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= 10;
  printf("U=%2d depth=%d wg/CU=%d  in flight per CU ~ %4d KB  %.3f ms  %.0f GB/s\n", U, DEPTH, wpc, U * 4 * (DEPTH - 1) * wpc, ms, 1.5 * nf4 * 16 / ms / 1e6);
}
int main() {
  const long nf4 = 64L * 1024 * 1024;   // 1 GiB in, 512 MiB out
  f4 *a, *o; hipMalloc(&a, nf4 * 16); hipMalloc(&o, nf4 * 8); hipMemset(a, 1, nf4 * 16);
  for (int wpc : {1, 2, 4}) {
    run<2, 2>(a, o, nf4, wpc); run<4, 2>(a, o, nf4, wpc); run<8, 2>(a, o, nf4, wpc); run<16, 2>(a, o, nf4, wpc);
    run<8, 3>(a, o, nf4, wpc); run<16, 3>(a, o, nf4, wpc); run<8, 4>(a, o, nf4, wpc);
  }
  return 0;
}
