// What sets the idle time between two dependent kernels of one stream on this part?  Sequences of streaming kernels with
// different properties (grid size, dirty lines left in L2, LDS footprint, store kind), run under rocprofv3 --kernel-trace;
// tools/scratch/gapbench/gaps.py turns the trace into end->start gaps per (previous, next) kernel pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); exit(1); } } while (0)

// copy n f4: grid-stride
__global__ __launch_bounds__(256) void k_copy(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_copy_nt(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
// read only (a store that never happens keeps the loads alive)
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n) {
  f4 s = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s.x == 1.2345e33f) b[0] = s;
}
// write only
__global__ __launch_bounds__(256) void k_write(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n) {
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = v;
}
// copy with a large LDS footprint (one workgroup per CU)
__global__ __launch_bounds__(256) void k_copy_lds(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n) {
  __shared__ f4 sm[8192];     // 128 KB
  sm[threadIdx.x] = f4{0, 0, 0, 0};
  __syncthreads();
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i] + sm[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_copy2(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read2(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n) {
  f4 s = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s.x == 1.2345e33f) b[0] = s;
}
__global__ __launch_bounds__(256) void k_read_lds(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n) {
  __shared__ f4 sm[4096];     // 64 KB
  sm[threadIdx.x] = f4{0, 0, 0, 0};
  __syncthreads();
  f4 s = sm[threadIdx.x ^ 1];
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s.x == 1.2345e33f) b[0] = s;
}
__global__ void k_tiny(float *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ void k_tiny2(float *p) { if (threadIdx.x == 0) p[blockIdx.x] += 2.f; }

int main() {
  const size_t bytes = 256ull << 20, n = bytes / 16;
  f4 *a, *b;
  float *t;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&t, 4096));
  CK(hipMemset(a, 0, bytes));
  CK(hipMemset(b, 0, bytes));
  CK(hipMemset(t, 0, 4096));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const int R = 12;
  auto sep = [&]() { CK(hipStreamSynchronize(s)); };
  // A: tiny x R
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, t);
  sep();
  // B: copy, large grid
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_copy, dim3(65536), dim3(256), 0, s, a, b, n);
  sep();
  // C: copy, persistent grid (1024 workgroups)
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n);
  sep();
  // D: read only, persistent
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
  sep();
  // E: write only, persistent
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, s, a, b, n);
  sep();
  // F: nontemporal copy, persistent
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_copy_nt, dim3(1024), dim3(256), 0, s, a, b, n);
  sep();
  // G: LDS-heavy copy, one workgroup per CU
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_copy_lds, dim3(256), dim3(256), 0, s, a, b, n);
  sep();
  // H: copy / tiny alternating
  for (int i = 0; i < R; ++i) {
    hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, t);
  }
  sep();
  // I: read / tiny alternating
  for (int i = 0; i < R; ++i) {
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_tiny2, dim3(1), dim3(64), 0, s, t);
  }
  sep();
  // J: small copies (4 MB), persistent grid: short big-ish kernels
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, (size_t)(4 << 20) / 16);
  sep();
  // K: the same in a graph
  {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, s, a, b, n);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    sep();
  }
  // L: alternating different kernels on the created stream
  for (int i = 0; i < R; ++i) {
    hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_copy_lds, dim3(256), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, s, a, b, n);
  }
  sep();
  // M: the same on the NULL stream
  for (int i = 0; i < R; ++i) {
    hipLaunchKernelGGL(k_copy, dim3(1000), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_read, dim3(2000), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_copy_lds, dim3(250), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(k_write, dim3(4000), dim3(256), 0, 0, a, b, n);
  }
  CK(hipDeviceSynchronize());
  // N: NULL stream, same kernel
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_read, dim3(3000), dim3(256), 0, 0, a, b, n);
  CK(hipDeviceSynchronize());
  // wall-clock per iteration of B..G without the profiler's help
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto wall = [&](const char *name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 50; ++i) launch();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.2f us per launch\n", name, ms * 1000 / 50);
  };
  wall("copy grid 65536", [&]() { hipLaunchKernelGGL(k_copy, dim3(65536), dim3(256), 0, s, a, b, n); });
  wall("copy grid 1024", [&]() { hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n); });
  wall("read grid 1024", [&]() { hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n); });
  wall("write grid 1024", [&]() { hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, s, a, b, n); });
  wall("copy nt grid 1024", [&]() { hipLaunchKernelGGL(k_copy_nt, dim3(1024), dim3(256), 0, s, a, b, n); });
  wall("copy lds grid 256", [&]() { hipLaunchKernelGGL(k_copy_lds, dim3(256), dim3(256), 0, s, a, b, n); });
  wall("mix of 4, stream", [&]() {
    hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_copy_lds, dim3(256), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, s, a, b, n);
  });
  wall("read x2 same symbol", [&]() {
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
  });
  wall("read x2 same symbol, grids 1024/2048", [&]() {
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, s, a, b, n);
  });
  wall("read + read2 (twin symbol)", [&]() {
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read2, dim3(1024), dim3(256), 0, s, a, b, n);
  });
  wall("read + read_lds (64 KB LDS)", [&]() {
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read_lds, dim3(1024), dim3(256), 0, s, a, b, n);
  });
  wall("read_lds x2", [&]() {
    hipLaunchKernelGGL(k_read_lds, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read_lds, dim3(1024), dim3(256), 0, s, a, b, n);
  });
  wall("copy + copy2 (twin symbol)", [&]() {
    hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_copy2, dim3(1024), dim3(256), 0, s, a, b, n);
  });
  wall("copy x2", [&]() {
    hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n);
  });
  wall("read + tiny", [&]() {
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, t);
  });
  wall("read + tiny + tiny2", [&]() {
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, t);
    hipLaunchKernelGGL(k_tiny2, dim3(1), dim3(64), 0, s, t);
  });
  wall("read + tiny + tiny", [&]() {
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, t);
    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, t);
  });
  wall("write(b) + read(b)  RAW", [&]() {
    hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, b, a, n);
  });
  wall("write(b) + read(a)  indep", [&]() {
    hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
  });
  wall("copy(a->b) + read(b) RAW", [&]() {
    hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, b, a, n);
  });
  wall("copy(a->b) + read(a)", [&]() {
    hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s, a, b, n);
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, a, b, n);
  });
  wall("write 64MB + read same 64MB (RAW, fits MALL)", [&]() {
    hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, s, a, b, n / 4);
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, b, a, n / 4);
  });
  wall("write 64MB alone", [&]() { hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, s, a, b, n / 4); });
  wall("read 64MB alone", [&]() { hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, s, b, a, n / 4); });
  wall("tiny", [&]() { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, t); });
  return 0;
}
