// Infinity-Cache (256 MiB LLC) reuse probe: does a kernel that reads what its PREDECESSOR just wrote / read find it in
// the LLC, and how much depends on the traversal order?  producer (write or read S bytes, ascending) -> consumer (read the same
// S bytes ascending | descending); only the consumer is timed.  Between pairs the cache is flushed by streaming a 2 GiB buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void write_k(f4* __restrict__ b, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { f4 v = {1.f, 2.f, 3.f, (float)i}; if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v; }
}
template <int REV, int NT>
__global__ __launch_bounds__(256) void read_k(const f4* __restrict__ a, f4* __restrict__ sink, long n, long nblk) {
  const long blk = REV ? nblk - 1 - blockIdx.x : blockIdx.x;
  const long i = blk * 256 + threadIdx.x;
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (i < n) v = NT ? __builtin_nontemporal_load(a + i) : a[i];
  if (v.x == 123.456f) sink[0] = v;
}
// read S bytes, write S/2 bytes somewhere else (a GEMM-like 2:1 consumer), ascending or descending
template <int REV>
__global__ __launch_bounds__(256) void rw_k(const f4* __restrict__ a, f4* __restrict__ o, long n, long nblk) {
  const long blk = REV ? nblk - 1 - blockIdx.x : blockIdx.x;
  const long i = blk * 256 + threadIdx.x;       // i indexes pairs of f4
  if (2 * i + 1 < n) { f4 v = a[2 * i] + a[2 * i + 1]; __builtin_nontemporal_store(v, o + i); }
}

static void flush(f4* big, long nbig, f4* sink) {
  const long nblk = (nbig + 255) / 256;
  hipLaunchKernelGGL((read_k<0, 0>), dim3((unsigned)nblk), dim3(256), 0, 0, big, sink, nbig, nblk);
}
template <class F>
static float timed(F&& f) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); hipEventDestroy(s); hipEventDestroy(e); return ms;
}
int main() {
  const long nbig = 128L * 1024 * 1024;   // 2 GiB
  f4 *big, *buf, *out, *sink;
  hipMalloc(&big, nbig * 16); hipMalloc(&buf, 2048L << 20); hipMalloc(&out, 1024L << 20); hipMalloc(&sink, 64);
  hipMemset(big, 1, nbig * 16); hipMemset(buf, 1, 2048L << 20);
  printf("# consumer time (ms) and effective GB/s; S = bytes the producer touched and the consumer reads\n");
  printf("# %6s | %-22s | %9s %9s | %9s %9s\n", "S(MB)", "producer", "asc ms", "GB/s", "desc ms", "GB/s");
  for (long mb : {64L, 128L, 192L, 256L, 384L, 512L, 768L, 1024L, 1536L}) {
    const long n = mb * 1024 * 1024 / 16, nblk = (n + 255) / 256;
    for (int prod = 0; prod < 4; ++prod) {
      const char* pname[] = {"none (cold)", "write plain", "write nt", "read plain"};
      float t[2];
      for (int rev = 0; rev < 2; ++rev) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          flush(big, nbig, sink);
          if (prod == 1) hipLaunchKernelGGL((write_k<0>), dim3((unsigned)nblk), dim3(256), 0, 0, buf, n);
          if (prod == 2) hipLaunchKernelGGL((write_k<1>), dim3((unsigned)nblk), dim3(256), 0, 0, buf, n);
          if (prod == 3) hipLaunchKernelGGL((read_k<0, 0>), dim3((unsigned)nblk), dim3(256), 0, 0, buf, sink, n, nblk);
          hipDeviceSynchronize();
          float ms = timed([&] {
            if (rev) hipLaunchKernelGGL((read_k<1, 0>), dim3((unsigned)nblk), dim3(256), 0, 0, buf, sink, n, nblk);
            else hipLaunchKernelGGL((read_k<0, 0>), dim3((unsigned)nblk), dim3(256), 0, 0, buf, sink, n, nblk);
          });
          if (ms < best) best = ms;
        }
        t[rev] = best;
      }
      printf("  %6ld | %-22s | %9.4f %9.0f | %9.4f %9.0f\n", mb, pname[prod], t[0], mb * 1.048576 / t[0], t[1], mb * 1.048576 / t[1]);
    }
  }
  printf("# GEMM-like consumer: reads S (just written, nt), writes S/2 elsewhere (nt)\n");
  for (long mb : {256L, 512L, 1024L}) {
    const long n = mb * 1024 * 1024 / 16, nblk = (n / 2 + 255) / 256, nblkw = (n + 255) / 256;
    float t[2];
    for (int rev = 0; rev < 2; ++rev) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        flush(big, nbig, sink);
        hipLaunchKernelGGL((write_k<1>), dim3((unsigned)nblkw), dim3(256), 0, 0, buf, n);
        hipDeviceSynchronize();
        float ms = timed([&] {
          if (rev) hipLaunchKernelGGL((rw_k<1>), dim3((unsigned)nblk), dim3(256), 0, 0, buf, out, n, nblk);
          else hipLaunchKernelGGL((rw_k<0>), dim3((unsigned)nblk), dim3(256), 0, 0, buf, out, n, nblk);
        });
        if (ms < best) best = ms;
      }
      t[rev] = best;
    }
    printf("  %6ld | asc %.4f ms %.0f GB/s | desc %.4f ms %.0f GB/s  (bytes = 1.5 S)\n", mb, t[0], 1.5 * mb * 1.048576 / t[0], t[1], 1.5 * mb * 1.048576 / t[1]);
  }
  return 0;
}
