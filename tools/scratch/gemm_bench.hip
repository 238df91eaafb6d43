// Standalone timing harness for sn_gemm.hip (no torch): hipcc --offload-arch=gfx950 -O3 -I../../include gemm_bench.hip -o gemm_bench
#include "../../surfacenetworks_amd/csrc/sn_gemm.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 627200;
  const int reps = 20;
  float *x, *W, *b, *y, *e, *res, *dy, *dx, *mu, *B, *C;
  hipMalloc(&x, rows * 256 * 4); hipMalloc(&y, rows * 128 * 4); hipMalloc(&e, rows * 256 * 4); hipMalloc(&res, rows * 128 * 4);
  hipMalloc(&dy, rows * 128 * 4); hipMalloc(&dx, rows * 256 * 4);
  hipMalloc(&W, 128 * 256 * 4); hipMalloc(&b, 1024); hipMalloc(&mu, 1024); hipMalloc(&B, 1024); hipMalloc(&C, 1024);
  std::vector<float> h(rows * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(x, h.data(), rows * 256 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dy, h.data(), rows * 128 * 4, hipMemcpyHostToDevice);
  hipMemcpy(res, h.data(), rows * 128 * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), 128 * 256 * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, h.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(mu, h.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(C, h.data(), 1024, hipMemcpyHostToDevice);
  hipEvent_t s, t; hipEventCreate(&s); hipEventCreate(&t);
  auto time = [&](const char *name, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(s);
    for (int i = 0; i < reps; ++i) fn();
    hipEventRecord(t); hipEventSynchronize(t);
    float ms; hipEventElapsedTime(&ms, s, t);
    printf("%-14s %8.1f us\n", name, ms / reps * 1e3);
  };
  time("fwd", [&] { sn_linear_fwd_f32(x, 256, W, 256, b, nullptr, 0, y, 128, nullptr, 0, rows, 256, 128, nullptr, nullptr); });
  time("fwd elu-only", [&] { sn_linear_fwd_f32(x, 256, W, 256, b, nullptr, 0, nullptr, 128, e, 256, rows, 256, 128, nullptr, nullptr); });
  time("fwd+res+elu", [&] { sn_linear_fwd_f32(x, 256, W, 256, b, res, 128, y, 128, e, 256, rows, 256, 128, nullptr, nullptr); });
  time("dgrad+affine", [&] { sn_linear_dgrad_f32(dy, 128, W, 256, x, 256, mu, B, C, dx, 256, rows, 128, 256, nullptr); });
  return 0;
}
