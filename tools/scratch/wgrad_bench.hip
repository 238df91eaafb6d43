// Standalone timing harness for the weight-gradient kernels of sn_dense.hip (no torch).
#include "../../surfacenetworks_amd/csrc/sn_dense.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 627200;
  const int reps = 10;
  float *x, *dy, *G, *ws, *mu;
  double *ds;
  hipMalloc(&x, rows * 256 * 4); hipMalloc(&dy, rows * 128 * 4); hipMalloc(&G, 128 * 256 * 4); hipMalloc(&mu, 1024);
  hipMalloc(&ds, 128 * 8);
  const size_t wsb = sn_wgrad_workspace_bytes(rows, 128, 256);
  hipMalloc(&ws, wsb);
  std::vector<float> h(rows * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(x, h.data(), rows * 256 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dy, h.data(), rows * 128 * 4, hipMemcpyHostToDevice);
  hipMemcpy(mu, h.data(), 1024, hipMemcpyHostToDevice);
  hipEvent_t s, t; hipEventCreate(&s); hipEventCreate(&t);
  auto fn = [&] { sn_wgrad_f32(dy, 128, x, 256, mu, rows, 128, 256, G, ds, ws, wsb, nullptr); };
  for (int i = 0; i < 3; ++i) fn();
  hipEventRecord(s);
  for (int i = 0; i < reps; ++i) fn();
  hipEventRecord(t); hipEventSynchronize(t);
  float ms; hipEventElapsedTime(&ms, s, t);
  printf("wgrad(+reduce) %8.1f us\n", ms / reps * 1e3);
  return 0;
}
