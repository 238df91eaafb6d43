// Read-only streaming of a (rows x 384)-float operand pair the way the weight gradient reads it: 256 persistent workgroups
// of 8 waves, each wave walking its share of a row slab in 32-row blocks; dword loads (256 B per wave instruction, what
// wgrad_u_k issues) against dwordx4 loads (1 KiB per instruction), same bytes in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: per wave and block 24 dword loads (3 slots x 8 rows, 64 consecutive columns each)
// MODE 1: per wave and block 6 dwordx4 loads (256 consecutive columns of a row each)
template <int MODE, int DEPTH, int NW = 8>
__global__ __launch_bounds__(64 * NW, 1) void read_k(const float* __restrict__ x, long rows, int C, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long per = ((rows + gridDim.x - 1) / gridDim.x + 31) / 32 * 32;
  const long r0 = (long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  float acc = 0.f;
  // a block = 32 rows x C floats = 32*C/64 dword-instructions or 32*C/256 x4-instructions, dealt round-robin to the 8 waves
  if (MODE == 0) {
    const int per_row = C / 64;                       // dword instructions per row
    const int n_inst = 32 * per_row / NW;              // per wave and block (24 at C = 384)
    for (long b = r0; b < r1; b += 32 * DEPTH) {
      float v[DEPTH][24];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 24; ++i) {
          const int id = wave * n_inst + i, r = id / per_row, cseg = id % per_row;
          const long row = b + 32 * d + r;
          v[d][i] = (i < n_inst && row < r1) ? x[row * C + cseg * 64 + lane] : 0.f;
        }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 24; ++i) acc += v[d][i];
    }
  } else {
    const int per_row = C / 256;                      // (C = 256 or 512 only for this mode; 384 = 256 + 128: see main)
    const int n_inst = 32 * per_row / NW;
    for (long b = r0; b < r1; b += 32 * DEPTH) {
      f4 v[DEPTH][64 / NW];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 64 / NW; ++i) {
          const int id = wave * n_inst + i, r = id / per_row, cseg = id % per_row;
          const long row = b + 32 * d + r;
          v[d][i] = (i < n_inst && row < r1) ? *reinterpret_cast<const f4*>(x + row * C + cseg * 256 + lane * 4) : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 64 / NW; ++i) acc += v[d][i].x + v[d][i].y + v[d][i].z + v[d][i].w;
    }
  }
  if (MODE == 2 || MODE == 3) {                       // MODE 2: dwordx2 (512 B / instruction); MODE 3: dword with NW waves
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int BPI = MODE == 2 ? 512 : 256;        // bytes per instruction
    const int per_row = C * 4 / BPI;
    const int n_inst = 32 * per_row / NW;             // per wave and block
    constexpr int MAXI = (32 * 256 * 4 / BPI) / NW;   // (C = 256)
    for (long b = r0; b < r1; b += 32 * DEPTH) {
      float v[DEPTH][MAXI];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
          const int id = wave * n_inst + i, r = id / per_row, cseg = id % per_row;
          const long row = b + 32 * d + r;
          float t = 0.f;
          if (i < n_inst && row < r1) {
            if (MODE == 2) { const f2 q = *reinterpret_cast<const f2*>(x + row * C + cseg * 128 + lane * 2); t = q.x + q.y; }
            else t = x[row * C + cseg * 64 + lane];
          }
          v[d][i] = t;
        }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < MAXI; ++i) acc += v[d][i];
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int DEPTH, int NW = 8>
void run(const char* tag, const float* x, long rows, int C, float* out) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((read_k<MODE, DEPTH, NW>), dim3(256), dim3(64 * NW), 0, 0, x, rows, C, out);
  hipEventRecord(s);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((read_k<MODE, DEPTH, NW>), dim3(256), dim3(64 * NW), 0, 0, x, rows, C, out);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= 20;
  printf("%-44s rows=%ld C=%d  %.4f ms  %.0f GB/s\n", tag, rows, C, ms, rows * (double)C * 4 / ms / 1e6);
}

int main() {
  const long rows = 627200;
  float *x, *out;
  hipMalloc(&x, rows * 512 * 4); hipMalloc(&out, 64);
  hipMemset(x, 0, rows * 512 * 4);
  run<0, 1>("dword loads, 1 block in flight", x, rows, 384, out);
  run<0, 2>("dword loads, 2 blocks in flight", x, rows, 384, out);
  run<1, 1>("dwordx4 loads, 1 block in flight (C=256)", x, rows, 256, out);
  run<1, 2>("dwordx4 loads, 2 blocks in flight (C=256)", x, rows, 256, out);
  run<1, 1>("dwordx4 loads, 1 block in flight (C=512)", x, rows, 512, out);
  run<1, 2>("dwordx4 loads, 2 blocks in flight (C=512)", x, rows, 512, out);
  run<0, 1>("dword loads, 1 block in flight (C=256)", x, rows, 256, out);
  run<0, 2>("dword loads, 2 blocks in flight (C=256)", x, rows, 256, out);
  run<1, 3>("dwordx4 loads, 3 blocks in flight (C=256)", x, rows, 256, out);
  run<1, 4>("dwordx4 loads, 4 blocks in flight (C=256)", x, rows, 256, out);
  // 4 waves per workgroup (the forward GEMM's shape): 8 x4 loads per wave and block
  run<1, 1, 4>("4 waves: dwordx4, 1 block in flight (C=256)", x, rows, 256, out);
  run<1, 2, 4>("4 waves: dwordx4, 2 blocks in flight (C=256)", x, rows, 256, out);
  run<1, 3, 4>("4 waves: dwordx4, 3 blocks in flight (C=256)", x, rows, 256, out);
  run<1, 4, 4>("4 waves: dwordx4, 4 blocks in flight (C=256)", x, rows, 256, out);
  run<2, 1, 4>("4 waves: dwordx2, 1 block in flight (C=256)", x, rows, 256, out);
  run<2, 2, 4>("4 waves: dwordx2, 2 blocks in flight (C=256)", x, rows, 256, out);
  run<3, 1, 4>("4 waves: dword, 1 block in flight (C=256)", x, rows, 256, out);
  run<3, 2, 4>("4 waves: dword, 2 blocks in flight (C=256)", x, rows, 256, out);
  run<2, 2, 8>("8 waves: dwordx2, 2 blocks in flight (C=256)", x, rows, 256, out);
  return 0;
}
