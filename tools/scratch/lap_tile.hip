// Prototype (scratch, not part of the library): Laplacian-type SpMM over row tiles whose X rows are staged in LDS.
// A tile = TR consecutive output rows x CH channels; U(tile) = sorted unique columns of the tile's entries; every entry
// carries its position in U (u16).  Phase 1 streams the CH-channel slice of X[U] into LDS (one 128-byte line per row at
// CH = 32), phase 2 multiplies out of LDS.  Built: hipcc -O3 --offload-arch=gfx950 -shared -fPIC lap_tile.hip -o liblaptile.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kWG = 256;

template <int CH, int TR>
__global__ __launch_bounds__(kWG) void spmm_rt_k(const int *__restrict__ tile_uptr, const int *__restrict__ ucols,
                                                 const int *__restrict__ rowptr, const unsigned short *__restrict__ lidx,
                                                 const float *__restrict__ vals, const float *__restrict__ X,
                                                 float *__restrict__ Y, int M, int ntiles, int N) {
  constexpr int LPR = CH / 4;                 // lanes per row (float4 each)
  constexpr int RS = CH + 4;                  // LDS row stride in floats (+16 bytes: spreads the banks)
  constexpr int NSPLIT_MAX = 8;
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int nsplit = N / CH;
  // XCD-aware order: the splits of one tile run on the same XCD (block b lands on XCD b % 8)
  const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
  const int tl = slot / nsplit, split = slot - tl * nsplit;
  const int tile = tl * 8 + xcd;
  if (tile >= ntiles) return;
  (void)NSPLIT_MAX;
  const int u0 = tile_uptr[tile], nu = tile_uptr[tile + 1] - u0;
  const int c0 = split * CH;
  const int lr = threadIdx.x / LPR, lc = threadIdx.x % LPR;
  for (int u = lr; u < nu; u += kWG / LPR) {
    const int col = ucols[u0 + u];
    *reinterpret_cast<f4 *>(xs + u * RS + 4 * lc) = *reinterpret_cast<const f4 *>(X + (int64_t)col * N + c0 + 4 * lc);
  }
  __syncthreads();
  const int r0 = tile * TR;
  for (int rr = lr; rr < TR; rr += kWG / LPR) {
    const int r = r0 + rr;
    if (r >= M) break;
    const int p0 = rowptr[r], p1 = rowptr[r + 1];
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = p0; p < p1; ++p) {
      const float a = vals[p];
      const f4 x = *reinterpret_cast<const f4 *>(xs + (int)lidx[p] * RS + 4 * lc);
      acc[0] = fmaf(a, x[0], acc[0]);
      acc[1] = fmaf(a, x[1], acc[1]);
      acc[2] = fmaf(a, x[2], acc[2]);
      acc[3] = fmaf(a, x[3], acc[3]);
    }
    __builtin_nontemporal_store(acc, reinterpret_cast<f4 *>(Y + (int64_t)r * N + c0 + 4 * lc));
  }
}

extern "C" int lt_spmm(const int *tile_uptr, const int *ucols, const int *rowptr, const unsigned short *lidx, const float *vals,
                       const float *X, float *Y, int M, int ntiles, int N, int CH, int TR, int max_u, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nsplit = N / CH;
  const int nt8 = (ntiles + 7) / 8;
  const unsigned grid = (unsigned)(nt8 * nsplit * 8);
  const size_t shm = (size_t)max_u * (CH + 4) * sizeof(float);
#define LT(C_, T_)                                                                                                     \
  if (CH == C_ && TR == T_) {                                                                                          \
    if (shm > 64 * 1024) hipFuncSetAttribute((const void *)spmm_rt_k<C_, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL((spmm_rt_k<C_, T_>), dim3(grid), dim3(kWG), shm, s, tile_uptr, ucols, rowptr, lidx, vals, X, Y, M, ntiles, N); \
    return (int)hipGetLastError();                                                                                     \
  }
  LT(32, 128) LT(32, 64) LT(32, 256) LT(64, 128) LT(64, 64) LT(64, 256) LT(128, 64) LT(128, 128)
  return -1;
}
