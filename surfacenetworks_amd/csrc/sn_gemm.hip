// sn_gemm.hip — row-streaming fp32 MFMA GEMMs for the per-node Linear layers (gfx950).
//
// Shape class: Out(rows x Nout) = In(rows x K) · Wmat, rows ~ 1e5..1e6, K and Nout in {128, 256}.  The weights are
// tiny (128 KB) and the operands are streamed once, so the kernel is "weights stationary IN REGISTERS":
//
//   v_mfma_f32_32x32x2_f32 computes D[i][n] += A[i][k]·B[k][n], k = 0,1, with lane l supplying A[i = l&31][k = l>>5] and
//   B[k = l>>5][n = l&31].  Take i = output column, n = data row:  a wave owns NT tiles of 32 output columns and keeps
//   the K/2 A-operands of each tile in VGPRs for the whole kernel (128 registers); it then streams 32-row tiles of the
//   input: lane l reads 16 bytes of row (r0 + l&31) per load — k = 8t + 4(l>>5) .. +3 — so the four registers of one load
//   feed four MFMA steps, and the A registers were loaded with the matching k.  The four waves of a workgroup cover
//   Nout = 128·NT columns and read the same input tile (L1 hits).  The whole next tile is prefetched into a second
//   register set while the current one is multiplied (K/2·NT MFMAs = 8192 cycles per tile ≫ HBM latency), so one wave
//   per SIMD keeps the matrix pipe busy: no LDS, no barriers.
//   D layout: lane (n, h = l>>5) holds, for its data row n, the 16 outputs at columns 8g + 4h + q (g, q = 0..3):
//   four 16-byte pieces per tile — the epilogue (bias / residual / ELU copy, or the BatchNorm tail) works on them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sn_spmm.h"

namespace {

constexpr int kWG = 256;
constexpr int kCUs = 256;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SN_OK : (int)e;
}
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// ELU in the GEMM epilogue runs while the matrix pipe of this SIMD idles (one wave per SIMD), so it uses the hardware
// exponential (v_exp_f32) instead of the ~30-instruction expm1f: |error| <= 1.2e-7 absolute, far below the 1e-5 parity bar.
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

enum { EPI_FWD = 0, EPI_DGRAD = 1 };

struct EpiArgs {
  // forward: bias[Nout], residual (rows x Nout, ldr) | NULL, y_elu (rows x Nout, lde) | NULL
  // dgrad  : x (rows x Nout, ldx2), center / B / Cc [Nout] (B == NULL: plain product)
  const float *v0, *v1, *v2, *v3;
  float *o2;
  int64_t ld1, ld2;
};

template <int K, int NT, bool TRANSW, int EPI>
__global__ __launch_bounds__(kWG, 1) void gemm_rows_k(const float *__restrict__ In, int64_t ldi,
                                                      const float *__restrict__ W, int64_t ldw,
                                                      float *__restrict__ Out, int64_t ldo, int64_t rows, EpiArgs ep) {
  constexpr int STEPS = K / 2;       // MFMA k-steps per output tile
  constexpr int LOADS = K / 8;       // 16-byte input loads per lane per row tile
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 31, h = lane >> 5;
  // ---- stationary A operands: a[t][4*t8 + s] = Wmat[col = 32*(wave*NT + t) + n][k = 8*t8 + 4*h + s] ----
  float a[NT][STEPS];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = 32 * (wave * NT + t) + n;
#pragma unroll
    for (int t8 = 0; t8 < LOADS; ++t8) {
      if constexpr (!TRANSW) {
        const f4 w4 = *reinterpret_cast<const f4 *>(W + (int64_t)col * ldw + 8 * t8 + 4 * h);
        a[t][4 * t8 + 0] = w4.x; a[t][4 * t8 + 1] = w4.y; a[t][4 * t8 + 2] = w4.z; a[t][4 * t8 + 3] = w4.w;
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) a[t][4 * t8 + s] = W[(int64_t)(8 * t8 + 4 * h + s) * ldw + col];
      }
    }
  }
  const int64_t ntiles = (rows + 31) / 32;
  const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
  int64_t tile = (int64_t)blockIdx.x * per;
  const int64_t tend = tile + per < ntiles ? tile + per : ntiles;
  if (tile >= tend) return;
  // ---- per-column epilogue constants, hoisted into registers once (indexed [t][g]: columns 32*(wave*NT+t) + 8g + 4h..) ----
  f4 e0[NT][4], e1[NT][4], e2[NT][4];
  const bool affine = (EPI == EPI_DGRAD) && ep.v2 != nullptr;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = 32 * (wave * NT + t) + 4 * h + 8 * g;
      const f4 z = {0.f, 0.f, 0.f, 0.f};
      if constexpr (EPI == EPI_FWD) {
        e0[t][g] = *reinterpret_cast<const f4 *>(ep.v0 + c);          // bias
        e1[t][g] = z;
        e2[t][g] = z;
      } else {
        e0[t][g] = (affine && ep.v1) ? *reinterpret_cast<const f4 *>(ep.v1 + c) : z;   // center
        e1[t][g] = affine ? *reinterpret_cast<const f4 *>(ep.v2 + c) : z;              // B
        e2[t][g] = affine ? *reinterpret_cast<const f4 *>(ep.v3 + c) : z;              // Cc
      }
    }
  const bool side = (EPI == EPI_FWD) ? (ep.v1 != nullptr) : affine;   // a per-element side operand (residual | x) exists
  const float *side_p = (EPI == EPI_FWD) ? ep.v1 : ep.v0;

  // two statically named register sets (a runtime-indexed array would be demoted to scratch)
  f4 b0[LOADS], b1[LOADS];
  auto row_ptr = [&](int64_t tl) {
    int64_t r = tl * 32 + n;
    r = r < rows ? r : rows - 1;                              // clamp: out-of-range lanes are masked at the store
    return In + r * ldi + 4 * h;
  };
  // One tile: the side operand of THIS tile is requested first (oldest loads), then every consumed input register set is
  // immediately re-filled with the NEXT tile's data, so that loads stay in flight under the MFMAs and every wait the
  // compiler inserts is a counted one (never a drain).
  auto do_tile = [&](f4 (&cur)[LOADS], f4 (&nxt)[LOADS], int64_t tl, bool has_next) {
    const int64_t r = tl * 32 + n;
    const int64_t rc = r < rows ? r : rows - 1;
    f4 sd[NT][4];
    if (side) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          sd[t][g] = *reinterpret_cast<const f4 *>(side_p + rc * ep.ld1 + 32 * (wave * NT + t) + 4 * h + 8 * g);
    }
    const float *pn = row_ptr(tl + 1);
    f16v acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int t8 = 0; t8 < LOADS; ++t8) {
      const f4 bv = cur[t8];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][4 * t8 + 0], bv.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][4 * t8 + 1], bv.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][4 * t8 + 2], bv.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][4 * t8 + 3], bv.w, acc[t], 0, 0, 0);
      }
      if (has_next) nxt[t8] = *reinterpret_cast<const f4 *>(pn + 8 * t8);
    }
    // ---- epilogue: lane (n, h) owns row r, columns 32*(wave*NT+t) + 8g + 4h .. +3 ----
    if (r < rows) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = 32 * (wave * NT + t) + 4 * h + 8 * g;
          f4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
          if constexpr (EPI == EPI_FWD) {
            v += e0[t][g];
            if (side) v += sd[t][g];
            *reinterpret_cast<f4 *>(Out + r * ldo + c) = v;
            if (ep.o2) {
              f4 e4 = {elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
              *reinterpret_cast<f4 *>(ep.o2 + r * ep.ld2 + c) = e4;
            }
          } else {
            if (side) {
              const f4 xv = sd[t][g] - e0[t][g];
              v.x += __builtin_fmaf(xv.x, e1[t][g].x, e2[t][g].x);
              v.y += __builtin_fmaf(xv.y, e1[t][g].y, e2[t][g].y);
              v.z += __builtin_fmaf(xv.z, e1[t][g].z, e2[t][g].z);
              v.w += __builtin_fmaf(xv.w, e1[t][g].w, e2[t][g].w);
            }
            *reinterpret_cast<f4 *>(Out + r * ldo + c) = v;
          }
        }
    }
  };
  {
    const float *p0 = row_ptr(tile);
#pragma unroll
    for (int t8 = 0; t8 < LOADS; ++t8) b0[t8] = *reinterpret_cast<const f4 *>(p0 + 8 * t8);
  }
  while (tile < tend) {
    do_tile(b0, b1, tile, tile + 1 < tend);
    if (++tile >= tend) break;
    do_tile(b1, b0, tile, tile + 1 < tend);
    ++tile;
  }
}

inline unsigned gemm_grid(int64_t rows) {
  const int64_t ntiles = (rows + 31) / 32;
  int64_t b = kCUs;                       // one 4-wave workgroup per CU: a single wave per SIMD owns the register file
  if (b > ntiles) b = ntiles;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int sn_linear_fwd_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *bias,
                      const float *residual, int64_t ldr, float *y, int64_t ldy, float *y_elu, int64_t lde,
                      int64_t rows, int32_t K, int32_t J, void *stream) {
  if (rows < 0 || K < 1 || J < 1 || ldx < K || ldw < K || ldy < J) return SN_E_SHAPE;
  if (J != 128 || (K != 128 && K != 256)) return SN_E_UNSUPPORTED;
  if (rows == 0) return SN_OK;
  if (!x || !W || !bias || !y) return SN_E_NULL;
  if (!aligned16(x) || !aligned16(W) || !aligned16(bias) || !aligned16(y) || (ldx % 4) || (ldw % 4) || (ldy % 4) ||
      (residual && (!aligned16(residual) || (ldr % 4) || ldr < J)) || (y_elu && (!aligned16(y_elu) || (lde % 4) || lde < J)))
    return SN_E_ALIGN;
  EpiArgs ep{bias, residual, nullptr, nullptr, y_elu, ldr, lde};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = gemm_grid(rows);
  if (K == 256)
    hipLaunchKernelGGL((gemm_rows_k<256, 1, false, EPI_FWD>), dim3(grid), dim3(kWG), 0, s, x, ldx, W, ldw, y, ldy, rows, ep);
  else
    hipLaunchKernelGGL((gemm_rows_k<128, 1, false, EPI_FWD>), dim3(grid), dim3(kWG), 0, s, x, ldx, W, ldw, y, ldy, rows, ep);
  return launch_status();
}

int sn_linear_dgrad_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                        const float *center, const float *B, const float *Cc, float *dx, int64_t lddx, int64_t rows,
                        int32_t J, int32_t C, void *stream) {
  if (rows < 0 || J < 1 || C < 1 || lddy < J || ldw < C || lddx < C) return SN_E_SHAPE;
  if (J != 128 || (C != 128 && C != 256)) return SN_E_UNSUPPORTED;
  if (rows == 0) return SN_OK;
  if (!dy || !W || !dx) return SN_E_NULL;
  if (B && (!x || !Cc)) return SN_E_NULL;
  if (!aligned16(dy) || !aligned16(W) || !aligned16(dx) || (lddy % 4) || (ldw % 4) || (lddx % 4)) return SN_E_ALIGN;
  if (B && (!aligned16(x) || !aligned16(B) || !aligned16(Cc) || (center && !aligned16(center)) || (ldx % 4) || ldx < C))
    return SN_E_ALIGN;
  EpiArgs ep{x, center, B, Cc, nullptr, ldx, 0};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = gemm_grid(rows);
  if (C == 256)
    hipLaunchKernelGGL((gemm_rows_k<128, 2, true, EPI_DGRAD>), dim3(grid), dim3(kWG), 0, s, dy, lddy, W, ldw, dx, lddx, rows, ep);
  else
    hipLaunchKernelGGL((gemm_rows_k<128, 1, true, EPI_DGRAD>), dim3(grid), dim3(kWG), 0, s, dy, lddy, W, ldw, dx, lddx, rows, ep);
  return launch_status();
}

}  // extern "C"
