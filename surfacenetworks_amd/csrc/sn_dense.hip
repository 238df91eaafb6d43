// sn_dense.hip — kernels around the per-node BatchNorm+Linear of the residual blocks (gfx950).
//
//   colstats_k / colstats_final_k   per-channel sum and sum of squares, fp64 accumulation (HBM-bound, one pass)
//   wgrad_mfma_k / wgrad_reduce_k   G = dyᵀ·x for tall-skinny operands: split-K over row slabs on the fp32 MFMA
//                                   (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD) — the only MFMA use in
//                                   this library, as the dense per-node MLP is the only GEMM-shaped work on the path
//   affine_cols_acc_k               dx += x*B[c] + C[c]  (tail of the BatchNorm backward)
//
// Interfaces and the reference code they replace: include/sn_spmm.h.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include "sn_spmm.h"

// per-launch timing shared with sn_kernels.hip (the facility behind sn_timing_*)
int sn_internal_cu_count();
hipError_t sn_internal_fill(void *dst, int value, size_t bytes, hipStream_t s);
hipError_t sn_internal_copy2d(void *dst, int64_t dpitch, const void *src, int64_t spitch, int64_t width, int64_t rows, hipStream_t s);
bool sn_internal_timing_slot(int kind, int64_t rows, int64_t width, int64_t bytes, int outw, hipEvent_t *s, hipEvent_t *e);

namespace {

constexpr int kWG = 256;
#define kCUs sn_internal_cu_count()      // compute units of the current device (256 on an MI355X in SPX mode)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f4 ld4_s(const float *p, int nt) {
  return nt ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p)) : *reinterpret_cast<const f4 *>(p);
}
__device__ __forceinline__ void st4_s(float *p, f4 v, int nt) {
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
  else *reinterpret_cast<f4 *>(p) = v;
}
// statistics passes (colstats_k, segstats_k): read-once operands
__device__ __forceinline__ f4 ld4_stat(const float *p) { return ld4_s(p, 0); }

constexpr int kStreamNT = 1;     // elementwise passes stream with non-temporal loads/stores (see sn_kernels.hip)

inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SN_OK : (int)e;
}
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ------------------------------------------------------------------------------------------------
// column statistics
// ------------------------------------------------------------------------------------------------
constexpr int kStatBlocks = 512;

// VEC: C % 4 == 0 and (C/4) divides 256: a thread owns 4 adjacent columns and every (256/(C/4))-th row.
template <bool VEC>
__global__ __launch_bounds__(kWG) void colstats_k(const float *__restrict__ x, int64_t ld, int64_t rows, int C,
                                                  double *__restrict__ partial /* [grid][2][C] */) {
  extern __shared__ double sm[];   // [lanes_r][2][C]
  const int cw = VEC ? C / 4 : C;              // column groups
  const int lanes_r = kWG / cw;                // row lanes per block (>= 1)
  const int cg = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = r0 + per < rows ? r0 + per : rows;
  if constexpr (VEC) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (rl < lanes_r) {
      const float *p = x + cg * 4;
      int64_t r = r0 + rl;
      for (; r + 3 * lanes_r < r1; r += 4 * lanes_r) {     // 4 independent 16-byte loads in flight
        const f4 a = ld4_stat(p + r * ld);
        const f4 b = ld4_stat(p + (r + lanes_r) * ld);
        const f4 c = ld4_stat(p + (r + 2 * lanes_r) * ld);
        const f4 d = ld4_stat(p + (r + 3 * lanes_r) * ld);
#define SN_ACC(v)                                                                    \
  s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;        \
  q0 += (double)v.x * v.x; q1 += (double)v.y * v.y; q2 += (double)v.z * v.z; q3 += (double)v.w * v.w;
        SN_ACC(a) SN_ACC(b) SN_ACC(c) SN_ACC(d)
      }
      for (; r < r1; r += lanes_r) {
        const f4 a = ld4_stat(p + r * ld);
        SN_ACC(a)
#undef SN_ACC
      }
      double *o = sm + (int64_t)rl * 2 * C;
      o[cg * 4] = s0; o[cg * 4 + 1] = s1; o[cg * 4 + 2] = s2; o[cg * 4 + 3] = s3;
      o[C + cg * 4] = q0; o[C + cg * 4 + 1] = q1; o[C + cg * 4 + 2] = q2; o[C + cg * 4 + 3] = q3;
    }
  } else {
    double s = 0, q = 0;
    if (rl < lanes_r) {
      for (int64_t r = r0 + rl; r < r1; r += lanes_r) {
        const double v = x[r * ld + cg];
        s += v;
        q += v * v;
      }
      sm[(int64_t)rl * 2 * C + cg] = s;
      sm[(int64_t)rl * 2 * C + C + cg] = q;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += kWG) {
    double t = 0;
    for (int l = 0; l < lanes_r; ++l) t += sm[(int64_t)l * 2 * C + i];      // fixed order
    partial[(int64_t)blockIdx.x * 2 * C + i] = t;
  }
}

// 32 columns x 8 partial groups per block; group g sums partial blocks g, g+8, ... ; groups combined in fixed order.
// Element i of a partial row is (kind = i / Ch, column = i % Ch), Ch = C2/2; it lands at out[kind * out_ld + out_off + column]
// (out_ld = Ch, out_off = 0: the plain [sums | squares] vector; otherwise one half of a wider statistics vector).
__global__ __launch_bounds__(kWG) void colstats_final_k(const double *__restrict__ partial, int nblk, int C2,
                                                        double *__restrict__ out, int64_t out_ld, int64_t out_off) {
  __shared__ double sm[8][32];
  const int cg = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + cg;
  double t = 0;
  if (i < C2) {
    // eight loads in flight, added in the original order (a dependent load-add chain costs ~0.3 us per term)
    int b = pg;
    for (; b + 56 < nblk; b += 64) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + 8 * u) * C2 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < nblk; b += 8) t += partial[(int64_t)b * C2 + i];
  }
  sm[pg][cg] = t;
  __syncthreads();
  if (pg == 0 && i < C2) {
    double r = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) r += sm[g][cg];
    const int Ch = C2 >> 1, kind = i >= Ch;
    out[(int64_t)kind * out_ld + out_off + (i - kind * Ch)] = r;
  }
}

// Both halves of a concat buffer's statistics in ONE launch: workgroups [0, nb_lo) finish the first producer's partials into
// out[kind][0 .. C_lo), the rest the second producer's into out[kind][C_lo ..) — colstats_final_k twice, same order of
// addition, one launch less in front of every Linear of a Dirac / Laplacian stage.
__global__ __launch_bounds__(kWG) void colstats_final2_k(const double *__restrict__ p_lo, int nblk_lo, int C_lo, int ld_lo,
                                                         const double *__restrict__ p_hi, int nblk_hi, int C_hi, int ld_hi,
                                                         double *__restrict__ out) {
  // a partial row is [sums | squares] of `ld` channels each, of which the first C are the producer's (ld > C: the forward GEMM
  // of a 64-output layer leaves its partials in the 128-column layout of the kernel)
  __shared__ double sm[8][32];
  const int nb_lo = (2 * C_lo + 31) / 32;
  const bool hi = (int)blockIdx.x >= nb_lo;
  const double *partial = hi ? p_hi : p_lo;
  const int nblk = hi ? nblk_hi : nblk_lo, Cx = hi ? C_hi : C_lo, C2 = 2 * Cx, ld = hi ? ld_hi : ld_lo;
  const int out_off = hi ? C_lo : 0, out_ld = C_lo + C_hi;
  const int cg = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int i = ((int)blockIdx.x - (hi ? nb_lo : 0)) * 32 + cg;
  double t = 0;
  if (i < C2) {
    const int64_t rs = 2 * (int64_t)ld;                       // doubles per partial row
    const double *p = partial + (i >= Cx ? ld + (i - Cx) : i);
    int b = pg;
    for (; b + 120 < nblk; b += 128) {          // sixteen, then eight loads in flight (in the step the partials are cold)
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p[(int64_t)(b + 8 * u) * rs];
#pragma unroll
      for (int u = 0; u < 16; ++u) t += v[u];
    }
    for (; b + 56 < nblk; b += 64) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(b + 8 * u) * rs];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < nblk; b += 8) t += p[(int64_t)b * rs];
  }
  sm[pg][cg] = t;
  __syncthreads();
  if (pg == 0 && i < C2) {
    double r = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) r += sm[g][cg];
    const int Ch = C2 >> 1, kind = i >= Ch;
    out[(int64_t)kind * out_ld + out_off + (i - kind * Ch)] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient G = dyᵀ·x on the fp32 MFMA.
//
// v_mfma_f32_32x32x2_f32: D[i][n] += A[i][k]·B[k][n], k = 0,1; lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][n = l&31].  With k = two consecutive ROWS of the operands, lanes 0-31 read row r, lanes 32-63 row r+1 —
// i.e. the natural row-major layout, no transposition.  A lane loads 4 adjacent columns with one 16-byte load
// (32 lanes x 16 B = one 512-byte row of 128 columns); register q of that load then stands for columns 4n+q, so
// MFMA tile (qa, qb) accumulates G[4i+qa][4n+qb] — a column permutation undone when the tile is written out.
// Workgroup = 4 waves = one slab of rows; wave w owns dy columns {4i+w} (A register w) and all x columns:
// (C/128)*4 tiles of 16 accumulators.  Partial 128 x C tiles go to the workspace, reduced in slab order.
// ------------------------------------------------------------------------------------------------
template <int CT /* C / 128: 1 or 2 */>
__global__ __launch_bounds__(kWG, 2) void wgrad_mfma_k(const float *__restrict__ dy, int64_t lddy,
                                                       const float *__restrict__ x, int64_t ldx,
                                                       const float *__restrict__ center, int64_t rows,
                                                       int J /* <= 128, multiple of 4 */, int C,
                                                       float *__restrict__ partial /* [grid][128][C] */,
                                                       float *__restrict__ colpart /* [grid][128] column sums of dy | NULL */) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 31, kk = lane >> 5;
  const int64_t per = (((rows + gridDim.x - 1) / gridDim.x) + 1) & ~(int64_t)1;   // even slab size
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = r0 + per < rows ? r0 + per : rows;
  f16v acc[CT * 4];
#pragma unroll
  for (int t = 0; t < CT * 4; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const bool jok = 4 * n < J;                       // dy narrower than 128 columns: missing columns read as 0
  const float *pd = dy + 4 * n;
  const float *px = x + 4 * n;
  f4 mu[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
    mu[c] = center ? *reinterpret_cast<const f4 *>(center + 4 * n + 128 * c) : f4{0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;                              // row pairs per group; two groups of loads are in flight
  int64_t r = r0;
  float dsum = 0.f;                                 // this lane's share of colsum(dy)[4n + wave] (the bias gradient)
  float a[2][U];                                    // this wave only needs dy column 4n + wave
  f4 b[2][U][CT];
  auto load_group = [&](int buf, int64_t rb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = rb + 2 * u + kk;
      a[buf][u] = jok ? pd[row * lddy + wave] : 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c) b[buf][u][c] = *reinterpret_cast<const f4 *>(px + row * ldx + 128 * c);
    }
  };
  auto mfma_group = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float av = a[buf][u];
      dsum += av;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const f4 bv = b[buf][u][c] - mu[c];
        acc[c * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.x, acc[c * 4 + 0], 0, 0, 0);
        acc[c * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.y, acc[c * 4 + 1], 0, 0, 0);
        acc[c * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.z, acc[c * 4 + 2], 0, 0, 0);
        acc[c * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv.w, acc[c * 4 + 3], 0, 0, 0);
      }
    }
  };
  const int64_t ngroups = (r1 - r0) / (2 * U);
  if (ngroups > 0) {
    load_group(0, r);
    for (int64_t g = 0; g + 1 < ngroups; g += 2) {      // steady state: loads of the next group fly under the MFMAs
      load_group(1, r + 2 * U);
      mfma_group(0);
      if (g + 2 < ngroups) load_group(0, r + 4 * U);
      mfma_group(1);
      r += 4 * U;
    }
    if (ngroups & 1) {
      mfma_group(0);
      r += 2 * U;
    }
  }
  for (; r < r1; r += 2) {                          // tail pairs (second row of the last pair may be past the end)
    const int64_t row = r + kk;
    const bool ok = row < r1;
    const f4 a = (ok && jok) ? *reinterpret_cast<const f4 *>(pd + row * lddy) : f4{0.f, 0.f, 0.f, 0.f};
    const float av = a[wave];
    dsum += av;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const f4 b = ok ? *reinterpret_cast<const f4 *>(px + row * ldx + 128 * c) - mu[c] : f4{0.f, 0.f, 0.f, 0.f};
      acc[c * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b.x, acc[c * 4 + 0], 0, 0, 0);
      acc[c * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b.y, acc[c * 4 + 1], 0, 0, 0);
      acc[c * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b.z, acc[c * 4 + 2], 0, 0, 0);
      acc[c * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b.w, acc[c * 4 + 3], 0, 0, 0);
    }
  }
  if (colpart) {
    dsum += __shfl_xor(dsum, 32, 64);               // even + odd rows
    if (kk == 0) colpart[(int64_t)blockIdx.x * 128 + 4 * n + wave] = dsum;
  }
  // D layout (32x32): column n = lane & 31, row i = (e & 3) + 8*(e >> 2) + 4*(lane >> 5), e = 0..15.
  float *P = partial + (int64_t)blockIdx.x * 128 * C;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * kk;
        P[(int64_t)(4 * i + wave) * C + 128 * c + 4 * n + qb] = acc[c * 4 + qb][e];
      }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient G = dy^T (x - center) on the 16-bit matrix pipe: the operands' rows are the contraction index, so a lane's
// MFMA fragment (EIGHT CONSECUTIVE k of one column) is a column walk and the operands go through LDS transposed — a unit of
// 8 rows x 1 column becomes one 16-byte slot per piece.  Slot of (column c, row group g): g·PL + (c%4)·QP + c/4 with QP =
// (#columns/4) + 4: consecutive writer lanes hit consecutive slots, the 16-lane groups of a fragment ds_read_b128 fall on 16
// different slot residues mod 16 — both directions conflict-free.  Columns 0..127 are dy (zero past J), the rest x - center.
// Two kernels: wgrad_h_k (two scaled fp16 pieces, needs bounds on its operands; the default of the training step) and
// wgrad_u_k (three bf16 pieces, no bounds needed: eval-mode BatchNorm, dy from a kernel that leaves no maxima).  Rounds 1-3's
// wave-specialised and LDS-DMA forms are described in LABNOTES (k23, wgrad_d).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16v mfma_bf16(const u4 &a, const u4 &b, const f16v &c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}
constexpr int kWgradThreads = 512;          // 8 waves, two per SIMD


// ------------------------------------------------------------------------------------------------
// wgrad_u_k — the three-piece bf16 weight gradient (six exact partial products per term), uniform waves.
//
// All 8 waves do both jobs: per 32-row block a wave converts its share of the operands — units of 8 rows x 1 column, one
// 16-byte LDS slot per piece: 3 units per lane at C = 256, 2 at C = 128, every lane of every wave busy — and multiplies
// 2 x CT output tiles (2 dy tiles x CT x tiles).
//   * conversion slot 0 of wave w = the dy half-block (row group w/2, 64 columns (w%2)·64 ..), slots 1.. = x half-blocks
//     w + 8(k-1) of the 4 row groups x 2·CT half-blocks: roles are compile-time, a slot has one row group; global loads are
//     dword-per-lane over 64 consecutive columns (256 contiguous bytes per row and instruction) through a raw buffer
//     resource whose extent ends with the slab (rows past it read 0: no masks, no clamps); block b+2 is requested into the
//     registers block b+1 was converted from, while block b is multiplied;
//   * one workgroup barrier per 32-row block, two LDS images of 4 row groups (154 KB at C = 256); slot layout as above;
//   * the two co-resident waves of a SIMD run the same stream, and a wave is in-order: what overlaps the matrix pipe, the
//     vector ALU and the memory pipe is the MIX inside the stream — a pair of rows converted every few MFMAs, one load at a
//     time behind it (program order pinned by sched_barrier).
// Measurements, the structures tried on the way and the ablation builds: LABNOTES k22.
// ------------------------------------------------------------------------------------------------
template <int I>
struct WIC {
  static constexpr int value = I;
};
template <int I, int N, class F>
__device__ __forceinline__ void wstatic_for(F &&f) {
  if constexpr (I < N) {
    f(WIC<I>{});
    wstatic_for<I + 1, N>(f);
  }
}

template <int CT /* C / 128: 1 or 2 */>
__global__ __launch_bounds__(kWgradThreads, 1) void wgrad_u_k(const float *__restrict__ dy, int64_t lddy,
                                                              const float *__restrict__ x, int64_t ldx,
                                                              const float *__restrict__ center, int64_t rows, int J, int C,
                                                              float *__restrict__ partial /* [grid][128][C] */,
                                                              float *__restrict__ colpart /* [grid][128] | NULL */,
                                                              int64_t seg_rows /* 0: even split of all rows */, int spm,
                                                              const int64_t *__restrict__ slab_off /* [grid + 1] | NULL */) {
  constexpr int NCG = 32 + 32 * CT;          // column groups of 4: 32 of dy, 32·CT of x
  constexpr int QP = NCG + 4;                // slots per (column % 4) plane; QP % 16 == 4 keeps fragment reads conflict-free
  constexpr int PL = 4 * QP;                 // slots per row group (8 rows)
  constexpr int NK = 1 + CT;                 // conversion slots per wave and block: one of dy, CT of x
  constexpr int NM = 24 * CT;                // MFMAs per wave and block
  static_assert(QP % 16 == 4, "slot permutation");
  __shared__ u4 img[2][3][4 * PL];           // [buffer][piece][slot]; one block = 32 rows = 4 row groups
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  int64_t r0, r1;                            // row slab of this workgroup: from the caller's table (ragged meshes), or an even split
  if (slab_off) {
    r0 = slab_off[blockIdx.x];
    r1 = slab_off[blockIdx.x + 1];
  } else if (seg_rows > 0) {
    const int64_t mesh = blockIdx.x / spm, part = blockIdx.x % spm;
    int64_t per = (seg_rows + spm - 1) / spm;
    per = (per + 15) & ~(int64_t)15;
    const int64_t mend = (mesh + 1) * seg_rows < rows ? (mesh + 1) * seg_rows : rows;
    r0 = mesh * seg_rows + part * per;
    r1 = r0 + per < mend ? r0 + per : mend;
  } else {
    int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    per = (per + 15) & ~(int64_t)15;
    r0 = (int64_t)blockIdx.x * per;
    r1 = r0 + per < rows ? r0 + per : rows;
  }
  const int nblocks = r1 > r0 ? (int)((r1 - r0 + 31) / 32) : 0;
  const int span = r1 > r0 ? (int)(r1 - r0) : 0;       // rows of the slab (far below 2^31)
  float *P = partial + (int64_t)blockIdx.x * 128 * C;

  // ---- matrix role: dy tiles 2·ga, 2·ga + 1  x  x tiles CT·gb .. CT·gb + CT - 1 ----
  const int i = lane & 31, kh = lane >> 5;
  const int ga = wave >> 2, gb = wave & 3;
  const int fo = kh * PL + (i & 3) * QP + (i >> 2);          // my fragment slot: + 2·PL·step + 8·(tile in column groups of 32)
  f16v acc[2][CT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // ---- conversion role ----
  // my column inside a 64-column half-block = my lane: consecutive lanes read consecutive dwords (a permutation that made
  // the ds_write_b128 groups conflict-free — lane = column is two-way conflicted on 30 % of the LDS cycles, still under the
  // store's own VGPR-transfer time — left each lane quad 16 bytes apart and cost the loads 20 %: 127 against 101 µs for the
  // memory path alone at 322 624 rows)
  const int lcol = lane;
  int s_rg[NK];                  // row group of the block (scalar)
  const float *s_cur[NK];        // operand + first column of the half-block + first row of the slot in the block to load next
  int l_voff[NK];                // my byte offset in a row of the half-block; past any extent if my dy column does not exist
  int l_slot[NK];
  float l_mu[NK];                // my column's centre (x slots)
  float l_sum = 0.f;             // running sum of my dy column (slot 0)
  const int dy_rstep = 4 * (int)lddy, x_rstep = 4 * (int)ldx;          // bytes per row
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int xhb = wave + 8 * (k - 1);
    const int rg = k == 0 ? wave >> 1 : xhb / (2 * CT);
    const int c = k == 0 ? 64 * (wave & 1) + lcol : 128 + 64 * (xhb % (2 * CT)) + lcol;      // column in the image
    s_rg[k] = rg;
    s_cur[k] = k == 0 ? dy + 64 * (wave & 1) + (r0 + 8 * rg) * lddy : x + 64 * (xhb % (2 * CT)) + (r0 + 8 * rg) * ldx;
    l_voff[k] = (k == 0 && c >= J) ? 0x7fffff00 : 4 * lcol;
    l_mu[k] = (k > 0 && center) ? center[c - 128] : 0.f;
    l_slot[k] = rg * PL + (c & 3) * QP + (c >> 2);
  }
  float raw[1][NK][8];           // rows in flight: one register set (a second one, two blocks ahead, measured no faster)
  u4 cH[NK], cM[NK], cL[NK];     // pieces of the slot being converted (filled pair by pair, written as three 16-byte slots)
  // The work of a slot, in chunks small enough to sit between two MFMAs:
  //   conv_pair   rows 2p, 2p+1: centre (x) | column sum (dy), then x = h + m + l by packed fp32 subtractions — each piece the
  //               upper 16 bits of an exact remainder (≈10 vector instructions);
  //   conv_write  the three pieces of the 8 rows as one LDS slot each;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u2v __attribute__((ext_vector_type(2)));
  auto conv_pair = [&](auto sc, auto kc, auto pc) {
    constexpr int set = decltype(sc)::value, k = decltype(kc)::value, p = decltype(pc)::value;
    f2 xv = {raw[set][k][2 * p], raw[set][k][2 * p + 1]};
    if constexpr (k == 0) l_sum += xv.x + xv.y;
    else xv -= f2{l_mu[k], l_mu[k]};
    const u2v xb = __builtin_bit_cast(u2v, xv);
    const f2 r = xv - __builtin_bit_cast(f2, xb & 0xFFFF0000u);
    const u2v rb = __builtin_bit_cast(u2v, r);
    const f2 l = r - __builtin_bit_cast(f2, rb & 0xFFFF0000u);
    const u2v lb = __builtin_bit_cast(u2v, l);
    cH[k][p] = __builtin_amdgcn_perm(xb.y, xb.x, 0x07060302u);
    cM[k][p] = __builtin_amdgcn_perm(rb.y, rb.x, 0x07060302u);
    cL[k][p] = __builtin_amdgcn_perm(lb.y, lb.x, 0x07060302u);
  };
  auto conv_write = [&](auto kc, int buf) {
    constexpr int k = decltype(kc)::value;
    img[buf][0][l_slot[k]] = cH[k];
    img[buf][1][l_slot[k]] = cM[k];
    img[buf][2][l_slot[k]] = cL[k];
  };
  // open_slot: the descriptor of slot k's rows in block b (b counts up by one per slot and call: s_cur runs along).  Its
  // extent ends with the slab: rows past it — whole blocks past it: the prefetch runs two blocks ahead — read as 0 without
  // traffic.  Such a row is 0 in BOTH operands' raw data; the centred x is then -mu, but the dy row is exactly 0 in every
  // piece, so it adds nothing to the products or to the column sums: no masks.
  __amdgpu_buffer_rsrc_t s_rs[NK];
  auto open_slot = [&](auto kc, int b) {
    constexpr int k = decltype(kc)::value;
    const int rstep = k == 0 ? dy_rstep : x_rstep;
    int left = span - 32 * b - 8 * s_rg[k];                           // rows of the slab from the slot's first row on
    left = left < 0 ? 0 : (left > 8 ? 8 : left);
    const int extent = __builtin_amdgcn_readfirstlane(left * rstep);      // (keeps the clamp's med3 out of the descriptor)
    s_rs[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s_cur[k]), 0, extent, 0x00020000);
    s_cur[k] += 8 * rstep;                                            // 32 rows on
  };
  auto load_row = [&](auto sc, auto kc, auto jc) {
    constexpr int set = decltype(sc)::value, k = decltype(kc)::value, j = decltype(jc)::value;
    const int rstep = k == 0 ? dy_rstep : x_rstep;
    raw[set][k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s_rs[k], l_voff[k] + j * rstep, 0, 0));
  };
  // The other work of a block, dealt out behind its MFMAs (m = 0 .. NM-1) — the two co-resident waves of a SIMD run this
  // same stream and a wave is in-order, so what overlaps the matrix pipe, the vector ALU and the memory pipe is the mix
  // INSIDE the stream:
  //   conversion of block b+1 into the other image: one pair of rows every PSTEP MFMAs;
  //   requests of block b+2 into the registers just converted: ONE load at a time — eight in a row fill the memory
  //     pipe's queue and stall the wave, hence the MFMAs behind them, for hundreds of cycles (the kernel is HBM-bound:
  //     a CU's share of the bandwidth is one 256-byte load per ≈26 cycles);
  //   the three LDS slots of a unit after its fourth pair.
  constexpr int PSTEP = NM / (4 * NK);      // 4 | 3
  auto behind_mfma = [&](auto sc, auto ic, auto mc, int b) {
    constexpr int set = decltype(sc)::value, image = decltype(ic)::value, m = decltype(mc)::value;
    if constexpr (m % PSTEP == 0 && m / PSTEP < 4 * NK) {
      constexpr int q = m / PSTEP, k = q / 4, p = q % 4;
      if constexpr (p == 0) open_slot(WIC<k>{}, b);
      conv_pair(sc, WIC<k>{}, WIC<p>{});
    }
    constexpr int L = CT == 2 ? (m % 2 == 1 ? (m - 1) / 2 : -1) : (m % 3 != 0 ? (m / 3) * 2 + (m % 3 - 1) : -1);
    if constexpr (L >= 0 && L < 8 * NK) load_row(sc, WIC<L / 8>{}, WIC<L % 8>{});
    constexpr int wk = CT == 2 ? (m % 16 == 14 ? m / 16 : -1) : (m % 12 == 11 ? m / 12 : -1);
    if constexpr (wk >= 0 && wk < NK) conv_write(WIC<wk>{}, image);
  };
#define SN_BLOCK_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  auto multiply_block = [&](auto sc, int b) {
    constexpr int buf = decltype(sc)::value;           // block b = image buf = b & 1
    SN_BLOCK_BARRIER();                   // image buf complete; image buf^1 (read during block b-1) may be overwritten
    // fragments of both 16-row steps are requested up front (before any write of this block in program order)
    u4 A[2][2][3], B[2][CT][3];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int a = 0; a < 2; ++a) A[st][a][p] = img[buf][p][fo + 2 * st * PL + 8 * (2 * ga + a)];
#pragma unroll
        for (int c = 0; c < CT; ++c) B[st][c][p] = img[buf][p][fo + 2 * st * PL + 32 + 8 * (CT * gb + c)];
      }
    // six partial products (pieces of dy, x): (l,h) (h,l) (m,m) (m,h) (h,m) (h,h) — small terms first; tile-inner, so
    // consecutive MFMAs never share an accumulator; the order of the stream is pinned by sched_barrier
    wstatic_for<0, NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value, g = m / (2 * CT), st = g / 6, t = g % 6;
      constexpr int a = (m % (2 * CT)) / CT, c = m % CT;
      constexpr int pa = (0x001102 >> (4 * t)) & 15, pb = (0x010120 >> (4 * t)) & 15;
      acc[a][c] = mfma_bf16(A[st][a][pa], B[st][c][pb], acc[a][c]);
      behind_mfma(WIC<0>{}, WIC<buf ^ 1>{}, mc, b + 2);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  if (nblocks > 0) {
    // block 0 requested, converted into image 0, block 1 requested into its registers
    wstatic_for<0, NK>([&](auto kc) {
      open_slot(kc, 0);
      wstatic_for<0, 8>([&](auto jc) { load_row(WIC<0>{}, kc, jc); });
    });
    wstatic_for<0, NM>([&](auto mc) { behind_mfma(WIC<0>{}, WIC<0>{}, mc, 1); });
    for (int b = 0; b < nblocks; b += 2) {
      multiply_block(WIC<0>{}, b);
      if (b + 1 < nblocks) multiply_block(WIC<1>{}, b + 1);
    }
  }
  if (colpart) {                             // bias gradient: column sums of dy — one (row group, column) entry per wave
    __syncthreads();
    float *sm = reinterpret_cast<float *>(&img[0][0][0]);
    sm[s_rg[0] * 128 + 64 * (wave & 1) + lcol] = l_sum;
    __syncthreads();
    if (tid < 128) colpart[(int64_t)blockIdx.x * 128 + tid] = (sm[tid] + sm[128 + tid]) + (sm[256 + tid] + sm[384 + tid]);
  }
  // D layout (32x32): column n = lane & 31 (x column within its tile), row (dy column within its tile)
  // = (e & 3) + 8 (e >> 2) + 4 (lane >> 5), e = 0..15
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int jr = 32 * (2 * ga + a) + (e & 3) + 8 * (e >> 2) + 4 * kh;
        P[(int64_t)jr * C + 32 * (CT * gb + c) + i] = acc[a][c][e];
      }
#undef SN_BLOCK_BARRIER
}

__global__ __launch_bounds__(kWG) void wgrad_reduce_k(const float *__restrict__ partial, int nslab, int J, int C,
                                                      float *__restrict__ G, const float *__restrict__ colpart,
                                                      double *__restrict__ dysum,
                                                      float *__restrict__ segsum /* [nslab/spm][J] | NULL */, int spm,
                                                      const int64_t *__restrict__ seg_slab_ptr /* [nseg + 1] | NULL */,
                                                      int nseg) {
  __shared__ double sm[4][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int nG = J * C;
  const int i = blockIdx.x * 64 + o;            // output element j*C + c  (partials are laid out [slab][128][C]);
  double t = 0;                                 // elements past J*C are the J column sums of dy ([slab][128]), and past
  const float *src = nullptr;                   // those the per-mesh column sums of dy — the mesh's consecutive slabs
  int64_t stride = 128;                         // (spm each, or from the table)
  int cnt = 0;
  if (i < nG) {
    src = partial + i, stride = (int64_t)128 * C, cnt = nslab;
  } else if (colpart && i < nG + J) {
    src = colpart + (i - nG), cnt = nslab;
  } else if (segsum && i >= nG + J && i < nG + J + (seg_slab_ptr ? nseg : nslab / spm) * J) {
    const int k = i - nG - J, mesh = k / J, j = k - mesh * J;
    const int64_t p0 = seg_slab_ptr ? seg_slab_ptr[mesh] : (int64_t)mesh * spm;
    const int64_t p1 = seg_slab_ptr ? seg_slab_ptr[mesh + 1] : p0 + spm;
    src = colpart + p0 * 128 + j, cnt = (int)(p1 - p0);
  }
  if (src) {
    // the four waves take every fourth slab, eight loads in flight each
    int sl = g;
    for (; sl + 28 < cnt; sl += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(sl + 4 * u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += (double)v[u];
    }
    for (; sl < cnt; sl += 4) t += (double)src[(int64_t)sl * stride];
  }
  sm[g][o] = t;
  __syncthreads();
  if (g == 0) {
    const double r = sm[0][o] + sm[1][o] + sm[2][o] + sm[3][o];
    if (i < nG) G[i] = (float)r;
    else if (colpart && i < nG + J) dysum[i - nG] = r;
    else if (src) segsum[i - nG - J] = (float)r;
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a Linear with a handful of input channels (the models' first layer, 3 or 6 coordinates -> C
// features):  G (J x C) = dy^T x,  db = colsum(dy).  One pass over dy at HBM rate — a GEMM library sees a 128 x 6 output
// with K = 3e5 and has no good tile for it.  A thread owns 4 adjacent dy columns and every (1024/J)-th row; the x row
// is a broadcast load.  fp32 per-thread accumulation over <= ~64 rows, fp64 from there on; two deterministic stages.
// ------------------------------------------------------------------------------------------------
constexpr int kThinBlocks = 1024;
constexpr int kThinMaxC = 8;

template <int C>
__global__ __launch_bounds__(kWG) void wgrad_thin_k(const float *__restrict__ dy, int64_t lddy,
                                                    const float *__restrict__ x, int64_t ldx, int64_t rows, int J,
                                                    double *__restrict__ partial /* [grid][C+1][J] */) {
  extern __shared__ float smf[];                 // [lanes_r][C+1][J]  (the threads' fp32 sums; added up in fp64)
  const int cw = J / 4, lanes_r = kWG / cw;
  const int cg = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  const int64_t r1 = r0 + per < rows ? r0 + per : rows;
  float acc[C + 1][4];
#pragma unroll
  for (int c = 0; c <= C; ++c) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f;
  if (rl < lanes_r) {
    const float *pd = dy + cg * 4;
    int64_t r = r0 + rl;
    for (; r + 3 * lanes_r < r1; r += 4 * lanes_r) {         // 4 rows in flight
      f4 d[4];
      float xv[4][C];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d[u] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(pd + (r + (int64_t)u * lanes_r) * lddy));
#pragma unroll
        for (int c = 0; c < C; ++c) xv[u][c] = x[(r + (int64_t)u * lanes_r) * ldx + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          acc[c][0] = __builtin_fmaf(d[u].x, xv[u][c], acc[c][0]);
          acc[c][1] = __builtin_fmaf(d[u].y, xv[u][c], acc[c][1]);
          acc[c][2] = __builtin_fmaf(d[u].z, xv[u][c], acc[c][2]);
          acc[c][3] = __builtin_fmaf(d[u].w, xv[u][c], acc[c][3]);
        }
        acc[C][0] += d[u].x; acc[C][1] += d[u].y; acc[C][2] += d[u].z; acc[C][3] += d[u].w;
      }
    }
    for (; r < r1; r += lanes_r) {
      const f4 d = *reinterpret_cast<const f4 *>(pd + r * lddy);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float xv = x[r * ldx + c];
        acc[c][0] = __builtin_fmaf(d.x, xv, acc[c][0]);
        acc[c][1] = __builtin_fmaf(d.y, xv, acc[c][1]);
        acc[c][2] = __builtin_fmaf(d.z, xv, acc[c][2]);
        acc[c][3] = __builtin_fmaf(d.w, xv, acc[c][3]);
      }
      acc[C][0] += d.x; acc[C][1] += d.y; acc[C][2] += d.z; acc[C][3] += d.w;
    }
    float *o = smf + (int64_t)rl * (C + 1) * J + cg * 4;
#pragma unroll
    for (int c = 0; c <= C; ++c) {
      o[c * J] = acc[c][0]; o[c * J + 1] = acc[c][1]; o[c * J + 2] = acc[c][2]; o[c * J + 3] = acc[c][3];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (C + 1) * J; i += kWG) {
    double t = 0;
    for (int l = 0; l < lanes_r; ++l) t += (double)smf[(int64_t)l * (C + 1) * J + i];      // fixed order
    partial[(int64_t)blockIdx.x * (C + 1) * J + i] = t;
  }
}

// element i = c*J + j of the (C+1) x J partial layout -> G[j][c] (c < C) or db[j] (c == C)
__global__ __launch_bounds__(kWG) void wgrad_thin_final_k(const double *__restrict__ partial, int nblk, int J, int C,
                                                          float *__restrict__ G, float *__restrict__ db) {
  __shared__ double sm[8][32];
  const int cl = threadIdx.x & 31, pg = threadIdx.x >> 5;
  const int n = (C + 1) * J;
  const int i = blockIdx.x * 32 + cl;
  double t = 0;
  if (i < n) {
    int b = pg;
    for (; b + 56 < nblk; b += 64) {             // eight loads in flight, added in block order
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + 8 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < nblk; b += 8) t += partial[(int64_t)b * n + i];
  }
  sm[pg][cl] = t;
  __syncthreads();
  if (pg == 0 && i < n) {
    double r = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) r += sm[g][cl];
    const int c = i / J, j = i - c * J;
    if (c < C) G[(int64_t)j * C + c] = (float)r;
    else if (db) db[j] = (float)r;
  }
}

// ------------------------------------------------------------------------------------------------
// Masked smooth-L1 training loss of the ARAP harness (src/as_rigid_as_possible/main.py:225-226):
//   loss = scale * sum_{r,c} l(out[r,c]*mask[r] - target[r,c]),   l(d) = d^2/2 (|d| < 1) | |d| - 1/2
// forward: one read of out and target, fp64 partial sums, two deterministic stages;
// backward: gout = (gloss*scale) * mask[r] * clamp(out*mask - target, -1, 1), one pass.
// (torch runs mask-multiply, loss, reduction, loss-backward, mask-multiply as five elementwise passes.)
// ------------------------------------------------------------------------------------------------
constexpr int kLossBlocks = 1024;

__device__ __forceinline__ double sl1(float d) {
  const float a = fabsf(d);
  return a < 1.f ? 0.5 * (double)d * (double)d : (double)a - 0.5;
}

template <bool VEC>
__global__ __launch_bounds__(kWG) void masked_sl1_fwd_k(const float *__restrict__ o, int64_t ldo,
                                                        const float *__restrict__ t, int64_t ldt,
                                                        const float *__restrict__ mask, int64_t rows, int C,
                                                        double *__restrict__ partial) {
  __shared__ double red[kWG];
  constexpr int W = VEC ? 4 : 1;
  const int cw = C / W;
  const int64_t total = rows * cw;
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x; i < total; i += (int64_t)gridDim.x * kWG) {
    const int64_t r = i / cw;
    const int c = (int)(i - r * cw) * W;
    const float m = mask ? mask[r] : 1.f;
    if constexpr (VEC) {
      const f4 ov = ld4_s(o + r * ldo + c, 1), tv = ld4_s(t + r * ldt + c, 1);
      acc += sl1(ov.x * m - tv.x) + sl1(ov.y * m - tv.y) + sl1(ov.z * m - tv.z) + sl1(ov.w * m - tv.w);
    } else {
      acc += sl1(o[r * ldo + c] * m - t[r * ldt + c]);
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kWG / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(kWG) void masked_sl1_final_k(const double *__restrict__ partial, int nblk, double scale,
                                                          float *__restrict__ loss) {
  __shared__ double red[kWG];
  double acc = 0;
  for (int i = threadIdx.x; i < nblk; i += kWG) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kWG / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] * scale);
}

__device__ __forceinline__ float sl1_grad(float d) { return d < -1.f ? -1.f : (d > 1.f ? 1.f : d); }

template <bool VEC>
__global__ __launch_bounds__(kWG) void masked_sl1_bwd_k(const float *__restrict__ o, int64_t ldo,
                                                        const float *__restrict__ t, int64_t ldt,
                                                        const float *__restrict__ mask, int64_t rows, int C, float scale,
                                                        const float *__restrict__ gloss, float *__restrict__ g,
                                                        int64_t ldg) {
  constexpr int W = VEC ? 4 : 1;
  const int cw = C / W;
  const int64_t total = rows * cw;
  const float gs = gloss[0] * scale;
  for (int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x; i < total; i += (int64_t)gridDim.x * kWG) {
    const int64_t r = i / cw;
    const int c = (int)(i - r * cw) * W;
    const float m = mask ? mask[r] : 1.f;
    const float gm = gs * m;
    if constexpr (VEC) {
      const f4 ov = ld4_s(o + r * ldo + c, 1), tv = ld4_s(t + r * ldt + c, 1);
      st4_s(g + r * ldg + c, f4{gm * sl1_grad(ov.x * m - tv.x), gm * sl1_grad(ov.y * m - tv.y),
                                gm * sl1_grad(ov.z * m - tv.z), gm * sl1_grad(ov.w * m - tv.w)}, 1);
    } else {
      g[r * ldg + c] = gm * sl1_grad(o[r * ldo + c] * m - t[r * ldt + c]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Sampler gather (src/as_rigid_as_possible/main.py:126-152 builds these tensors on the host, frame by frame): item i of
// a batch is rows_per_item segments of `len` consecutive floats, segment r at src[base[i] + r*row_stride .. + len).
// One thread per output float: consecutive threads read consecutive floats of a segment (coalesced, no alignment needed).
// ------------------------------------------------------------------------------------------------
// W = 4 (len % 4 == 0, out 16-byte aligned): a thread gathers four consecutive floats (the source run has no alignment)
// and stores them as one 16-byte word.
template <int W>
__global__ __launch_bounds__(kWG) void gather_segments_k(const float *__restrict__ src, const int64_t *__restrict__ base,
                                                         int64_t rows_per_item, int64_t row_stride, int len,
                                                         int64_t total, float *__restrict__ out) {
  const int lw = len / W;
  const int64_t per_item = rows_per_item * lw;
  for (int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x; i < total; i += (int64_t)gridDim.x * kWG) {
    const int64_t item = i / per_item, w = i - item * per_item;
    const int64_t r = w / lw;
    const int c = (int)(w - r * lw) * W;
    const float *p = src + base[item] + r * row_stride + c;
    if constexpr (W == 4) {
      __builtin_nontemporal_store(f4{p[0], p[1], p[2], p[3]}, reinterpret_cast<f4 *>(out) + i);
    } else {
      __builtin_nontemporal_store(p[0], out + i);
    }
  }
}

// The same gather straight into a PACKED batch: item i contributes item_off[i+1] - item_off[i] segments (its real rows), written
// to output rows item_off[i] ..; a thread finds its item by bisection of the (short) offset table.
template <int W>
__global__ __launch_bounds__(kWG) void gather_segments_ragged_k(const float *__restrict__ src, const int64_t *__restrict__ base,
                                                                const int64_t *__restrict__ item_off, int nitems,
                                                                int64_t row_stride, int len, int64_t total,
                                                                float *__restrict__ out) {
  const int lw = len / W;
  for (int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x; i < total; i += (int64_t)gridDim.x * kWG) {
    const int64_t row = i / lw;
    const int c = (int)(i - row * lw) * W;
    int lo = 0, hi = nitems - 1;
    while (lo < hi) {                              // last item with item_off[item] <= row
      const int mid = (lo + hi + 1) >> 1;
      if (item_off[mid] <= row) lo = mid;
      else hi = mid - 1;
    }
    const float *p = src + base[lo] + (row - item_off[lo]) * row_stride + c;
    if constexpr (W == 4) {
      __builtin_nontemporal_store(f4{p[0], p[1], p[2], p[3]}, reinterpret_cast<f4 *>(out) + i);
    } else {
      __builtin_nontemporal_store(p[0], out + i);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Forward of the same first layer: y = x·W^T + b with 1..8 input channels, and optionally elu(y) straight into the next
// block's concat buffer.  Pure output streaming (a thread owns 4 adjacent output columns, its 4 x C weights stay in
// registers, the x row is a broadcast load).
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(kWG) void linear_thin_fwd_k(const float *__restrict__ x, int64_t ldx,
                                                         const float *__restrict__ W, int64_t ldw,
                                                         const float *__restrict__ b, int64_t rows, int J,
                                                         float *__restrict__ y, int64_t ldy, float *__restrict__ ye,
                                                         int64_t lde) {
  const int cw = J / 4, lanes_r = kWG / cw;
  const int cg = threadIdx.x % cw, rl = threadIdx.x / cw;
  if (rl >= lanes_r) return;
  float w[4][C];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < C; ++c) w[q][c] = W[(int64_t)(4 * cg + q) * ldw + c];
  const f4 bias = b ? *reinterpret_cast<const f4 *>(b + 4 * cg) : f4{0.f, 0.f, 0.f, 0.f};
  for (int64_t r = (int64_t)blockIdx.x * lanes_r + rl; r < rows; r += (int64_t)gridDim.x * lanes_r) {
    float xv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xv[c] = x[r * ldx + c];
    f4 acc = bias;
#pragma unroll
    for (int c = 0; c < C; ++c) {                    // ascending-k FMA chain on top of the bias
      acc.x = __builtin_fmaf(xv[c], w[0][c], acc.x);
      acc.y = __builtin_fmaf(xv[c], w[1][c], acc.y);
      acc.z = __builtin_fmaf(xv[c], w[2][c], acc.z);
      acc.w = __builtin_fmaf(xv[c], w[3][c], acc.w);
    }
    if (y) __builtin_nontemporal_store(acc, reinterpret_cast<f4 *>(y + r * ldy + 4 * cg));
    if (ye) {
      const f4 e = {acc.x > 0.f ? acc.x : __expf(acc.x) - 1.0f, acc.y > 0.f ? acc.y : __expf(acc.y) - 1.0f,
                    acc.z > 0.f ? acc.z : __expf(acc.z) - 1.0f, acc.w > 0.f ? acc.w : __expf(acc.w) - 1.0f};
      __builtin_nontemporal_store(e, reinterpret_cast<f4 *>(ye + r * lde + 4 * cg));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ELU: x is the OUTPUT of an ELU (the operand of the BatchNorm) and the result continues through that activation:
//      dx = (dx + (x - center)·B + Cc) · elu'(.) with elu' = 1 (x > 0) | x + 1;  B == NULL: no BatchNorm tail (eval mode).
template <bool VEC, bool ELU>
__global__ __launch_bounds__(kWG) void affine_cols_acc_k(float *__restrict__ dx, int64_t lddx,
                                                         const float *__restrict__ x, int64_t ldx,
                                                         const float *__restrict__ center,
                                                         const float *__restrict__ B, const float *__restrict__ Cc,
                                                         int64_t rows, int C, int nt) {
  constexpr int W = VEC ? 4 : 1;
  const int cw = C / W;
  const int64_t total = rows * cw;
  for (int64_t t = (int64_t)blockIdx.x * kWG + threadIdx.x; t < total; t += (int64_t)gridDim.x * kWG) {
    const int64_t r = t / cw;
    const int c = (int)(t - r * cw) * W;
    if constexpr (VEC) {
      f4 d = ld4_s(dx + r * lddx + c, nt);
      const f4 xr = ld4_s(x + r * ldx + c, nt);
      if (B) {
        f4 xv = xr;
        if (center) xv -= *reinterpret_cast<const f4 *>(center + c);
        const f4 b = *reinterpret_cast<const f4 *>(B + c);
        const f4 k = *reinterpret_cast<const f4 *>(Cc + c);
        d.x += __builtin_fmaf(xv.x, b.x, k.x);
        d.y += __builtin_fmaf(xv.y, b.y, k.y);
        d.z += __builtin_fmaf(xv.z, b.z, k.z);
        d.w += __builtin_fmaf(xv.w, b.w, k.w);
      }
      if constexpr (ELU)
        d = f4{d.x * (xr.x > 0.f ? 1.f : xr.x + 1.f), d.y * (xr.y > 0.f ? 1.f : xr.y + 1.f),
               d.z * (xr.z > 0.f ? 1.f : xr.z + 1.f), d.w * (xr.w > 0.f ? 1.f : xr.w + 1.f)};
      st4_s(dx + r * lddx + c, d, nt);
    } else {
      const float xr = x[r * ldx + c];
      float d = dx[r * lddx + c];
      if (B) d += __builtin_fmaf(xr - (center ? center[c] : 0.f), B[c], Cc[c]);
      if constexpr (ELU) d *= xr > 0.f ? 1.f : xr + 1.f;
      dx[r * lddx + c] = d;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// BatchNorm folding coefficients.  One workgroup per output row j of Wf (plus its bias dot product);
// every workgroup recomputes the C per-channel scalars in double (C <= 1024), workgroup 0 publishes them.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void bn_fold_k(const double *__restrict__ stats, int64_t rows,
                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                 const float *__restrict__ W, const float *__restrict__ b, int C,
                                                 double eps, double momentum, int training,
                                                 float *__restrict__ running_mean, float *__restrict__ running_var,
                                                 float *__restrict__ mean_o, float *__restrict__ invstd_o,
                                                 float *__restrict__ s_o, float *__restrict__ t_o,
                                                 float *__restrict__ Wf, float *__restrict__ bf,
                                                 int64_t *__restrict__ num_batches_tracked,
                                                 const float *__restrict__ seg_m = nullptr /* [nseg][C/2] */, int nseg = 0,
                                                 float *__restrict__ segb = nullptr /* [nseg][J] */, int J = 0) {
  __shared__ float ss[1024], st[1024];
  __shared__ double red[kWG];
  const int j = blockIdx.x;
  if (j == 0 && threadIdx.x == 0 && training && num_batches_tracked) num_batches_tracked[0] += 1;   // nn.BatchNorm1d's counter
  for (int c = threadIdx.x; c < C; c += kWG) {
    double mean, var;
    if (training) {
      mean = stats[c] / (double)rows;
      var = stats[C + c] / (double)rows - mean * mean;
      var = var > 0 ? var : 0;
    } else {
      mean = running_mean[c];
      var = running_var[c];
    }
    const double invstd = 1.0 / sqrt(var + eps);
    const double sc = (double)gamma[c] * invstd;
    const double tc = (double)beta[c] - mean * sc;
    ss[c] = (float)sc;
    st[c] = (float)tc;
    if (j == 0) {
      mean_o[c] = (float)mean;
      invstd_o[c] = (float)invstd;
      s_o[c] = (float)sc;
      t_o[c] = (float)tc;
      if (training && running_mean) {
        const double unbiased = var * ((double)rows / (double)(rows > 1 ? rows - 1 : 1));
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
      }
    }
  }
  __syncthreads();
  double dot = 0;
  for (int c = threadIdx.x; c < C; c += kWG) {
    const float w = W[(int64_t)j * C + c];
    Wf[(int64_t)j * C + c] = w * ss[c];
    dot += (double)w * (double)st[c];
  }
  red[threadIdx.x] = dot;
  __syncthreads();
  for (int o = kWG / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) bf[j] = (float)((b ? (double)b[j] : 0.0) + red[0]);
  if (seg_m) {
    // A global-average stage (sn_bn_fold_seg_f32): the per-mesh bias  segb[g][j] = bf[j] + sum_c m[g][c] Wf[j][C/2 + c]  of MY
    // output row, from the folded row this workgroup has just formed — sn_seg_affine_f32's numbers (same operands, same order)
    // without its launch.  64 meshes at a time through LDS (coalesced), one thread per mesh walks the row.
    const int C2 = C >> 1;                        // (C2 <= 128: checked by the launcher)
    __shared__ float s_w[128], s_mt[64][129];
    __shared__ float s_bf;
    if (threadIdx.x == 0) s_bf = (float)((b ? (double)b[j] : 0.0) + red[0]);
    for (int c = threadIdx.x; c < C2; c += kWG) s_w[c] = W[(int64_t)j * C + C2 + c] * ss[C2 + c];
    for (int g0 = 0; g0 < nseg; g0 += 64) {
      __syncthreads();                            // (s_w, s_bf written; the previous tile read)
      const int ng = nseg - g0 < 64 ? nseg - g0 : 64;
      for (int i = threadIdx.x; i < ng * C2; i += kWG) s_mt[i / C2][i % C2] = seg_m[(int64_t)g0 * C2 + i];
      __syncthreads();
      if ((int)threadIdx.x < ng) {
        double acc = (double)s_bf;
#pragma unroll 16
        for (int c = 0; c < C2; ++c) acc += (double)s_mt[threadIdx.x][c] * (double)s_w[c];
        segb[(int64_t)(g0 + threadIdx.x) * J + j] = (float)acc;
      }
    }
  }
}

// 32 channels x 8 row-groups per workgroup: group g handles output rows j = g, g+8, ...; the two per-channel
// reductions over j are combined across the 8 groups in a fixed order.
__global__ __launch_bounds__(kWG) void bn_bwd_coeffs_k(const float *__restrict__ Gc, const double *__restrict__ sdy,
                                                       const float *__restrict__ W, const float *__restrict__ s,
                                                       const float *__restrict__ invstd, const float *__restrict__ beta,
                                                       int64_t rows, int J, int C, float *__restrict__ dW,
                                                       float *__restrict__ db, float *__restrict__ dgamma,
                                                       float *__restrict__ dbeta, float *__restrict__ Bc,
                                                       float *__restrict__ Cc) {
  __shared__ double sa[8][32], sp[8][32];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  if (blockIdx.x == 0 && db)
    for (int j = threadIdx.x; j < J; j += kWG) db[j] = (float)sdy[j];
  double a = 0, p = 0;
  if (c < C) {
    const double sc = s[c], bc = beta[c];
#pragma unroll 4
    for (int j = g; j < J; j += 8) {
      const double w = W[(int64_t)j * C + c], gg = Gc[(int64_t)j * C + c];
      a += sdy[j] * w;
      p += w * gg;
      dW[(int64_t)j * C + c] = (float)(gg * sc + sdy[j] * bc);
    }
  }
  sa[g][cl] = a;
  sp[g][cl] = p;
  __syncthreads();
  if (g == 0 && c < C) {
    double at = 0, pt = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      at += sa[i][cl];
      pt += sp[i][cl];
    }
    const double sc = s[c], is = invstd[c];
    const double dg = is * pt;
    dgamma[c] = (float)dg;
    dbeta[c] = (float)at;
    Bc[c] = (float)(-(sc * is * dg) / (double)rows);
    Cc[c] = (float)(-(sc * at) / (double)rows);
  }
}

// ------------------------------------------------------------------------------------------------
// per-mesh (segment) masked column sums, broadcast, and ELU backward with a broadcast term
// ------------------------------------------------------------------------------------------------
constexpr int kSegSlabs = 16;      // row slabs per mesh in stage 1

// grid (kSegSlabs, nseg); C % 4 == 0 and (C/4) divides 256
__global__ __launch_bounds__(kWG) void segsum_k(const float *__restrict__ x, int64_t ld, const float *__restrict__ mask,
                                                int64_t rows_per_seg, int C, double *__restrict__ partial) {
  extern __shared__ double sm[];               // [lanes_r][C]
  const int cw = C / 4, lanes_r = kWG / cw;
  const int cg = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int64_t seg = blockIdx.y;
  const int64_t per = (rows_per_seg + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = seg * rows_per_seg + (int64_t)blockIdx.x * per;
  int64_t r1 = r0 + per;
  const int64_t rend = (seg + 1) * rows_per_seg;
  r1 = r1 < rend ? r1 : rend;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int64_t r = r0 + rl; r < r1; r += lanes_r) {
    const float m = mask ? mask[r] : 1.f;
    if (m != 0.f) {
      const f4 v = *reinterpret_cast<const f4 *>(x + r * ld + cg * 4);
      s0 += (double)(m * v.x); s1 += (double)(m * v.y); s2 += (double)(m * v.z); s3 += (double)(m * v.w);
    }
  }
  double *o = sm + (int64_t)rl * C + cg * 4;
  o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += kWG) {
    double t = 0;
    for (int l = 0; l < lanes_r; ++l) t += sm[(int64_t)l * C + i];
    partial[((int64_t)seg * gridDim.x + blockIdx.x) * C + i] = t;
  }
}

__global__ __launch_bounds__(kWG) void segsum_final_k(const double *__restrict__ partial, int nslab, int64_t total,
                                                      int C, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x;       // seg*C + c
  if (i >= total) return;
  const int64_t seg = i / C;
  const int c = (int)(i - seg * C);
  double t = 0;
  for (int sl = 0; sl < nslab; ++sl) t += partial[(seg * nslab + sl) * C + c];
  out[i] = (float)t;
}

// ---- half-width global-average stage: the second half of its concat buffer is a per-mesh constant (the masked mean m
// of the first half, broadcast over the mesh's rows), so everything that concerns that half is nseg x C algebra ----
// m = Ssum * inv_count;  stats (2 x 2C, fp64) = [ stats1 | per * sum_mesh m , per * sum_mesh m^2 ]
__global__ __launch_bounds__(kWG) void avg_fwd_prep_k(const float *__restrict__ ssum, const float *__restrict__ inv_count,
                                                      int nseg, int C, double per, const double *__restrict__ stats1,
                                                      float *__restrict__ m, double *__restrict__ stats) {
  const int c = blockIdx.x * kWG + threadIdx.x;
  if (c >= C) return;
  double s1 = 0, s2 = 0;
#pragma unroll 8                        // (independent loads: keep several in flight — these kernels are latency-, not work-bound)
  for (int g = 0; g < nseg; ++g) {
    const float mv = ssum[(int64_t)g * C + c] * inv_count[g];
    m[(int64_t)g * C + c] = mv;
    s1 += (double)mv;
    s2 += (double)mv * (double)mv;
  }
  stats[c] = stats1[c];
  stats[C + c] = per * s1;
  stats[2 * C + c] = stats1[C + c];
  stats[3 * C + c] = per * s2;
}
// out[g][j] = bias[j] + sum_c A[g][c] * W[j][c]      (W: J rows, leading dimension ldw, already offset to the half used)
__global__ __launch_bounds__(kWG) void seg_affine_k(const float *__restrict__ A, int nseg, int K, const float *__restrict__ W,
                                                    int64_t ldw, const float *__restrict__ bias, int J,
                                                    float *__restrict__ out) {
  const int t = blockIdx.x * kWG + threadIdx.x;
  if (t >= nseg * J) return;
  const int g = t / J, j = t - g * J;
  double acc = bias ? (double)bias[j] : 0.0;
#pragma unroll 16
  for (int c = 0; c < K; ++c) acc += (double)A[(int64_t)g * K + c] * (double)W[(int64_t)j * ldw + c];
  out[t] = (float)acc;
}
// Gc (J x 2C): [ G1 | sum_mesh Sg[mesh][j] * (m[mesh][c] - mu2[c]) ]
__global__ __launch_bounds__(kWG) void avg_bwd_gc_k(const float *__restrict__ G1, const float *__restrict__ Sg,
                                                    const float *__restrict__ m, const float *__restrict__ mu2, int nseg,
                                                    int J, int C, float *__restrict__ Gc) {
  const int t = blockIdx.x * kWG + threadIdx.x;
  if (t >= J * 2 * C) return;
  const int j = t / (2 * C), c2 = t - j * 2 * C;
  if (c2 < C) {
    Gc[t] = G1[(int64_t)j * C + c2];
    return;
  }
  const int c = c2 - C;
  double acc = 0;
#pragma unroll 16
  for (int g = 0; g < nseg; ++g) acc += (double)Sg[(int64_t)g * J + j] * ((double)m[(int64_t)g * C + c] - (double)mu2[c]);
  Gc[t] = (float)acc;
}
// segvec[mesh][c] = inv_count[mesh] * ( sum_j Sg[mesh][j] * Wf2[j][c] + per * ((m[mesh][c] - mu2[c]) * B2[c] + C2[c]) )
__global__ __launch_bounds__(kWG) void avg_bwd_segvec_k(const float *__restrict__ Sg, const float *__restrict__ Wf2,
                                                        int64_t ldw, const float *__restrict__ m,
                                                        const float *__restrict__ mu2, const float *__restrict__ B2,
                                                        const float *__restrict__ C2, const float *__restrict__ inv_count,
                                                        double per, int nseg, int J, int C, float *__restrict__ out,
                                                        const int64_t *__restrict__ segoff /* ragged: rows of mesh g | NULL */) {
  const int t = blockIdx.x * kWG + threadIdx.x;
  if (t >= nseg * C) return;
  const int g = t / C, c = t - g * C;
  if (segoff) per = (double)(segoff[g + 1] - segoff[g]);
  double acc = 0;
#pragma unroll 16
  for (int j = 0; j < J; ++j) acc += (double)Sg[(int64_t)g * J + j] * (double)Wf2[(int64_t)j * ldw + c];
  acc += per * (((double)m[t] - (double)mu2[c]) * (double)B2[c] + (double)C2[c]);
  out[t] = (float)(acc * (double)inv_count[g]);
}

// A global-average stage's  avg_bwd_gc_k -> bn_bwd_coeffs_k -> avg_bwd_segvec_k  in ONE launch (sn_avg_bn_bwd_f32): each of the
// three is per channel — the broadcast half of G, the BatchNorm sums over j, the per-mesh vector of the mean path —, so the
// workgroup that owns 32 channels runs all three for them; nothing passes between workgroups (no ticket, no fence).  Same
// operands, same order of every sum as the three kernels: bit-identical (tests/test_dense_gpu.py).  Grid: the first C / 32
// workgroups are bn_bwd_coeffs_k over G1; then (C / 32) x nchunk workgroups for the broadcast half: workgroup (channel block, mesh
// chunk q) forms its 32 columns of the broadcast half of G (all meshes, 64 at a time through LDS) and the coefficients —
// every chunk redundantly (a few thousand FMAs), chunk 0 writes them — and then the per-mesh vector of ITS meshes
// [q mpc, (q + 1) mpc): the 128-term dot products per (mesh, channel) are what the launch costs, eight meshes in flight per
// workgroup; with one workgroup per channel block (round 4) 64 meshes were eight serial rounds and the merged launch lost to
// the three it replaces beyond 8 meshes.
__global__ __launch_bounds__(kWG) void avg_bn_bwd_k(const float *__restrict__ G1, const double *__restrict__ sdy,
                                                    const float *__restrict__ Sg, const float *__restrict__ m,
                                                    const float *__restrict__ mu2v, const float *__restrict__ W,
                                                    const float *__restrict__ s, const float *__restrict__ invstd,
                                                    const float *__restrict__ beta, int64_t rows, int J, int C, int nseg,
                                                    const float *__restrict__ Wf2, int64_t ldw, const float *__restrict__ inv_count,
                                                    double per, const int64_t *__restrict__ segoff, int mpc /* meshes per chunk */, float *__restrict__ dW,
                                                    float *__restrict__ db, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                    float *__restrict__ Bc, float *__restrict__ Cc, float *__restrict__ segvec) {
  __shared__ double sa[8][32], sp[8][32];
  // per-mesh column sums of dy, 64 meshes at a time, columns PERMUTED: column j sits at (j % 8) * 16 + j / 8, so that the 16
  // rows j = gq, gq + 8, ... a lane group owns are 16 consecutive floats (four 16-byte LDS reads per mesh instead of sixteen)
  __shared__ __attribute__((aligned(16))) float s_sg[64][132];
  __shared__ float s_m[64][33];                  // per-mesh means of my 32 channels, the same meshes
  __shared__ float s_wf[128][33];                // my 32 columns of Wf2
  __shared__ float s_b[32], s_c[32];
  const int tid = threadIdx.x, cl = tid & 31, gq = tid >> 5;
  const int Ct = 2 * C;
  // ng meshes' column sums of dy (ng x J floats from Sg row g0 on) into s_sg: eight loads in flight per thread, then the stores —
  // issued one by one (a load, its index arithmetic, its LDS store per iteration) the 8 192 values of 64 meshes were 32
  // dependent round trips to memory: 30 of the launch's 57 us at 64 meshes
  auto stage_sg = [&](int g0, int ng) {
    const int n = ng * J;
    const float *src = Sg + (int64_t)g0 * J;
    for (int base = 0; base < n; base += 8 * kWG) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * kWG + tid;
        v[u] = i < n ? src[i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * kWG + tid;
        if (i < n) {
          const int j = i % J;
          s_sg[i / J][(j & 7) * 16 + (j >> 3)] = v[u];
        }
      }
    }
  };
  const int nfirst = C / 32;                     // (C % 32 == 0)
  const bool first = (int)blockIdx.x < nfirst;   // (workgroup-uniform)
  const int cb = first ? (int)blockIdx.x : ((int)blockIdx.x - nfirst) % nfirst;     // 32-channel block inside my half
  const int q = first ? 0 : ((int)blockIdx.x - nfirst) / nfirst;                    // my mesh chunk
  const int c2 = cb * 32 + cl;                   // channel inside the half
  const int c = (first ? 0 : C) + c2;            // < Ct
  const bool writer = first || q == 0;           // (the chunks of one channel block all hold the same coefficients)
  if (blockIdx.x == 0 && db)
    for (int j = tid; j < J; j += kWG) db[j] = (float)sdy[j];
  // Everything that depends on nothing is requested FIRST — the coefficient operands of my 16 rows (W, colsum(dy)) and, for the
  // broadcast half, my 32 columns of Wf2 —: the launch is a chain of load -> LDS -> barrier round trips (2-3 us each), and
  // these used to start only after the meshes had been walked.
  float wv[16];
  double sd[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {                 // (row index clamped, not branched on: a branch per row serialises the loads)
    const int j = gq + 8 * i, jj = j < J ? j : J - 1;
    wv[i] = W[(int64_t)jj * Ct + c];
    sd[i] = sdy[jj];
  }
  if (!first)
    for (int i = tid; i < J * 32; i += kWG) s_wf[i >> 5][i & 31] = Wf2[(int64_t)(i >> 5) * ldw + cb * 32 + (i & 31)];
  float gv[16];                                  // G of my channel in rows gq, gq + 8, ... (J <= 128)
  if (first) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {               // (row index clamped, not branched on: sixteen loads in flight)
      const int j = gq + 8 * i;
      gv[i] = G1[(int64_t)(j < J ? j : J - 1) * C + c];
    }
  } else {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    const double mu = (double)mu2v[c2];
    for (int g0 = 0; g0 < nseg; g0 += 64) {
      const int ng = nseg - g0 < 64 ? nseg - g0 : 64;
      __syncthreads();
      stage_sg(g0, ng);
      for (int i = tid; i < ng * 32; i += kWG) s_m[i >> 5][i & 31] = m[(int64_t)(g0 + (i >> 5)) * C + cb * 32 + (i & 31)];
      __syncthreads();
      for (int k = 0; k < ng; ++k) {             // meshes in ascending order: avg_bwd_gc_k's sum
        const double d = (double)s_m[k][cl] - mu;
        // all 16 rows unconditionally (rows >= J read staging space nobody wrote and feed accumulators nobody reads): a
        // condition per row made every LDS read wait for the previous one's branch — 43 of the launch's 57 us at 64 meshes
        const f4 *r = reinterpret_cast<const f4 *>(&s_sg[k][gq * 16]);
        const f4 v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3];
        const float sv[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += (double)sv[i] * d;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) gv[i] = (float)acc[i];
  }
  // ---- bn_bwd_coeffs_k for my 32 channels ----
  double a = 0, p = 0;
  {
    const double sc = s[c], bc = beta[c];
    // (operands loaded above; the sums in bn_bwd_coeffs_k's order)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int j = gq + 8 * i;
      if (j < J) {
        const double w = wv[i], gg = gv[i];
        a += sd[i] * w;
        p += w * gg;
        if (writer) dW[(int64_t)j * Ct + c] = (float)(gg * sc + sd[i] * bc);
      }
    }
  }
  sa[gq][cl] = a;
  sp[gq][cl] = p;
  __syncthreads();
  if (gq == 0) {
    double at = 0, pt = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      at += sa[i][cl];
      pt += sp[i][cl];
    }
    const double sc = s[c], is = invstd[c];
    const double dg = is * pt;
    const float bq = (float)(-(sc * is * dg) / (double)rows), cq = (float)(-(sc * at) / (double)rows);
    if (writer) {
      dgamma[c] = (float)dg;
      dbeta[c] = (float)at;
      Bc[c] = bq;
      Cc[c] = cq;
    }
    s_b[cl] = bq;
    s_c[cl] = cq;
  }
  if (first) return;
  // ---- avg_bwd_segvec_k for my 32 channels: segvec[g][c2] = inv_count[g] (sum_j Sg[g][j] Wf2[j][c2] + per ((m - mu2) B2 + C2)) ----
  const double mu = (double)mu2v[c2];
  const int g_lo = q * mpc, g_hi = g_lo + mpc < nseg ? g_lo + mpc : nseg;        // my meshes
  const bool resident = nseg <= 64;              // the one staging round above holds every mesh: rows g_lo .. of s_sg / s_m are mine
  for (int g0 = g_lo; g0 < g_hi; g0 += 64) {
    const int ng = g_hi - g0 < 64 ? g_hi - g0 : 64;
    __syncthreads();                             // (s_b, s_c, s_wf written; the tiles of the previous round read)
    if (!resident) {
      stage_sg(g0, ng);
      for (int i = tid; i < ng * 32; i += kWG) s_m[i >> 5][i & 31] = m[(int64_t)(g0 + (i >> 5)) * C + cb * 32 + (i & 31)];
      __syncthreads();
    }
    const int r0 = resident ? g0 : 0;            // staging row of mesh g0
    for (int kk = gq; kk < ng; kk += 8) {
      const int g = g0 + kk, k = r0 + kk;
      double acc = 0;
#pragma unroll 16
      for (int j = 0; j < J; ++j) acc += (double)s_sg[k][(j & 7) * 16 + (j >> 3)] * (double)s_wf[j][cl];
      const double pr = segoff ? (double)(segoff[g + 1] - segoff[g]) : per;
      acc += pr * (((double)s_m[k][cl] - mu) * (double)s_b[cl] + (double)s_c[cl]);
      segvec[(int64_t)g * C + c2] = (float)(acc * (double)inv_count[g]);
    }
  }
}

// The same quantities WITHOUT a pass over e, when the GEMM that wrote e left (a) the column sums of every 32-row tile
// (EpiArgs::tile_sums, sn_linear_fwd_tiles_f32) and (b) its per-workgroup column sums / sums of squares (the statistics
// partials).  Mesh g owns rows [g·per, (g+1)·per): the tiles that lie entirely inside it contribute their stored sums, the
// rows of the (at most two) tiles it shares with its neighbours are read from e; with a row mask every tile's 32 mask values
// are read first and a tile that holds a masked-out row is summed from e as well (prefix masks: one tile per mesh).
// grid (C / 32, nseg), 256 threads = 8 four-column chunks x 32 tile lanes; ssum[nseg][C] fp32 (fp64 across tiles).
// segoff != NULL: ragged meshes — mesh g owns rows [segoff[g], segoff[g+1]) (no mask).
__global__ __launch_bounds__(kWG) void segsum_tiles_k(const float *__restrict__ tile_sums, const float *__restrict__ e, int64_t lde,
                                                      const float *__restrict__ mask, int64_t per, int C,
                                                      float *__restrict__ ssum, const int64_t *__restrict__ segoff = nullptr) {
  __shared__ double sm[32][8][4];
  const int chunk = threadIdx.x & 7, tl = threadIdx.x >> 3;
  const int c = blockIdx.x * 32 + 4 * chunk;
  const int64_t g = blockIdx.y;
  const int64_t a = segoff ? segoff[g] : g * per, b = segoff ? segoff[g + 1] : a + per;      // rows of the mesh
  const int64_t t0 = (a + 31) / 32, t1 = b / 32;                 // whole tiles: [t0, t1)
  double s[4] = {0, 0, 0, 0};
  auto rows_from_e = [&](int64_t r0, int64_t r1) {               // masked sums of rows [r0, r1) of e, my 4 columns
    for (int64_t r = r0; r < r1; ++r) {
      const float mk = mask ? mask[r] : 1.f;
      if (mk != 0.f) {
        const f4 v = *reinterpret_cast<const f4 *>(e + r * lde + c);
        s[0] += (double)(mk * v.x); s[1] += (double)(mk * v.y); s[2] += (double)(mk * v.z); s[3] += (double)(mk * v.w);
      }
    }
  };
  if (t0 >= t1) {                                                // (a mesh shorter than a tile)
    if (tl == 0) rows_from_e(a, b);
  } else {
    const int64_t nt = t1 - t0;
    for (int64_t i = 0; i < (nt + 31) / 32 * 32; i += 32) {      // (uniform trip count: the vote below needs the whole wave)
      const int64_t t = t0 + i + tl;
      bool clean = true;
      // the tile's stored sums are requested BEFORE its mask values are known (used only for a clean tile): asked for after
      // the vote, every round of 32 tiles was two dependent memory round trips
      f4 tv = {0.f, 0.f, 0.f, 0.f};
      if (t < t1) tv = *reinterpret_cast<const f4 *>(tile_sums + t * 128 + c);
      if (mask) {
        // the tile's 32 mask values: one 16-byte piece per chunk lane, combined by a vote among the 8 lanes of the tile
        bool mine = true;
        if (t < t1) {
          const f4 mv = *reinterpret_cast<const f4 *>(mask + 32 * t + 4 * chunk);      // (32 t: 128-byte aligned)
          mine = mv.x == 1.f && mv.y == 1.f && mv.z == 1.f && mv.w == 1.f;
        }
        const unsigned long long votes = __ballot(mine);
        clean = ((votes >> (8 * ((threadIdx.x & 63) >> 3))) & 0xffull) == 0xffull;
      }
      if (t >= t1) continue;
      if (clean) {
        s[0] += (double)tv.x; s[1] += (double)tv.y; s[2] += (double)tv.z; s[3] += (double)tv.w;
      } else {
        rows_from_e(32 * t, 32 * t + 32);                        // (a tile with masked-out rows: one per mesh with prefix masks)
      }
    }
    // the rows before the first / after the last whole tile (fewer than 32 each): one row per tile lane, all in flight at once
    if (a + tl < 32 * t0) rows_from_e(a + tl, a + tl + 1);
    if (32 * t1 + tl < b) rows_from_e(32 * t1 + tl, 32 * t1 + tl + 1);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sm[tl][chunk][k] = s[k];
  __syncthreads();
  if (threadIdx.x < 32) {
    const int ch = threadIdx.x >> 2, k = threadIdx.x & 3;
    double t = 0;
#pragma unroll 8
    for (int l = 0; l < 32; ++l) t += sm[l][ch][k];
    ssum[g * C + blockIdx.x * 32 + 4 * ch + k] = (float)t;
  }
}
// m and the BatchNorm statistics of [e | m broadcast] from the per-mesh sums and the producer's statistics partials
// (part[nblk][2][C] fp64): avg_fwd_prep_k with the merge of the partials in front.  8 columns x 32 lanes per workgroup: the
// lanes share the partial blocks and the meshes (a few dozen independent loads each), fixed-order sums through LDS.
// segoff != NULL (ragged meshes, sn_avg_prep_ragged_f32): `ssum` holds the per-mesh MEANS already (nothing is written to m) and
// mesh g counts segoff[g+1] - segoff[g] rows in the statistics of the broadcast half.
__global__ __launch_bounds__(kWG) void avg_prep_parts_k(const float *__restrict__ ssum, const float *__restrict__ inv_count, int nseg,
                                                        int C, double per, const double *__restrict__ part, int nblk,
                                                        float *__restrict__ m, double *__restrict__ stats,
                                                        const int64_t *__restrict__ segoff = nullptr) {
  __shared__ double sm[4][32][8];
  const int cl = threadIdx.x & 7, ln = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double u = 0, q = 0, s1 = 0, s2 = 0;
#pragma unroll 4
  for (int b = ln; b < nblk; b += 32) {
    u += part[(int64_t)b * 2 * C + c];
    q += part[(int64_t)b * 2 * C + C + c];
  }
  if (segoff) {
    per = 1.0;
#pragma unroll 2
    for (int g = ln; g < nseg; g += 32) {
      float mf = ssum[(int64_t)g * C + c];
      if (inv_count) mf *= inv_count[g];          // (sums in, means out: sn_avg_stats_from_tiles_ragged_f32)
      if (m) m[(int64_t)g * C + c] = mf;
      const double mv = (double)mf, len = (double)(segoff[g + 1] - segoff[g]);
      s1 += len * mv;
      s2 += len * mv * mv;
    }
  } else
#pragma unroll 2
  for (int g = ln; g < nseg; g += 32) {
    const float mv = ssum[(int64_t)g * C + c] * inv_count[g];
    m[(int64_t)g * C + c] = mv;
    s1 += (double)mv;
    s2 += (double)mv * (double)mv;
  }
  sm[0][ln][cl] = u; sm[1][ln][cl] = q; sm[2][ln][cl] = s1; sm[3][ln][cl] = s2;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int what = threadIdx.x >> 3, col = threadIdx.x & 7;
    double t = 0;
#pragma unroll 8
    for (int l = 0; l < 32; ++l) t += sm[what][l][col];
    const int cc = blockIdx.x * 8 + col;
    // stats layout (2 x 2C): [ sum e | per sum m ; sum e^2 | per sum m^2 ]
    if (what == 0) stats[cc] = t;
    else if (what == 1) stats[2 * C + cc] = t;
    else if (what == 2) stats[C + cc] = per * t;
    else stats[3 * C + cc] = per * t;
  }
}

// One pass over e for everything the half-width global-average stage needs from it: per-mesh MASKED column sums (-> the
// mean m), and the unmasked column sums / sums of squares of all rows (-> BatchNorm statistics of the first half).
// Stage 1: grid (kSegSlabs, nseg), partial[mesh][slab][3][C] fp64 (masked sum | sum | sum of squares).
__global__ __launch_bounds__(kWG) void segstats_k(const float *__restrict__ x, int64_t ld, const float *__restrict__ mask,
                                                  int64_t rows_per_seg, int C, double *__restrict__ partial) {
  extern __shared__ double sm[];               // [lanes_r][3C]
  const int cw = C / 4, lanes_r = kWG / cw;
  const int cg = threadIdx.x % cw, rl = threadIdx.x / cw;
  const int64_t seg = blockIdx.y;
  const int64_t per = (rows_per_seg + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = seg * rows_per_seg + (int64_t)blockIdx.x * per;
  int64_t r1 = r0 + per;
  const int64_t rend = (seg + 1) * rows_per_seg;
  r1 = r1 < rend ? r1 : rend;
  double s[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
#pragma unroll 4
  for (int64_t r = r0 + rl; r < r1; r += lanes_r) {
    const float mk = mask ? mask[r] : 1.f;
    const f4 v = ld4_stat(x + r * ld + cg * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double d = (double)v[k];
      s[k] += (double)(mk * v[k]);
      u[k] += d;
      q[k] += d * d;
    }
  }
  double *o = sm + (int64_t)rl * 3 * C + cg * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[k] = s[k];
    o[C + k] = u[k];
    o[2 * C + k] = q[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += kWG) {
    double t = 0;
    for (int l = 0; l < lanes_r; ++l) t += sm[(int64_t)l * 3 * C + i];
    partial[((int64_t)seg * gridDim.x + blockIdx.x) * 3 * C + i] = t;
  }
}
// Stage 2: m = masked sum * inv_count, and
// stats (2 x 2C fp64) = [ sum | per * sum_mesh m ;  sum of squares | per * sum_mesh m^2 ]  (the layout sn_bn_fold_f32 reads)
// 256 threads = 8 columns x 32 mesh lanes, C/8 workgroups: the 3 MB of partials are pulled by 16 CUs instead of 4 and a
// thread handles 2 instead of 8 meshes in sequence (48 loads in flight each) — this tiny kernel is bound by both.
constexpr int kSegFinalLanes = 32, kSegFinalCols = 8;
constexpr int kSegFewMeshes = 4, kSegSlabsFew = 128;      // few meshes: more slabs each, so that stage 1 still fills the chip
inline int seg_slabs(int64_t nseg) { return nseg <= kSegFewMeshes ? kSegSlabsFew : kSegSlabs; }
__global__ __launch_bounds__(kSegFinalCols * kSegFinalLanes) void segstats_final_k(const double *__restrict__ partial, int nslab,
                                                                                   int nseg, int C,
                                                                                   const float *__restrict__ inv_count,
                                                                                   double per, float *__restrict__ m,
                                                                                   double *__restrict__ stats) {
  __shared__ double sm[4][kSegFinalLanes][kSegFinalCols];
  const int cl = threadIdx.x % kSegFinalCols, gg = threadIdx.x / kSegFinalCols;
  const int c = blockIdx.x * kSegFinalCols + cl;
  double U = 0, Q = 0, S1 = 0, S2 = 0;
  if (nseg <= kSegFewMeshes) {
    // a handful of meshes (a FAUST pair is two launches of one): the 32 lanes of a column split a mesh's slabs instead of
    // the meshes; the mesh's totals are combined by lane 0, which then carries U, Q, S1, S2 alone
    for (int g = 0; g < nseg; ++g) {
      double ms = 0, u = 0, q = 0;
      if (c < C) {
        const double *p = partial + (int64_t)g * nslab * 3 * C + c;
        for (int sl = gg; sl < nslab; sl += kSegFinalLanes) {
          ms += p[(int64_t)sl * 3 * C];
          u += p[(int64_t)sl * 3 * C + C];
          q += p[(int64_t)sl * 3 * C + 2 * C];
        }
      }
      sm[0][gg][cl] = ms; sm[1][gg][cl] = u; sm[2][gg][cl] = q;
      __syncthreads();
      if (gg == 0 && c < C) {
        double t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
        for (int l = 0; l < kSegFinalLanes; ++l) t0 += sm[0][l][cl], t1 += sm[1][l][cl], t2 += sm[2][l][cl];
        const float mv = (float)t0 * inv_count[g];
        m[(int64_t)g * C + c] = mv;
        U += t1;
        Q += t2;
        S1 += (double)mv;
        S2 += (double)mv * (double)mv;
      }
      __syncthreads();
    }
  } else if (c < C)
    for (int g = gg; g < nseg; g += kSegFinalLanes) {
      double ms = 0;
      const double *p = partial + (int64_t)g * nslab * 3 * C + c;
      if (nslab == kSegSlabs) {                  // the launch below always uses kSegSlabs: all 48 loads of a mesh in flight
        double v[kSegSlabs][3];
#pragma unroll
        for (int sl = 0; sl < kSegSlabs; ++sl) {
          v[sl][0] = p[(int64_t)sl * 3 * C];
          v[sl][1] = p[(int64_t)sl * 3 * C + C];
          v[sl][2] = p[(int64_t)sl * 3 * C + 2 * C];
        }
#pragma unroll
        for (int sl = 0; sl < kSegSlabs; ++sl) {
          ms += v[sl][0];
          U += v[sl][1];
          Q += v[sl][2];
        }
      } else {
#pragma unroll 8
        for (int sl = 0; sl < nslab; ++sl) {
          ms += p[(int64_t)sl * 3 * C];
          U += p[(int64_t)sl * 3 * C + C];
          Q += p[(int64_t)sl * 3 * C + 2 * C];
        }
      }
      const float mv = (float)ms * inv_count[g];
      m[(int64_t)g * C + c] = mv;
      S1 += (double)mv;
      S2 += (double)mv * (double)mv;
    }
  sm[0][gg][cl] = U; sm[1][gg][cl] = Q; sm[2][gg][cl] = S1; sm[3][gg][cl] = S2;
  __syncthreads();
  if (gg == 0 && c < C) {
    double t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      t[k] = 0;
#pragma unroll
      for (int l = 0; l < kSegFinalLanes; ++l) t[k] += sm[k][l][cl];
    }
    stats[c] = t[0];
    stats[C + c] = per * t[2];
    stats[2 * C + c] = t[1];
    stats[3 * C + c] = per * t[3];
  }
}

__global__ __launch_bounds__(kWG) void bcast_rows_k(const float *__restrict__ src, float *__restrict__ dst, int64_t ldd,
                                                    int64_t rows_per_seg, int64_t rows, int C, int nt) {
  const int cw = C / 4;
  const int64_t total = rows * cw;
  for (int64_t t = (int64_t)blockIdx.x * kWG + threadIdx.x; t < total; t += (int64_t)gridDim.x * kWG) {
    const int64_t r = t / cw;
    const int c = (int)(t - r * cw) * 4;
    st4_s(dst + r * ldd + c, *reinterpret_cast<const f4 *>(src + (r / rows_per_seg) * C + c), nt);
  }
}

__global__ __launch_bounds__(kWG) void elu_bwd_bcast_k(const float *__restrict__ gdst, int64_t ldg,
                                                       const float *__restrict__ out, int64_t ldo,
                                                       const float *__restrict__ bias, const float *__restrict__ mask,
                                                       const float *__restrict__ gadd, int64_t ldga,
                                                       float *__restrict__ gsrc, int64_t ldgs, int64_t rows_per_seg,
                                                       int64_t rows, int C, int nt) {
  const int cw = C / 4;
  const int64_t total = rows * cw;
  for (int64_t t = (int64_t)blockIdx.x * kWG + threadIdx.x; t < total; t += (int64_t)gridDim.x * kWG) {
    const int64_t r = t / cw;
    const int c = (int)(t - r * cw) * 4;
    const f4 g = ld4_s(gdst + r * ldg + c, nt);
    const f4 o = ld4_s(out + r * ldo + c, nt);
    const f4 b = *reinterpret_cast<const f4 *>(bias + (r / rows_per_seg) * C + c);
    const float m = mask ? mask[r] : 1.f;
    f4 d;
    d.x = __builtin_fmaf(m, b.x, g.x) * (o.x > 0.f ? 1.f : o.x + 1.f);
    d.y = __builtin_fmaf(m, b.y, g.y) * (o.y > 0.f ? 1.f : o.y + 1.f);
    d.z = __builtin_fmaf(m, b.z, g.z) * (o.z > 0.f ? 1.f : o.z + 1.f);
    d.w = __builtin_fmaf(m, b.w, g.w) * (o.w > 0.f ? 1.f : o.w + 1.f);
    if (gadd) d += ld4_s(gadd + r * ldga + c, nt);
    st4_s(gsrc + r * ldgs + c, d, nt);
  }
}

inline int stat_blocks(int64_t rows) {
  int64_t b = (rows + 63) / 64;
  if (b > kStatBlocks) b = kStatBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

// SN_GEMM_VARIANT (shared with sn_gemm.hip): 0 = the fp32-MFMA kernels, the ONE A/B baseline of the Linear kernels; anything
// else (default) = the 16-bit matrix-pipe kernels
inline int gemm_variant() {
  static const int v = [] {
    const char *e = getenv("SN_GEMM_VARIANT");
    return (e && atoi(e) == 0) ? 0 : 2;
  }();
  return v;
}

// ------------------------------------------------------------------------------------------------
// Target of the dense-correspondence loss:  t[r] = argmin_j ( GA[r][pa[j]] + GB[pb[r]][j] )  — the torch.min over the
// sum of two gathered (NA x NB) geodesic matrices of src/dense_correspondence/main.py:236-237, without materialising
// them (3 x 190 MB at 6890 vertices).  A workgroup takes kArgRows rows of the result: pa[j] is read once for the four,
// row pb[r] of GB streams, row r of GA (27 KB) is gathered through L1/L2.  Ties go to the lowest j and a NaN wins, as
// in torch.min.
// ------------------------------------------------------------------------------------------------
constexpr int kArgRows = 4;
__device__ inline bool arg_better(float av, int aj, float bv, int bj) {
  const bool an = av != av, bn = bv != bv;
  if (an != bn) return an;
  if (!an && av != bv) return av < bv;
  return aj < bj;
}
__global__ __launch_bounds__(kWG) void pair_argmin_k(const float *__restrict__ GA, int64_t ldA, const int64_t *__restrict__ pa,
                                                     const float *__restrict__ GB, int64_t ldB, const int64_t *__restrict__ pb,
                                                     int NA, int NB, int64_t *__restrict__ out) {
  __shared__ float sv[kArgRows][kWG / 64];
  __shared__ int sj[kArgRows][kWG / 64];
  const int r0 = blockIdx.x * kArgRows;
  const float *ga[kArgRows], *gb[kArgRows];
#pragma unroll
  for (int q = 0; q < kArgRows; ++q) {
    const int r = r0 + q < NA ? r0 + q : NA - 1;
    ga[q] = GA + (int64_t)r * ldA;
    gb[q] = GB + pb[r] * ldB;
  }
  float bv[kArgRows];
  int bj[kArgRows];
#pragma unroll
  for (int q = 0; q < kArgRows; ++q) bv[q] = __builtin_inff(), bj[q] = INT_MAX;
#pragma unroll 2
  for (int j = threadIdx.x; j < NB; j += kWG) {
    const int64_t pj = pa[j];
#pragma unroll
    for (int q = 0; q < kArgRows; ++q) {
      const float v = ga[q][pj] + gb[q][j];
      if (arg_better(v, j, bv[q], bj[q])) bv[q] = v, bj[q] = j;
    }
  }
#pragma unroll
  for (int q = 0; q < kArgRows; ++q) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float ov = __shfl_xor(bv[q], d);
      const int oj = __shfl_xor(bj[q], d);
      if (arg_better(ov, oj, bv[q], bj[q])) bv[q] = ov, bj[q] = oj;
    }
    if ((threadIdx.x & 63) == 0) sv[q][threadIdx.x >> 6] = bv[q], sj[q][threadIdx.x >> 6] = bj[q];
  }
  __syncthreads();
  if (threadIdx.x < kArgRows && r0 + (int)threadIdx.x < NA) {
    const int q = threadIdx.x;
    float v = sv[q][0];
    int j = sj[q][0];
    for (int w = 1; w < kWG / 64; ++w)
      if (arg_better(sv[q][w], sj[q][w], v, j)) v = sv[q][w], j = sj[q][w];
    out[r0 + q] = j;
  }
}

// The same with row r of GA staged in LDS (one coalesced pass) and gathered from there: a gather straight from global
// memory moves a whole cache line per 4-byte element once the rows of a CU's workgroups have pushed each other out of L1
// (measured: 550 us for 6890 x 6890, against ~75 us of HBM time for the two matrices).  One row per workgroup, 27 KB of LDS
// at 6890 columns: five workgroups per CU overlap each other's staging and gathering.
constexpr size_t kArgLdsMax = 60 * 1024;           // row image; with the shift slack and the static arrays under the 64 KB a launch gets by default
__global__ __launch_bounds__(kWG) void pair_argmin_lds_k(const float *__restrict__ GA, int64_t ldA, int colsA,
                                                         const int64_t *__restrict__ pa, const float *__restrict__ GB, int64_t ldB,
                                                         const int64_t *__restrict__ pb, int NB, int64_t *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float rowA[];
  __shared__ float sv[kWG / 64];
  __shared__ int sj[kWG / 64];
  const int r = blockIdx.x;
  const float *ga = GA + (int64_t)r * ldA;
  const float *gb = GB + pb[r] * ldB;
  // quads aligned in global memory; the LDS image is shifted by the row's misalignment so that both sides are 16-byte aligned
  const int sh = (int)((reinterpret_cast<uintptr_t>(ga) >> 2) & 3);
  for (int c = 4 * (int)threadIdx.x - sh; c < colsA; c += 4 * kWG) {
    if (c >= 0 && c + 3 < colsA) {
      *reinterpret_cast<f4 *>(rowA + sh + c) = *reinterpret_cast<const f4 *>(ga + c);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e >= 0 && c + e < colsA) rowA[sh + c + e] = ga[c + e];
    }
  }
  __syncthreads();
  float bv = __builtin_inff();
  int bj = INT_MAX;
#pragma unroll 4
  for (int j = threadIdx.x; j < NB; j += kWG) {
    const float v = rowA[sh + pa[j]] + gb[j];
    if (arg_better(v, j, bv, bj)) bv = v, bj = j;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const float ov = __shfl_xor(bv, d);
    const int oj = __shfl_xor(bj, d);
    if (arg_better(ov, oj, bv, bj)) bv = ov, bj = oj;
  }
  if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = bv, sj[threadIdx.x >> 6] = bj;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kWG / 64; ++w)
      if (arg_better(sv[w], sj[w], bv, bj)) bv = sv[w], bj = sj[w];
    out[r] = bj;
  }
}

// ------------------------------------------------------------------------------------------------
// Cross entropy of the dense-correspondence loss (src/dense_correspondence/main.py:238-239:
// F.cross_entropy(outputs[0, :NA, :NB], GAB), mean over the NA rows) on the (N x N) score matrix.  torch runs
// log_softmax (read + write 196 MB at N = 7000), nll_loss, their two backward passes and the zero-padding of the slice's
// gradient; here the forward reads the scores once (row maximum and sum from registers) and keeps one log-sum-exp per row,
// the backward reads them once more and writes the gradient of the WHOLE padded matrix in the same pass.
// One workgroup per row; a row of up to kCeRegs x 1024 columns lives in registers between the two reductions.
// ------------------------------------------------------------------------------------------------
constexpr int kCeRegs = 8;                   // float4 per thread held between the max and the sum pass (8192 columns)
__device__ inline float wg_reduce_max(float v, float *sm) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  __syncthreads();
  return v;
}
__device__ inline float wg_reduce_sum(float v, float *sm) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return v;
}
template <bool VEC>
__global__ __launch_bounds__(kWG) void pair_ce_fwd_k(const float *__restrict__ S, int64_t ld, const int64_t *__restrict__ target,
                                                     int NB, float *__restrict__ lse, float *__restrict__ rowloss) {
  __shared__ float sm[kWG / 64];
  const float *row = S + (int64_t)blockIdx.x * ld;
  float mx = -__builtin_inff(), sum = 0.f;
  if (VEC && NB <= kCeRegs * 4 * kWG) {
    f4 v[kCeRegs];
#pragma unroll
    for (int u = 0; u < kCeRegs; ++u) {
      const int c = 4 * (threadIdx.x + u * kWG);
      if (c + 3 < NB) {
        v[u] = *reinterpret_cast<const f4 *>(row + c);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = c + e < NB ? row[c + e] : -__builtin_inff();
      }
      mx = fmaxf(fmaxf(mx, fmaxf(v[u][0], v[u][1])), fmaxf(v[u][2], v[u][3]));
    }
    mx = wg_reduce_max(mx, sm);
#pragma unroll
    for (int u = 0; u < kCeRegs; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += expf(v[u][e] - mx);        // exp(-inf) = 0 for the columns past NB
  } else {
    for (int c = threadIdx.x; c < NB; c += kWG) mx = fmaxf(mx, row[c]);
    mx = wg_reduce_max(mx, sm);
    for (int c = threadIdx.x; c < NB; c += kWG) sum += expf(row[c] - mx);
  }
  sum = wg_reduce_sum(sum, sm);
  if (threadIdx.x == 0) {
    const float l = mx + logf(sum);
    lse[blockIdx.x] = l;
    rowloss[blockIdx.x] = l - row[target[blockIdx.x]];
  }
}
// dS[r][j] = gs * (exp(S[r][j] - lse[r]) - [j == target[r]]) inside NA x NB, 0 in the padding; gs = *gloss / NA
template <bool VEC>
__global__ __launch_bounds__(kWG) void pair_ce_bwd_k(const float *__restrict__ S, int64_t ld, const int64_t *__restrict__ target,
                                                     const float *__restrict__ lse, const float *__restrict__ gloss, int NA,
                                                     int NB, int cols, float *__restrict__ dS, int64_t ldd) {
  const int r = blockIdx.x;
  float *out = dS + (int64_t)r * ldd;
  if (r >= NA) {
    if (VEC) {
      for (int c = 4 * threadIdx.x; c < cols; c += 4 * kWG) {
        if (c + 3 < cols) *reinterpret_cast<f4 *>(out + c) = f4{0.f, 0.f, 0.f, 0.f};
        else for (int e = 0; c + e < cols; ++e) out[c + e] = 0.f;
      }
    } else {
      for (int c = threadIdx.x; c < cols; c += kWG) out[c] = 0.f;
    }
    return;
  }
  const float *row = S + (int64_t)r * ld;
  const float l = lse[r], gs = gloss[0] / (float)NA;
  const int t = (int)target[r];
  if (VEC) {
    for (int c = 4 * threadIdx.x; c < cols; c += 4 * kWG) {
      if (c + 3 < NB) {
        const f4 v = *reinterpret_cast<const f4 *>(row + c);
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gs * (expf(v[e] - l) - (c + e == t ? 1.f : 0.f));
        *reinterpret_cast<f4 *>(out + c) = o;
      } else {
        for (int e = 0; e < 4 && c + e < cols; ++e)
          out[c + e] = c + e < NB ? gs * (expf(row[c + e] - l) - (c + e == t ? 1.f : 0.f)) : 0.f;
      }
    }
  } else {
    for (int c = threadIdx.x; c < cols; c += kWG) out[c] = c < NB ? gs * (expf(row[c] - l) - (c == t ? 1.f : 0.f)) : 0.f;
  }
}

inline int wgrad_slabs(int64_t rows) {
  // One 8-wave workgroup is resident per CU (its two LDS images take 104 / 154 KB), so slab counts are whole ROUNDS of the
  // chip: 302 slabs were a full round plus a round that kept 46 CUs busy (Mesh-MNIST batch, 77 k rows: 29 us against 19).
  // Small products: at least two 32-row blocks per slab until every CU has one (a 7000-row mesh on 27 slabs of 256 rows
  // kept 27 CUs busy for 24 us; on 110 slabs the same product takes 13).  A second round only when its slabs still have
  // 256 rows (each slab costs a 128 x C tile of partials, written and read again).
  int64_t b = (rows + 63) / 64;
  if (b > kCUs) b = rows >= (int64_t)2 * 256 * kCUs ? 2 * kCUs : kCUs;
  if (b < 1) b = 1;
  return (int)b;
}


// ------------------------------------------------------------------------------------------------
// Dense-correspondence loss WITHOUT the score matrix (src/dense_correspondence/models.py:203 bmm(FA, FB^T) and
// main.py:229-240 argmin-target cross entropy over outputs[0, :NA, :NB]): the 7000 x 7000 scores are formed tile by tile on
// the fp16 matrix pipe and reduced on the spot; forward and backward never write them.
//
// Arithmetic: the two-piece fp16 split of the Linear kernels (sn_gemm.hip).  x·up = h + l with h = rn16(x·up),
// l = rn16(x·up - h) holds 22+ significant bits, `up` an exact power of two taken from the MATRIX's absolute maximum (so it
// factors out of every contraction); a product is the three partial products l·h + h·l + h·h, each exact in the fp32
// accumulator of v_mfma_f32_32x32x16_f16 (the dropped l·l is below 2^-24 of the term).  Elements more than 2^16 below the
// matrix maximum lose low-order bits of l: an ABSOLUTE error below 2^-39 of the maximum, nothing next to the fp32 rounding of
// a 120-term sum.  The soft-max factor P = softmax - onehot in [-1, 1] is split the same way after scaling by 2^14.
//
// Layout: every operand is stored in MFMA FRAGMENT ORDER, 32 rows (a "tile") at a time — [tile][k-step][piece][lane][8 halfs],
// lane (i, kh) holding row i's elements 8 kh .. 8 kh + 7 of the k-step — so that one wave-wide LDS-DMA instruction moves 1 KiB
// of contiguous global memory into 1 KiB of LDS that ds_read_b128 then reads without bank conflicts: no transposition, no
// address arithmetic per element.  R holds the features for the score product (contraction over the feature index), T holds
// them transposed for the gradient product (contraction over the streamed rows, in the order the accumulator of the score
// tile hands them over: pair_perm).
//   pair_maxabs_k   absolute maximum of both feature matrices -> the two scales
//   pair_split_k    F -> R, T of both sides
//   pair_lse_k      a workgroup owns 128 rows of A (4 waves x 32, fragments in registers) and streams a RANGE of B's tiles
//                   through a double-buffered LDS stage; the tile is computed TRANSPOSED (lane = row of A), so the soft-max
//                   reductions of a row stay inside a lane; (max, sum, target logit) per row and range -> pair_combine_k
//   pair_grad_k     both gradients in one launch: a workgroup owns 128 rows of one side and streams a range of the other
//                   side's tiles (R and T); scores recomputed, P split in registers — the accumulator layout of the transposed
//                   tile IS the operand layout of the second product — dOwn += P·Other; partial sums per range
//   pair_reduce_k   sums the ranges in fixed order, applies gloss / NA and the scales, zero-fills the padding rows
// ------------------------------------------------------------------------------------------------
constexpr int kPairKP = 128;                       // padded feature count (K <= 128)
constexpr int kPairTile = 32 * kPairKP * 2;        // halfs of one tile of R (or T): 8 (or 4 x 2) k-steps x 2 pieces x 64 lanes x 8
constexpr int kPairChunk = 512;                    // halfs per DMA instruction (64 lanes x 16 B)
constexpr int kPairMaxLseSplits = 8, kPairMaxGradSplits = 4;
constexpr int kPairHeader = 256;                   // bytes: [0] max|FA| bits, [1] max|FB| bits

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16v mfma_f16(const u4 &a, const u4 &b, const f16v &c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <int N_>
__device__ __forceinline__ void pair_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
// up = 2^(14 - E), down = 2^(E - 14) for an absolute maximum m = f·2^E, f in [0.5, 1)
__device__ __forceinline__ void pair_scales(unsigned mbits, float &up, float &down) {
  int e = (int)((mbits >> 23) & 0xffu) - 126;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);       // zero / denormal / non-finite matrices: any finite scale will do
  up = __uint_as_float((unsigned)(127 + 14 - e) << 23);
  down = __uint_as_float((unsigned)(127 - 14 + e) << 23);
}

__global__ __launch_bounds__(kWG) void pair_maxabs_k(const float *__restrict__ FA, int64_t lda, int rowsA, const float *__restrict__ FB,
                                                     int64_t ldb, int rowsB, int K, unsigned *__restrict__ header) {
  const float *F = blockIdx.y ? FB : FA;
  const int64_t ld = blockIdx.y ? ldb : lda;
  const int64_t total = (int64_t)(blockIdx.y ? rowsB : rowsA) * K;
  unsigned m = 0;
  for (int64_t id = (int64_t)blockIdx.x * kWG + threadIdx.x; id < total; id += (int64_t)gridDim.x * kWG) {
    const unsigned b = __float_as_uint(F[id / K * ld + id % K]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    const unsigned q = (unsigned)__shfl_xor((int)m, o);
    m = q > m ? q : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(header + blockIdx.y, m);
}

// one thread per (row, 4 features) of a side (blockIdx.y): rows [0, npad), feature quads [0, 32)
__global__ __launch_bounds__(kWG) void pair_split_k(const float *__restrict__ FA, int64_t lda, int rowsA, int npadA, unsigned short *__restrict__ RA,
                                                    unsigned short *__restrict__ TA, const float *__restrict__ FB, int64_t ldb, int rowsB,
                                                    int npadB, unsigned short *__restrict__ RB, unsigned short *__restrict__ TB, int K,
                                                    const unsigned *__restrict__ header) {
  const bool sb = blockIdx.y != 0;
  const float *F = sb ? FB : FA;
  const int64_t ld = sb ? ldb : lda;
  const int n = sb ? rowsB : rowsA, npad = sb ? npadB : npadA;
  unsigned short *R = sb ? RB : RA, *T = sb ? TB : TA;
  const int64_t id = (int64_t)blockIdx.x * kWG + threadIdx.x;
  if (id >= (int64_t)npad * 32) return;
  float up, down;
  pair_scales(header[sb ? 1 : 0], up, down);
  const int row = (int)(id >> 5), kq = (int)(id & 31) * 4;
  _Float16 h[4], l[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float v = ((row < n && kq + c < K) ? F[(int64_t)row * ld + kq + c] : 0.f) * up;
    h[c] = (_Float16)v;
    l[c] = (_Float16)(v - (float)h[c]);
  }
  const int t = row >> 5, i = row & 31;
  {  // R: [t][ks][p][kh*32 + i][j], k = 16 ks + 8 kh + j
    const int ks = kq >> 4, kh = (kq >> 3) & 1, j = kq & 7;
    _Float16 *r = reinterpret_cast<_Float16 *>(R) + (size_t)t * kPairTile + ((size_t)(ks * 2) * 64 + kh * 32 + i) * 8 + j;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      r[c] = h[c];
      r[512 + c] = l[c];
    }
  }
  {  // T: [t][f][s2][p][kh*32 + kfl][j], feature 32 f + kfl, streamed row 16 s2 + 8 (j >> 2) + 4 kh + (j & 3)  (pair_perm)
    const int s2 = i >> 4, r16 = i & 15, kh = (r16 >> 2) & 1, j = 4 * (r16 >> 3) + (r16 & 3);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int kf = kq + c, f = kf >> 5, kfl = kf & 31;
      _Float16 *q = reinterpret_cast<_Float16 *>(T) + (size_t)t * kPairTile + ((size_t)((f * 2 + s2) * 2) * 64 + kh * 32 + kfl) * 8 + j;
      q[0] = h[c];
      q[512] = l[c];
    }
  }
}

// the wave's share (chunks wave, wave + 4, ...) of NCH 1 KiB chunks from global memory into the LDS stage
template <int NCH>
__device__ __forceinline__ void pair_stage(const unsigned short *__restrict__ src, unsigned short *dst, int wave, int lane) {
#pragma unroll
  for (int q = 0; q < NCH / 4; ++q) {
    const int c = wave + 4 * q;
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const u4 *>(src + (size_t)c * kPairChunk) + lane, dst + c * kPairChunk, 16, 0, 0);
  }
}

// transposed score tile from the staged R tile: D[i][n] = sum_k Other[i][k] · Own[n][k]   (lane & 31 = n; element e <-> streamed
// row i = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)); two accumulators (even / odd k-steps) halve the dependent chain
__device__ __forceinline__ f16v pair_tile(const u4 (&own)[8][2], const unsigned short *st, int lane) {
  const u4 *s4 = reinterpret_cast<const u4 *>(st) + lane;
  f16v a0, a1;
#pragma unroll
  for (int e = 0; e < 16; ++e) a0[e] = a1[e] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ks += 2) {
    const u4 h0 = s4[(ks * 2) * 64], l0 = s4[(ks * 2 + 1) * 64], h1 = s4[(ks * 2 + 2) * 64], l1 = s4[(ks * 2 + 3) * 64];
    a0 = mfma_f16(l0, own[ks][0], a0);
    a1 = mfma_f16(l1, own[ks + 1][0], a1);
    a0 = mfma_f16(h0, own[ks][1], a0);
    a1 = mfma_f16(h1, own[ks + 1][1], a1);
    a0 = mfma_f16(h0, own[ks][0], a0);
    a1 = mfma_f16(h1, own[ks + 1][0], a1);
  }
  return a0 + a1;
}

__device__ __forceinline__ void pair_load_own(u4 (&own)[8][2], const unsigned short *__restrict__ R, int tile, int lane) {
  const u4 *g = reinterpret_cast<const u4 *>(R + (size_t)tile * kPairTile) + lane;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    own[ks][0] = g[(ks * 2) * 64];
    own[ks][1] = g[(ks * 2 + 1) * 64];
  }
}

// grid (ceil(tilesA / 4), splits); part[split][row][4] = (max, sum, target logit, -)
__global__ __launch_bounds__(kWG, 2) void pair_lse_k(const unsigned short *__restrict__ RA, const unsigned short *__restrict__ RB,
                                                     const int64_t *__restrict__ target, int NA, int NB, int npadA,
                                                     const unsigned *__restrict__ header, float *__restrict__ part) {
  __shared__ __attribute__((aligned(16))) unsigned short stage[2][kPairTile];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 31, kh = lane >> 5;
  const int tilesA = (NA + 31) / 32, tilesB = (NB + 31) / 32;
  const int mytile = blockIdx.x * 4 + wave;
  const int n0 = mytile * 32;
  u4 own[8][2];
  pair_load_own(own, RA, mytile < tilesA ? mytile : tilesA - 1, lane);
  float upA, downA, upB, downB;
  pair_scales(header[0], upA, downA);
  pair_scales(header[1], upB, downB);
  const float sAB = downA * downB;
  const int tgt = (n0 + n < NA) ? (int)target[n0 + n] : -1;
  const int per = (tilesB + (int)gridDim.y - 1) / (int)gridDim.y;
  const int t0 = blockIdx.y * per, t1 = min(tilesB, t0 + per);
  float m = -INFINITY, l = 0.f, tl = 0.f;
  pair_wait_vmcnt<0>();                              // (own fragments, target: out of the way of the counted stage loads)
  if (t0 < t1) pair_stage<16>(RB + (size_t)t0 * kPairTile, stage[0], wave, lane);
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    if (t + 1 < t1) {
      pair_stage<16>(RB + (size_t)(t + 1) * kPairTile, stage[buf ^ 1], wave, lane);
      pair_wait_vmcnt<4>();
    } else {
      pair_wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();                    // every wave's share of tile t has landed
    const f16v acc = pair_tile(own, stage[buf], lane);
    float sv[16], tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
      const float s = acc[e] * sAB;
      sv[e] = i < NB ? s : -INFINITY;
      tl += i == tgt ? s : 0.f;
      tmax = fmaxf(tmax, sv[e]);
    }
    if (tmax > -INFINITY) {
      const float mn = fmaxf(m, tmax);
      float add = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) add += __expf(sv[e] - mn);          // (exp(-inf) = 0 for the columns past NB)
      l = l * __expf(m - mn) + add;
      m = mn;
    }
    __builtin_amdgcn_s_barrier();                    // all waves are done with stage[buf] before tile t + 2 lands in it
  }
  // the two half-waves hold different columns of the same row
  const float m2 = __shfl_xor(m, 32), l2 = __shfl_xor(l, 32), t2 = __shfl_xor(tl, 32);
  const float mn = fmaxf(m, m2);
  l = (m > -INFINITY ? l * __expf(m - mn) : 0.f) + (m2 > -INFINITY ? l2 * __expf(m2 - mn) : 0.f);
  tl += t2;
  if (kh == 0 && n0 + n < NA) {
    float *p = part + ((size_t)blockIdx.y * npadA + n0 + n) * 4;
    *reinterpret_cast<f4 *>(p) = f4{mn, l, tl, 0.f};
  }
}

__global__ __launch_bounds__(kWG) void pair_combine_k(const float *__restrict__ part, int splits, int npadA, int NA, float *__restrict__ lse,
                                                      float *__restrict__ rowloss) {
  const int r = blockIdx.x * kWG + threadIdx.x;
  if (r >= NA) return;
  float mm = -INFINITY;
  for (int s = 0; s < splits; ++s) mm = fmaxf(mm, part[((size_t)s * npadA + r) * 4]);
  float ll = 0.f, tt = 0.f;
  for (int s = 0; s < splits; ++s) {
    const f4 v = *reinterpret_cast<const f4 *>(part + ((size_t)s * npadA + r) * 4);
    ll += v.x > -INFINITY ? v.y * __expf(v.x - mm) : 0.f;
    tt += v.z;
  }
  const float ls = mm + __logf(ll);
  lse[r] = ls;
  rowloss[r] = ls - tt;
}

struct PairGradSide {
  const unsigned short *Rown, *Roth, *Toth;
  float *part;             // [splits][npad_own][128]
  int Nown, Noth, npad_own, nblk, splits;
};

// grid: side A's nblk x splits workgroups, then side B's.  Own rows n of side A carry lse / target themselves; for side B
// (own rows are COLUMNS of the score matrix) they belong to the streamed rows and come through the stage.
__global__ __launch_bounds__(kWG, 2) void pair_grad_k(PairGradSide A, PairGradSide B, const int64_t *__restrict__ target,
                                                      const float *__restrict__ lse, const unsigned *__restrict__ header, int NA) {
  extern __shared__ __attribute__((aligned(16))) unsigned short gstage[];      // 2 x (R tile | T tile | 4 x 256 B lse / target)
  constexpr int kStage = 2 * kPairTile + 4 * 128;                              // halfs
  const bool ownA = blockIdx.x < (unsigned)(A.nblk * A.splits);
  const PairGradSide &S = ownA ? A : B;
  const int bid = ownA ? blockIdx.x : blockIdx.x - A.nblk * A.splits;
  const int blk = bid % S.nblk, split = bid / S.nblk;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = lane & 31, kh = lane >> 5;
  const int tiles_own = (S.Nown + 31) / 32, tiles_oth = (S.Noth + 31) / 32;
  const int mytile = blk * 4 + wave;
  const int n0 = mytile * 32;
  const bool own_ok = n0 + n < S.Nown;
  u4 own[8][2];
  pair_load_own(own, S.Rown, mytile < tiles_own ? mytile : tiles_own - 1, lane);
  float upA, downA, upB, downB;
  pair_scales(header[0], upA, downA);
  pair_scales(header[1], upB, downB);
  const float sAB = downA * downB;
  float my_lse = 0.f;
  int my_tgt = -1;
  if (ownA && own_ok) {
    my_lse = lse[n0 + n];
    my_tgt = (int)target[n0 + n];
  }
  const int per = (tiles_oth + S.splits - 1) / S.splits;
  const int t0 = split * per, t1 = min(tiles_oth, t0 + per);
  f16v g[4];                                   // dOwn[n][32 f + (e&3) + 8 (e>>2) + 4 kh], f = 0..3
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int e = 0; e < 16; ++e) g[f][e] = 0.f;
  auto issue = [&](int t, int buf) {
    unsigned short *st = gstage + buf * kStage;
    pair_stage<16>(S.Roth + (size_t)t * kPairTile, st, wave, lane);
    pair_stage<16>(S.Toth + (size_t)t * kPairTile, st + kPairTile, wave, lane);
    // lanes 0..31: lse of the streamed rows, lanes 32..63: their targets (low words) — a private copy per wave
    const int i = min(t * 32 + n, NA - 1);
    const void *src = kh ? static_cast<const void *>(target + i) : static_cast<const void *>(lse + i);
    __builtin_amdgcn_global_load_lds(static_cast<const unsigned *>(src), st + 2 * kPairTile + wave * 128, 4, 0, 0);
  };
  pair_wait_vmcnt<0>();
  if (t0 < t1) issue(t0, 0);
  for (int t = t0; t < t1; ++t) {
    const int buf = (t - t0) & 1;
    if (t + 1 < t1) {
      issue(t + 1, buf ^ 1);
      pair_wait_vmcnt<9>();
    } else {
      pair_wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    const unsigned short *st = gstage + buf * kStage;
    const f16v acc = pair_tile(own, st, lane);
    const float *aux = reinterpret_cast<const float *>(st + 2 * kPairTile + wave * 128);
    // P (times 2^14) for my own row and the 16 streamed rows this lane holds, as two fp16 pieces: slot (s2, j) = element 8 s2 + j
    u4 PH[2], PL[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float pv[8];
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) {
        const int il = 16 * s2 + 8 * jq + 4 * kh;                       // streamed rows il .. il + 3 of the tile (elements 8 s2 + 4 jq + 0..3)
        f4 ls4 = f4{my_lse, my_lse, my_lse, my_lse};
        int tg[4] = {my_tgt, my_tgt, my_tgt, my_tgt};
        if (!ownA) {
          ls4 = *reinterpret_cast<const f4 *>(aux + il);
          const int4 q4 = *reinterpret_cast<const int4 *>(aux + 32 + il);
          tg[0] = q4.x; tg[1] = q4.y; tg[2] = q4.z; tg[3] = q4.w;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int e = 8 * s2 + 4 * jq + c;
          const int i = t * 32 + il + c;                                // streamed (other) row
          const bool hit = ownA ? (i == tg[c]) : (tg[c] == n0 + n);
          const float p = __expf(acc[e] * sAB - ls4[c]) - (hit ? 1.f : 0.f);
          pv[4 * jq + c] = (own_ok && i < S.Noth) ? p * 16384.f : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const h2v h = __builtin_convertvector(f2v{pv[2 * q], pv[2 * q + 1]}, h2v);
        const h2v lo = __builtin_convertvector(f2v{pv[2 * q] - (float)h.x, pv[2 * q + 1] - (float)h.y}, h2v);
        PH[s2][q] = __builtin_bit_cast(unsigned, h);
        PL[s2][q] = __builtin_bit_cast(unsigned, lo);
      }
    }
    // dOwn[n][kf] += sum_i P[n][i] Other[i][kf]: D2[kf][n], operand A = T tile (feature-major, pair_perm order), operand B = P
    const u4 *t4 = reinterpret_cast<const u4 *>(st + kPairTile) + lane;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      u4 th[4], tl_[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        th[f] = t4[((f * 2 + s2) * 2) * 64];
        tl_[f] = t4[((f * 2 + s2) * 2 + 1) * 64];
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) g[f] = mfma_f16(tl_[f], PH[s2], g[f]);
#pragma unroll
      for (int f = 0; f < 4; ++f) g[f] = mfma_f16(th[f], PL[s2], g[f]);
#pragma unroll
      for (int f = 0; f < 4; ++f) g[f] = mfma_f16(th[f], PH[s2], g[f]);
    }
    __builtin_amdgcn_s_barrier();
  }
  if (mytile < tiles_own) {
    float *p = S.part + ((size_t)split * S.npad_own + n0 + n) * kPairKP;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f4 *>(p + 32 * f + 8 * q + 4 * kh) = f4{g[f][4 * q], g[f][4 * q + 1], g[f][4 * q + 2], g[f][4 * q + 3]};
  }
}

// dOwn[r][k] = (gloss / NA) 2^-14 down_other · sum_splits part[s][r][k] for r < Nown, 0 for the padding rows; one thread per
// (row, 4 features) of a side (blockIdx.y)
__global__ __launch_bounds__(kWG) void pair_reduce_k(PairGradSide A, PairGradSide B, float *__restrict__ dFA, int64_t ldda, int rowsA,
                                                     float *__restrict__ dFB, int64_t lddb, int rowsB, int K,
                                                     const float *__restrict__ gloss, const unsigned *__restrict__ header, int NA) {
  const bool sb = blockIdx.y != 0;
  const PairGradSide &S = sb ? B : A;
  float *d = sb ? dFB : dFA;
  const int64_t ldd = sb ? lddb : ldda;
  const int rows = sb ? rowsB : rowsA;
  const int64_t id = (int64_t)blockIdx.x * kWG + threadIdx.x;
  const int r = (int)(id >> 5), k = (int)(id & 31) * 4;
  if (r >= rows || k >= K) return;
  f4 v = f4{0.f, 0.f, 0.f, 0.f};
  if (r < S.Nown) {
    for (int s = 0; s < S.splits; ++s) v += *reinterpret_cast<const f4 *>(S.part + ((size_t)s * S.npad_own + r) * kPairKP + k);
    float up, down;
    pair_scales(header[sb ? 0 : 1], up, down);      // the OTHER side's features were scaled up
    v *= gloss[0] / (float)NA * (1.f / 16384.f) * down;
  }
  float *o = d + (int64_t)r * ldd + k;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (k + c < K) o[c] = v[c];
}


// ------------------------------------------------------------------------------------------------
// wgrad_h_k — the uniform-wave weight gradient on TWO fp16 pieces (three partial products instead of six).
//
// Ablation of wgrad_u_k on one box (round 4, tools/scratch/wgrad_ab.sh, 627 200 rows x 256 columns): the six bf16 products
// alone (no loads, no conversion) take 150 of the kernel's 293 us at the 1.8 GHz the chip sustains under it, the memory path
// alone 191, and the two do not hide each other; with three of the six products issued the kernel runs in 232.  Halving the
// matrix work needs the fp16 split of the forward kernels (sn_gemm.hip: x = h + l, h = rn16(x), l = rn16(x - h), three exact
// products h·h + (h·l + l·h)), and fp16 needs RANGE control.  The contraction runs over the rows, so the powers of two
// that bring the operands into range must be constant along a COLUMN, and they must be known before the first row is
// converted.  Both come from bounds the callers already hold:
//   x - mean   BatchNorm's own statistics: sum_r (x[r][c] - mean[c])^2 = n·var[c], hence |x[r][c] - mean[c]| <= sqrt(n / invstd[c]^2)
//              for EVERY row — rigorous, no pass over x (n = rows behind the statistics; the caller passes xfac >= sqrt(n));
//   dy         one number for the whole operand, an upper bound of max |dy| left by the kernel that PRODUCED dy (the
//              input-gradient GEMM's epilogue, the transposed sparse product's store, the loss backward: sn_absmax_* in
//              sn_spmm.h); one global scale suffices for rigour, and accuracy survives it — below.
// Scaled operands: bound -> [2^14, 2^15) (fp16 tops out at 65504), so nothing can overflow.  LOW11: the low pieces carry a
// further 2^11 (their two products go to a second accumulator, folded in with 2^-11 at the end, as in sn_gemm.hip): an
// element keeps 22 significant bits while its scaled magnitude is >= 2^-14, i.e. down to 2^-28 of the operand's bound, and
// 2^-36 of the bound in absolute terms below that.  !LOW11: one accumulator, the low piece unscaled: 22 bits down to 2^-17
// of the bound, 2^-25 (scaled) absolute below.  Results leave multiplied by the exact inverse scales.
// Structure (slots, images, barriers, slab tables, column sums of dy) as wgrad_u_k; two pieces: two LDS images of 51 KB at
// C = 256 instead of three.  SETS = 2: two 32-row blocks in flight in registers instead of one.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pow2_up_for(float bound) {      // 2^(15 - E) for bound = f·2^E, f in [0.5, 1): bound -> [2^14, 2^15)
  int e = (int)((__float_as_uint(bound) >> 23) & 0xffu) - 126;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);                     // zero / denormal / non-finite bounds: any finite scale
  return __uint_as_float((unsigned)(127 + 15 - e) << 23);
}
__device__ __forceinline__ float pow2_inv(float p) {             // 1 / p for p = 2^k, k in [-115, 115]
  return __uint_as_float((254u << 23) - __float_as_uint(p));
}

template <int CT /* C / 128 */, bool LOW11, int SETS>
__global__ __launch_bounds__(kWgradThreads, 1) void wgrad_h_k(const float *__restrict__ dy, int64_t lddy,
                                                              const float *__restrict__ x, int64_t ldx,
                                                              const float *__restrict__ center, int64_t rows, int J, int C,
                                                              float *__restrict__ partial /* [grid][128][C] */,
                                                              float *__restrict__ colpart /* [grid][128] | NULL */,
                                                              int64_t seg_rows, int spm, const int64_t *__restrict__ slab_off,
                                                              const float *__restrict__ dybound /* [ndy]: max >= max |dy| */,
                                                              int ndy, const float *__restrict__ xinvstd /* [C] */, float xfac,
                                                              int interleave /* plain slabs only: 32-row blocks b, b + grid, ... */) {
  constexpr int NCG = 32 + 32 * CT;          // column groups of 4: 32 of dy, 32·CT of x
  constexpr int QP = NCG + 4;                // slots per (column % 4) plane; QP % 16 == 4 keeps fragment reads conflict-free
  constexpr int PL = 4 * QP;                 // slots per row group (8 rows)
  constexpr int NK = 1 + CT;                 // conversion slots per wave and block: one of dy, CT of x
  constexpr int NM = 12 * CT;                // MFMAs per wave and block: 2 steps x 3 products x 2·CT tiles
  constexpr int NPAIR = 4 * NK, NLOAD = 8 * NK;
  constexpr int NACC = LOW11 ? 2 : 1;
  static_assert(QP % 16 == 4, "slot permutation");
  __shared__ u4 img[2][2][4 * PL];           // [buffer][piece][slot]; one block = 32 rows = 4 row groups
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  int64_t r0, r1;                            // row slab of this workgroup (as wgrad_u_k)
  if (slab_off) {
    r0 = slab_off[blockIdx.x];
    r1 = slab_off[blockIdx.x + 1];
  } else if (seg_rows > 0) {
    const int64_t mesh = blockIdx.x / spm, part = blockIdx.x % spm;
    int64_t per = (seg_rows + spm - 1) / spm;
    per = (per + 15) & ~(int64_t)15;
    const int64_t mend = (mesh + 1) * seg_rows < rows ? (mesh + 1) * seg_rows : rows;
    r0 = mesh * seg_rows + part * per;
    r1 = r0 + per < mend ? r0 + per : mend;
  } else if (interleave) {
    r0 = 32 * (int64_t)blockIdx.x;             // my first block; the following ones lie G blocks apart, up to the operand's end
    r1 = rows;
  } else {
    int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    per = (per + 15) & ~(int64_t)15;
    r0 = (int64_t)blockIdx.x * per;
    r1 = r0 + per < rows ? r0 + per : rows;
  }
  const int G = (interleave && !slab_off && seg_rows <= 0) ? (int)gridDim.x : 1;      // 32-row blocks between two of mine
  const int nblocks = r1 > r0 ? (int)(((r1 - r0 + 31) / 32 + G - 1) / G) : 0;
  const int span = r1 > r0 ? (int)(r1 - r0) : 0;
  float *P = partial + (int64_t)blockIdx.x * 128 * C;

  // ---- matrix role: dy tiles 2·ga, 2·ga + 1  x  x tiles CT·gb .. CT·gb + CT - 1 ----
  const int i = lane & 31, kh = lane >> 5;
  const int ga = wave >> 2, gb = wave & 3;
  const int fo = kh * PL + (i & 3) * QP + (i >> 2);
  f16v acc[NACC][2][CT];
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < CT; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[n][a][b][e] = 0.f;
  // ---- conversion role (as wgrad_u_k: slot 0 = a dy half-block, slots 1.. = x half-blocks) ----
  const int lcol = lane;
  int s_rg[NK];
  const float *s_cur[NK];
  int l_voff[NK];
  int l_slot[NK];
  float l_mu[NK], l_sc[NK];
  float l_sum = 0.f;
  const int dy_rstep = 4 * (int)lddy, x_rstep = 4 * (int)ldx;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int xhb = wave + 8 * (k - 1);
    const int rg = k == 0 ? wave >> 1 : xhb / (2 * CT);
    const int c = k == 0 ? 64 * (wave & 1) + lcol : 128 + 64 * (xhb % (2 * CT)) + lcol;      // column in the image
    s_rg[k] = rg;
    s_cur[k] = k == 0 ? dy + 64 * (wave & 1) + (r0 + 8 * rg) * lddy : x + 64 * (xhb % (2 * CT)) + (r0 + 8 * rg) * ldx;
    l_voff[k] = (k == 0 && c >= J) ? 0x7fffff00 : 4 * lcol;
    l_mu[k] = (k > 0 && center) ? center[c - 128] : 0.f;
    l_sc[k] = k == 0 ? 1.f : pow2_up_for(xfac / xinvstd[c - 128]);          // (slot 0: dy's scale, set after the first loads are out)
    l_slot[k] = rg * PL + (c & 3) * QP + (c >> 2);
  }
  float raw[SETS][NK][8];
  u4 cH[NK], cL[NK];
  auto conv_pair = [&](auto sc, auto kc, auto pc) {
    constexpr int set = decltype(sc)::value, k = decltype(kc)::value, p = decltype(pc)::value;
    f2v xv = {raw[set][k][2 * p], raw[set][k][2 * p + 1]};
    if constexpr (k == 0) l_sum += xv.x + xv.y;
    else xv -= f2v{l_mu[k], l_mu[k]};
    xv *= l_sc[k];                                                   // exact (power of two; the bound keeps it below 2^15)
    if constexpr (k > 0) {
      // rows past the slab's end load as 0 and leave the centring as -mean, which no bound covers (a column sitting at
      // -1 +- 1e-4 behind an ELU: |mean| / bound = 17): scaled it may pass fp16's range, and inf x the exact 0 of the padded dy row
      // is NaN.  Clamped to the largest fp16 value the product with 0 is 0; rows inside the bound are untouched.
      xv.x = __builtin_amdgcn_fmed3f(xv.x, -65504.f, 65504.f);
      xv.y = __builtin_amdgcn_fmed3f(xv.y, -65504.f, 65504.f);
    }
    const h2v h = __builtin_convertvector(xv, h2v);                  // round to nearest
    f2v r = xv - __builtin_convertvector(h, f2v);                    // exact remainder
    if constexpr (LOW11) r *= 2048.f;
    const h2v l = __builtin_convertvector(r, h2v);
    cH[k][p] = __builtin_bit_cast(unsigned, h);
    cL[k][p] = __builtin_bit_cast(unsigned, l);
  };
  auto conv_write = [&](auto kc, int buf) {
    constexpr int k = decltype(kc)::value;
    img[buf][0][l_slot[k]] = cH[k];
    img[buf][1][l_slot[k]] = cL[k];
  };
  __amdgpu_buffer_rsrc_t s_rs[NK];
  auto open_slot = [&](auto kc, int b) {
    constexpr int k = decltype(kc)::value;
    const int rstep = k == 0 ? dy_rstep : x_rstep;
    int left = span - 32 * G * b - 8 * s_rg[k];
    left = left < 0 ? 0 : (left > 8 ? 8 : left);
    const int extent = __builtin_amdgcn_readfirstlane(left * rstep);
    s_rs[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s_cur[k]), 0, extent, 0x00020000);
    s_cur[k] += (int64_t)8 * rstep * G;
  };
  auto load_row = [&](auto sc, auto kc, auto jc) {
    constexpr int set = decltype(sc)::value, k = decltype(kc)::value, j = decltype(jc)::value;
    const int rstep = k == 0 ? dy_rstep : x_rstep;
    raw[set][k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s_rs[k], l_voff[k] + j * rstep, 0, 0));
  };
  // Work dealt out behind MFMA m of a block: pair q of the conversion at m_pair(q), the reload of the two rows it freed right
  // behind it (load L = 8k + j at m_pair(4k + j/2) + j % 2), the two LDS slots of a unit after its fourth pair.
  auto behind_mfma = [&](auto sc, auto ic, auto mc, int b) {
    constexpr int image = decltype(ic)::value, m = decltype(mc)::value;
    wstatic_for<0, NPAIR>([&](auto qc) {
      constexpr int q = decltype(qc)::value, k = q / 4, p = q % 4;
      constexpr int mq = q * NM / NPAIR;
      if constexpr (mq == m) {
        conv_pair(sc, WIC<k>{}, WIC<p>{});
        if constexpr (p == 3) conv_write(WIC<k>{}, image);
      }
      // (j = 2p reloads with its pair, j = 2p + 1 one MFMA later; the last slot's last row stays inside the block)
      constexpr int m0 = mq, m1 = mq + 1 < NM ? mq + 1 : NM - 1;
      if constexpr (m0 == m) {
        if constexpr (p == 0) open_slot(WIC<k>{}, b);
        load_row(sc, WIC<k>{}, WIC<2 * p>{});
      }
      if constexpr (m1 == m) load_row(sc, WIC<k>{}, WIC<2 * p + 1>{});
    });
  };
#define SN_BLOCK_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  auto multiply_block = [&](auto uc, int b) {
    constexpr int u = decltype(uc)::value;             // b % lcm(2, SETS): block b = image b & 1, block n lives in set n % SETS
    constexpr int buf = u & 1, set = (u + 1) % SETS;   // block b + 1 converts from its set while block b is multiplied
    SN_BLOCK_BARRIER();
    u4 A[2][2][2], B[2][CT][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int a = 0; a < 2; ++a) A[st][a][p] = img[buf][p][fo + 2 * st * PL + 8 * (2 * ga + a)];
#pragma unroll
        for (int c = 0; c < CT; ++c) B[st][c][p] = img[buf][p][fo + 2 * st * PL + 32 + 8 * (CT * gb + c)];
      }
    // three products per step and tile, small terms first: (l, h), (h, l) -> the correction accumulator (LOW11) | the one
    // accumulator; (h, h) last; tile-inner, so consecutive MFMAs never share an accumulator
    wstatic_for<0, NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value, g = m / (2 * CT), st = g / 3, t = g % 3;
      constexpr int a = (m % (2 * CT)) / CT, c = m % CT;
      constexpr int pa = t == 0 ? 1 : 0, pb = t == 1 ? 1 : 0;
      constexpr int n = (LOW11 && t < 2) ? 1 : 0;
      acc[n][a][c] = mfma_f16(A[st][a][pa], B[st][c][pb], acc[n][a][c]);
      behind_mfma(WIC<set>{}, WIC<buf ^ 1>{}, mc, b + 1 + SETS);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // the first blocks' rows are requested BEFORE the scales are formed (the reduction of dy's maxima then runs under them)
  if (nblocks > 0) {
    wstatic_for<0, SETS>([&](auto sc) {
      wstatic_for<0, NK>([&](auto kc) {
        open_slot(kc, decltype(sc)::value);
        wstatic_for<0, 8>([&](auto jc) { load_row(sc, kc, jc); });
      });
    });
  }
  // scales: one for dy — from the maximum of the producer's per-workgroup maxima —, one per x column (mine as a converter,
  // mine as the owner of output columns)
  float sdy;
  {
    float *sb = reinterpret_cast<float *>(&img[0][0][0]);             // (the images are not in use yet)
    // up to ~20 000 maxima (one per workgroup of the transposed product that wrote dy): 16-byte loads, eight in flight per
    // lane — one dependent load per iteration cost this prologue ≈12 us per launch (fixed cost 17 -> 29 us against the bf16
    // kernel, from the launch times at two row counts)
    float m = 0.f;
    const int n4 = (reinterpret_cast<uintptr_t>(dybound) & 15u) == 0 ? ndy / 4 : 0;
    const f4 *d4 = reinterpret_cast<const f4 *>(dybound);
    int q = tid;
    for (; q + 7 * kWgradThreads < n4; q += 8 * kWgradThreads) {
      f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = d4[q + u * kWgradThreads];
#pragma unroll
      for (int u = 0; u < 8; ++u) m = fmaxf(m, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
    }
    for (; q < n4; q += kWgradThreads) {
      const f4 v = d4[q];
      m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    for (int r = 4 * n4 + tid; r < ndy; r += kWgradThreads) m = fmaxf(m, dybound[r]);
    unsigned mb = __float_as_uint(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned other = (unsigned)__shfl_xor((int)mb, o);
      mb = other > mb ? other : mb;
    }
    if (lane == 0) sb[wave] = __uint_as_float(mb);
    __syncthreads();
    m = fmaxf(fmaxf(fmaxf(sb[0], sb[1]), fmaxf(sb[2], sb[3])), fmaxf(fmaxf(sb[4], sb[5]), fmaxf(sb[6], sb[7])));
    __syncthreads();                                                   // (all have read before the first image write)
    sdy = pow2_up_for(m);
  }
  float osc[CT];                             // inverse scale of my output column in each of my x tiles, times dy's
#pragma unroll
  for (int c = 0; c < CT; ++c) osc[c] = pow2_inv(pow2_up_for(xfac / xinvstd[32 * (CT * gb + c) + i])) * pow2_inv(sdy);
  l_sc[0] = sdy;
  if (nblocks > 0) {
    wstatic_for<0, NM>([&](auto mc) { behind_mfma(WIC<0>{}, WIC<0>{}, mc, SETS); });      // block 0 -> image 0, block SETS requested
    constexpr int U = (SETS % 2 == 0) ? SETS : 2 * SETS;                                    // lcm(2, SETS)
    for (int b = 0; b < nblocks; b += U) {
      wstatic_for<0, U>([&](auto uc) {
        if (b + decltype(uc)::value < nblocks) multiply_block(uc, b + decltype(uc)::value);
      });
    }
  }
  if (colpart) {
    __syncthreads();
    float *sm = reinterpret_cast<float *>(&img[0][0][0]);
    sm[s_rg[0] * 128 + 64 * (wave & 1) + lcol] = l_sum;
    __syncthreads();
    if (tid < 128) colpart[(int64_t)blockIdx.x * 128 + tid] = (sm[tid] + sm[128 + tid]) + (sm[256 + tid] + sm[384 + tid]);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int jr = 32 * (2 * ga + a) + (e & 3) + 8 * (e >> 2) + 4 * kh;
        float v = acc[0][a][c][e];
        if constexpr (LOW11) v = __builtin_fmaf(acc[1][a][c][e], 1.f / 2048.f, v);
        P[(int64_t)jr * C + 32 * (CT * gb + c) + i] = v * osc[c];
      }
#undef SN_BLOCK_BARRIER
}

}  // namespace

extern "C" {

size_t sn_colstats_workspace_bytes(int64_t rows, int32_t C) {
  if (C < 1) return 0;
  return (size_t)stat_blocks(rows) * 2 * (size_t)C * sizeof(double);
}

static int colstats_launch(const float *x, int64_t ld, int64_t rows, int32_t C, double *out, int64_t out_ld, int64_t out_off,
                           void *workspace, size_t workspace_bytes, void *stream) {
  if (rows < 0 || C < 1 || C > 4096 || ld < C || out_off < 0 || out_ld < out_off + C) return SN_E_SHAPE;
  if (!out) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows == 0) {
    hipError_t e = sn_internal_fill(out + out_off, 0, (size_t)C * sizeof(double), s);
    if (e == hipSuccess) e = sn_internal_fill(out + out_ld + out_off, 0, (size_t)C * sizeof(double), s);
    return e == hipSuccess ? SN_OK : (int)e;
  }
  if (!x || !workspace) return SN_E_NULL;
  if (workspace_bytes < sn_colstats_workspace_bytes(rows, C)) return SN_E_WORKSPACE;
  const int nblk = stat_blocks(rows);
  double *partial = static_cast<double *>(workspace);
  const bool vec = (C % 4 == 0) && (kWG % (C / 4) == 0) && (ld % 4 == 0) && aligned16(x);
  const int cw = vec ? C / 4 : C;
  if (!vec && cw > kWG) return SN_E_UNSUPPORTED;
  const int lanes_r = kWG / cw;
  const size_t shm = (size_t)lanes_r * 2 * C * sizeof(double);
  if (vec)
    hipLaunchKernelGGL((colstats_k<true>), dim3(nblk), dim3(kWG), shm, s, x, ld, rows, (int)C, partial);
  else
    hipLaunchKernelGGL((colstats_k<false>), dim3(nblk), dim3(kWG), shm, s, x, ld, rows, (int)C, partial);
  hipLaunchKernelGGL(colstats_final_k, dim3((2 * C + 31) / 32), dim3(kWG), 0, s, partial, nblk, 2 * (int)C, out, out_ld, out_off);
  return launch_status();
}

int sn_colstats_f32(const float *x, int64_t ld, int64_t rows, int32_t C, double *out, void *workspace,
                    size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  return colstats_launch(x, ld, rows, C, out, C, 0, workspace, workspace_bytes, stream);
}

int sn_colstats_into_f32(const float *x, int64_t ld, int64_t rows, int32_t C, double *out, int64_t out_ld, int64_t out_off,
                         void *workspace, size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  return colstats_launch(x, ld, rows, C, out, out_ld, out_off, workspace, workspace_bytes, stream);
}

int sn_colstats_merge_f64(const double *part, int32_t nblk, int32_t C, double *out, int64_t out_ld, int64_t out_off,
                          void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nblk < 0 || C < 1 || out_off < 0 || out_ld < out_off + C) return SN_E_SHAPE;
  if (!out || (nblk > 0 && !part)) return SN_E_NULL;
  hipLaunchKernelGGL(colstats_final_k, dim3((2 * C + 31) / 32), dim3(kWG), 0, static_cast<hipStream_t>(stream), part, (int)nblk,
                     2 * (int)C, out, out_ld, out_off);
  return launch_status();
}

int sn_colstats_merge2_f64(const double *part_lo, int32_t nblk_lo, int32_t C_lo, int32_t ld_lo, const double *part_hi,
                           int32_t nblk_hi, int32_t C_hi, int32_t ld_hi, double *out, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nblk_lo < 0 || nblk_hi < 0 || C_lo < 1 || C_hi < 1 || ld_lo < C_lo || ld_hi < C_hi) return SN_E_SHAPE;
  if (!out || (nblk_lo > 0 && !part_lo) || (nblk_hi > 0 && !part_hi)) return SN_E_NULL;
  const unsigned grid = (unsigned)((2 * C_lo + 31) / 32 + (2 * C_hi + 31) / 32);
  hipLaunchKernelGGL(colstats_final2_k, dim3(grid), dim3(kWG), 0, static_cast<hipStream_t>(stream), part_lo, (int)nblk_lo, (int)C_lo,
                     (int)ld_lo, part_hi, (int)nblk_hi, (int)C_hi, (int)ld_hi, out);
  return launch_status();
}

size_t sn_wgrad_workspace_bytes(int64_t rows, int32_t J, int32_t C) {
  (void)J;
  if (C < 1) return 0;
  return (size_t)wgrad_slabs(rows) * 128 * ((size_t)C + 1) * sizeof(float);      // tile partials + column-sum partials
}

// Bounds that let the weight gradient run on two fp16 pieces (wgrad_h_k): dybound[0] >= max |dy| (device), xinvstd[C] the
// BatchNorm inverse standard deviations of x's columns about `center` (device), xfac >= sqrt(rows behind those statistics).
struct WgradBounds {
  const float *dybound;      // ndy floats: their maximum bounds |dy|
  int ndy;
  const float *xinvstd;
  float xfac;
};
static int wgrad_launch(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                        int32_t J, int32_t C, float *G, double *dysum, int64_t rows_per_seg, float *seg_dysum,
                        void *workspace, size_t workspace_bytes, void *stream, const int64_t *slab_off = nullptr,
                        int32_t nslab_tab = 0, const int64_t *seg_slab_ptr = nullptr, int32_t nseg_tab = 0,
                        const WgradBounds *bounds = nullptr) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || J < 1 || C < 1 || lddy < J || ldx < C) return SN_E_SHAPE;
  if (J > 128 || (J % 4) || (C != 128 && C != 256)) return SN_E_UNSUPPORTED;
  if (!G) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool x3 = gemm_variant() != 0;
  const bool ragged = slab_off != nullptr;                  // slabs and their meshes from the caller's tables
  const bool segmented = rows_per_seg > 0 && !ragged;
  if (segmented && (!x3 || !dysum || !seg_dysum || rows % rows_per_seg)) return x3 ? SN_E_SHAPE : SN_E_UNSUPPORTED;
  if (ragged && (nslab_tab < 1 || nseg_tab < 1 || !seg_slab_ptr || !dysum || !seg_dysum)) return SN_E_SHAPE;
  if (rows == 0) {
    hipError_t e = sn_internal_fill(G, 0, (size_t)J * C * sizeof(float), s);
    if (e == hipSuccess && dysum) e = sn_internal_fill(dysum, 0, (size_t)J * sizeof(double), s);
    return e == hipSuccess ? SN_OK : (int)e;
  }
  if (!dy || !x || !workspace) return SN_E_NULL;
  if (!aligned16(dy) || !aligned16(x) || (center && !aligned16(center)) || (lddy % 4) || (ldx % 4)) return SN_E_ALIGN;
  int nslab = wgrad_slabs(rows), spm = 1;
  if (x3 && nslab > kCUs) nslab = kCUs;      // one resident workgroup per CU
  if (segmented) {
    const int64_t nseg = rows / rows_per_seg;
    spm = (int)(kCUs / nseg);
    if (spm < 1) spm = 1;
    if ((int64_t)spm * 16 > rows_per_seg) spm = (int)((rows_per_seg + 15) / 16);
    if (nseg * spm > INT_MAX) return SN_E_RANGE;
    nslab = (int)(nseg * spm);
  }
  if (ragged) nslab = nslab_tab;
  if (workspace_bytes < (size_t)nslab * 128 * ((size_t)C + 1) * sizeof(float)) return SN_E_WORKSPACE;
  float *partial = static_cast<float *>(workspace);
  float *colpart = dysum ? partial + (size_t)nslab * 128 * C : nullptr;
  const int64_t sr = segmented ? rows_per_seg : 0;
  const bool uni = x3;                                      // the 16-bit matrix-pipe kernels (raw buffer windows)
  if (uni && (lddy >= ((int64_t)1 << 24) || ldx >= ((int64_t)1 << 24))) return SN_E_UNSUPPORTED;
  if (ragged && !uni) return SN_E_UNSUPPORTED;              // slab tables: not in the fp32-MFMA baseline
  hipEvent_t t_start = nullptr, t_stop = nullptr;
  if (uni) sn_internal_timing_slot(0x400 | (segmented ? 1 : 0) | (ragged ? 2 : 0), rows, C, rows * 4 * ((int64_t)J + C), J, &t_start, &t_stop);
  const bool half = uni && bounds;
  if (bounds && ((bounds->ndy > 0 && !bounds->dybound) || bounds->ndy < 0 || !bounds->xinvstd || !(bounds->xfac > 0.f))) return SN_E_NULL;
  if (half) {
#define SN_WGH(CT_)                                                                                                                    \
  do {                                                                                                                                 \
    if (t_start)                                                                                                                       \
      hipExtLaunchKernelGGL((wgrad_h_k<CT_, false, 2>), dim3(nslab), dim3(kWgradThreads), 0, s, t_start, t_stop, 0, dy, lddy, x, ldx,  \
                            center, rows, (int)J, (int)C, partial, colpart, sr, spm, slab_off, bounds->dybound, bounds->ndy, bounds->xinvstd, \
                            bounds->xfac, 0);                                                                                          \
    else                                                                                                                               \
      hipLaunchKernelGGL((wgrad_h_k<CT_, false, 2>), dim3(nslab), dim3(kWgradThreads), 0, s, dy, lddy, x, ldx, center, rows, (int)J,   \
                         (int)C, partial, colpart, sr, spm, slab_off, bounds->dybound, bounds->ndy, bounds->xinvstd, bounds->xfac, 0); \
  } while (0)
    // one accumulator, two 32-row blocks in flight (LABNOTES r4wgrad: the other combinations measured slower)
    if (C == 256) SN_WGH(2);
    else SN_WGH(1);
#undef SN_WGH
  }
  else if (uni && C == 128 && t_start)
    hipExtLaunchKernelGGL((wgrad_u_k<1>), dim3(nslab), dim3(kWgradThreads), 0, s, t_start, t_stop, 0, dy, lddy, x, ldx, center, rows, (int)J, (int)C, partial, colpart, sr, spm, slab_off);
  else if (uni && t_start)
    hipExtLaunchKernelGGL((wgrad_u_k<2>), dim3(nslab), dim3(kWgradThreads), 0, s, t_start, t_stop, 0, dy, lddy, x, ldx, center, rows, (int)J, (int)C, partial, colpart, sr, spm, slab_off);
  else if (uni && C == 128)
    hipLaunchKernelGGL((wgrad_u_k<1>), dim3(nslab), dim3(kWgradThreads), 0, s, dy, lddy, x, ldx, center, rows, (int)J, (int)C, partial, colpart, sr, spm, slab_off);
  else if (uni)
    hipLaunchKernelGGL((wgrad_u_k<2>), dim3(nslab), dim3(kWgradThreads), 0, s, dy, lddy, x, ldx, center, rows, (int)J, (int)C, partial, colpart, sr, spm, slab_off);
  else if (C == 128)
    hipLaunchKernelGGL((wgrad_mfma_k<1>), dim3(nslab), dim3(kWG), 0, s, dy, lddy, x, ldx, center, rows, (int)J, (int)C, partial, colpart);
  else
    hipLaunchKernelGGL((wgrad_mfma_k<2>), dim3(nslab), dim3(kWG), 0, s, dy, lddy, x, ldx, center, rows, (int)J, (int)C, partial, colpart);
  const int64_t extra = (dysum ? J : 0) + (segmented ? (int64_t)(nslab / spm) * J : 0) + (ragged ? (int64_t)nseg_tab * J : 0);
  hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)((J * C + extra + 63) / 64)), dim3(kWG), 0, s, partial, nslab, (int)J, (int)C, G,
                     colpart, dysum, (segmented || ragged) ? seg_dysum : nullptr, spm, ragged ? seg_slab_ptr : nullptr,
                     (int)nseg_tab);
  return launch_status();
}

int sn_wgrad_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                 int32_t J, int32_t C, float *G, double *dysum, void *workspace, size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  return wgrad_launch(dy, lddy, x, ldx, center, rows, J, C, G, dysum, 0, nullptr, workspace, workspace_bytes, stream);
}

size_t sn_wgrad_seg_workspace_bytes(int64_t rows, int64_t rows_per_seg, int32_t J, int32_t C) {
  (void)J;
  if (C < 1 || rows_per_seg < 1 || rows < 0) return 0;
  const int64_t nseg = rows / rows_per_seg;
  int64_t spm = kCUs / (nseg > 0 ? nseg : 1);
  if (spm < 1) spm = 1;
  return (size_t)(nseg * spm) * 128 * ((size_t)C + 1) * sizeof(float);
}

int sn_wgrad_seg_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                     int64_t rows_per_seg, int32_t J, int32_t C, float *G, double *dysum, float *seg_dysum, void *workspace,
                     size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows_per_seg < 1) return SN_E_SHAPE;
  return wgrad_launch(dy, lddy, x, ldx, center, rows, J, C, G, dysum, rows_per_seg, seg_dysum, workspace, workspace_bytes, stream);
}

// Weight gradient over caller-defined row slabs (ragged meshes): slab b = rows [slab_off[b], slab_off[b+1]) — consecutive,
// none crossing a mesh boundary — and mesh m owns slabs [seg_slab_ptr[m], seg_slab_ptr[m+1]); seg_dysum receives the per-mesh
// column sums of dy ([nseg][J]).  Workspace: nslab·128·(C+1) floats.  Uniform-wave kernel only (SN_E_UNSUPPORTED otherwise).
int sn_wgrad_slabs_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                       const int64_t *slab_off, int32_t nslab, const int64_t *seg_slab_ptr, int32_t nseg, int32_t J, int32_t C,
                       float *G, double *dysum, float *seg_dysum, void *workspace, size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (!slab_off || !seg_slab_ptr) return SN_E_NULL;
  return wgrad_launch(dy, lddy, x, ldx, center, rows, J, C, G, dysum, 0, seg_dysum, workspace, workspace_bytes, stream, slab_off,
                      nslab, seg_slab_ptr, nseg);
}

// The same three products with bounds (see WgradBounds / wgrad_h_k): two fp16 pieces, half the matrix work.  dybound: one
// float on the device, >= max |dy| over the operand (sn_absmax_* leave it); xinvstd: the BatchNorm inverse standard deviations
// of x's columns about `center`, from statistics over stat_rows rows that include every row of x (the local rows, or the
// global batch under synchronised statistics).  Results agree with the bf16 forms to fp32 rounding.
static WgradBounds make_bounds(const float *dybound, int64_t n_dybound, const float *xinvstd, int64_t stat_rows) {
  // |x - mean| <= sqrt(n·var) <= sqrt(n) / invstd; the factor 1.0625 covers the rounding of the statistics and of this product
  return WgradBounds{dybound, (n_dybound < 0 || n_dybound > INT_MAX) ? -1 : (int)n_dybound, xinvstd,
                     stat_rows > 0 ? sqrtf((float)stat_rows) * 1.0625f : 0.f};
}
int sn_wgrad_bounded_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                         int32_t J, int32_t C, float *G, double *dysum, void *workspace, size_t workspace_bytes,
                         const float *dybound, int64_t n_dybound, const float *xinvstd, int64_t stat_rows, void *stream) {
  (void)hipGetLastError();
  if (stat_rows < rows) return SN_E_SHAPE;
  const WgradBounds b = make_bounds(dybound, n_dybound, xinvstd, stat_rows);
  return wgrad_launch(dy, lddy, x, ldx, center, rows, J, C, G, dysum, 0, nullptr, workspace, workspace_bytes, stream, nullptr, 0,
                      nullptr, 0, &b);
}
int sn_wgrad_seg_bounded_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                             int64_t rows_per_seg, int32_t J, int32_t C, float *G, double *dysum, float *seg_dysum,
                             void *workspace, size_t workspace_bytes, const float *dybound, int64_t n_dybound,
                             const float *xinvstd, int64_t stat_rows, void *stream) {
  (void)hipGetLastError();
  if (rows_per_seg < 1 || stat_rows < rows) return SN_E_SHAPE;
  const WgradBounds b = make_bounds(dybound, n_dybound, xinvstd, stat_rows);
  return wgrad_launch(dy, lddy, x, ldx, center, rows, J, C, G, dysum, rows_per_seg, seg_dysum, workspace, workspace_bytes, stream,
                      nullptr, 0, nullptr, 0, &b);
}
int sn_wgrad_slabs_bounded_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                               const int64_t *slab_off, int32_t nslab, const int64_t *seg_slab_ptr, int32_t nseg, int32_t J,
                               int32_t C, float *G, double *dysum, float *seg_dysum, void *workspace, size_t workspace_bytes,
                               const float *dybound, int64_t n_dybound, const float *xinvstd, int64_t stat_rows, void *stream) {
  (void)hipGetLastError();
  if (!slab_off || !seg_slab_ptr) return SN_E_NULL;
  if (stat_rows < rows) return SN_E_SHAPE;
  const WgradBounds b = make_bounds(dybound, n_dybound, xinvstd, stat_rows);
  return wgrad_launch(dy, lddy, x, ldx, center, rows, J, C, G, dysum, 0, seg_dysum, workspace, workspace_bytes, stream, slab_off,
                      nslab, seg_slab_ptr, nseg, &b);
}

static int thin_blocks(int64_t rows, int J) {
  const int lanes_r = kWG / (J / 4);
  int64_t nb = (rows + (int64_t)lanes_r * 8 - 1) / ((int64_t)lanes_r * 8);     // >= 8 rows per row lane
  if (nb > kThinBlocks) nb = kThinBlocks;
  return nb < 1 ? 1 : (int)nb;
}

size_t sn_wgrad_thin_workspace_bytes(int64_t rows, int32_t J, int32_t C) {
  if (rows < 1 || J < 4 || (J % 4) || C < 1 || C > kThinMaxC || (kWG % (J / 4))) return 0;
  return (size_t)thin_blocks(rows, J) * ((size_t)C + 1) * J * sizeof(double);
}

int sn_wgrad_thin_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, int64_t rows, int32_t J, int32_t C,
                      float *G, float *db, void *workspace, size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || J < 1 || C < 1 || lddy < J || ldx < C) return SN_E_SHAPE;
  if (C > kThinMaxC || (J % 4) || J / 4 > kWG || (kWG % (J / 4))) return SN_E_UNSUPPORTED;
  if (!G) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows == 0) {
    hipError_t e = sn_internal_fill(G, 0, (size_t)J * C * sizeof(float), s);
    if (e == hipSuccess && db) e = sn_internal_fill(db, 0, (size_t)J * sizeof(float), s);
    return e == hipSuccess ? SN_OK : (int)e;
  }
  if (!dy || !x || !workspace) return SN_E_NULL;
  if (!aligned16(dy) || (lddy % 4)) return SN_E_ALIGN;
  const int nblk = thin_blocks(rows, J);
  if (workspace_bytes < (size_t)nblk * ((size_t)C + 1) * J * sizeof(double)) return SN_E_WORKSPACE;
  double *partial = static_cast<double *>(workspace);
  const int lanes_r = kWG / (J / 4);
  const size_t shm = (size_t)lanes_r * (C + 1) * J * sizeof(float);
#define SN_THIN(CC)                                                                                                       \
  case CC:                                                                                                                \
    hipLaunchKernelGGL((wgrad_thin_k<CC>), dim3(nblk), dim3(kWG), shm, s, dy, lddy, x, ldx, rows, (int)J, partial);      \
    break;
  switch (C) {
    SN_THIN(1) SN_THIN(2) SN_THIN(3) SN_THIN(4) SN_THIN(5) SN_THIN(6) SN_THIN(7) SN_THIN(8)
    default: return SN_E_UNSUPPORTED;
  }
#undef SN_THIN
  hipLaunchKernelGGL(wgrad_thin_final_k, dim3((unsigned)(((C + 1) * J + 31) / 32)), dim3(kWG), 0, s, partial, nblk, (int)J,
                     (int)C, G, db);
  return launch_status();
}

int sn_avg_fwd_prep_f32(const float *segsum, const float *inv_count, int64_t nseg, int32_t C, int64_t rows_per_seg,
                        const double *stats1, float *m, double *stats, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nseg < 1 || C < 1 || rows_per_seg < 1) return SN_E_SHAPE;
  if (!segsum || !inv_count || !stats1 || !m || !stats) return SN_E_NULL;
  hipLaunchKernelGGL(avg_fwd_prep_k, dim3((C + kWG - 1) / kWG), dim3(kWG), 0, static_cast<hipStream_t>(stream), segsum,
                     inv_count, (int)nseg, (int)C, (double)rows_per_seg, stats1, m, stats);
  return launch_status();
}

int sn_seg_affine_f32(const float *A, int64_t nseg, int32_t K, const float *W, int64_t ldw, const float *bias, int32_t J,
                      float *out, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nseg < 1 || K < 1 || J < 1 || ldw < K) return SN_E_SHAPE;
  if (!A || !W || !out) return SN_E_NULL;
  hipLaunchKernelGGL(seg_affine_k, dim3((unsigned)((nseg * J + kWG - 1) / kWG)), dim3(kWG), 0,
                     static_cast<hipStream_t>(stream), A, (int)nseg, (int)K, W, ldw, bias, (int)J, out);
  return launch_status();
}

int sn_avg_bwd_gc_f32(const float *G1, const float *seg_dy, const float *m, const float *mu2, int64_t nseg, int32_t J,
                      int32_t C, float *Gc, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nseg < 1 || J < 1 || C < 1) return SN_E_SHAPE;
  if (!G1 || !seg_dy || !m || !mu2 || !Gc) return SN_E_NULL;
  hipLaunchKernelGGL(avg_bwd_gc_k, dim3((unsigned)((J * 2 * C + kWG - 1) / kWG)), dim3(kWG), 0,
                     static_cast<hipStream_t>(stream), G1, seg_dy, m, mu2, (int)nseg, (int)J, (int)C, Gc);
  return launch_status();
}

int sn_avg_bwd_segvec_f32(const float *seg_dy, const float *Wf2, int64_t ldw, const float *m, const float *mu2,
                          const float *B2, const float *C2, const float *inv_count, int64_t rows_per_seg, int64_t nseg,
                          int32_t J, int32_t C, float *out, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nseg < 1 || J < 1 || C < 1 || ldw < C || rows_per_seg < 1) return SN_E_SHAPE;
  if (!seg_dy || !Wf2 || !m || !mu2 || !B2 || !C2 || !inv_count || !out) return SN_E_NULL;
  hipLaunchKernelGGL(avg_bwd_segvec_k, dim3((unsigned)((nseg * C + kWG - 1) / kWG)), dim3(kWG), 0,
                     static_cast<hipStream_t>(stream), seg_dy, Wf2, ldw, m, mu2, B2, C2, inv_count, (double)rows_per_seg,
                     (int)nseg, (int)J, (int)C, out, (const int64_t *)nullptr);
  return launch_status();
}

// the same per-mesh vector for RAGGED meshes: mesh g has segoff[g+1] - segoff[g] rows
int sn_avg_bwd_segvec_ragged_f32(const float *seg_dy, const float *Wf2, int64_t ldw, const float *m, const float *mu2,
                                 const float *B2, const float *C2, const float *inv_count, const int64_t *segoff, int64_t nseg,
                                 int32_t J, int32_t C, float *out, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nseg < 1 || J < 1 || C < 1 || ldw < C) return SN_E_SHAPE;
  if (!seg_dy || !Wf2 || !m || !mu2 || !B2 || !C2 || !inv_count || !out || !segoff) return SN_E_NULL;
  hipLaunchKernelGGL(avg_bwd_segvec_k, dim3((unsigned)((nseg * C + kWG - 1) / kWG)), dim3(kWG), 0,
                     static_cast<hipStream_t>(stream), seg_dy, Wf2, ldw, m, mu2, B2, C2, inv_count, 0.0, (int)nseg, (int)J, (int)C,
                     out, segoff);
  return launch_status();
}

static int affine_cols_launch(float *dx, int64_t lddx, const float *x, int64_t ldx, const float *center, const float *B,
                              const float *Cc, int64_t rows, int32_t C, bool elu, void *stream) {
  if (rows < 0 || C < 1 || lddx < C || ldx < C) return SN_E_SHAPE;
  if (rows == 0) return SN_OK;
  if (!dx || !x || (!elu && !B) || (B && !Cc)) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (C % 4 == 0) && (lddx % 4 == 0) && (ldx % 4 == 0) && aligned16(dx) && aligned16(x) &&
                   (!B || (aligned16(B) && aligned16(Cc))) && (!center || aligned16(center));
  int64_t items = vec ? rows * (C / 4) : rows * (int64_t)C;
  int64_t blocks = (items + kWG - 1) / kWG;
#define SN_AFF(V, E) hipLaunchKernelGGL((affine_cols_acc_k<V, E>), dim3((unsigned)blocks), dim3(kWG), 0, s, dx, lddx, x, ldx, center, B, Cc, rows, (int)C, kStreamNT)
  if (vec && elu) SN_AFF(true, true);
  else if (vec) SN_AFF(true, false);
  else if (elu) SN_AFF(false, true);
  else SN_AFF(false, false);
#undef SN_AFF
  return launch_status();
}

int sn_affine_cols_acc_f32(float *dx, int64_t lddx, const float *x, int64_t ldx, const float *center, const float *B,
                           const float *Cc, int64_t rows, int32_t C, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  return affine_cols_launch(dx, lddx, x, ldx, center, B, Cc, rows, C, false, stream);
}

int sn_affine_cols_elu_bwd_f32(float *dx, int64_t lddx, const float *x, int64_t ldx, const float *center, const float *B,
                               const float *Cc, int64_t rows, int32_t C, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  return affine_cols_launch(dx, lddx, x, ldx, center, B, Cc, rows, C, true, stream);
}

int sn_bn_fold_f32(const double *stats, int64_t rows, const float *gamma, const float *beta, const float *W,
                   const float *b, int32_t J, int32_t C, double eps, double momentum, int32_t training,
                   float *running_mean, float *running_var, float *mean, float *invstd, float *s, float *t,
                   float *Wf, float *bf, int64_t *num_batches_tracked, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || J < 1 || C < 1 || C > 1024) return SN_E_SHAPE;
  if (!gamma || !beta || !W || !mean || !invstd || !s || !t || !Wf || !bf) return SN_E_NULL;
  if (training ? !stats : (!running_mean || !running_var)) return SN_E_NULL;
  hipLaunchKernelGGL(bn_fold_k, dim3(J), dim3(kWG), 0, static_cast<hipStream_t>(stream), stats, rows, gamma, beta, W,
                     b, (int)C, eps, momentum, (int)training, running_mean, running_var, mean, invstd, s, t, Wf, bf,
                     num_batches_tracked);
  return launch_status();
}

int sn_bn_fold_seg_f32(const double *stats, int64_t rows, const float *gamma, const float *beta, const float *W, const float *b,
                       int32_t J, int32_t C, double eps, double momentum, float *running_mean, float *running_var, float *mean,
                       float *invstd, float *s, float *t, float *Wf, float *bf, int64_t *num_batches_tracked, const float *seg_mean,
                       int64_t nseg, float *segbias, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 1 || J < 1 || C < 2 || (C & 1) || nseg < 1 || nseg > INT_MAX) return SN_E_SHAPE;
  if (C > 256) return SN_E_UNSUPPORTED;          // (the folded half row is held in LDS)
  if (!stats || !gamma || !beta || !W || !mean || !invstd || !s || !t || !Wf || !bf || !seg_mean || !segbias) return SN_E_NULL;
  hipLaunchKernelGGL(bn_fold_k, dim3(J), dim3(kWG), 0, static_cast<hipStream_t>(stream), stats, rows, gamma, beta, W, b, (int)C, eps,
                     momentum, 1, running_mean, running_var, mean, invstd, s, t, Wf, bf, num_batches_tracked, seg_mean, (int)nseg,
                     segbias, (int)J);
  return launch_status();
}

int sn_avg_bn_bwd_f32(const float *G1, const double *dystats, const float *seg_dy, const float *seg_mean, const float *mu2,
                      const float *W, const float *s, const float *invstd, const float *beta, int64_t rows, int32_t J, int32_t C,
                      int64_t nseg, const float *Wf2, int64_t ldw, const float *inv_count, int64_t rows_per_seg, const int64_t *segoff,
                      float *dW, float *db, float *dgamma, float *dbeta, float *Bc, float *Cc, float *segvec, void *stream) {
  (void)hipGetLastError();
  if (rows < 1 || J < 1 || C < 1 || nseg < 1 || nseg > INT_MAX || ldw < C || (!segoff && rows_per_seg < 1)) return SN_E_SHAPE;
  if (J > 128 || (C % 32)) return SN_E_UNSUPPORTED;
  if (!G1 || !dystats || !seg_dy || !seg_mean || !mu2 || !W || !s || !invstd || !beta || !Wf2 || !inv_count || !dW || !dgamma ||
      !dbeta || !Bc || !Cc || !segvec)
    return SN_E_NULL;
  // mesh chunks of the broadcast half: eight meshes per workgroup (one per 32-lane group) up to 64 chunks, more per chunk beyond
  int64_t nch = (nseg + 7) / 8;
  if (nch > 64) nch = 64;
  const int mpc = (int)(((nseg + nch - 1) / nch + 7) / 8 * 8);
  nch = (nseg + mpc - 1) / mpc;
  hipLaunchKernelGGL(avg_bn_bwd_k, dim3((unsigned)((C / 32) * (1 + nch))), dim3(kWG), 0, static_cast<hipStream_t>(stream), G1, dystats,
                     seg_dy, seg_mean, mu2, W, s, invstd, beta, rows, (int)J, (int)C, (int)nseg, Wf2, ldw, inv_count,
                     (double)rows_per_seg, segoff, mpc, dW, db, dgamma, dbeta, Bc, Cc, segvec);
  return launch_status();
}

// 1 when sn_wgrad_*bounded_f32 take the two-piece fp16 kernel (i.e. producers of dy should leave their maxima), 0 when the
// bounds would be ignored (SN_GEMM_VARIANT=0: the fp32-MFMA baseline) — the ONE place that decision is made
int32_t sn_wgrad_bounded_enabled(void) { return gemm_variant() != 0 ? 1 : 0; }

int32_t sn_colstats_blocks(int64_t rows) { return rows > 0 ? stat_blocks(rows) : 0; }

int sn_colstats_partial_f32(const float *x, int64_t ld, int64_t rows, int32_t C, double *partial, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || C < 1 || C > 4096 || ld < C) return SN_E_SHAPE;
  if (rows == 0) return SN_OK;
  if (!x || !partial) return SN_E_NULL;
  const int nblk = stat_blocks(rows);
  const bool vec = (C % 4 == 0) && (kWG % (C / 4) == 0) && (ld % 4 == 0) && aligned16(x);
  const int cw = vec ? C / 4 : C;
  if (!vec && cw > kWG) return SN_E_UNSUPPORTED;
  const size_t shm = (size_t)(kWG / cw) * 2 * C * sizeof(double);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec)
    hipLaunchKernelGGL((colstats_k<true>), dim3(nblk), dim3(kWG), shm, s, x, ld, rows, (int)C, partial);
  else
    hipLaunchKernelGGL((colstats_k<false>), dim3(nblk), dim3(kWG), shm, s, x, ld, rows, (int)C, partial);
  return launch_status();
}

int sn_bn_bwd_coeffs_f32(const float *Gc, const double *dystats, const float *W, const float *s, const float *invstd,
                         const float *beta, int64_t rows, int32_t J, int32_t C, float *dW, float *db, float *dgamma,
                         float *dbeta, float *Bc, float *Cc, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 1 || J < 1 || C < 1) return SN_E_SHAPE;
  if (!Gc || !dystats || !W || !s || !invstd || !beta || !dW || !dgamma || !dbeta || !Bc || !Cc) return SN_E_NULL;
  hipLaunchKernelGGL(bn_bwd_coeffs_k, dim3((C + 31) / 32), dim3(kWG), 0, static_cast<hipStream_t>(stream), Gc,
                     dystats, W, s, invstd, beta, rows, (int)J, (int)C, dW, db, dgamma, dbeta, Bc, Cc);
  return launch_status();
}

size_t sn_segment_colsum_workspace_bytes(int64_t rows_per_seg, int64_t nseg, int32_t C) {
  (void)rows_per_seg;
  if (nseg < 0 || C < 1) return 0;
  return (size_t)nseg * kSegSlabs * (size_t)C * sizeof(double);
}

static bool seg_shape_ok(int32_t C) { return C >= 4 && (C % 4 == 0) && (kWG % (C / 4) == 0); }

static unsigned ew_grid(int64_t items) {
  int64_t b = (items + kWG - 1) / kWG;
  return (unsigned)(b < 1 ? 1 : b);
}

int sn_segment_colsum_f32(const float *x, int64_t ld, const float *mask, int64_t rows_per_seg, int64_t nseg, int32_t C,
                          float *out, void *workspace, size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows_per_seg < 0 || nseg < 0 || C < 1 || ld < C) return SN_E_SHAPE;
  if (!seg_shape_ok(C) || (ld % 4)) return SN_E_UNSUPPORTED;
  if (nseg == 0) return SN_OK;
  if (!x || !out || !workspace) return SN_E_NULL;
  if (!aligned16(x)) return SN_E_ALIGN;
  if (workspace_bytes < sn_segment_colsum_workspace_bytes(rows_per_seg, nseg, C)) return SN_E_WORKSPACE;
  if (nseg > 65535) return SN_E_RANGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  double *partial = static_cast<double *>(workspace);
  const size_t shm = (size_t)(kWG / (C / 4)) * C * sizeof(double);
  hipLaunchKernelGGL(segsum_k, dim3(kSegSlabs, (unsigned)nseg), dim3(kWG), shm, s, x, ld, mask, rows_per_seg, (int)C, partial);
  const int64_t total = nseg * C;
  hipLaunchKernelGGL(segsum_final_k, dim3((unsigned)((total + kWG - 1) / kWG)), dim3(kWG), 0, s, partial, kSegSlabs, total,
                     (int)C, out);
  return launch_status();
}

size_t sn_avg_stats_workspace_bytes(int64_t rows_per_seg, int64_t nseg, int32_t C) {
  (void)rows_per_seg;
  if (nseg < 0 || C < 1) return 0;
  return (size_t)nseg * seg_slabs(nseg) * 3 * (size_t)C * sizeof(double);
}

int sn_avg_stats_f32(const float *e, int64_t ld, const float *mask, const float *inv_count, int64_t rows_per_seg, int64_t nseg,
                     int32_t C, float *m, double *stats, void *workspace, size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows_per_seg < 1 || nseg < 1 || C < 1 || ld < C) return SN_E_SHAPE;
  if (!seg_shape_ok(C) || (ld % 4)) return SN_E_UNSUPPORTED;
  if (!e || !inv_count || !m || !stats || !workspace) return SN_E_NULL;
  if (!aligned16(e)) return SN_E_ALIGN;
  if (workspace_bytes < sn_avg_stats_workspace_bytes(rows_per_seg, nseg, C)) return SN_E_WORKSPACE;
  if (nseg > 65535) return SN_E_RANGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  double *partial = static_cast<double *>(workspace);
  const size_t shm = (size_t)(kWG / (C / 4)) * 3 * C * sizeof(double);
  const int nslab = seg_slabs(nseg);
  hipLaunchKernelGGL(segstats_k, dim3(nslab, (unsigned)nseg), dim3(kWG), shm, s, e, ld, mask, rows_per_seg, (int)C, partial);
  hipLaunchKernelGGL(segstats_final_k, dim3((unsigned)((C + kSegFinalCols - 1) / kSegFinalCols)), dim3(kSegFinalCols * kSegFinalLanes), 0, s, partial, nslab, (int)nseg, (int)C,
                     inv_count, (double)rows_per_seg, m, stats);
  return launch_status();
}

int sn_avg_stats_from_tiles_f32(const float *tile_sums, const double *stats_part, int32_t nblk, const float *e, int64_t ld,
                                const float *mask, const float *inv_count, int64_t rows_per_seg, int64_t nseg, int32_t C,
                                float *m, double *stats, float *workspace, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows_per_seg < 1 || nseg < 1 || C != 128 || ld < C || nblk < 0) return SN_E_SHAPE;
  if (ld % 4) return SN_E_UNSUPPORTED;
  if (!tile_sums || !stats_part || !e || !inv_count || !m || !stats || !workspace) return SN_E_NULL;
  if (!aligned16(e) || !aligned16(tile_sums) || (mask && !aligned16(mask))) return SN_E_ALIGN;
  if (nseg > 65535) return SN_E_RANGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(segsum_tiles_k, dim3((unsigned)(C / 32), (unsigned)nseg), dim3(kWG), 0, s, tile_sums, e, ld, mask, rows_per_seg,
                     (int)C, workspace);
  hipLaunchKernelGGL(avg_prep_parts_k, dim3((unsigned)(C / 8)), dim3(kWG), 0, s, workspace, inv_count, (int)nseg, (int)C,
                     (double)rows_per_seg, stats_part, (int)nblk, m, stats);
  return launch_status();
}

// BatchNorm statistics (2 x 2C fp64, sn_bn_fold_f32's layout) of [e | per-mesh mean broadcast] for RAGGED meshes from the means
// (sn_segment_colsum_ragged_f32 with scale = 1 / rows), the meshes' row offsets and the statistics partials of the kernel that
// wrote e ([nblk][2][C] fp64; one block holding ready statistics works too) — one launch for what was a merge launch and six
// elementwise / reduction launches of the framework.
int sn_avg_prep_ragged_f32(const float *seg_mean, const int64_t *segoff, int64_t nseg, int32_t C, const double *stats_part,
                           int32_t nblk, double *stats, void *stream) {
  (void)hipGetLastError();
  if (nseg < 1 || nseg > INT_MAX || C < 8 || (C % 8) || nblk < 0) return SN_E_SHAPE;
  if (!seg_mean || !segoff || !stats || (nblk > 0 && !stats_part)) return SN_E_NULL;
  hipLaunchKernelGGL(avg_prep_parts_k, dim3((unsigned)(C / 8)), dim3(kWG), 0, static_cast<hipStream_t>(stream), seg_mean,
                     (const float *)nullptr, (int)nseg, (int)C, 1.0, stats_part, (int)nblk, (float *)nullptr, stats, segoff);
  return launch_status();
}

// sn_avg_stats_from_tiles_f32 for RAGGED meshes: per-mesh means (inv_count[g] = 1 / rows of mesh g) and the statistics of
// [e | mean broadcast] from the per-tile column sums and statistics partials of the GEMM that wrote e — no pass over e (the rows
// of the two tiles a mesh shares with its neighbours are read from e).
int sn_avg_stats_from_tiles_ragged_f32(const float *tile_sums, const double *stats_part, int32_t nblk, const float *e, int64_t ld,
                                       const int64_t *segoff, const float *inv_count, int64_t nseg, int32_t C, float *m,
                                       double *stats, float *workspace, void *stream) {
  (void)hipGetLastError();
  if (nseg < 1 || C != 128 || ld < C || nblk < 0) return SN_E_SHAPE;
  if (ld % 4) return SN_E_UNSUPPORTED;
  if (!tile_sums || !stats_part || !e || !segoff || !inv_count || !m || !stats || !workspace) return SN_E_NULL;
  if (!aligned16(e) || !aligned16(tile_sums)) return SN_E_ALIGN;
  if (nseg > 65535) return SN_E_RANGE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(segsum_tiles_k, dim3((unsigned)(C / 32), (unsigned)nseg), dim3(kWG), 0, s, tile_sums, e, ld, (const float *)nullptr,
                     (int64_t)0, (int)C, workspace, segoff);
  hipLaunchKernelGGL(avg_prep_parts_k, dim3((unsigned)(C / 8)), dim3(kWG), 0, s, workspace, inv_count, (int)nseg, (int)C, 1.0,
                     stats_part, (int)nblk, m, stats, segoff);
  return launch_status();
}

int sn_bcast_rows_f32(const float *src, float *dst, int64_t ldd, int64_t rows_per_seg, int64_t nseg, int32_t C,
                      void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows_per_seg < 0 || nseg < 0 || C < 1 || ldd < C) return SN_E_SHAPE;
  if ((C % 4) || (ldd % 4)) return SN_E_UNSUPPORTED;
  const int64_t rows = rows_per_seg * nseg;
  if (rows == 0) return SN_OK;
  if (!src || !dst) return SN_E_NULL;
  if (!aligned16(src) || !aligned16(dst)) return SN_E_ALIGN;
  hipLaunchKernelGGL(bcast_rows_k, dim3(ew_grid(rows * (C / 4))), dim3(kWG), 0, static_cast<hipStream_t>(stream), src, dst,
                     ldd, rows_per_seg, rows, (int)C, kStreamNT);
  return launch_status();
}

int sn_elu_bwd_bcast_f32(const float *gdst, int64_t ldg, const float *out, int64_t ldo, const float *bias,
                         const float *mask, const float *gadd, int64_t ldga, float *gsrc, int64_t ldgs,
                         int64_t rows_per_seg, int64_t nseg, int32_t C, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows_per_seg < 0 || nseg < 0 || C < 1 || ldg < C || ldo < C || ldgs < C || (gadd && ldga < C)) return SN_E_SHAPE;
  if ((C % 4) || (ldg % 4) || (ldo % 4) || (ldgs % 4) || (gadd && (ldga % 4))) return SN_E_UNSUPPORTED;
  const int64_t rows = rows_per_seg * nseg;
  if (rows == 0) return SN_OK;
  if (!gdst || !out || !bias || !gsrc) return SN_E_NULL;
  if (!aligned16(gdst) || !aligned16(out) || !aligned16(bias) || !aligned16(gsrc) || (gadd && !aligned16(gadd))) return SN_E_ALIGN;
  hipLaunchKernelGGL(elu_bwd_bcast_k, dim3(ew_grid(rows * (C / 4))), dim3(kWG), 0, static_cast<hipStream_t>(stream), gdst,
                     ldg, out, ldo, bias, mask, gadd, ldga, gsrc, ldgs, rows_per_seg, rows, (int)C, kStreamNT);
  return launch_status();
}

static int loss_blocks(int64_t items) {
  int64_t b = (items + kWG - 1) / kWG;
  if (b > kLossBlocks) b = kLossBlocks;
  return b < 1 ? 1 : (int)b;
}

size_t sn_masked_smooth_l1_workspace_bytes(int64_t rows, int32_t C) {
  (void)rows; (void)C;
  return (size_t)kLossBlocks * sizeof(double);
}

int sn_masked_smooth_l1_fwd_f32(const float *out, int64_t ldo, const float *target, int64_t ldt, const float *rowmask,
                                int64_t rows, int32_t C, double scale, float *loss, void *workspace, size_t workspace_bytes,
                                void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || C < 1 || ldo < C || ldt < C) return SN_E_SHAPE;
  if (!loss) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows == 0) {
    const hipError_t e = sn_internal_fill(loss, 0, sizeof(float), s);
    return e == hipSuccess ? SN_OK : (int)e;
  }
  if (!out || !target || !workspace) return SN_E_NULL;
  if (workspace_bytes < (size_t)kLossBlocks * sizeof(double)) return SN_E_WORKSPACE;
  const bool vec = (C % 4 == 0) && (ldo % 4 == 0) && (ldt % 4 == 0) && aligned16(out) && aligned16(target);
  const int nblk = loss_blocks(rows * (vec ? C / 4 : C));
  double *partial = static_cast<double *>(workspace);
  if (vec)
    hipLaunchKernelGGL((masked_sl1_fwd_k<true>), dim3(nblk), dim3(kWG), 0, s, out, ldo, target, ldt, rowmask, rows, (int)C, partial);
  else
    hipLaunchKernelGGL((masked_sl1_fwd_k<false>), dim3(nblk), dim3(kWG), 0, s, out, ldo, target, ldt, rowmask, rows, (int)C, partial);
  hipLaunchKernelGGL(masked_sl1_final_k, dim3(1), dim3(kWG), 0, s, partial, nblk, scale, loss);
  return launch_status();
}

int sn_masked_smooth_l1_bwd_f32(const float *out, int64_t ldo, const float *target, int64_t ldt, const float *rowmask,
                                int64_t rows, int32_t C, double scale, const float *gloss, float *gout, int64_t ldg,
                                void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || C < 1 || ldo < C || ldt < C || ldg < C) return SN_E_SHAPE;
  if (rows == 0) return SN_OK;
  if (!out || !target || !gloss || !gout) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (C % 4 == 0) && (ldo % 4 == 0) && (ldt % 4 == 0) && (ldg % 4 == 0) && aligned16(out) &&
                   aligned16(target) && aligned16(gout);
  const int64_t items = rows * (vec ? C / 4 : C);
  int64_t blocks = (items + kWG - 1) / kWG;
  if (blocks > 16 * 1024) blocks = 16 * 1024;
  if (vec)
    hipLaunchKernelGGL((masked_sl1_bwd_k<true>), dim3((unsigned)blocks), dim3(kWG), 0, s, out, ldo, target, ldt, rowmask, rows, (int)C,
                       (float)scale, gloss, gout, ldg);
  else
    hipLaunchKernelGGL((masked_sl1_bwd_k<false>), dim3((unsigned)blocks), dim3(kWG), 0, s, out, ldo, target, ldt, rowmask, rows, (int)C,
                       (float)scale, gloss, gout, ldg);
  return launch_status();
}

int sn_gather_segments_f32(const float *src, const int64_t *base, int64_t nitems, int64_t rows_per_item, int64_t row_stride,
                           int32_t len, float *out, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nitems < 0 || rows_per_item < 0 || len < 1 || row_stride < 0) return SN_E_SHAPE;
  if (nitems * rows_per_item * len == 0) return SN_OK;
  if (!src || !base || !out) return SN_E_NULL;
  const bool vec = (len % 4 == 0) && aligned16(out);
  const int64_t total = nitems * rows_per_item * (vec ? len / 4 : len);
  int64_t blocks = (total + kWG - 1) / kWG;
  if (blocks > 64 * 1024) blocks = 64 * 1024;
  if (vec)
    hipLaunchKernelGGL((gather_segments_k<4>), dim3((unsigned)blocks), dim3(kWG), 0, static_cast<hipStream_t>(stream), src, base,
                       rows_per_item, row_stride, (int)len, total, out);
  else
    hipLaunchKernelGGL((gather_segments_k<1>), dim3((unsigned)blocks), dim3(kWG), 0, static_cast<hipStream_t>(stream), src, base,
                       rows_per_item, row_stride, (int)len, total, out);
  return launch_status();
}

// sn_gather_segments_f32 into a packed batch: item i supplies rows [0, item_off[i+1] - item_off[i]) of its window (item_off:
// int64[nitems + 1] on the device, item_off[0] = 0, total_rows = item_off[nitems] — passed by the caller, who built the table).
int sn_gather_segments_ragged_f32(const float *src, const int64_t *base, const int64_t *item_off, int64_t nitems,
                                  int64_t total_rows, int64_t row_stride, int32_t len, float *out, void *stream) {
  (void)hipGetLastError();
  if (nitems < 0 || nitems > INT_MAX || total_rows < 0 || len < 1 || row_stride < 0) return SN_E_SHAPE;
  if (nitems == 0 || total_rows == 0) return SN_OK;
  if (!src || !base || !item_off || !out) return SN_E_NULL;
  const bool vec = (len % 4 == 0) && aligned16(out);
  const int64_t total = total_rows * (vec ? len / 4 : len);
  int64_t blocks = (total + kWG - 1) / kWG;
  if (blocks > 64 * 1024) blocks = 64 * 1024;
  if (vec)
    hipLaunchKernelGGL((gather_segments_ragged_k<4>), dim3((unsigned)blocks), dim3(kWG), 0, static_cast<hipStream_t>(stream), src,
                       base, item_off, (int)nitems, row_stride, (int)len, total, out);
  else
    hipLaunchKernelGGL((gather_segments_ragged_k<1>), dim3((unsigned)blocks), dim3(kWG), 0, static_cast<hipStream_t>(stream), src,
                       base, item_off, (int)nitems, row_stride, (int)len, total, out);
  return launch_status();
}

namespace {
struct PairWs {
  int pa, pb;
  unsigned *header;
  unsigned short *RA, *TA, *RB, *TB;
  float *lse_part, *gradA, *gradB;
  size_t bytes;
};
PairWs pair_ws(void *workspace, int64_t rowsA, int64_t rowsB) {
  PairWs w;
  w.pa = (int)((rowsA + 31) / 32 * 32);
  w.pb = (int)((rowsB + 31) / 32 * 32);
  char *p = static_cast<char *>(workspace);
  w.header = reinterpret_cast<unsigned *>(p);
  p += kPairHeader;
  const size_t fa = (size_t)w.pa * kPairKP * 2 * sizeof(unsigned short), fb = (size_t)w.pb * kPairKP * 2 * sizeof(unsigned short);
  w.RA = reinterpret_cast<unsigned short *>(p); p += fa;
  w.TA = reinterpret_cast<unsigned short *>(p); p += fa;
  w.RB = reinterpret_cast<unsigned short *>(p); p += fb;
  w.TB = reinterpret_cast<unsigned short *>(p); p += fb;
  w.lse_part = reinterpret_cast<float *>(p); p += (size_t)kPairMaxLseSplits * w.pa * 4 * sizeof(float);
  w.gradA = reinterpret_cast<float *>(p); p += (size_t)kPairMaxGradSplits * w.pa * kPairKP * sizeof(float);
  w.gradB = reinterpret_cast<float *>(p); p += (size_t)kPairMaxGradSplits * w.pb * kPairKP * sizeof(float);
  w.bytes = (size_t)(p - static_cast<char *>(workspace));
  return w;
}
}  // namespace

size_t sn_pair_fused_workspace_bytes(int64_t rowsA, int64_t rowsB) {
  if (rowsA < 0 || rowsB < 0 || rowsA > INT_MAX - 64 || rowsB > INT_MAX - 64) return 0;
  return pair_ws(nullptr, rowsA, rowsB).bytes;
}

int sn_pair_fused_fwd_f32(const float *FA, int64_t lda, const float *FB, int64_t ldb, const int64_t *target, int64_t NA, int64_t NB,
                          int64_t rowsA, int64_t rowsB, int32_t K, float *lse, float *rowloss, void *workspace,
                          size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (NA < 1 || NB < 1 || rowsA < NA || rowsB < NB || K < 1 || lda < K || ldb < K) return SN_E_SHAPE;
  if (K > kPairKP) return SN_E_UNSUPPORTED;
  if (rowsA > INT_MAX - 64 || rowsB > INT_MAX - 64) return SN_E_RANGE;
  if (!FA || !FB || !target || !workspace || !lse || !rowloss) return SN_E_NULL;
  if (!aligned16(workspace)) return SN_E_ALIGN;
  if (workspace_bytes < sn_pair_fused_workspace_bytes(rowsA, rowsB)) return SN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const PairWs w = pair_ws(workspace, rowsA, rowsB);
  hipError_t e = sn_internal_fill(w.header, 0, kPairHeader, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(pair_maxabs_k, dim3((unsigned)std::min<int64_t>(1024, (std::max(rowsA, rowsB) * K + 4 * kWG - 1) / (4 * kWG)), 2), dim3(kWG), 0, s, FA, lda, (int)rowsA, FB, ldb, (int)rowsB, (int)K, w.header);
  const int64_t quads = (int64_t)std::max(w.pa, w.pb) * 32;
  hipLaunchKernelGGL(pair_split_k, dim3((unsigned)((quads + kWG - 1) / kWG), 2), dim3(kWG), 0, s, FA, lda, (int)rowsA, w.pa, w.RA, w.TA, FB, ldb,
                     (int)rowsB, w.pb, w.RB, w.TB, (int)K, w.header);
  const int tilesA = (int)((NA + 31) / 32), tilesB = (int)((NB + 31) / 32);
  const int nblk = (tilesA + 3) / 4;
  const int splits = std::max(1, std::min({kPairMaxLseSplits, 512 / nblk, tilesB}));
  hipLaunchKernelGGL(pair_lse_k, dim3((unsigned)nblk, (unsigned)splits), dim3(kWG), 0, s, w.RA, w.RB, target, (int)NA, (int)NB, w.pa, w.header,
                     w.lse_part);
  hipLaunchKernelGGL(pair_combine_k, dim3((unsigned)((NA + kWG - 1) / kWG)), dim3(kWG), 0, s, w.lse_part, splits, w.pa, (int)NA, lse, rowloss);
  return launch_status();
}

int sn_pair_fused_bwd_f32(const int64_t *target, const float *lse, const float *gloss, int64_t NA, int64_t NB, int64_t rowsA,
                          int64_t rowsB, int32_t K, float *dFA, int64_t ldda, float *dFB, int64_t lddb, void *workspace,
                          size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (NA < 1 || NB < 1 || rowsA < NA || rowsB < NB || K < 1 || K > kPairKP || ldda < K || lddb < K) return SN_E_SHAPE;
  if (rowsA > INT_MAX - 64 || rowsB > INT_MAX - 64) return SN_E_RANGE;
  if (!target || !lse || !gloss || !dFA || !dFB || !workspace) return SN_E_NULL;
  if (workspace_bytes < sn_pair_fused_workspace_bytes(rowsA, rowsB)) return SN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const PairWs w = pair_ws(workspace, rowsA, rowsB);
  const int tilesA = (int)((NA + 31) / 32), tilesB = (int)((NB + 31) / 32);
  PairGradSide A{w.RA, w.RB, w.TB, w.gradA, (int)NA, (int)NB, w.pa, (tilesA + 3) / 4, 1};
  PairGradSide B{w.RB, w.RA, w.TA, w.gradB, (int)NB, (int)NA, w.pb, (tilesB + 3) / 4, 1};
  const int want = std::max(1, 512 / (A.nblk + B.nblk));
  A.splits = std::max(1, std::min({kPairMaxGradSplits, want, tilesB}));
  B.splits = std::max(1, std::min({kPairMaxGradSplits, want, tilesA}));
  constexpr size_t lds = (size_t)2 * (2 * kPairTile + 4 * 128) * sizeof(unsigned short);
  static const hipError_t attr =
      hipFuncSetAttribute(reinterpret_cast<const void *>(pair_grad_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL(pair_grad_k, dim3((unsigned)(A.nblk * A.splits + B.nblk * B.splits)), dim3(kWG), lds, s, A, B, target, lse, w.header,
                     (int)NA);
  const int64_t quads = (int64_t)std::max(rowsA, rowsB) * 32;
  hipLaunchKernelGGL(pair_reduce_k, dim3((unsigned)((quads + kWG - 1) / kWG), 2), dim3(kWG), 0, s, A, B, dFA, ldda, (int)rowsA, dFB, lddb,
                     (int)rowsB, (int)K, gloss, w.header, (int)NA);
  return launch_status();
}

int sn_pair_ce_fwd_f32(const float *S, int64_t ld, const int64_t *target, int64_t NA, int64_t NB, float *lse, float *rowloss,
                       void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (NA < 0 || NB < 1 || ld < NB) return SN_E_SHAPE;
  if (NA > INT_MAX || NB > INT_MAX) return SN_E_RANGE;
  if (NA == 0) return SN_OK;
  if (!S || !target || !lse || !rowloss) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (aligned16(S) && (ld % 4) == 0)
    hipLaunchKernelGGL((pair_ce_fwd_k<true>), dim3((unsigned)NA), dim3(kWG), 0, s, S, ld, target, (int)NB, lse, rowloss);
  else
    hipLaunchKernelGGL((pair_ce_fwd_k<false>), dim3((unsigned)NA), dim3(kWG), 0, s, S, ld, target, (int)NB, lse, rowloss);
  return launch_status();
}

int sn_pair_ce_bwd_f32(const float *S, int64_t ld, const int64_t *target, const float *lse, const float *gloss, int64_t NA,
                       int64_t NB, int64_t rows, int64_t cols, float *dS, int64_t ldd, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (NA < 0 || NB < 1 || rows < NA || cols < NB || ld < NB || ldd < cols) return SN_E_SHAPE;
  if (rows > INT_MAX || cols > INT_MAX) return SN_E_RANGE;
  if (rows == 0) return SN_OK;
  if (!S || !target || !lse || !gloss || !dS) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (aligned16(S) && aligned16(dS) && (ld % 4) == 0 && (ldd % 4) == 0)
    hipLaunchKernelGGL((pair_ce_bwd_k<true>), dim3((unsigned)rows), dim3(kWG), 0, s, S, ld, target, lse, gloss, (int)NA, (int)NB,
                       (int)cols, dS, ldd);
  else
    hipLaunchKernelGGL((pair_ce_bwd_k<false>), dim3((unsigned)rows), dim3(kWG), 0, s, S, ld, target, lse, gloss, (int)NA, (int)NB,
                       (int)cols, dS, ldd);
  return launch_status();
}

int sn_pair_argmin_f32(const float *GA, int64_t ldA, int64_t colsA, const int64_t *pa, const float *GB, int64_t ldB,
                       const int64_t *pb, int64_t NA, int64_t NB, int64_t *out, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (NA < 0 || NB < 1 || colsA < 1 || ldA < colsA || ldB < NB) return SN_E_SHAPE;
  if (NA > INT_MAX - kArgRows || NB > INT_MAX || colsA > INT_MAX) return SN_E_RANGE;
  if (NA == 0) return SN_OK;
  if (!GA || !pa || !GB || !pb || !out) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)colsA * sizeof(float);
  if (lds <= kArgLdsMax) {
    hipLaunchKernelGGL(pair_argmin_lds_k, dim3((unsigned)NA), dim3(kWG), lds + 16, s, GA, ldA, (int)colsA, pa, GB, ldB, pb, (int)NB,
                       out);
  } else {
    hipLaunchKernelGGL(pair_argmin_k, dim3((unsigned)((NA + kArgRows - 1) / kArgRows)), dim3(kWG), 0, s, GA, ldA, pa, GB, ldB, pb,
                       (int)NA, (int)NB, out);
  }
  return launch_status();
}

int sn_linear_thin_fwd_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *bias, int64_t rows,
                           int32_t C, int32_t J, float *y, int64_t ldy, float *y_elu, int64_t lde, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || J < 1 || C < 1 || ldx < C || ldw < C || (y && ldy < J) || (y_elu && lde < J)) return SN_E_SHAPE;
  if (C > kThinMaxC || (J % 4) || J / 4 > kWG || (kWG % (J / 4))) return SN_E_UNSUPPORTED;
  if (rows == 0) return SN_OK;
  if (!x || !W || (!y && !y_elu)) return SN_E_NULL;
  if ((y && (!aligned16(y) || (ldy % 4))) || (y_elu && (!aligned16(y_elu) || (lde % 4))) || (bias && !aligned16(bias)))
    return SN_E_ALIGN;
  const int lanes_r = kWG / (J / 4);
  int64_t blocks = (rows + lanes_r - 1) / lanes_r;
  if (blocks > 16 * kCUs) blocks = 16 * kCUs;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define SN_THINF(CC)                                                                                                      \
  case CC:                                                                                                                \
    hipLaunchKernelGGL((linear_thin_fwd_k<CC>), dim3((unsigned)blocks), dim3(kWG), 0, s, x, ldx, W, ldw, bias, rows, (int)J, y, ldy, \
                       y_elu, lde);                                                                                       \
    break;
  switch (C) {
    SN_THINF(1) SN_THINF(2) SN_THINF(3) SN_THINF(4) SN_THINF(5) SN_THINF(6) SN_THINF(7) SN_THINF(8)
    default: return SN_E_UNSUPPORTED;
  }
#undef SN_THINF
  return launch_status();
}

}  // extern "C"
