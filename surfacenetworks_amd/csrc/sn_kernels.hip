// sn_kernels.hip — CDNA4 (gfx950, wave64) kernels + C-ABI of the Surface-Network operator layer.
//
// Written for MI355X only: 64-lane wavefronts, 256 CUs in 8 XCDs (private 4 MiB L2 each),
// HBM3E-bound arithmetic (≈2 flop/B), so everything here is about coalescing, bytes in flight
// and L2 locality — there is no MFMA in this file on purpose.
//
// Kernel inventory (one kernel per storage form)
//   spmm_csr_rows<N,XG,YG> Y = A·X, CSR, N ∈ {16,32,64,128}: N/4 lanes own one row (float4 each), a wave owns 256/N·iters
//                          consecutive rows, its entries staged in LDS; no cross-lane reduction (a lane owns whole columns).
//   spmm_bsr4_lds          the same product for 4x4-block operators: N/4 lanes own one block row (4 output rows).
//   spmm_q3_lds            quaternion-packed Dirac operators (the default of the Dirac path).
//   spmm_rb4 / spmm_ring_k row-blocked and sliding-window Laplacian kernels.
//   spmm_csr_any           any N, one thread per output element (correctness fallback).
//   coo_to_csr / transpose / bsr4 count+fill / blockdiag concat / scan / elu helpers.
//
// Reference semantics being replaced are cited per entry point in include/sn_spmm.h.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "sn_spmm.h"

int sn_internal_cu_count();
hipError_t sn_internal_fill(void *dst, int value, size_t bytes, hipStream_t s);
hipError_t sn_internal_copy2d(void *dst, int64_t dpitch, const void *src, int64_t spitch, int64_t width, int64_t rows, hipStream_t s);

namespace {

constexpr int kWG = 256;    // 4 wavefronts per workgroup
// XCDs of the device the kernels run on (workgroup b is dispatched to XCD b % c_xcd): 8 on an MI355X in SPX mode — the value
// this constant is built with; xcd_count() below reads hipDeviceAttributeNumberOfXccs once per device and rewrites it where the
// partition mode says otherwise (4 / 2 / 1).  Only LOCALITY depends on it: every map below is a bijection for any value that
// divides the grid.
__constant__ int c_xcd = 8;
#define kCUs sn_internal_cu_count()      // compute units of the current device (256 on an MI355X in SPX mode)

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ void st4(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
__device__ __forceinline__ void st4_stream(float *p, f4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p)); }
// streaming (non-temporal) forms for data that is touched once per kernel: measured +15-30 % on 1:1 copy-like passes
constexpr int kStreamNT = 1;   // streamed-once operands use non-temporal loads/stores
__device__ __forceinline__ f4 ld4_s(const float *p, int nt) {
  return nt ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p)) : *reinterpret_cast<const f4 *>(p);
}
__device__ __forceinline__ void st4_s(float *p, f4 v, int nt) {
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
  else *reinterpret_cast<f4 *>(p) = v;
}
__device__ __forceinline__ f4 fma4(float a, f4 x, f4 acc) {
  acc.x = __builtin_fmaf(a, x.x, acc.x);
  acc.y = __builtin_fmaf(a, x.y, acc.y);
  acc.z = __builtin_fmaf(a, x.z, acc.z);
  acc.w = __builtin_fmaf(a, x.w, acc.w);
  return acc;
}

// Offset (floats) of dense row r under the (ld, group) addressing of sn_spmm.h.
template <int G, int N>
__device__ __forceinline__ int64_t row_off(int r, int64_t ld) {
  if constexpr (G == 1) return (int64_t)r * ld;
  else return (int64_t)(r >> 2) * ld + (int64_t)(r & 3) * N;
}

// XCD-aware chunk map.  Workgroup b runs on XCD b % 8; giving each XCD one contiguous eighth of the row chunks keeps the
// X rows shared by neighbouring mesh rows in ONE 4 MiB L2 instead of eight.  One workgroup per chunk: a capped persistent
// grid measured 5 % slower (late workgroups lose L2 reuse with their spatial neighbours) — the grid is a multiple of 8.
__device__ __forceinline__ int my_chunk(int nchunks) {
  const int X = c_xcd;
  const int cpx = (nchunks + X - 1) / X;
  return (blockIdx.x % X) * cpx + blockIdx.x / X;      // may be >= nchunks for the padding workgroups: callers test rows
}


// Optional fused epilogue of the LDS SpMM kernels (the backward of an ELU-activated propagation stage):
//     Y = (A·X) ∘ elu'(E) + G,    elu'(·) through the activation OUTPUT E: 1 where E > 0, E + 1 elsewhere,
// E and G laid out and addressed like Y (same row grouping, own leading dimensions); G may be NULL.  e == NULL: plain product.
struct SpmmEpi {
  const float *e;
  int64_t lde;
  const float *g;
  int64_t ldg;
  // max |Y| over the rows a workgroup (quaternion-packed kernel: [gridDim.x]) or a wave (row-blocked and sliding-window Laplacian
  // kernels: [gridDim.x * waves]) wrote — sn_spmm_q3_elubwd_absmax_f32, sn_spmm_rb4_elubwd_absmax_f32, sn_spmm_csr_ring_elubwd_absmax_f32
  float *absmax = nullptr;
};
__device__ __forceinline__ f4 fabs4(const f4 &a) {
  return f4{__builtin_fabsf(a.x), __builtin_fabsf(a.y), __builtin_fabsf(a.z), __builtin_fabsf(a.w)};
}
__device__ __forceinline__ float hmax_abs4(const f4 &a) {
  return fmaxf(fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), fmaxf(__builtin_fabsf(a.z), __builtin_fabsf(a.w)));
}
// the maximum of a non-negative float over the wave (non-negative floats order like their bit patterns), in every lane
__device__ __forceinline__ float wave_max_nonneg(float m) {
  unsigned mb = __float_as_uint(m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned other = (unsigned)__shfl_xor((int)mb, o);
    mb = other > mb ? other : mb;
  }
  return __uint_as_float(mb);
}
__device__ __forceinline__ f4 elu_bwd4(const f4 &a, const f4 &o) {
  return f4{a.x * (o.x > 0.f ? 1.f : o.x + 1.f), a.y * (o.y > 0.f ? 1.f : o.y + 1.f), a.z * (o.z > 0.f ? 1.f : o.z + 1.f),
            a.w * (o.w > 0.f ? 1.f : o.w + 1.f)};
}

// ------------------------------------------------------------------------------------------------
// CSR SpMM, several row passes per wave ("rows" kernel) — the Laplacian kernel (N = 128: the products at utils_pt.py:167,176).
//
// N/4 lanes per row, a float4 column slice per lane.  A wave owns R = P*iters consecutive rows (P = 256/N): ONE coalesced load
// fetches its R+1 row pointers, the whole entry run of those rows goes to the wave's LDS slice with a few DMA instructions,
// and the wave then walks its passes with LDS reads, gathers and stores only.  Measured on the Laplacian batches (N = 128):
// +3 % at 2 passes, slower from 8 passes on (the row ranges the resident waves walk concurrently outgrow the XCD's L2) — the
// leading round trips are NOT what bounds this product; the gather stream through the 32 KiB vector cache is (PMC,
// DESIGN.md §4), which is what the RB4 form below attacks.  This kernel remains the generic CSR path and carries the
// statistics epilogue.  Same k-ascending FMA chain per row as every other CSR kernel (bit-identical results).  A wave whose
// entry run exceeds the LDS slice (rows far longer than a mesh operator's) stages each pass's entries in tiles instead.
// STATS (N = 128, YG = 1): the workgroup also leaves the column sums / sums of squares of its output rows in
// stats_part[blockIdx.x][2][128] (fp32 over <= 256 rows, combined in fp64 by spmm_stats_reduce_k) — the BatchNorm statistics
// of the propagated half [e | L·e] of a Laplacian stage, so that no statistics pass reads it back.
// ------------------------------------------------------------------------------------------------
template <int N, int XG, int YG, bool EPI, bool STATS>
__device__ __forceinline__ void spmm_csr_rows_body(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                   const float *__restrict__ vals, int M, const float *__restrict__ X,
                                                   int64_t ldx, float *__restrict__ Y, int64_t ldy, int nchunks, int iters,
                                                   SpmmEpi epi, float *__restrict__ stats_part) {
  constexpr int LPR = N / 4;          // lanes per row
  constexpr int P = 64 / LPR;         // rows per pass
  constexpr int WAVES = kWG / 64;
  constexpr int CAP = 512;            // entries of a wave's rows held in LDS (4 KiB per wave)
  constexpr int KB = 8;               // gathers in flight per lane
  __shared__ int s_col[WAVES][CAP];
  __shared__ float s_val[WAVES][CAP];
  __shared__ int s_rp[WAVES][68];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane % LPR, grp = lane / LPR;
  const float *xb = X + sub * 4;
  int *sc = s_col[wave];
  float *sv = s_val[wave];
  int *rp = s_rp[wave];
  const int R = P * iters;                                        // rows of this wave (<= 64)
  const int r0 = (my_chunk(nchunks) * WAVES + wave) * R;
  if constexpr (!STATS) {
    if (r0 >= M) return;                                          // wave-uniform
  }
  {
    int rl = r0 + lane;
    rl = rl < M ? rl : M;
    rp[lane] = rowptr[rl];
    int re = r0 + R;
    re = re < M ? re : M;
    if (lane == 0) rp[64] = rowptr[r0 < M ? re : M];
  }
  __builtin_amdgcn_wave_barrier();
  const int k0 = rp[0];
  const int k1 = R < 64 ? rp[R] : rp[64];
  const bool fast = (k1 - k0) <= CAP;                             // wave-uniform
  if (fast) {
    for (int p0 = 0; p0 < k1 - k0; p0 += 64) {
      int p = p0 + lane;
      p = p < k1 - k0 ? p : k1 - k0 - 1;                          // tail lanes re-read the last entry into spare slots
      __builtin_amdgcn_global_load_lds(colind + k0 + p, sc + p0, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(vals + k0 + p, sv + p0, 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  f4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = ssum;
  for (int i = 0; i < iters; ++i) {
    const int lr = i * P + grp;
    const int r = r0 + lr;
    const int kb = rp[lr];
    const int ke = (lr + 1 < 64) ? rp[lr + 1] : rp[64];
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (fast) {
      for (int k = kb; k < ke; k += KB) {
        int c[KB];
        float a[KB];
        f4 x[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          const bool in = k + j < ke;
          const int o = (in ? k + j : ke - 1) - k0;
          c[j] = sc[o];
          a[j] = in ? sv[o] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < KB; ++j) x[j] = ld4(xb + row_off<XG, N>(c[j], ldx));
#pragma unroll
        for (int j = 0; j < KB; ++j) acc = fma4(a[j], x[j], acc);
      }
    } else {
      // entries of this pass's P rows, in tiles of CAP through the same LDS slice
      const int pk0 = __builtin_amdgcn_readfirstlane(kb);
      const int pk1 = __builtin_amdgcn_readlane(ke, 63);
      for (int t0 = pk0; t0 < pk1; t0 += CAP) {
        const int nt = (pk1 - t0) < CAP ? (pk1 - t0) : CAP;
        for (int p0 = 0; p0 < nt; p0 += 64) {
          int p = p0 + lane;
          p = p < nt ? p : nt - 1;
          __builtin_amdgcn_global_load_lds(colind + t0 + p, sc + p0, 4, 0, 0);
          __builtin_amdgcn_global_load_lds(vals + t0 + p, sv + p0, 4, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        int k = kb > t0 ? kb : t0;
        const int kend = ke < t0 + nt ? ke : t0 + nt;
        for (; k < kend; k += KB) {
          int c[KB];
          float a[KB];
          f4 x[KB];
#pragma unroll
          for (int j = 0; j < KB; ++j) {
            const bool in = k + j < kend;
            const int o = (in ? k + j : kend - 1) - t0;
            c[j] = sc[o];
            a[j] = in ? sv[o] : 0.f;
          }
#pragma unroll
          for (int j = 0; j < KB; ++j) x[j] = ld4(xb + row_off<XG, N>(c[j], ldx));
#pragma unroll
          for (int j = 0; j < KB; ++j) acc = fma4(a[j], x[j], acc);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (r < M) {
      if constexpr (EPI) {
        acc = elu_bwd4(acc, ld4_s(epi.e + row_off<YG, N>(r, epi.lde) + sub * 4, kStreamNT));
        if (epi.g) acc += ld4_s(epi.g + row_off<YG, N>(r, epi.ldg) + sub * 4, kStreamNT);
      }
      st4_stream(Y + row_off<YG, N>(r, ldy) + sub * 4, acc);
      if constexpr (STATS) {
        ssum += acc;
        ssq.x = __builtin_fmaf(acc.x, acc.x, ssq.x); ssq.y = __builtin_fmaf(acc.y, acc.y, ssq.y);
        ssq.z = __builtin_fmaf(acc.z, acc.z, ssq.z); ssq.w = __builtin_fmaf(acc.w, acc.w, ssq.w);
      }
    }
  }
  if constexpr (STATS) {
    static_assert(N == 128 && YG == 1 && !EPI, "statistics: 128-column rows in the plain row-major layout");
    // lane (grp, sub) holds the sums of columns 4*sub .. 4*sub+3 over its rows: [wave][grp][sum | squares][128] -> one
    // (2 x 128) partial per workgroup, added in a fixed order
    __shared__ float s_st[WAVES * P][256];
    float *st = s_st[wave * P + grp];
    *reinterpret_cast<f4 *>(st + sub * 4) = ssum;
    *reinterpret_cast<f4 *>(st + 128 + sub * 4) = ssq;
    __syncthreads();
    const int t = threadIdx.x;
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES * P; ++w) tot += s_st[w][t];
    stats_part[(int64_t)blockIdx.x * 256 + t] = tot;
  }
}
template <int N, int XG, int YG>
__global__ __launch_bounds__(kWG) void spmm_csr_rows(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                     const float *__restrict__ vals, int M, const float *__restrict__ X,
                                                     int64_t ldx, float *__restrict__ Y, int64_t ldy, int nchunks, int iters) {
  spmm_csr_rows_body<N, XG, YG, false, false>(rowptr, colind, vals, M, X, ldx, Y, ldy, nchunks, iters,
                                              SpmmEpi{nullptr, 0, nullptr, 0}, nullptr);
}
template <int N, int XG, int YG>
__global__ __launch_bounds__(kWG) void spmm_csr_rows_epi(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                         const float *__restrict__ vals, int M, const float *__restrict__ X,
                                                         int64_t ldx, float *__restrict__ Y, int64_t ldy, int nchunks, int iters,
                                                         SpmmEpi epi) {
  spmm_csr_rows_body<N, XG, YG, true, false>(rowptr, colind, vals, M, X, ldx, Y, ldy, nchunks, iters, epi, nullptr);
}
template <int XG>
__global__ __launch_bounds__(kWG) void spmm_csr_rows_stats(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                           const float *__restrict__ vals, int M, const float *__restrict__ X,
                                                           int64_t ldx, float *__restrict__ Y, int64_t ldy, int nchunks, int iters,
                                                           float *__restrict__ stats_part) {
  spmm_csr_rows_body<128, XG, 1, false, true>(rowptr, colind, vals, M, X, ldx, Y, ldy, nchunks, iters,
                                              SpmmEpi{nullptr, 0, nullptr, 0}, stats_part);
}

// ------------------------------------------------------------------------------------------------
// BSR4 SpMM with the operator tile staged through LDS.
//
// N/4 lanes per block row; each lane keeps the 4 output rows of its column slice.  Blocks re-read from global memory by
// every lane group (the round-1 kernel) are 16-byte loads whose address is shared by the N/4 lanes of the group: each occupies
// the texture-addresser like a full 1 KiB gather but delivers 128 unique bytes (rocprof: SQ_WAIT_INST_ANY 39 % — vector-memory-
// ISSUE bound, not HBM bound; LABNOTES).  Here each WAVE copies the contiguous run of blocks owned by its 256/N block rows
// (b_vals[16*k0 .. 16*k1), b_colind[k0..k1)) into its private LDS slice with fully coalesced 16-byte
// loads, and the lane groups then read their blocks back with broadcast ds_read_b128.  Waves stay
// independent (no workgroup barrier): DS operations of one wave complete in order.
// ------------------------------------------------------------------------------------------------
template <int N, int XG, int YG, bool EPI>
__device__ __forceinline__ void spmm_bsr4_lds_body(const int *__restrict__ b_rowptr,
                                                   const int *__restrict__ b_colind,
                                                   const float *__restrict__ b_vals, int Mb,
                                                   const float *__restrict__ X, int64_t ldx,
                                                   float *__restrict__ Y, int64_t ldy, int nchunks, SpmmEpi epi) {
  constexpr int LPR = N / 4;          // lanes per block row
  constexpr int RPW = 64 / LPR;       // block rows per wave pass
  constexpr int WAVES = kWG / 64;
  constexpr int TILE = 64;            // blocks staged per wave per tile (4 KiB of values)
  __shared__ f4 s_vals[WAVES][TILE * 4];
  __shared__ int s_col[WAVES][TILE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane % LPR, grp = lane / LPR;
  const int64_t xq = (XG == 4) ? ldx : 4 * ldx;
  const int64_t xs = (XG == 4) ? (int64_t)N : ldx;
  const int64_t yq = (YG == 4) ? ldy : 4 * ldy;
  const int64_t ys = (YG == 4) ? (int64_t)N : ldy;
  const float *xb = X + sub * 4;
  const f4 *gv = reinterpret_cast<const f4 *>(b_vals);
  f4 *sv = s_vals[wave];
  int *sc = s_col[wave];
  {
    const int r0 = (my_chunk(nchunks) * WAVES + wave) * RPW;  // first block row of this wave (a chunk = WAVES * RPW block rows)
    if (r0 >= Mb) return;                                     // wave-uniform
    const int br = r0 + grp;
    const int brc = br < Mb ? br : Mb;
    const int kb = b_rowptr[brc];
    const int ke = b_rowptr[brc + 1 <= Mb ? brc + 1 : Mb];
    const int k0 = __builtin_amdgcn_readfirstlane(kb);
    const int k1 = __builtin_amdgcn_readlane(ke, 63);
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (int t0 = k0; t0 < k1; t0 += TILE) {
      const int nt = (k1 - t0) < TILE ? (k1 - t0) : TILE;
      // ---- stage nt blocks: 4*nt contiguous 16-byte pieces + nt column indices ----
      for (int p0 = 0; p0 < 4 * nt; p0 += 64) {
        int p = p0 + lane;
        p = p < 4 * nt ? p : 4 * nt - 1;        // tail lanes re-read the last piece into spare LDS slots
        __builtin_amdgcn_global_load_lds(gv + (int64_t)t0 * 4 + p, sv + p0, 16, 0, 0);
      }
      if (lane < nt) sc[lane] = b_colind[t0 + lane];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      // ---- my blocks inside this tile ----
      int k = kb > t0 ? kb : t0;
      const int kend = ke < t0 + nt ? ke : t0 + nt;
#pragma unroll 2
      for (; k < kend; ++k) {
        const int o = k - t0;
        const int bc = sc[o];
        const f4 a0 = sv[4 * o], a1 = sv[4 * o + 1], a2 = sv[4 * o + 2], a3 = sv[4 * o + 3];
        const float *xp = xb + (int64_t)bc * xq;
        const f4 x0 = ld4(xp), x1 = ld4(xp + xs), x2 = ld4(xp + 2 * xs), x3 = ld4(xp + 3 * xs);
        acc0 = fma4(a0.x, x0, acc0); acc0 = fma4(a0.y, x1, acc0); acc0 = fma4(a0.z, x2, acc0); acc0 = fma4(a0.w, x3, acc0);
        acc1 = fma4(a1.x, x0, acc1); acc1 = fma4(a1.y, x1, acc1); acc1 = fma4(a1.z, x2, acc1); acc1 = fma4(a1.w, x3, acc1);
        acc2 = fma4(a2.x, x0, acc2); acc2 = fma4(a2.y, x1, acc2); acc2 = fma4(a2.z, x2, acc2); acc2 = fma4(a2.w, x3, acc2);
        acc3 = fma4(a3.x, x0, acc3); acc3 = fma4(a3.y, x1, acc3); acc3 = fma4(a3.z, x2, acc3); acc3 = fma4(a3.w, x3, acc3);
      }
      __builtin_amdgcn_wave_barrier();          // all reads of this tile done before it is overwritten
    }
    if constexpr (EPI) {
      if (br < Mb) {                  // operands read here, not before the loop: the plain kernel's register budget stays
        const int64_t eq = (YG == 4) ? epi.lde : 4 * epi.lde, es = (YG == 4) ? (int64_t)N : epi.lde;
        const float *ep = epi.e + (int64_t)br * eq + sub * 4;
        const f4 e0 = ld4_s(ep, kStreamNT), e1 = ld4_s(ep + es, kStreamNT), e2 = ld4_s(ep + 2 * es, kStreamNT),
                 e3 = ld4_s(ep + 3 * es, kStreamNT);
        acc0 = elu_bwd4(acc0, e0); acc1 = elu_bwd4(acc1, e1); acc2 = elu_bwd4(acc2, e2); acc3 = elu_bwd4(acc3, e3);
        if (epi.g) {
          const int64_t gq = (YG == 4) ? epi.ldg : 4 * epi.ldg, gs = (YG == 4) ? (int64_t)N : epi.ldg;
          const float *gp = epi.g + (int64_t)br * gq + sub * 4;
          acc0 += ld4_s(gp, kStreamNT); acc1 += ld4_s(gp + gs, kStreamNT);
          acc2 += ld4_s(gp + 2 * gs, kStreamNT); acc3 += ld4_s(gp + 3 * gs, kStreamNT);
        }
      }
    }
    if (br < Mb) {
      float *yp = Y + (int64_t)br * yq + sub * 4;
      st4_stream(yp, acc0);
      st4_stream(yp + ys, acc1);
      st4_stream(yp + 2 * ys, acc2);
      st4_stream(yp + 3 * ys, acc3);
    }
  }
}
// Four waves per SIMD (= 4 workgroups per CU), on purpose: left alone the compiler fits the kernel into 80 VGPRs and 6
// workgroups per CU, whose larger combined working set re-reads 15-20 % of X from HBM on the vertex-row products
// (PMC TCC_EA0_RDREQ: 519 MB against the compulsory 451 MB).
#define SN_FOUR_WAVES __attribute__((amdgpu_waves_per_eu(4, 4)))
#define SN_SIX_WAVES __attribute__((amdgpu_waves_per_eu(6, 6)))
#define SN_FIVE_WAVES __attribute__((amdgpu_waves_per_eu(5, 5)))      // (the statistics variant's LDS transposition fits 5 workgroups per CU)
template <int N, int XG, int YG>
__global__ __launch_bounds__(kWG) SN_FOUR_WAVES void spmm_bsr4_lds(const int *__restrict__ b_rowptr, const int *__restrict__ b_colind,
                                                     const float *__restrict__ b_vals, int Mb,
                                                     const float *__restrict__ X, int64_t ldx,
                                                     float *__restrict__ Y, int64_t ldy, int nchunks) {
  spmm_bsr4_lds_body<N, XG, YG, false>(b_rowptr, b_colind, b_vals, Mb, X, ldx, Y, ldy, nchunks, SpmmEpi{nullptr, 0, nullptr, 0});
}
template <int N, int XG, int YG>
__global__ __launch_bounds__(kWG) SN_FOUR_WAVES void spmm_bsr4_lds_epi(const int *__restrict__ b_rowptr, const int *__restrict__ b_colind,
                                                         const float *__restrict__ b_vals, int Mb,
                                                         const float *__restrict__ X, int64_t ldx,
                                                         float *__restrict__ Y, int64_t ldy, int nchunks, SpmmEpi epi) {
  spmm_bsr4_lds_body<N, XG, YG, true>(b_rowptr, b_colind, b_vals, Mb, X, ldx, Y, ldy, nchunks, epi);
}

// ------------------------------------------------------------------------------------------------
// Quaternion-packed Dirac SpMM ("Q3").  Every 4x4 block of Di, DiA and their transposes is the matrix of a multiplication
// by a PURE quaternion (src/utils/mesh.py:28-33,55-58: -Q(0,e)/(2 Af), its transpose times Af/Av, ...):
//        M(p) = [[ 0,  p1,  p2,  p3], [-p1, 0,  p3, -p2], [-p2, -p3, 0,  p1], [-p3,  p2, -p1, 0]]
// so a block is three floats.  The packed record is one 16-byte word (p1, p2, p3, block column as int bits): the operator
// stream shrinks from 68 to 16 bytes per block (21 % -> 6 % of the kernel's HBM traffic at 128 channels) and a block costs
// 48 instead of 64 FMAs per lane.  Same k-ascending FMA order as the CSR oracle (the diagonal zero and the explicit zeros
// of the BSR4 form contribute fma(0, x, acc) = acc), so results stay bit-identical for finite X.
// Work decomposition, LDS staging (one 1 KiB DMA instruction per 64 blocks) and epilogue as spmm_bsr4_lds.
// ------------------------------------------------------------------------------------------------
// STATS (N = 32, YG = 4, i.e. the (rows/4, 128) view of a 128-channel tensor): the workgroup also leaves the column sums and
// sums of squares of its 32 output rows — per channel c = 32·(component) + column — in stats_part[blockIdx.x][2][128] (fp32
// over the 32 rows; added up in fp64 by sn_spmm_stats_reduce): the BatchNorm statistics of the propagated half of a stage's
// concat buffer, so that no statistics pass has to read it back.
template <int N, int XG, int YG, bool EPI, bool STATS = false, int UNROLL = 3>
__device__ __forceinline__ void spmm_q3_lds_body(const int *__restrict__ b_rowptr, const f4 *__restrict__ q_blk, int Mb,
                                                 const float *__restrict__ X, int64_t ldx, float *__restrict__ Y,
                                                 int64_t ldy, int nchunks, SpmmEpi epi, float *__restrict__ stats_part = nullptr) {
  constexpr int LPR = N / 4;          // lanes per block row
  constexpr int RPW = 64 / LPR;       // block rows per wave pass
  constexpr int WAVES = kWG / 64;
  constexpr int TILE = 64;            // blocks staged per wave per tile (1 KiB)
  __shared__ f4 s_blk[WAVES][TILE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane % LPR, grp = lane / LPR;
  const int64_t xq = (XG == 4) ? ldx : 4 * ldx;
  const int64_t xs = (XG == 4) ? (int64_t)N : ldx;
  const int64_t yq = (YG == 4) ? ldy : 4 * ldy;
  const int64_t ys = (YG == 4) ? (int64_t)N : ldy;
  const float *xb = X + sub * 4;
  f4 *sv = s_blk[wave];
  const int r0 = (my_chunk(nchunks) * WAVES + wave) * RPW;    // first block row of this wave
  const bool stays = STATS || (EPI && epi.absmax != nullptr);  // the workgroup meets again after the product (uniform)
  if (!stays) {
    if (r0 >= Mb) return;                                     // wave-uniform
  }
  const int br = r0 + grp;
  const int brc = br < Mb ? br : Mb;
  const int kb = b_rowptr[brc];
  const int ke = b_rowptr[brc + 1 <= Mb ? brc + 1 : Mb];
  const int k0 = __builtin_amdgcn_readfirstlane(kb);
  const int k1 = (stays && r0 >= Mb) ? k0 : __builtin_amdgcn_readlane(ke, 63);   // (a wave past the end has no blocks)
  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  for (int t0 = k0; t0 < k1; t0 += TILE) {
    const int nt = (k1 - t0) < TILE ? (k1 - t0) : TILE;
    {
      const int p = lane < nt ? lane : nt - 1;                // tail lanes re-read the last record into spare LDS slots
      __builtin_amdgcn_global_load_lds(q_blk + t0 + p, sv, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int k = kb > t0 ? kb : t0;
    const int kend = ke < t0 + nt ? ke : t0 + nt;
#pragma unroll UNROLL
    for (; k < kend; ++k) {
      const f4 q = sv[k - t0];
      const int bc = __float_as_int(q.w);
      const float *xp = xb + (int64_t)bc * xq;
      const f4 x0 = ld4(xp), x1 = ld4(xp + xs), x2 = ld4(xp + 2 * xs), x3 = ld4(xp + 3 * xs);
      acc0 = fma4(q.x, x1, acc0);  acc0 = fma4(q.y, x2, acc0);  acc0 = fma4(q.z, x3, acc0);
      acc1 = fma4(-q.x, x0, acc1); acc1 = fma4(q.z, x2, acc1);  acc1 = fma4(-q.y, x3, acc1);
      acc2 = fma4(-q.y, x0, acc2); acc2 = fma4(-q.z, x1, acc2); acc2 = fma4(q.x, x3, acc2);
      acc3 = fma4(-q.z, x0, acc3); acc3 = fma4(q.y, x1, acc3);  acc3 = fma4(-q.x, x2, acc3);
    }
    __builtin_amdgcn_wave_barrier();            // all reads of this tile done before it is overwritten
  }
  if constexpr (EPI) {
    if (br < Mb) {
      const int64_t eq = (YG == 4) ? epi.lde : 4 * epi.lde, es = (YG == 4) ? (int64_t)N : epi.lde;
      const float *ep = epi.e + (int64_t)br * eq + sub * 4;
      const f4 e0 = ld4_s(ep, kStreamNT), e1 = ld4_s(ep + es, kStreamNT), e2 = ld4_s(ep + 2 * es, kStreamNT),
               e3 = ld4_s(ep + 3 * es, kStreamNT);
      acc0 = elu_bwd4(acc0, e0); acc1 = elu_bwd4(acc1, e1); acc2 = elu_bwd4(acc2, e2); acc3 = elu_bwd4(acc3, e3);
      if (epi.g) {
        const int64_t gq = (YG == 4) ? epi.ldg : 4 * epi.ldg, gs = (YG == 4) ? (int64_t)N : epi.ldg;
        const float *gp = epi.g + (int64_t)br * gq + sub * 4;
        acc0 += ld4_s(gp, kStreamNT); acc1 += ld4_s(gp + gs, kStreamNT);
        acc2 += ld4_s(gp + 2 * gs, kStreamNT); acc3 += ld4_s(gp + 3 * gs, kStreamNT);
      }
    }
  }
  if (br < Mb) {
    float *yp = Y + (int64_t)br * yq + sub * 4;
    st4_stream(yp, acc0);
    st4_stream(yp + ys, acc1);
    st4_stream(yp + 2 * ys, acc2);
    st4_stream(yp + 3 * ys, acc3);
  }
  if constexpr (EPI) {
    if (epi.absmax) {                 // max |Y| over this workgroup's rows (non-negative floats order like their bit patterns)
      __shared__ float s_am[WAVES];
      float m = 0.f;
      if (br < Mb) {
        const f4 a = fabs4(acc0), b = fabs4(acc1), c = fabs4(acc2), d = fabs4(acc3);
        m = fmaxf(fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w))),
                  fmaxf(fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w)), fmaxf(fmaxf(d.x, d.y), fmaxf(d.z, d.w))));
      }
      unsigned mb = __float_as_uint(m);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)mb, o);
        mb = other > mb ? other : mb;
      }
      if (lane == 0) s_am[wave] = __uint_as_float(mb);
      __syncthreads();
      if (threadIdx.x == 0) {
        float t = s_am[0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) t = fmaxf(t, s_am[w]);
        epi.absmax[blockIdx.x] = t;
      }
    }
  }
  if constexpr (STATS) {
    static_assert((N == 32 || N == 16) && YG == 4 && !EPI, "statistics: 128- or 64-channel rows in the group-4 layout");
    // wave-private transposition: lane (grp, sub) writes its 16 channel values of block row grp (channel = N·component + 4·sub
    // + j; row stride ST floats: the block rows a 16-lane write phase touches land on disjoint banks), then every lane sums
    // its channel(s) — two at N = 32 (128 channels), one at N = 16 (64) — over the RPW block rows of the pass
    constexpr int CH = 4 * N;                   // channels
    constexpr int ST = N == 32 ? 160 : 80;
    __shared__ float s_st[WAVES][RPW * ST];
    __shared__ float s_wv[WAVES][2 * CH];
    float *st = s_st[wave];
    const bool live = br < Mb;
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    f4 *row = reinterpret_cast<f4 *>(st + grp * ST + sub * 4);
    row[0] = live ? acc0 : z;
    row[N / 4] = live ? acc1 : z;
    row[2 * N / 4] = live ? acc2 : z;
    row[3 * N / 4] = live ? acc3 : z;
    __builtin_amdgcn_wave_barrier();            // (same-wave LDS operations complete in order)
    if constexpr (N == 32) {
      float sa = 0.f, sb = 0.f, qa = 0.f, qb = 0.f;
#pragma unroll
      for (int g = 0; g < RPW; ++g) {
        const float2 v = *reinterpret_cast<const float2 *>(st + g * ST + 2 * lane);
        sa += v.x; sb += v.y;
        qa = __builtin_fmaf(v.x, v.x, qa); qb = __builtin_fmaf(v.y, v.y, qb);
      }
      s_wv[wave][2 * lane] = sa; s_wv[wave][2 * lane + 1] = sb;
      s_wv[wave][CH + 2 * lane] = qa; s_wv[wave][CH + 2 * lane + 1] = qb;
    } else {
      float sa = 0.f, qa = 0.f;
#pragma unroll
      for (int g = 0; g < RPW; ++g) {
        const float v = st[g * ST + lane];
        sa += v;
        qa = __builtin_fmaf(v, v, qa);
      }
      s_wv[wave][lane] = sa;
      s_wv[wave][CH + lane] = qa;
    }
    __syncthreads();
    const int t = threadIdx.x;                  // [sums | squares] x CH channels; waves added in order
    if (t < 2 * CH) stats_part[(int64_t)blockIdx.x * (2 * CH) + t] = (s_wv[0][t] + s_wv[1][t]) + (s_wv[2][t] + s_wv[3][t]);
  }
}
// Two launch shapes of every Q3 kernel (measured on the config-5 and config-3 batches, profiles/r2_q3_variants.txt):
//   deep  4 waves per SIMD, 3 blocks (12 gathers) in flight per lane — the VERTEX-output products (DiA, Di^T: ~6 blocks per
//         block row, output half the size of the input): at 6 waves the larger combined working set re-reads X from HBM;
//   wide  6 waves per SIMD, 2 blocks in flight — the FACE-output products (Di, DiA^T: 3 blocks per block row, output twice
//         the input, i.e. write-heavy): more waves hide the store stream; 0.84 -> 0.91 and 0.80 -> 0.92 of the roofline.
// The launcher picks by the operator's blocks per block row (<= 4: wide).
template <int N, int XG, int YG>
__global__ __launch_bounds__(kWG) SN_FOUR_WAVES void spmm_q3_lds_stats(const int *__restrict__ b_rowptr,
                                                                       const f4 *__restrict__ q_blk, int Mb,
                                                                       const float *__restrict__ X, int64_t ldx,
                                                                       float *__restrict__ Y, int64_t ldy, int nchunks,
                                                                       float *__restrict__ stats_part) {
  spmm_q3_lds_body<N, XG, YG, false, true, 3>(b_rowptr, q_blk, Mb, X, ldx, Y, ldy, nchunks, SpmmEpi{nullptr, 0, nullptr, 0},
                                              stats_part);
}
template <int N, int XG, int YG>
__global__ __launch_bounds__(kWG) SN_FIVE_WAVES void spmm_q3_lds_stats_wide(const int *__restrict__ b_rowptr,
                                                                           const f4 *__restrict__ q_blk, int Mb,
                                                                           const float *__restrict__ X, int64_t ldx,
                                                                           float *__restrict__ Y, int64_t ldy, int nchunks,
                                                                           float *__restrict__ stats_part) {
  spmm_q3_lds_body<N, XG, YG, false, true, 2>(b_rowptr, q_blk, Mb, X, ldx, Y, ldy, nchunks, SpmmEpi{nullptr, 0, nullptr, 0},
                                              stats_part);
}
// stage 1 of the reduction of those partials: workgroup b adds rows b, b + gridDim.x, ... (fixed order) in fp64
__global__ __launch_bounds__(kWG) void spmm_stats_reduce_k(const float *__restrict__ part, int64_t n,
                                                           double *__restrict__ out /* [gridDim.x][w] */, int w = 256) {
  if ((int)threadIdx.x >= w) return;            // (w = 256: 128 channels; 128: the 64-channel products)
  double t = 0.0;
  int64_t r = blockIdx.x;
  // sixteen, then four loads in flight, added in row order (the 19 600 partial rows of a config-3 product are 153 per
  // workgroup: 10 round trips to the partials instead of 38)
  for (; r + 15 * (int64_t)gridDim.x < n; r += 16 * (int64_t)gridDim.x) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = part[(r + u * (int64_t)gridDim.x) * w + threadIdx.x];
#pragma unroll
    for (int u = 0; u < 16; ++u) t += (double)v[u];
  }
  for (; r + 3 * (int64_t)gridDim.x < n; r += 4 * (int64_t)gridDim.x) {
    const float a = part[r * w + threadIdx.x], b = part[(r + gridDim.x) * w + threadIdx.x],
                c = part[(r + 2 * (int64_t)gridDim.x) * w + threadIdx.x],
                d = part[(r + 3 * (int64_t)gridDim.x) * w + threadIdx.x];
    t += (double)a; t += (double)b; t += (double)c; t += (double)d;
  }
  for (; r < n; r += gridDim.x) t += (double)part[r * w + threadIdx.x];
  out[(int64_t)blockIdx.x * w + threadIdx.x] = t;
}
template <int N, int XG, int YG>
__global__ __launch_bounds__(kWG) SN_FOUR_WAVES void spmm_q3_lds(const int *__restrict__ b_rowptr, const f4 *__restrict__ q_blk,
                                                                 int Mb, const float *__restrict__ X, int64_t ldx,
                                                                 float *__restrict__ Y, int64_t ldy, int nchunks) {
  spmm_q3_lds_body<N, XG, YG, false, false, 3>(b_rowptr, q_blk, Mb, X, ldx, Y, ldy, nchunks, SpmmEpi{nullptr, 0, nullptr, 0});
}
template <int N, int XG, int YG>
__global__ __launch_bounds__(kWG) SN_SIX_WAVES void spmm_q3_lds_wide(const int *__restrict__ b_rowptr, const f4 *__restrict__ q_blk,
                                                                     int Mb, const float *__restrict__ X, int64_t ldx,
                                                                     float *__restrict__ Y, int64_t ldy, int nchunks) {
  spmm_q3_lds_body<N, XG, YG, false, false, 2>(b_rowptr, q_blk, Mb, X, ldx, Y, ldy, nchunks, SpmmEpi{nullptr, 0, nullptr, 0});
}
template <int N, int XG, int YG>
__global__ __launch_bounds__(kWG) SN_FOUR_WAVES void spmm_q3_lds_epi(const int *__restrict__ b_rowptr,
                                                                     const f4 *__restrict__ q_blk, int Mb,
                                                                     const float *__restrict__ X, int64_t ldx,
                                                                     float *__restrict__ Y, int64_t ldy, int nchunks,
                                                                     SpmmEpi epi) {
  spmm_q3_lds_body<N, XG, YG, true, false, 3>(b_rowptr, q_blk, Mb, X, ldx, Y, ldy, nchunks, epi);
}
template <int N, int XG, int YG>
__global__ __launch_bounds__(kWG) SN_SIX_WAVES void spmm_q3_lds_epi_wide(const int *__restrict__ b_rowptr,
                                                                         const f4 *__restrict__ q_blk, int Mb,
                                                                         const float *__restrict__ X, int64_t ldx,
                                                                         float *__restrict__ Y, int64_t ldy, int nchunks,
                                                                         SpmmEpi epi) {
  spmm_q3_lds_body<N, XG, YG, true, false, 2>(b_rowptr, q_blk, Mb, X, ldx, Y, ldy, nchunks, epi);
}

// BSR4 -> Q3: pack blocks that are exactly M(p); *flag is raised (atomically OR-ed) if any block is not.
__global__ __launch_bounds__(kWG) void bsr4_to_q3_k(const int *__restrict__ b_colind, const float *__restrict__ b_vals,
                                                    int64_t nblocks, f4 *__restrict__ q_blk, int *__restrict__ flag) {
  const int64_t k = (int64_t)blockIdx.x * kWG + threadIdx.x;
  if (k >= nblocks) return;
  const f4 *b = reinterpret_cast<const f4 *>(b_vals + 16 * k);
  const f4 r0 = b[0], r1 = b[1], r2 = b[2], r3 = b[3];
  const float p1 = r0.y, p2 = r0.z, p3 = r0.w;
  const bool ok = r0.x == 0.f && r1.y == 0.f && r2.z == 0.f && r3.w == 0.f &&
                  r1.x == -p1 && r1.z == p3 && r1.w == -p2 &&
                  r2.x == -p2 && r2.y == -p3 && r2.w == p1 &&
                  r3.x == -p3 && r3.y == p2 && r3.z == -p1;
  if (!ok) atomicOr(flag, 1);
  q_blk[k] = f4{p1, p2, p3, __int_as_float(b_colind[k])};
}

// ------------------------------------------------------------------------------------------------
// Any-N fallback: one thread per output element (j fastest => coalesced along a dense row).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void spmm_csr_any(const int *__restrict__ rowptr,
                                                    const int *__restrict__ colind,
                                                    const float *__restrict__ vals, int64_t M, int N,
                                                    const float *__restrict__ X, int64_t ldx, int xg,
                                                    float *__restrict__ Y, int64_t ldy, int yg) {
  const int64_t total = M * (int64_t)N;
  for (int64_t t = (int64_t)blockIdx.x * kWG + threadIdx.x; t < total; t += (int64_t)gridDim.x * kWG) {
    const int64_t r = t / N;
    const int j = (int)(t - r * N);
    float acc = 0.f;
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int64_t c = colind[k];
      acc = __builtin_fmaf(vals[k], X[(c / xg) * ldx + (c % xg) * N + j], acc);
    }
    Y[(r / yg) * ldy + (r % yg) * N + j] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// sorted COO -> CSR (binary search per row: interior empty rows come out right).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void coo_to_csr_k(const int64_t *__restrict__ ib,
                                                    const int64_t *__restrict__ ir,
                                                    const int64_t *__restrict__ ic, int64_t nnz,
                                                    int64_t M, int64_t R, int64_t Kb,
                                                    int *__restrict__ rowptr, int *__restrict__ colind) {
  const int64_t n = (M + 1 > nnz) ? M + 1 : nnz;
  for (int64_t t = (int64_t)blockIdx.x * kWG + threadIdx.x; t < n; t += (int64_t)gridDim.x * kWG) {
    if (t <= M) {
      // first k with key(k) >= t
      int64_t lo = 0, hi = nnz;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t key = (ib ? ib[mid] * R : 0) + ir[mid];
        if (key < t) lo = mid + 1; else hi = mid;
      }
      rowptr[t] = (int)lo;
    }
    if (t < nnz) colind[t] = (int)((ib ? ib[t] * Kb : 0) + ic[t]);
  }
}

// ------------------------------------------------------------------------------------------------
// int32 exclusive scan, 3 launches (block sums -> scan of sums -> rescan with offsets).
// ------------------------------------------------------------------------------------------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kWG * kScanItems;

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan of one value per thread across the 256-thread workgroup; returns the prefix,
// *total receives the workgroup sum.
__device__ __forceinline__ int block_excl_scan(int v, int *total) {
  __shared__ int wsum[kWG / 64];
  const int incl = wave_incl_scan(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kWG / 64; ++i) {
    if (i < wave) off += wsum[i];
    tot += wsum[i];
  }
  *total = tot;
  return off + incl - v;
}

__global__ __launch_bounds__(kWG) void scan_block_sums(const int *__restrict__ in, int64_t n,
                                                       int *__restrict__ sums) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  int s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) s += in[base + i];
  int tot;
  block_excl_scan(s, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kWG) void scan_sums_inplace(int *__restrict__ sums, int nblk) {
  int carry = 0;
  for (int b0 = 0; b0 < nblk; b0 += kWG) {
    const int i = b0 + threadIdx.x;
    const int v = (i < nblk) ? sums[i] : 0;
    int tot;
    const int ex = block_excl_scan(v, &tot);
    if (i < nblk) sums[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kWG) void scan_apply(const int *__restrict__ in, int64_t n,
                                                  const int *__restrict__ sums, int *__restrict__ out) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  int v[kScanItems];
  int s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    s += v[i];
  }
  int tot;
  int run = block_excl_scan(s, &tot) + sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
}

// ------------------------------------------------------------------------------------------------
// CSR transpose.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void histogram_cols(const int *__restrict__ colind, int64_t nnz,
                                                      int *__restrict__ counts) {
  for (int64_t k = (int64_t)blockIdx.x * kWG + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * kWG)
    atomicAdd(&counts[colind[k]], 1);
}

__global__ __launch_bounds__(kWG) void transpose_scatter(const int *__restrict__ rowptr,
                                                         const int *__restrict__ colind,
                                                         const float *__restrict__ vals, int64_t M,
                                                         int *__restrict__ cursor,
                                                         int *__restrict__ t_colind,
                                                         float *__restrict__ t_vals) {
  // one thread per source row keeps the row id for free; rows are short on meshes.
  for (int64_t r = (int64_t)blockIdx.x * kWG + threadIdx.x; r < M; r += (int64_t)gridDim.x * kWG) {
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int pos = atomicAdd(&cursor[colind[k]], 1);
      t_colind[pos] = (int)r;
      t_vals[pos] = vals[k];
    }
  }
}

// The scatter order inside an output row depends on atomic timing; sorting each (short) row by
// its column index restores a unique, deterministic layout (entries are distinct: A is coalesced).
__global__ __launch_bounds__(kWG) void sort_rows_by_col(const int *__restrict__ rowptr, int64_t M,
                                                        int *__restrict__ colind,
                                                        float *__restrict__ vals) {
  for (int64_t r = (int64_t)blockIdx.x * kWG + threadIdx.x; r < M; r += (int64_t)gridDim.x * kWG) {
    const int b = rowptr[r], e = rowptr[r + 1];
    for (int i = b + 1; i < e; ++i) {
      const int c = colind[i];
      const float v = vals[i];
      int j = i - 1;
      while (j >= b && colind[j] > c) {
        colind[j + 1] = colind[j];
        vals[j + 1] = vals[j];
        --j;
      }
      colind[j + 1] = c;
      vals[j + 1] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// CSR -> BSR4: 4-way merge of the (sorted) block-column lists of 4 consecutive rows.
// ------------------------------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(kWG) void bsr4_merge(const int *__restrict__ rowptr,
                                                  const int *__restrict__ colind,
                                                  const float *__restrict__ vals, int64_t Mb,
                                                  int *__restrict__ counts_or_rowptr,
                                                  int *__restrict__ b_colind,
                                                  float *__restrict__ b_vals) {
  for (int64_t br = (int64_t)blockIdx.x * kWG + threadIdx.x; br < Mb; br += (int64_t)gridDim.x * kWG) {
    int p[4], e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      p[q] = rowptr[4 * br + q];
      e[q] = rowptr[4 * br + q + 1];
    }
    int n = 0;
    int out = FILL ? counts_or_rowptr[br] : 0;
    while (true) {
      int cur = INT_MAX;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (p[q] < e[q]) {
          const int bc = colind[p[q]] >> 2;
          cur = bc < cur ? bc : cur;
        }
      if (cur == INT_MAX) break;
      if constexpr (FILL) {
        b_colind[out] = cur;
        float blk[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) blk[i] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          while (p[q] < e[q] && (colind[p[q]] >> 2) == cur) {
            const int c = colind[p[q]] & 3;
            // static indexing keeps blk[] in registers
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
              if (cc == c) blk[q * 4 + cc] = vals[p[q]];
            ++p[q];
          }
        f4 *dst = reinterpret_cast<f4 *>(b_vals + 16 * (int64_t)out);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = f4{blk[4 * q], blk[4 * q + 1], blk[4 * q + 2], blk[4 * q + 3]};
        ++out;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          while (p[q] < e[q] && (colind[p[q]] >> 2) == cur) ++p[q];
        ++n;
      }
    }
    if constexpr (!FILL) counts_or_rowptr[br] = n;
  }
}

// ------------------------------------------------------------------------------------------------
// CSR -> RB4 ("row-blocked", 4x1 blocks): rows 4b .. 4b+3 share ONE list of the columns any of them touches (sorted),
// with four coefficients per listed column (zero where a row lacks it).  On a mesh Laplacian four consecutive rows reach
// ~16 distinct columns with 28 entries between them, so the product gathers each X row once per group instead of once per
// row: the gather stream — which, not HBM, bounds these kernels (PMC: the vector-cache miss queue) — shrinks by ~45 %.
// Rows past M (M not a multiple of 4) are treated as empty.
// ------------------------------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(kWG) void rb4_merge(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                 const float *__restrict__ vals, int64_t M, int64_t Mb,
                                                 int *__restrict__ counts_or_ptr, int *__restrict__ b_col,
                                                 float *__restrict__ b_val) {
  for (int64_t br = (int64_t)blockIdx.x * kWG + threadIdx.x; br < Mb; br += (int64_t)gridDim.x * kWG) {
    int p[4], e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t r = 4 * br + q;
      p[q] = r < M ? rowptr[r] : 0;
      e[q] = r < M ? rowptr[r + 1] : 0;
    }
    int n = 0;
    int out = FILL ? counts_or_ptr[br] : 0;
    while (true) {
      int cur = INT_MAX;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (p[q] < e[q]) {
          const int c = colind[p[q]];
          cur = c < cur ? c : cur;
        }
      if (cur == INT_MAX) break;
      if constexpr (FILL) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (p[q] < e[q] && colind[p[q]] == cur) {
            a[q] = vals[p[q]];
            ++p[q];
          }
        b_col[out] = cur;
        reinterpret_cast<f4 *>(b_val)[out] = f4{a[0], a[1], a[2], a[3]};
        ++out;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (p[q] < e[q] && colind[p[q]] == cur) ++p[q];
        ++n;
      }
    }
    if constexpr (!FILL) counts_or_ptr[br] = n;
  }
}

// ------------------------------------------------------------------------------------------------
// RB4 SpMM (group-1 operands, N in {64, 128}): N/4 lanes own one 4-row group and a float4 column slice (4 accumulators),
// a wave owns P = 256/N groups per pass and `iters` passes (as spmm_csr_rows: row pointers and the whole entry run of the
// wave staged once).  Per listed column one gather of the X row serves all four output rows.  k-ascending FMA chain per
// row; the zero coefficients contribute fma(0, x, acc) = acc, so results are bit-identical to the CSR kernels for finite X.
// ------------------------------------------------------------------------------------------------
template <int N, bool EPI, bool STATS, int KB = 8>
__device__ __forceinline__ void spmm_rb4_body(const int *__restrict__ b_ptr, const int *__restrict__ b_col,
                                              const f4 *__restrict__ b_val, int M, int Mb, const float *__restrict__ X,
                                              int64_t ldx, float *__restrict__ Y, int64_t ldy, int nchunks, int iters,
                                              SpmmEpi epi, float *__restrict__ stats_part) {
  constexpr int LPR = N / 4;          // lanes per 4-row group
  constexpr int P = 64 / LPR;         // groups per pass
  constexpr int WAVES = kWG / 64;
  constexpr int CAP = 256;            // listed columns of a wave's groups held in LDS (5 KiB per wave)
                                      // KB: gathers in flight per lane
  __shared__ int s_col[WAVES][CAP];
  __shared__ f4 s_val[WAVES][CAP];
  __shared__ int s_rp[WAVES][68];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane % LPR, grp = lane / LPR;
  const float *xb = X + sub * 4;
  int *sc = s_col[wave];
  f4 *sv = s_val[wave];
  int *rp = s_rp[wave];
  const int R = P * iters;                                        // groups of this wave (<= 64)
  const int g0 = (my_chunk(nchunks) * WAVES + wave) * R;
  if constexpr (!STATS) {
    if (g0 >= Mb) {                                               // wave-uniform
      if constexpr (EPI)
        if (epi.absmax && lane == 0) epi.absmax[(int64_t)blockIdx.x * WAVES + wave] = 0.f;
      return;
    }
  }
  float am = 0.f;                                                 // EPI with epi.absmax: max |Y| over this wave's rows
  {
    int gl = g0 + lane;
    gl = gl < Mb ? gl : Mb;
    rp[lane] = b_ptr[gl];
    int ge = g0 + R;
    ge = ge < Mb ? ge : Mb;
    if (lane == 0) rp[64] = b_ptr[g0 < Mb ? ge : Mb];
  }
  __builtin_amdgcn_wave_barrier();
  const int k0 = rp[0];
  const int k1 = R < 64 ? rp[R] : rp[64];
  const bool fast = (k1 - k0) <= CAP;                             // wave-uniform
  if (fast) {
    for (int p0 = 0; p0 < k1 - k0; p0 += 64) {
      int p = p0 + lane;
      p = p < k1 - k0 ? p : k1 - k0 - 1;
      __builtin_amdgcn_global_load_lds(b_col + k0 + p, sc + p0, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(b_val + k0 + p, sv + p0, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  f4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = ssum;
  for (int i = 0; i < iters; ++i) {
    const int lg = i * P + grp;
    const int g = g0 + lg;
    const int kb = rp[lg];
    const int ke = (lg + 1 < 64) ? rp[lg + 1] : rp[64];
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const int pk0 = fast ? k0 : __builtin_amdgcn_readfirstlane(kb);
    const int pk1 = fast ? k1 : __builtin_amdgcn_readlane(ke, 63);
    for (int t0 = pk0; t0 < pk1; t0 += CAP) {
      int kk = kb, kend = ke;
      if (!fast) {                      // this pass's columns in tiles of CAP through the same LDS slice
        const int nt = (pk1 - t0) < CAP ? (pk1 - t0) : CAP;
        for (int p0 = 0; p0 < nt; p0 += 64) {
          int p = p0 + lane;
          p = p < nt ? p : nt - 1;
          __builtin_amdgcn_global_load_lds(b_col + t0 + p, sc + p0, 4, 0, 0);
          __builtin_amdgcn_global_load_lds(b_val + t0 + p, sv + p0, 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        kk = kb > t0 ? kb : t0;
        kend = ke < t0 + nt ? ke : t0 + nt;
      }
      const int base = fast ? k0 : t0;
      for (; kk < kend; kk += KB) {
        int c[KB];
        f4 x[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          const int o = (kk + j < kend ? kk + j : kend - 1) - base;
          c[j] = sc[o];
        }
#pragma unroll
        for (int j = 0; j < KB; ++j) x[j] = ld4(xb + (int64_t)c[j] * ldx);
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          if (kk + j < kend) {            // (uniform within a lane group; a skipped slot re-read a column of this group)
            const f4 a = sv[kk + j - base];
            acc0 = fma4(a.x, x[j], acc0);
            acc1 = fma4(a.y, x[j], acc1);
            acc2 = fma4(a.z, x[j], acc2);
            acc3 = fma4(a.w, x[j], acc3);
          }
        }
      }
      if (fast) break;                  // (everything was staged up front: one trip)
      __builtin_amdgcn_wave_barrier();
    }
    if (g < Mb) {
      const int r = 4 * g;
      auto finish = [&](f4 v, int rr) {
        if (rr < M) {
          if constexpr (EPI) {
            v = elu_bwd4(v, ld4_s(epi.e + (int64_t)rr * epi.lde + sub * 4, kStreamNT));
            if (epi.g) v += ld4_s(epi.g + (int64_t)rr * epi.ldg + sub * 4, kStreamNT);
            if (epi.absmax) am = fmaxf(am, hmax_abs4(v));
          }
          st4_stream(Y + (int64_t)rr * ldy + sub * 4, v);
          if constexpr (STATS) {
            ssum += v;
            ssq.x = __builtin_fmaf(v.x, v.x, ssq.x); ssq.y = __builtin_fmaf(v.y, v.y, ssq.y);
            ssq.z = __builtin_fmaf(v.z, v.z, ssq.z); ssq.w = __builtin_fmaf(v.w, v.w, ssq.w);
          }
        }
      };
      finish(acc0, r);
      finish(acc1, r + 1);
      finish(acc2, r + 2);
      finish(acc3, r + 3);
    }
  }
  if constexpr (EPI) {
    if (epi.absmax) {
      am = wave_max_nonneg(am);
      if (lane == 0) epi.absmax[(int64_t)blockIdx.x * WAVES + wave] = am;
    }
  }
  if constexpr (STATS) {
    static_assert(N == 128 && !EPI, "statistics: 128-column rows");
    __shared__ float s_st[WAVES * P][256];
    float *st = s_st[wave * P + grp];
    *reinterpret_cast<f4 *>(st + sub * 4) = ssum;
    *reinterpret_cast<f4 *>(st + 128 + sub * 4) = ssq;
    __syncthreads();
    const int t = threadIdx.x;
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES * P; ++w) tot += s_st[w][t];
    stats_part[(int64_t)blockIdx.x * 256 + t] = tot;
  }
}
template <int N, int KB>
__global__ __launch_bounds__(kWG) void spmm_rb4(const int *__restrict__ b_ptr, const int *__restrict__ b_col,
                                                const f4 *__restrict__ b_val, int M, int Mb, const float *__restrict__ X,
                                                int64_t ldx, float *__restrict__ Y, int64_t ldy, int nchunks, int iters) {
  spmm_rb4_body<N, false, false, KB>(b_ptr, b_col, b_val, M, Mb, X, ldx, Y, ldy, nchunks, iters, SpmmEpi{nullptr, 0, nullptr, 0}, nullptr);
}
template <int N>
__global__ __launch_bounds__(kWG) void spmm_rb4_epi(const int *__restrict__ b_ptr, const int *__restrict__ b_col,
                                                    const f4 *__restrict__ b_val, int M, int Mb, const float *__restrict__ X,
                                                    int64_t ldx, float *__restrict__ Y, int64_t ldy, int nchunks, int iters,
                                                    SpmmEpi epi) {
  spmm_rb4_body<N, true, false, 4>(b_ptr, b_col, b_val, M, Mb, X, ldx, Y, ldy, nchunks, iters, epi, nullptr);
}
__global__ __launch_bounds__(kWG) void spmm_rb4_stats(const int *__restrict__ b_ptr, const int *__restrict__ b_col,
                                                      const f4 *__restrict__ b_val, int M, int Mb, const float *__restrict__ X,
                                                      int64_t ldx, float *__restrict__ Y, int64_t ldy, int nchunks, int iters,
                                                      float *__restrict__ stats_part) {
  spmm_rb4_body<128, false, true, 4>(b_ptr, b_col, b_val, M, Mb, X, ldx, Y, ldy, nchunks, iters, SpmmEpi{nullptr, 0, nullptr, 0},
                                     stats_part);
}

// ------------------------------------------------------------------------------------------------
// Sliding-window ("ring") CSR SpMM for banded square operators — the Laplacian products at 64 / 128 dense columns
// (the products at src/utils/utils_pt.py:167,176) on batches large enough to fill the chip.
//
// What bounds the gather kernels above (csr_rows, rb4) is the number of L2 requests: a mesh Laplacian in row-major vertex
// order touches three index bands, a band is re-used ~m rows later, and 128-channel rows do not survive that long in the
// 32 KiB vector cache — every gather goes to L2 (3.4 requests per compulsory line with RB4, rocprofv3 PMC).
// Here a PERSISTENT workgroup walks a strip of consecutive rows in steps of R and keeps the X rows [r - H, r + R + H) of a
// 64-column slice in an LDS ring addressed by (row mod W): every X line and every CSR entry is requested ONCE per slice,
// by fully coalesced LDS-DMA (global_load_lds: 1 KiB per wave instruction), and all irregular access happens in LDS.
//   * NLW loader waves only issue DMA (the X piece, the entries and the row pointers of step t + D) and wait with a
//     partial s_waitcnt vmcnt — they never store, so their counter counts DMA only and D steps stay in flight;
//   * R/8 compute waves (one 8-row round per step: 8 lanes x 2 float4 per row) read LDS, multiply and store; they never
//     wait for their stores; one s_barrier per step (no implicit vmcnt(0));
//   * a column outside the window (|c - r| > H: wrap-around rows of closed meshes, unordered meshes) is gathered from
//     global memory inside the same batch ("mixed" rounds); rows of more than 32 entries take an entry-by-entry path.
// Same k-ascending FMA chain per row as every other CSR kernel: bit-identical results (slots past a row's end multiply the
// row's own first column by 0).  Measured (MI355X, 128 channels): 0.65-0.67 of the HBM roofline on the
// config-5 Laplacian batch (RB4: 0.55-0.58), 0.83 on the config-3-sized batch (0.72-0.79), 0.81 on config 4's (0.57).
// ------------------------------------------------------------------------------------------------
constexpr int kRingXAux = 2;         // cache policy of the X DMA: non-temporal (every X line is requested once per slice: +2-3 % over the default)
constexpr int kRingEAux = 0;         // ... of the entry / row-pointer DMA: default
constexpr int kRingCS = 64;          // dense columns per slice (one 256-byte piece of an X row)
constexpr int kRingW = 512;          // ring rows (128 KiB)
constexpr int kRingR = 64;           // rows per step
constexpr int kRingH = 160;          // half window: columns within +-H of the row come from the ring
constexpr int kRingD = 2;            // steps of DMA in flight
constexpr int kRingNLW = 4;   // loader waves
constexpr int kRingNCW = kRingR / 8; // compute waves
constexpr int kRingThreads = (kRingNCW + kRingNLW) * 64;
constexpr int kRingECap = kRingR * 8;                     // entry slots per step buffer
constexpr int kRingRPS = kRingR + 64;                     // row-pointer slots per step buffer
constexpr size_t kRingLds = (size_t)kRingW * kRingCS * 4 + (size_t)(kRingD + 1) * kRingECap * 8 + (size_t)(kRingD + 1) * kRingRPS * 4;
static_assert(kRingW >= (kRingD + 1) * kRingR + 2 * kRingH && (kRingW & (kRingW - 1)) == 0, "ring too small");
static_assert(kRingLds <= 160 * 1024, "LDS");

template <int N_>
__device__ __forceinline__ void wait_vmcnt_imm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// (A x) * elu'(e) + g with the product rounded BEFORE the addition, as the unfused composition and the other fused kernels do
// (with the operands in registers the compiler would otherwise contract the two into one fma)
__device__ __forceinline__ f4 ring_epilogue(f4 a, const f4 &e, const f4 &g, bool has_g) {
#pragma clang fp contract(off)
  f4 p = elu_bwd4(a, e);
  if (has_g) p = p + g;
  return p;
}

template <bool EPI, bool STATS>
__global__ __launch_bounds__(kRingThreads) void spmm_ring_k(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                            const float *__restrict__ vals, int M, int K, int nnz,
                                                            const float *__restrict__ X, int64_t ldx, float *__restrict__ Y,
                                                            int64_t ldy, int nstrips, int cps, int nsl, SpmmEpi epi,
                                                            float *__restrict__ stats_part) {
  constexpr int CS = kRingCS, W = kRingW, R = kRingR, H = kRingH, D = kRingD, NLW = kRingNLW, NCW = kRingNCW;
  constexpr int ECAP = kRingECap, NB = D + 1, RPS = kRingRPS;
  constexpr int LPX = CS / 4, RPI = 64 / LPX;   // lanes per X row in a DMA instruction, rows per instruction
  constexpr int NV = CS / 32;                   // float4 pieces per lane
  // DMA instructions of ONE step per loader wave: a constant, so that the partial wait is an immediate
  constexpr int NX = (R / RPI + NLW - 1) / NLW;
  constexpr int NE = 2 * ((ECAP / 64 + NLW - 1) / NLW);
  constexpr int NR = ((R + 1 + 63) / 64 + NLW - 1) / NLW;
  constexpr int NSTEP = NX + NE + NR;
  static_assert((D - 1) * NSTEP <= 63 && R % RPI == 0 && H % RPI == 0 && ECAP % 64 == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_smem[];
  float *xs = reinterpret_cast<float *>(ring_smem);                   // W x CS floats
  int *sc = reinterpret_cast<int *>(xs + W * CS);                     // [NB][ECAP]
  float *sv = reinterpret_cast<float *>(sc + NB * ECAP);              // [NB][ECAP]
  int *rp = reinterpret_cast<int *>(sv + NB * ECAP);                  // [NB][RPS]

  // workgroup b runs on XCD b % 8: an XCD owns a contiguous eighth of the strips, the slices of a strip sit next to each
  // other in dispatch order (they read the same entries: the second reader hits L2)
  const int nx = c_xcd;                                               // (nstrips is a multiple of it: ring_strips)
  const int b = blockIdx.x, xcd = b % nx, li = b / nx;
  const int spx = nstrips / nx;
  const int strip = xcd * spx + li / nsl, sl = li % nsl;
  const int nsteps = (M + R - 1) / R;
  const int t0 = strip * cps;
  const int t1 = (t0 + cps) < nsteps ? (t0 + cps) : nsteps;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = sl * CS;
  const bool idle = li / nsl >= spx || t0 >= t1;                      // (a strip past the end: nothing to do)
  if (idle) {
    if constexpr (STATS) {
      if (li / nsl < spx && threadIdx.x < 2 * CS)
        stats_part[(int64_t)strip * 256 + (threadIdx.x / CS) * 128 + c0 + (threadIdx.x % CS)] = 0.f;
    }
    if constexpr (EPI) {
      if (epi.absmax && threadIdx.x < NCW) epi.absmax[(int64_t)blockIdx.x * NCW + threadIdx.x] = 0.f;
    }
    return;
  }

  if (wave >= NCW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wave - NCW;
    const float *xg = X + c0 + (lane % LPX) * 4;
    auto issue_rows = [&](int row0, int i) {                          // RPI rows from row0 + RPI*i -> ring
      int row = row0 + RPI * i + lane / LPX;
      row = row < 0 ? 0 : (row < K ? row : K - 1);
      __builtin_amdgcn_global_load_lds(xg + (int64_t)row * ldx, xs + ((row0 + RPI * i) & (W - 1)) * CS, 16, 0, kRingXAux);
    };
    auto issue_step = [&](int t) {                                    // X piece, entries, row pointers of step t: NSTEP instr.
      const int buf = t % NB;
      const int x0 = t * R + H;
#pragma unroll
      for (int q = 0; q < NX; ++q) {
        int i = lw + q * NLW;
        i = i < R / RPI ? i : R / RPI - 1;
        issue_rows(x0, i);
      }
      const int64_t ra = (int64_t)t * R, rb = ra + R;
      const int k0 = rowptr[ra < M ? ra : M], k1 = rowptr[rb < M ? rb : M];
      int ne = k1 - k0;
      ne = ne < ECAP ? ne : ECAP;                                     // (entries past the buffer are read from global memory)
#pragma unroll
      for (int q = 0; q < NE / 2; ++q) {
        int p0 = (lw + q * NLW) * 64;
        p0 = p0 < ECAP ? p0 : ECAP - 64;
        int p = p0 + lane;
        p = p < ne ? p : (ne > 0 ? ne - 1 : 0);
        int k = k0 + p;
        k = k < nnz ? k : nnz - 1;
        __builtin_amdgcn_global_load_lds(colind + k, sc + buf * ECAP + p0, 4, 0, kRingEAux);
        __builtin_amdgcn_global_load_lds(vals + k, sv + buf * ECAP + p0, 4, 0, kRingEAux);
      }
      const int r0 = t * R;
      int nr = M - r0;
      nr = nr < R ? (nr > 0 ? nr : 0) : R;
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        int p0 = (lw + q * NLW) * 64;
        p0 = p0 < RPS ? p0 : RPS - 64;
        int p = p0 + lane;
        p = p < nr + 1 ? p : nr;
        int r = r0 + p;
        r = r < M ? r : M;
        __builtin_amdgcn_global_load_lds(rowptr + r, rp + buf * RPS + p0, 4, 0, 0);
      }
    };
    // prologue: the first window [t0 R - H, t0 R + H) and the steps t0 .. t0 + D - 1 (steps past the end are loaded too:
    // clamped addresses, dead ring slots — the instruction counts stay uniform)
    for (int i = lw; i < 2 * H / RPI; i += NLW) issue_rows(t0 * R - H, i);
#pragma unroll
    for (int d = 0; d < D; ++d) issue_step(t0 + d);
    wait_vmcnt_imm<(D - 1) * NSTEP>();
    __builtin_amdgcn_s_barrier();
    for (int t = t0; t < t1; ++t) {
      issue_step(t + D);
      wait_vmcnt_imm<(D - 1) * NSTEP>();                              // step t + 1 has landed
      __builtin_amdgcn_s_barrier();
    }
    wait_vmcnt_imm<0>();                                              // nothing may land in LDS after the workgroup has gone
    // STATS: the compute waves re-use ring rows 0..127 as scratch for the column sums.  The DMA of the steps past t1 (issued
    // above with clamped addresses to keep the instruction counts uniform) targets ring rows that can overlap that scratch, so
    // every loader drains FIRST (the wait above) and only then meets the compute waves at one more barrier, which they pass
    // before their first scratch store.
    if constexpr (STATS) __builtin_amdgcn_s_barrier();
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int g = lane >> 3, sub = lane & 7;
  const float *xl = xs + sub * 4;
  const float *xgl = X + c0 + sub * 4;
  f4 ssum[NV], ssq[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) ssum[v] = ssq[v] = f4{0.f, 0.f, 0.f, 0.f};
  float am = 0.f;                                                     // EPI with epi.absmax: max |Y| over this wave's rows
  f4 evn[NV], gvn[NV];                                                // epilogue operands of the NEXT step (requested a step ahead)
  if constexpr (EPI) {
    const int rf = t0 * R + wave * 8 + g;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      evn[v] = gvn[v] = f4{0.f, 0.f, 0.f, 0.f};
      if (rf < M) {
        evn[v] = ld4_s(epi.e + (int64_t)rf * epi.lde + c0 + v * 32 + sub * 4, kStreamNT);
        if (epi.g) gvn[v] = ld4_s(epi.g + (int64_t)rf * epi.ldg + c0 + v * 32 + sub * 4, kStreamNT);
      }
    }
  }
  __builtin_amdgcn_s_barrier();
  for (int t = t0; t < t1; ++t) {
    asm volatile("" ::: "memory");
    const int buf = t % NB;
    const int *scb = sc + buf * ECAP;
    const float *svb = sv + buf * ECAP;
    const int *rpb = rp + buf * RPS;
    const int r0 = t * R;
    const int nr = (M - r0) < R ? (M - r0) : R;
    const int wlo = r0 - H;
    const int lr = wave * 8 + g;
    const bool live = lr < nr;
    const int r = r0 + lr;
    const int k0 = rpb[0];                                            // entry held by slot 0 of the buffer
    int kb = 0, ke = 0;
    if (live) {
      kb = rpb[lr] - k0;
      ke = rpb[lr + 1] - k0;
    }
    f4 ev[NV], gv[NV];
    if constexpr (EPI) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        ev[v] = evn[v];
        gv[v] = gvn[v];
      }
      const int rn = r + R;                                           // this lane group's row of the next step
      if (rn < M && t + 1 < t1) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          evn[v] = ld4_s(epi.e + (int64_t)rn * epi.lde + c0 + v * 32 + sub * 4, kStreamNT);
          if (epi.g) gvn[v] = ld4_s(epi.g + (int64_t)rn * epi.ldg + c0 + v * 32 + sub * 4, kStreamNT);
        }
      }
    }
    const int len = ke - kb;
    const bool fits = ke <= ECAP && len <= 32;
    int cf = r0 < K ? r0 : K - 1, cl = cf;                            // (an empty row: any loaded column, multiplied by 0)
    if (len > 0 && fits) {
      cf = scb[kb];
      cl = scb[ke - 1];                                               // columns ascend within a row: first and last bound the rest
    }
    const bool inwin = (unsigned)(cf - wlo) < (unsigned)(R + 2 * H) && (unsigned)(cl - wlo) < (unsigned)(R + 2 * H);
    f4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = f4{0.f, 0.f, 0.f, 0.f};
    if (__builtin_amdgcn_ballot_w64(!fits) == 0) {
      // batches of 8 slots per row, read from kb onwards whatever the row's length: a slot past the row's end holds the next
      // rows' entries (or spare buffer words) and becomes (first column, 0) — fma(0, x, acc) == acc
      int kmax = 8;
      if (__builtin_amdgcn_ballot_w64(len > 8)) kmax = 16;
      if (__builtin_amdgcn_ballot_w64(len > 16)) kmax = 32;
      const bool allin = __builtin_amdgcn_ballot_w64(!inwin) == 0;    // wave-uniform
      for (int k = 0; k < kmax; k += 8) {
        int c[8];
        float a[8];
        f4 x[8][NV];
        const int *cp = scb + kb + k;
        const float *ap = svb + kb + k;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          c[j] = cp[j];
          a[j] = ap[j];
        }
        if (allin) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int cj = k + j < len ? c[j] : cf;
            a[j] = k + j < len ? a[j] : 0.f;
            const float *xp = xl + (cj & (W - 1)) * CS;
#pragma unroll
            for (int v = 0; v < NV; ++v) x[j][v] = *reinterpret_cast<const f4 *>(xp + v * 32);
          }
        } else {
          // mixed round: every slot is read from the ring AND, where its column lies outside the window, from global memory
          // (lanes inside are masked off), all loads in flight together
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int cj = k + j < len ? c[j] : cf;
            a[j] = k + j < len ? a[j] : 0.f;
            const bool in = (unsigned)(cj - wlo) < (unsigned)(R + 2 * H);
            const float *xp = xl + (cj & (W - 1)) * CS;
            const float *gp = xgl + (int64_t)cj * ldx;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              x[j][v] = *reinterpret_cast<const f4 *>(xp + v * 32);
              if (!in) x[j][v] = ld4(gp + v * 32);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
          for (int v = 0; v < NV; ++v) acc[v] = fma4(a[j], x[j][v], acc[v]);
        }
      }
      if (len <= 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = f4{0.f, 0.f, 0.f, 0.f};  // (an empty row is 0 whatever the placeholder column held)
      }
    } else {
      for (int k = kb; k < ke; ++k) {                                 // rows of more than 32 entries / past the buffer: one by one
        int cc;
        float aa;
        if (k < ECAP) {
          cc = scb[k];
          aa = svb[k];
        } else {
          cc = colind[k0 + k];
          aa = vals[k0 + k];
        }
        const bool in = (unsigned)(cc - wlo) < (unsigned)(R + 2 * H);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          f4 x;
          if (in) x = *reinterpret_cast<const f4 *>(xl + (cc & (W - 1)) * CS + v * 32);
          else x = ld4(xgl + (int64_t)cc * ldx + v * 32);
          acc[v] = fma4(aa, x, acc[v]);
        }
      }
    }
    if (live) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        f4 o = acc[v];
        if constexpr (EPI) {
          o = ring_epilogue(o, ev[v], gv[v], epi.g != nullptr);
          if (epi.absmax) am = fmaxf(am, hmax_abs4(o));
        }
        st4_stream(Y + (int64_t)r * ldy + c0 + v * 32 + sub * 4, o);
        if constexpr (STATS) {
          ssum[v] += o;
          ssq[v].x = __builtin_fmaf(o.x, o.x, ssq[v].x); ssq[v].y = __builtin_fmaf(o.y, o.y, ssq[v].y);
          ssq[v].z = __builtin_fmaf(o.z, o.z, ssq[v].z); ssq[v].w = __builtin_fmaf(o.w, o.w, ssq[v].w);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // every LDS read of this step has returned
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (EPI) {
    if (epi.absmax) {                                                 // one slot per compute wave: no meeting, no scratch
      am = wave_max_nonneg(am);
      if (lane == 0) epi.absmax[(int64_t)blockIdx.x * NCW + wave] = am;
    }
  }
  if constexpr (STATS) {
    // column sums / sums of squares of this workgroup's output rows -> stats_part[strip][sum | squares][128] (fp32 over the
    // strip per lane, then a fixed-order sum; fp64 above).  First the loaders' drain barrier (every DMA they issued has
    // landed: nothing can overwrite the scratch below); after it the loader waves leave, and a finished wave no longer counts
    // at s_barrier.
    __builtin_amdgcn_s_barrier();
    float *st = xs + (wave * 8 + g) * (2 * CS);                       // [NCW*8][sum | squares][CS] in the (now free) ring
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      *reinterpret_cast<f4 *>(st + v * 32 + sub * 4) = ssum[v];
      *reinterpret_cast<f4 *>(st + CS + v * 32 + sub * 4) = ssq[v];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int tt = threadIdx.x;
    if (tt < 2 * CS) {
      float tot = 0.f;
      for (int w = 0; w < NCW * 8; ++w) tot += xs[w * (2 * CS) + tt];
      stats_part[(int64_t)strip * 256 + (tt / CS) * 128 + c0 + (tt % CS)] = tot;
    }
  }
}

// max |column - row| over the entries, the longest row, and the number of rows with an entry outside the ring kernel's window
// (|column - row| > kRingH) of a CSR operator: what decides between the ring kernel and the gather kernels
// (out[0..2]; all start at 0)
__global__ __launch_bounds__(kWG) void csr_band_k(const int *__restrict__ rowptr, const int *__restrict__ colind, int64_t M,
                                                  int *__restrict__ out) {
  int band = 0, longest = 0, outside = 0;
  for (int64_t r = (int64_t)blockIdx.x * kWG + threadIdx.x; r < M; r += (int64_t)gridDim.x * kWG) {
    const int kb = rowptr[r], ke = rowptr[r + 1];
    if (ke > kb) {
      // The ring kernel bounds a row by its first and last entry, i.e. it needs strictly ascending columns.  A row that is
      // not (an operator built straight from unsorted CSR arrays) reports INT32_MAX as the longest row, which no caller's
      // "longest row <= 32" rule lets through to the ring kernel; the band is measured over all entries of the row.
      int cmin = colind[kb], cmax = cmin, prev = cmin;
      bool asc = true;
      for (int k = kb + 1; k < ke; ++k) {
        const int c = colind[k];
        asc = asc && c > prev;
        prev = c;
        cmin = c < cmin ? c : cmin;
        cmax = c > cmax ? c : cmax;
      }
      const int lo = (int)r - cmin, hi = cmax - (int)r;
      const int far = lo > hi ? lo : hi;
      band = far > band ? far : band;
      outside += far > kRingH ? 1 : 0;
      const int len = asc ? ke - kb : 0x7fffffff;
      longest = len > longest ? len : longest;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int b2 = __shfl_xor(band, o), l2 = __shfl_xor(longest, o);
    band = b2 > band ? b2 : band;
    longest = l2 > longest ? l2 : longest;
    outside += __shfl_xor(outside, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out, band);
    atomicMax(out + 1, longest);
    if (outside) atomicAdd(out + 2, outside);
  }
}

// ------------------------------------------------------------------------------------------------
// Block-diagonal batch assembly from the resident operator pool.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void blockdiag_rowptr(const int *__restrict__ pool_rowptr,
                                                        const int64_t *__restrict__ desc, int64_t B,
                                                        int64_t size0, int64_t total,
                                                        int *__restrict__ out_rowptr) {
  const int64_t n = B * size0 + 1;
  for (int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += (int64_t)gridDim.x * kWG) {
    if (i == n - 1) {
      out_rowptr[i] = (int)total;
      continue;
    }
    const int64_t b = i / size0, r = i - b * size0;
    const int64_t *d = desc + 4 * b;
    const int64_t rows = d[2];
    const int local = pool_rowptr[d[0] + (r < rows ? r : rows)];
    out_rowptr[i] = (int)(d[3] + local);
  }
}

template <int VPE>
__global__ __launch_bounds__(kWG) void blockdiag_entries(const int *__restrict__ pool_colind,
                                                         const float *__restrict__ pool_vals,
                                                         const int64_t *__restrict__ desc, int64_t B,
                                                         int64_t size1, int64_t total,
                                                         int *__restrict__ out_colind,
                                                         float *__restrict__ out_vals) {
  for (int64_t k = (int64_t)blockIdx.x * kWG + threadIdx.x; k < total; k += (int64_t)gridDim.x * kWG) {
    // last b with desc[b].out_off <= k
    int64_t lo = 0, hi = B;
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (desc[4 * mid + 3] <= k) lo = mid; else hi = mid;
    }
    const int64_t *d = desc + 4 * lo;
    const int64_t src = d[1] + (k - d[3]);
    if constexpr (VPE == 4) {         // Q3 record: (p1, p2, p3, block column as int bits) — no separate index array
      f4 q = reinterpret_cast<const f4 *>(pool_vals)[src];
      q.w = __int_as_float(__float_as_int(q.w) + (int)(lo * size1));
      reinterpret_cast<f4 *>(out_vals)[k] = q;
      continue;
    }
    out_colind[k] = pool_colind[src] + (int)(lo * size1);
    if constexpr (VPE == 1) {
      out_vals[k] = pool_vals[src];
    } else {
      const f4 *s = reinterpret_cast<const f4 *>(pool_vals + (int64_t)VPE * src);
      f4 *o = reinterpret_cast<f4 *>(out_vals + (int64_t)VPE * k);
#pragma unroll
      for (int i = 0; i < VPE / 4; ++i) o[i] = s[i];
    }
  }
}

// Ragged form of the same assembly: mesh b owns the output rows [d[4], next mesh's d[4]) — of which the first d[2] carry
// its entries and the rest (if any) are empty — and its column indices are shifted by d[5].  With d[4] = prefix sum of the
// meshes' own row counts and d[5] = prefix sum of their column counts the batch is PACKED: no padding rows or columns
// exist, so the product neither scans empty rows nor writes zeros into them (SURVEY.md §7 "ragged not padded").
// desc is (B x 6) int64: { rowptr offset, entry offset, rows with entries, output entry offset, first output row, column shift }.
__global__ __launch_bounds__(kWG) void blockdiag_rowptr_ragged(const int *__restrict__ pool_rowptr,
                                                               const int64_t *__restrict__ desc, int64_t B,
                                                               int64_t total_rows, int64_t total,
                                                               int *__restrict__ out_rowptr) {
  const int64_t n = total_rows + 1;
  for (int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x; i < n; i += (int64_t)gridDim.x * kWG) {
    if (i == n - 1) {
      out_rowptr[i] = (int)total;
      continue;
    }
    int64_t lo = 0, hi = B;                      // last b with first output row <= i
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (desc[6 * mid + 4] <= i) lo = mid; else hi = mid;
    }
    const int64_t *d = desc + 6 * lo;
    const int64_t r = i - d[4], rows = d[2];
    const int local = pool_rowptr[d[0] + (r < rows ? r : rows)];
    out_rowptr[i] = (int)(d[3] + local);
  }
}

template <int VPE>
__global__ __launch_bounds__(kWG) void blockdiag_entries_ragged(const int *__restrict__ pool_colind,
                                                                const float *__restrict__ pool_vals,
                                                                const int64_t *__restrict__ desc, int64_t B,
                                                                int64_t total, int *__restrict__ out_colind,
                                                                float *__restrict__ out_vals) {
  for (int64_t k = (int64_t)blockIdx.x * kWG + threadIdx.x; k < total; k += (int64_t)gridDim.x * kWG) {
    int64_t lo = 0, hi = B;                      // last b with output entry offset <= k
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (desc[6 * mid + 3] <= k) lo = mid; else hi = mid;
    }
    const int64_t *d = desc + 6 * lo;
    const int64_t src = d[1] + (k - d[3]);
    const int shift = (int)d[5];
    if constexpr (VPE == 4) {
      f4 q = reinterpret_cast<const f4 *>(pool_vals)[src];
      q.w = __int_as_float(__float_as_int(q.w) + shift);
      reinterpret_cast<f4 *>(out_vals)[k] = q;
      continue;
    }
    out_colind[k] = pool_colind[src] + shift;
    if constexpr (VPE == 1) {
      out_vals[k] = pool_vals[src];
    } else {
      const f4 *s = reinterpret_cast<const f4 *>(pool_vals + (int64_t)VPE * src);
      f4 *o = reinterpret_cast<f4 *>(out_vals + (int64_t)VPE * k);
#pragma unroll
      for (int i = 0; i < VPE / 4; ++i) o[i] = s[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Debug validation of a CSR operator (the reference kernels have no bounds checks at all: sparse_bmm.cu:16-61).
// flags: bit 0 rowptr[0] != 0, bit 1 rowptr decreasing, bit 2 rowptr[M] != nnz, bit 3 column index out of [0, K),
//        bit 4 column indices of a row not strictly ascending (operator not coalesced), bit 5 non-finite value.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWG) void validate_csr_k(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                      const float *__restrict__ vals, int64_t M, int64_t K, int64_t nnz,
                                                      int *__restrict__ flags) {
  int bad = 0;
  for (int64_t r = (int64_t)blockIdx.x * kWG + threadIdx.x; r < M; r += (int64_t)gridDim.x * kWG) {
    const int b = rowptr[r], e = rowptr[r + 1];
    if (r == 0 && b != 0) bad |= 1;
    if (e < b) bad |= 2;
    if (r == M - 1 && e != (int)nnz) bad |= 4;
    if (b < 0 || e > nnz || e < b) continue;             // do not follow a broken pointer
    int prev = -1;
    for (int k = b; k < e; ++k) {
      const int c = colind[k];
      if (c < 0 || c >= K) bad |= 8;
      if (c <= prev) bad |= 16;
      prev = c;
      if (vals) {
        const float v = vals[k];
        if (!(v == v) || v - v != 0.f) bad |= 32;
      }
    }
  }
  if (bad) atomicOr(flags, bad);
}

// ------------------------------------------------------------------------------------------------
// Ragged per-mesh column means for PACKED batches (global_average, utils_pt.py:120-122, on meshes of different sizes without
// padding).  The rows of mesh g are cut into tiles of <= kRagTile rows; `tiles` is an (ntiles x 3) int64 table
// {mesh, first row, rows} with the tiles of a mesh consecutive, seg_tile_ptr[g] the first tile of mesh g.  Stage 1: one
// workgroup per tile, fp64 column sums of its rows; stage 2: one thread per (mesh, column) adds the mesh's tiles in order
// and applies the optional per-mesh scale (1 / vertex count).  Deterministic (no atomics).
// ------------------------------------------------------------------------------------------------
constexpr int kRagTile = 256;

__global__ __launch_bounds__(kWG) void seg_colsum_ragged_tiles_k(const float *__restrict__ x, int64_t ld,
                                                                 const int64_t *__restrict__ tiles, int C,
                                                                 double *__restrict__ partial /* [ntiles][C] */) {
  const int64_t *t = tiles + 3 * (int64_t)blockIdx.x;
  const int64_t r0 = t[1];
  const int n = (int)t[2];
  __shared__ double s_acc[kWG];
  // thread (lane = column within a 256/CL-row interleave): CL = columns handled per sweep
  for (int c0 = 0; c0 < C; c0 += kWG) {
    const int CL = (C - c0) < kWG ? (C - c0) : kWG;           // columns in this sweep
    const int RL = kWG / CL > 0 ? kWG / CL : 1;              // row lanes sharing a column
    const int col = threadIdx.x % CL, rl = threadIdx.x / CL;
    double acc = 0.0;
    if (rl < RL)
      for (int r = rl; r < n; r += RL) acc += (double)x[(r0 + r) * ld + c0 + col];
    s_acc[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < CL) {
      double tot = 0.0;
      for (int k = 0; k < RL; ++k) tot += s_acc[k * CL + threadIdx.x];    // fixed order
      partial[(int64_t)blockIdx.x * C + c0 + threadIdx.x] = tot;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kWG) void seg_colsum_ragged_final_k(const double *__restrict__ partial,
                                                                 const int64_t *__restrict__ seg_tile_ptr, int64_t nseg, int C,
                                                                 const float *__restrict__ scale, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x;
  if (i >= nseg * C) return;
  const int64_t g = i / C;
  const int c = (int)(i - g * C);
  double tot = 0.0;
  for (int64_t k = seg_tile_ptr[g]; k < seg_tile_ptr[g + 1]; ++k) tot += partial[k * C + c];
  if (scale) tot *= (double)scale[g];
  out[i] = (float)tot;
}

// dst[r, :] = src[mesh(r), :] for the rows of every tile (the per-mesh mean broadcast into a concat buffer's second half)
__global__ __launch_bounds__(kWG) void bcast_rows_ragged_k(const float *__restrict__ src, const int64_t *__restrict__ tiles,
                                                           float *__restrict__ dst, int64_t ldd, int C) {
  const int64_t *t = tiles + 3 * (int64_t)blockIdx.x;
  const float *row = src + t[0] * C;
  const int64_t r0 = t[1];
  const int n = (int)t[2];
  const int c4 = C / 4;
  for (int i = threadIdx.x; i < n * c4; i += kWG) {
    const int r = i / c4, c = (i - r * c4) * 4;
    st4_s(dst + (r0 + r) * ldd + c, ld4(row + c), kStreamNT);
  }
}

// ------------------------------------------------------------------------------------------------
// ELU helpers (alpha = 1, as F.elu defaults in the reference).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }

template <bool VEC>
__global__ __launch_bounds__(kWG) void elu_into_k(const float *__restrict__ src, int64_t lds,
                                                  float *__restrict__ dst, int64_t ldd, int64_t rows,
                                                  int C, int nt) {
  constexpr int W = VEC ? 4 : 1;
  const int cw = C / W;
  const int64_t total = rows * cw;
  for (int64_t t = (int64_t)blockIdx.x * kWG + threadIdx.x; t < total; t += (int64_t)gridDim.x * kWG) {
    const int64_t r = t / cw;
    const int c = (int)(t - r * cw) * W;
    if constexpr (VEC) {
      f4 v = ld4_s(src + r * lds + c, nt);
      v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w);
      st4_s(dst + r * ldd + c, v, nt);
    } else {
      dst[r * ldd + c] = elu1(src[r * lds + c]);
    }
  }
}

template <bool VEC, bool ACC>
__global__ __launch_bounds__(kWG) void elu_bwd_k(const float *__restrict__ gdst, int64_t ldg,
                                                 const float *__restrict__ gdst2, int64_t ldg2,
                                                 const float *__restrict__ gadd, int64_t ldga,
                                                 const float *__restrict__ out, int64_t ldo,
                                                 float *__restrict__ gsrc, int64_t ldgs, int64_t rows,
                                                 int C, int nt) {
  constexpr int W = VEC ? 4 : 1;
  const int cw = C / W;
  const int64_t total = rows * cw;
  for (int64_t t = (int64_t)blockIdx.x * kWG + threadIdx.x; t < total; t += (int64_t)gridDim.x * kWG) {
    const int64_t r = t / cw;
    const int c = (int)(t - r * cw) * W;
    if constexpr (VEC) {
      f4 g = ld4_s(gdst + r * ldg + c, nt);
      if (gdst2) g += ld4_s(gdst2 + r * ldg2 + c, nt);
      const f4 o = ld4_s(out + r * ldo + c, nt);
      f4 d;
      d.x = g.x * (o.x > 0.f ? 1.f : o.x + 1.f);
      d.y = g.y * (o.y > 0.f ? 1.f : o.y + 1.f);
      d.z = g.z * (o.z > 0.f ? 1.f : o.z + 1.f);
      d.w = g.w * (o.w > 0.f ? 1.f : o.w + 1.f);
      if (gadd) d += ld4_s(gadd + r * ldga + c, nt);
      float *p = gsrc + r * ldgs + c;
      if constexpr (ACC) {
        const f4 a = ld4_s(p, nt);
        d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
      }
      st4_s(p, d, nt);
    } else {
      const float o = out[r * ldo + c];
      const float d = (gdst[r * ldg + c] + (gdst2 ? gdst2[r * ldg2 + c] : 0.f)) * (o > 0.f ? 1.f : o + 1.f) +
                      (gadd ? gadd[r * ldga + c] : 0.f);
      float *p = gsrc + r * ldgs + c;
      *p = ACC ? *p + d : d;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host-side helpers
// ------------------------------------------------------------------------------------------------
inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SN_OK : (int)e;
}

// Elementwise passes: one 16-byte item per thread and as many workgroups as that takes (measured faster than a capped
// grid-stride loop: 6.5 vs 4.9 TB/s on a 1 GiB copy, tools/scratch/copybench.hip), non-temporal loads/stores (kStreamNT).
inline unsigned grid_for(int64_t work_items, int per_block) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  return (unsigned)(b < (int64_t)INT_MAX ? b : (int64_t)INT_MAX);
}

// XCDs of the current device: hipDeviceAttributeNumberOfXccs, read once per device (a value that is not 1, 2, 4 or 8 — or a
// failed query — keeps 8); the device-side copy c_xcd is rewritten the first time a device reports something else.
inline int xcd_count() {
  static std::mutex mu;
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 8;
  std::lock_guard<std::mutex> lk(mu);
  if (cached[dev] == 0) {
    int x = 8;
    if (hipDeviceGetAttribute(&x, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || !(x == 1 || x == 2 || x == 4 || x == 8)) x = 8;
    if (x != 8 && hipMemcpyToSymbol(HIP_SYMBOL(c_xcd), &x, sizeof(int)) != hipSuccess) x = 8;
    (void)hipGetLastError();
    cached[dev] = x;
  }
  return cached[dev];
}

// grid of the chunked kernels: one workgroup per chunk, rounded up to a multiple of the XCD count (see my_chunk)
inline unsigned chunk_grid(int64_t nchunks) {
  const int X = xcd_count();
  int64_t b = ((nchunks + X - 1) / X) * X;
  if (b < X) b = X;
  return (unsigned)b;
}

inline bool fits_i32(int64_t v) { return v >= 0 && v <= (int64_t)INT_MAX; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int check_dense(const float *P, int64_t ld, int group, int N) {
  if (!P) return SN_E_NULL;
  if (group != 1 && group != 4) return SN_E_LD;
  if (group == 1 && ld < N) return SN_E_LD;
  if (group == 4 && ld < 4 * (int64_t)N) return SN_E_LD;
  return SN_OK;
}

int exclusive_scan_i32(const int *in, int64_t n, int *out, void *ws, size_t ws_bytes, hipStream_t s) {
  const int64_t nblk = (n + kScanTile - 1) / kScanTile;
  if (ws_bytes < (size_t)nblk * sizeof(int)) return SN_E_WORKSPACE;
  if (n <= 0) return SN_OK;
  int *sums = static_cast<int *>(ws);
  hipLaunchKernelGGL(scan_block_sums, dim3((unsigned)nblk), dim3(kWG), 0, s, in, n, sums);
  hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(kWG), 0, s, sums, (int)nblk);
  hipLaunchKernelGGL(scan_apply, dim3((unsigned)nblk), dim3(kWG), 0, s, in, n, sums, out);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------
// Optional per-launch timing of the SpMM kernels (profiling aid, off by default): hipExtLaunchKernelGGL stamps the
// kernel's own start/stop into two events, so the duration carries no marker or kernel-boundary overhead and agrees
// with rocprofv3's kernel trace.  The only global state of the library; guarded by a mutex.
// ------------------------------------------------------------------------------------------------
struct TimedLaunch {
  hipEvent_t start, stop;
  int64_t meta[5];          // kind (0 csr, 1 bsr4), M, K, nnz (csr) or nblocks (bsr4), N
};
std::mutex g_timing_mu;
bool g_timing_on = false;
std::vector<TimedLaunch> g_timing;

inline bool timing_slot(int kind, int64_t M, int64_t K, int64_t nnz, int N, hipEvent_t *s, hipEvent_t *e) {
  *s = *e = nullptr;
  std::lock_guard<std::mutex> lk(g_timing_mu);
  if (!g_timing_on) return false;
  TimedLaunch t;
  if (hipEventCreate(&t.start) != hipSuccess) return false;
  if (hipEventCreate(&t.stop) != hipSuccess) {
    (void)hipEventDestroy(t.start);
    return false;
  }
  t.meta[0] = kind; t.meta[1] = M; t.meta[2] = K; t.meta[3] = nnz; t.meta[4] = N;
  g_timing.push_back(t);
  *s = t.start;
  *e = t.stop;
  return true;
}

// launch with (t_start, t_stop) of the enclosing entry point when timing is on
#define SN_KLAUNCH(KERNEL, grid, stream, ...)                                                                  \
  do {                                                                                                         \
    if (t_start) hipExtLaunchKernelGGL(KERNEL, dim3(grid), dim3(kWG), 0, stream, t_start, t_stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(kWG), 0, stream, __VA_ARGS__);                            \
  } while (0)

#define SN_DISPATCH_G(KERNEL, N, xg, yg, grid, stream, ...)                                                    \
  do {                                                                                                         \
    if (xg == 1 && yg == 1) SN_KLAUNCH((KERNEL<N, 1, 1>), grid, stream, __VA_ARGS__);                          \
    else if (xg == 4 && yg == 4) SN_KLAUNCH((KERNEL<N, 4, 4>), grid, stream, __VA_ARGS__);                     \
    else if (xg == 1 && yg == 4) SN_KLAUNCH((KERNEL<N, 1, 4>), grid, stream, __VA_ARGS__);                     \
    else SN_KLAUNCH((KERNEL<N, 4, 1>), grid, stream, __VA_ARGS__);                                             \
  } while (0)
#define SN_DISPATCH_N(KERNEL, N, xg, yg, grid, stream, ...)                                \
  do {                                                                                     \
    switch (N) {                                                                           \
      case 16: SN_DISPATCH_G(KERNEL, 16, xg, yg, grid, stream, __VA_ARGS__); break;        \
      case 32: SN_DISPATCH_G(KERNEL, 32, xg, yg, grid, stream, __VA_ARGS__); break;        \
      case 64: SN_DISPATCH_G(KERNEL, 64, xg, yg, grid, stream, __VA_ARGS__); break;        \
      default: SN_DISPATCH_G(KERNEL, 128, xg, yg, grid, stream, __VA_ARGS__); break;       \
    }                                                                                      \
  } while (0)

}  // namespace

// Compute units of the current device (hipDeviceAttributeMultiprocessorCount, read once per device): what every "one workgroup
// per CU" / "whole rounds of the chip" grid of the three translation units is sized by — 256 on an MI355X in SPX mode, fewer in
// the CPX / DPX / QPX partition modes.  256 when there is no device to ask (host-only callers of the *_blocks / *_bytes queries).
int sn_internal_cu_count() {
  static std::mutex mu;
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    (void)hipGetLastError();
    return 256;
  }
  std::lock_guard<std::mutex> lk(mu);
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    (void)hipGetLastError();
    cached[dev] = n;
  }
  return cached[dev];
}

namespace {

// Fills and device-to-device copies are kernels of this library, not hipMemsetAsync / hipMemcpy2DAsync: a hipMemsetAsync of more
// than 4 bytes captured into a hipGraph replays wrongly on this runtime (ROCm 7.2; tools/scratch/plan_memset_graph.py: from the second
// replay on 3 bytes of every 16 are not the fill value), and every entry point has to write the same bytes launched directly, under
// stream capture and on replay.
template <class U>
__global__ void __launch_bounds__(256) fill2d_k(char *__restrict__ dst, int64_t pitch, int64_t wunits, int64_t total, U v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / wunits, c = i - r * wunits;
    *(U *)(dst + r * pitch + c * (int64_t)sizeof(U)) = v;
  }
}
template <class U>
__global__ void __launch_bounds__(256) copy2d_k(char *__restrict__ dst, int64_t dpitch, const char *__restrict__ src, int64_t spitch,
                                                     int64_t wunits, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / wunits, c = i - r * wunits;
    *(U *)(dst + r * dpitch + c * (int64_t)sizeof(U)) = *(const U *)(src + r * spitch + c * (int64_t)sizeof(U));
  }
}
inline unsigned fill_grid(int64_t total) {
  const int64_t want = (total + 255) / 256, cap = (int64_t)sn_internal_cu_count() * 16;
  return (unsigned)(want < cap ? (want < 1 ? 1 : want) : cap);
}
inline int fill_unit(uint64_t bits) { return (bits & 15) == 0 ? 16 : (bits & 3) == 0 ? 4 : 1; }

}  // namespace

hipError_t sn_internal_fill2d(void *dst, int64_t pitch, int value, int64_t width, int64_t rows, hipStream_t s) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows == 1) pitch = 0;
  const int u = fill_unit((uint64_t)(uintptr_t)dst | (uint64_t)pitch | (uint64_t)width);
  const int64_t wunits = width / u, total = wunits * rows;
  const uint32_t b = (uint32_t)(value & 0xff) * 0x01010101u;
  if (u == 16) fill2d_k<uint4><<<fill_grid(total), 256, 0, s>>>((char *)dst, pitch, wunits, total, make_uint4(b, b, b, b));
  else if (u == 4) fill2d_k<uint32_t><<<fill_grid(total), 256, 0, s>>>((char *)dst, pitch, wunits, total, b);
  else fill2d_k<uint8_t><<<fill_grid(total), 256, 0, s>>>((char *)dst, pitch, wunits, total, (uint8_t)b);
  return hipGetLastError();
}
hipError_t sn_internal_copy2d(void *dst, int64_t dpitch, const void *src, int64_t spitch, int64_t width, int64_t rows, hipStream_t s) {
  (void)hipGetLastError();
  if (rows == 1) dpitch = spitch = 0;
  const int u = fill_unit((uint64_t)(uintptr_t)dst | (uint64_t)(uintptr_t)src | (uint64_t)dpitch | (uint64_t)spitch | (uint64_t)width);
  const int64_t wunits = width / u, total = wunits * rows;
  if (u == 16) copy2d_k<uint4><<<fill_grid(total), 256, 0, s>>>((char *)dst, dpitch, (const char *)src, spitch, wunits, total);
  else if (u == 4) copy2d_k<uint32_t><<<fill_grid(total), 256, 0, s>>>((char *)dst, dpitch, (const char *)src, spitch, wunits, total);
  else copy2d_k<uint8_t><<<fill_grid(total), 256, 0, s>>>((char *)dst, dpitch, (const char *)src, spitch, wunits, total);
  return hipGetLastError();
}

hipError_t sn_internal_fill(void *dst, int value, size_t bytes, hipStream_t s) {
  return bytes ? sn_internal_fill2d(dst, 0, value, (int64_t)bytes, 1, s) : hipSuccess;
}

// The same timing slot for the Linear-layer launchers of the other translation units (sn_gemm.hip, sn_dense.hip): kind
// 0x100 forward / 0x200 input gradient / 0x400 weight gradient (+ a variant number in the low byte), then rows, the contraction
// or input width, the ALGORITHMIC bytes of the launch (operands read + results written, weights excluded) and the output width.
bool g_timing_linear = true;      // sn_timing_enable(2): the sparse products only
bool sn_internal_timing_on() {
  std::lock_guard<std::mutex> lk(g_timing_mu);
  return g_timing_on;
}
bool sn_internal_timing_slot(int kind, int64_t rows, int64_t width, int64_t bytes, int outw, hipEvent_t *s, hipEvent_t *e) {
  {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (!g_timing_linear) return false;
  }
  return timing_slot(kind, rows, width, bytes, outw, s, e);
}

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

int sn_abi_version(void) { return SN_ABI_VERSION; }

const char *sn_status_string(int status) {
  switch (status) {
    case SN_OK: return "ok";
    case SN_E_NULL: return "null pointer argument";
    case SN_E_SHAPE: return "invalid shape";
    case SN_E_RANGE: return "dimension or nnz exceeds int32 index range";
    case SN_E_LD: return "invalid leading dimension / group layout";
    case SN_E_ALIGN: return "misaligned pointer";
    case SN_E_WORKSPACE: return "workspace too small";
    case SN_E_UNSUPPORTED: return "unsupported configuration";
    default: break;
  }
  if (status > 0) return hipGetErrorString((hipError_t)status);
  return "unknown sn status";
}

constexpr int kSpmmStatsBlocks = 128;        // partial rows after stage 1 of the statistics reduction

// rows kernel: passes per wave.  As many as keep >= 8 workgroups per CU in the grid (small batches stay wide), at most 64 rows.
static int csr_rows_iters(int64_t M, int N) {
  const int P = 64 / (N / 4);
  int it = 2;                              // (measured best on the config-4 / config-5 Laplacian batches: 2..4 passes)
  if (it * P > 64) it = 64 / P;
  while (it > 1 && (M + (int64_t)4 * P * it - 1) / ((int64_t)4 * P * it) < 8 * kCUs) it >>= 1;
  return it < 1 ? 1 : it;
}
static int64_t csr_rows_chunks(int64_t M, int N, int iters) {
  const int64_t rows_per_wg = (int64_t)(kWG / 64) * (64 / (N / 4)) * iters;
  return (M + rows_per_wg - 1) / rows_per_wg;
}

static int spmm_csr_launch(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K,
                           int64_t nnz, const float *X, int64_t ldx, int32_t x_group, int32_t N, float *Y, int64_t ldy,
                           int32_t y_group, SpmmEpi epi, void *stream, float *stats_part = nullptr, double *stats_out = nullptr) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (M < 0 || K < 0 || nnz < 0 || N < 1) return SN_E_SHAPE;
  if (!fits_i32(M + 1) || !fits_i32(K) || !fits_i32(nnz)) return SN_E_RANGE;
  if (M == 0) return SN_OK;
  if (!rowptr || (nnz > 0 && (!colind || !vals))) return SN_E_NULL;
  int st = check_dense(Y, ldy, y_group, N);
  if (st) return st;
  if (K > 0 || nnz > 0) {
    st = check_dense(X, ldx, x_group, N);
    if (st) return st;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (N == 16 || N == 32 || N == 64 || N == 128) && aligned16(X) && aligned16(Y) &&
                   (ldx % 4 == 0) && (ldy % 4 == 0);
  if (epi.e) {                         // fused epilogue: vector kernels only, operands like Y
    if (!vec) return SN_E_UNSUPPORTED;
    st = check_dense(epi.e, epi.lde, y_group, N);
    if (!st && epi.g) st = check_dense(epi.g, epi.ldg, y_group, N);
    if (st) return st;
    if (!aligned16(epi.e) || epi.lde % 4 || (epi.g && (!aligned16(epi.g) || epi.ldg % 4))) return SN_E_ALIGN;
  }
  hipEvent_t t_start, t_stop;
  timing_slot(0 | (epi.e ? 2 : 0) | (epi.g ? 4 : 0) | (stats_part ? 16 : 0), M, K, nnz, N, &t_start, &t_stop);
  if (stats_part) {
    if (!vec || N != 128 || y_group != 1 || epi.e) return SN_E_UNSUPPORTED;
    if (!stats_out) return SN_E_NULL;
    const int iters = csr_rows_iters(M, N);
    const int64_t nchunks = csr_rows_chunks(M, N, iters);
    const unsigned grid = chunk_grid(nchunks);
    if (x_group == 1) SN_KLAUNCH((spmm_csr_rows_stats<1>), grid, s, rowptr, colind, vals, (int)M, X, ldx, Y, ldy, (int)nchunks, iters, stats_part);
    else SN_KLAUNCH((spmm_csr_rows_stats<4>), grid, s, rowptr, colind, vals, (int)M, X, ldx, Y, ldy, (int)nchunks, iters, stats_part);
    hipLaunchKernelGGL(spmm_stats_reduce_k, dim3(kSpmmStatsBlocks), dim3(kWG), 0, s, stats_part, (int64_t)grid, stats_out);
    return launch_status();
  }
  if (vec) {
    const int iters = csr_rows_iters(M, N);
    const int64_t nchunks = csr_rows_chunks(M, N, iters);
    const unsigned grid = chunk_grid(nchunks);
    if (epi.e)
      SN_DISPATCH_N(spmm_csr_rows_epi, N, x_group, y_group, grid, s, rowptr, colind, vals, (int)M, X, ldx, Y, ldy, (int)nchunks, iters, epi);
    else
      SN_DISPATCH_N(spmm_csr_rows, N, x_group, y_group, grid, s, rowptr, colind, vals, (int)M, X, ldx, Y, ldy, (int)nchunks, iters);
  } else {
    SN_KLAUNCH(spmm_csr_any, grid_for(M * (int64_t)N, kWG), s, rowptr, colind, vals, M, (int)N, X, ldx, (int)x_group, Y,
               ldy, (int)y_group);
  }
  return launch_status();
}

int sn_spmm_csr_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M,
                    int64_t K, int64_t nnz, const float *X, int64_t ldx, int32_t x_group, int32_t N,
                    float *Y, int64_t ldy, int32_t y_group, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  return spmm_csr_launch(rowptr, colind, vals, M, K, nnz, X, ldx, x_group, N, Y, ldy, y_group, SpmmEpi{nullptr, 0, nullptr, 0},
                         stream);
}

int sn_spmm_csr_elubwd_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K,
                           int64_t nnz, const float *X, int64_t ldx, int32_t x_group, int32_t N, const float *E,
                           int64_t lde, const float *G, int64_t ldg, float *Y, int64_t ldy, int32_t y_group,
                           void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (!E) return SN_E_NULL;
  return spmm_csr_launch(rowptr, colind, vals, M, K, nnz, X, ldx, x_group, N, Y, ldy, y_group, SpmmEpi{E, lde, G, ldg}, stream);
}

size_t sn_spmm_csr_stats_workspace_bytes(int64_t M) {
  if (M < 1) return 0;
  return (size_t)chunk_grid(csr_rows_chunks(M, 128, csr_rows_iters(M, 128))) * 256 * sizeof(float);
}

int sn_spmm_csr_stats_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                          const float *X, int64_t ldx, int32_t x_group, int32_t N, float *Y, int64_t ldy, int32_t y_group,
                          double *stats_part, void *workspace, size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (!stats_part || !workspace) return SN_E_NULL;
  if (M < 1) return SN_E_SHAPE;
  if (workspace_bytes < sn_spmm_csr_stats_workspace_bytes(M)) return SN_E_WORKSPACE;
  return spmm_csr_launch(rowptr, colind, vals, M, K, nnz, X, ldx, x_group, N, Y, ldy, y_group, SpmmEpi{nullptr, 0, nullptr, 0},
                         stream, static_cast<float *>(workspace), stats_part);
}

// rb4 kernels: passes per wave, as csr_rows_iters but counted in 4-row groups
static int rb4_iters(int64_t Mb, int N) {
  const int P = 64 / (N / 4);
  // One pass per wave by default: with more, the row ranges the resident waves of an XCD walk concurrently no longer fit
  // its 4 MiB L2 together with their neighbours' X rows (measured: 0.59 -> 0.47 of the roofline from 1 to 8 passes).
  (void)Mb;
  (void)P;
  return 1;
}
static int64_t rb4_chunks(int64_t Mb, int N, int iters) {
  const int64_t per_wg = (int64_t)(kWG / 64) * (64 / (N / 4)) * iters;
  return (Mb + per_wg - 1) / per_wg;
}

int sn_rb4_count(const int32_t *rowptr, const int32_t *colind, int64_t M, int64_t K, int32_t *b_ptr, void *workspace,
                 size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (M < 0 || K < 0) return SN_E_SHAPE;
  if (!fits_i32(M + 4) || !fits_i32(K)) return SN_E_RANGE;
  if (!rowptr || !b_ptr) return SN_E_NULL;
  const int64_t Mb = (M + 3) / 4;
  if (workspace_bytes < sn_scan_workspace_bytes(Mb + 1) || !workspace) return SN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = sn_internal_fill(b_ptr + Mb, 0, sizeof(int), s);
  if (e != hipSuccess) return (int)e;
  if (Mb > 0)
    hipLaunchKernelGGL((rb4_merge<false>), dim3(grid_for(Mb, kWG)), dim3(kWG), 0, s, rowptr, colind, (const float *)nullptr, M, Mb,
                       b_ptr, (int *)nullptr, (float *)nullptr);
  return exclusive_scan_i32(b_ptr, Mb + 1, b_ptr, workspace, workspace_bytes, s);
}

int sn_rb4_fill(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, const int32_t *b_ptr,
                int32_t *b_col, float *b_val, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (M < 0 || K < 0) return SN_E_SHAPE;
  if (!rowptr || !b_ptr) return SN_E_NULL;
  const int64_t Mb = (M + 3) / 4;
  if (Mb == 0) return SN_OK;
  if (!colind || !vals || !b_col || !b_val) return SN_E_NULL;
  if (!aligned16(b_val)) return SN_E_ALIGN;
  hipLaunchKernelGGL((rb4_merge<true>), dim3(grid_for(Mb, kWG)), dim3(kWG), 0, static_cast<hipStream_t>(stream), rowptr, colind,
                     vals, M, Mb, const_cast<int *>(b_ptr), b_col, b_val);
  return launch_status();
}

static int spmm_rb4_launch(const int32_t *b_ptr, const int32_t *b_col, const float *b_val, int64_t M, int64_t K, int64_t capacity,
                           const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, SpmmEpi epi, void *stream,
                           float *stats_part = nullptr, double *stats_out = nullptr) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (M < 0 || K < 0 || capacity < 0 || N < 1) return SN_E_SHAPE;
  if (!fits_i32(M + 4) || !fits_i32(K) || !fits_i32(capacity)) return SN_E_RANGE;
  if (M == 0) return SN_OK;
  if (!b_ptr || (capacity > 0 && (!b_col || !b_val))) return SN_E_NULL;
  int st = check_dense(Y, ldy, 1, N);
  if (st) return st;
  if (K > 0 || capacity > 0) {
    st = check_dense(X, ldx, 1, N);
    if (st) return st;
  }
  if (N != 64 && N != 128) return SN_E_UNSUPPORTED;
  if (!aligned16(X) || !aligned16(Y) || !aligned16(b_val) || ldx % 4 || ldy % 4) return SN_E_ALIGN;
  if (epi.e) {
    st = check_dense(epi.e, epi.lde, 1, N);
    if (!st && epi.g) st = check_dense(epi.g, epi.ldg, 1, N);
    if (st) return st;
    if (!aligned16(epi.e) || epi.lde % 4 || (epi.g && (!aligned16(epi.g) || epi.ldg % 4))) return SN_E_ALIGN;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t t_start, t_stop;
  timing_slot(32 | (epi.e ? 2 : 0) | (epi.g ? 4 : 0) | (stats_part ? 16 : 0), M, K, capacity, N, &t_start, &t_stop);
  const int64_t Mb = (M + 3) / 4;
  const int iters = rb4_iters(Mb, N);
  const int64_t nchunks = rb4_chunks(Mb, N, iters);
  const unsigned grid = chunk_grid(nchunks);
  const f4 *bv = reinterpret_cast<const f4 *>(b_val);
  if (stats_part) {
    if (epi.e || N != 128) return SN_E_UNSUPPORTED;
    if (!stats_out) return SN_E_NULL;
    SN_KLAUNCH(spmm_rb4_stats, grid, s, b_ptr, b_col, bv, (int)M, (int)Mb, X, ldx, Y, ldy, (int)nchunks, iters, stats_part);
    hipLaunchKernelGGL(spmm_stats_reduce_k, dim3(kSpmmStatsBlocks), dim3(kWG), 0, s, stats_part, (int64_t)grid, stats_out);
    return launch_status();
  }
  if (epi.e) {
    if (N == 128) SN_KLAUNCH((spmm_rb4_epi<128>), grid, s, b_ptr, b_col, bv, (int)M, (int)Mb, X, ldx, Y, ldy, (int)nchunks, iters, epi);
    else SN_KLAUNCH((spmm_rb4_epi<64>), grid, s, b_ptr, b_col, bv, (int)M, (int)Mb, X, ldx, Y, ldy, (int)nchunks, iters, epi);
  } else {
    // 4 gathers in flight per lane: measured best (config-4 / config-5 batches, MI355X) against 2 / 8 / 12 / 16 — the
    // vector cache (32 KiB per CU) holds 64 X rows of 128 channels, and every additional gather in flight evicts lines that
    // neighbouring row groups are about to re-use
    if (N == 128) SN_KLAUNCH((spmm_rb4<128, 4>), grid, s, b_ptr, b_col, bv, (int)M, (int)Mb, X, ldx, Y, ldy, (int)nchunks, iters);
    else SN_KLAUNCH((spmm_rb4<64, 4>), grid, s, b_ptr, b_col, bv, (int)M, (int)Mb, X, ldx, Y, ldy, (int)nchunks, iters);
  }
  return launch_status();
}

int sn_spmm_rb4_f32(const int32_t *b_ptr, const int32_t *b_col, const float *b_val, int64_t M, int64_t K, int64_t capacity,
                    const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, void *stream) {
  return spmm_rb4_launch(b_ptr, b_col, b_val, M, K, capacity, X, ldx, N, Y, ldy, SpmmEpi{nullptr, 0, nullptr, 0}, stream);
}

int sn_spmm_rb4_elubwd_f32(const int32_t *b_ptr, const int32_t *b_col, const float *b_val, int64_t M, int64_t K,
                           int64_t capacity, const float *X, int64_t ldx, int32_t N, const float *E, int64_t lde,
                           const float *G, int64_t ldg, float *Y, int64_t ldy, void *stream) {
  if (!E) return SN_E_NULL;
  return spmm_rb4_launch(b_ptr, b_col, b_val, M, K, capacity, X, ldx, N, Y, ldy, SpmmEpi{E, lde, G, ldg}, stream);
}

// the same launch, and max |Y| of every wave in y_absmax[sn_spmm_rb4_absmax_blocks(M, N)] (all entries written): the bound the
// two-piece weight gradient of the layer below needs for its dy operand (sn_wgrad_bounded_f32)
int64_t sn_spmm_rb4_absmax_blocks(int64_t M, int32_t N) {
  if (M < 1 || (N != 64 && N != 128)) return 0;
  const int64_t Mb = (M + 3) / 4;
  return (int64_t)chunk_grid(rb4_chunks(Mb, N, rb4_iters(Mb, N))) * (kWG / 64);
}
int sn_spmm_rb4_elubwd_absmax_f32(const int32_t *b_ptr, const int32_t *b_col, const float *b_val, int64_t M, int64_t K,
                                  int64_t capacity, const float *X, int64_t ldx, int32_t N, const float *E, int64_t lde,
                                  const float *G, int64_t ldg, float *Y, int64_t ldy, float *y_absmax, void *stream) {
  if (!E || !y_absmax) return SN_E_NULL;
  if (M < 1) return SN_E_SHAPE;
  return spmm_rb4_launch(b_ptr, b_col, b_val, M, K, capacity, X, ldx, N, Y, ldy, SpmmEpi{E, lde, G, ldg, y_absmax}, stream);
}

size_t sn_spmm_rb4_stats_workspace_bytes(int64_t M) {
  if (M < 1) return 0;
  const int64_t Mb = (M + 3) / 4;
  return (size_t)chunk_grid(rb4_chunks(Mb, 128, rb4_iters(Mb, 128))) * 256 * sizeof(float);
}

int sn_spmm_rb4_stats_f32(const int32_t *b_ptr, const int32_t *b_col, const float *b_val, int64_t M, int64_t K,
                          int64_t capacity, const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, double *stats_part,
                          void *workspace, size_t workspace_bytes, void *stream) {
  if (!stats_part || !workspace) return SN_E_NULL;
  if (M < 1) return SN_E_SHAPE;
  if (workspace_bytes < sn_spmm_rb4_stats_workspace_bytes(M)) return SN_E_WORKSPACE;
  return spmm_rb4_launch(b_ptr, b_col, b_val, M, K, capacity, X, ldx, N, Y, ldy, SpmmEpi{nullptr, 0, nullptr, 0}, stream,
                         static_cast<float *>(workspace), stats_part);
}

// ---- ring kernel (banded square CSR operators, N in {64, 128}) ------------------------------------------------------
// strips: a multiple of 8 (one contiguous eighth per XCD) chosen so that strips x slices = one workgroup per CU
static int ring_strips(int64_t M, int N) {
  const int nsl = N / kRingCS;
  int nstrips = kCUs / nsl;
  const int64_t nsteps = (M + kRingR - 1) / kRingR;
  while (nstrips > xcd_count() && nsteps < (int64_t)nstrips * 4) nstrips >>= 1;      // (a strip shorter than 4 steps is mostly prologue)
  return nstrips;
}

static int spmm_ring_launch(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                            const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, SpmmEpi epi, void *stream,
                            float *stats_ws = nullptr, double *stats_out = nullptr) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (M < 0 || K < 0 || nnz < 0 || N < 1) return SN_E_SHAPE;
  if (!fits_i32(M + kRingR + 1) || !fits_i32(K) || !fits_i32(nnz)) return SN_E_RANGE;
  if (M != K) return SN_E_UNSUPPORTED;                    // the window follows the diagonal
  if (N != 64 && N != 128) return SN_E_UNSUPPORTED;
  if (M == 0) return SN_OK;
  if (!rowptr || (nnz > 0 && (!colind || !vals))) return SN_E_NULL;
  int st = check_dense(Y, ldy, 1, N);
  if (!st) st = check_dense(X, ldx, 1, N);
  if (st) return st;
  if (!aligned16(X) || !aligned16(Y) || ldx % 4 || ldy % 4) return SN_E_ALIGN;
  if (epi.e) {
    st = check_dense(epi.e, epi.lde, 1, N);
    if (!st && epi.g) st = check_dense(epi.g, epi.ldg, 1, N);
    if (st) return st;
    if (!aligned16(epi.e) || epi.lde % 4 || (epi.g && (!aligned16(epi.g) || epi.ldg % 4))) return SN_E_ALIGN;
  }
  if (stats_ws && (N != 128 || epi.e)) return SN_E_UNSUPPORTED;
  if (stats_ws && !stats_out) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (nnz == 0) {                                         // an all-zero operator: Y = 0 (+ the epilogue's G), through the generic kernel
    return spmm_csr_launch(rowptr, colind, vals, M, K, nnz, X, ldx, 1, N, Y, ldy, 1, epi, stream, stats_ws, stats_out);
  }
  // kernels that ask for more than 64 KiB of dynamic LDS have to be told once (per kernel, idempotent)
  static const hipError_t attr_status = [] {
    hipError_t e = hipFuncSetAttribute((const void *)spmm_ring_k<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRingLds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)spmm_ring_k<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRingLds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void *)spmm_ring_k<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRingLds);
    return e;
  }();
  if (attr_status != hipSuccess) return (int)attr_status;
  hipEvent_t t_start, t_stop;
  timing_slot(64 | (epi.e ? 2 : 0) | (epi.g ? 4 : 0) | (stats_ws ? 16 : 0), M, K, nnz, N, &t_start, &t_stop);
  const int nsl = N / kRingCS;
  const int nstrips = ring_strips(M, N);
  const int64_t nsteps = (M + kRingR - 1) / kRingR;
  const int cps = (int)((nsteps + nstrips - 1) / nstrips);
  const dim3 grid((unsigned)(nstrips * nsl)), block(kRingThreads);
#define SN_RING_LAUNCH(EPI_, ST_, PART_)                                                                                       \
  do {                                                                                                                         \
    if (t_start)                                                                                                               \
      hipExtLaunchKernelGGL((spmm_ring_k<EPI_, ST_>), grid, block, kRingLds, s, t_start, t_stop, 0, rowptr, colind, vals,       \
                            (int)M, (int)K, (int)nnz, X, ldx, Y, ldy, nstrips, cps, nsl, epi, PART_);                          \
    else                                                                                                                       \
      hipLaunchKernelGGL((spmm_ring_k<EPI_, ST_>), grid, block, kRingLds, s, rowptr, colind, vals, (int)M, (int)K, (int)nnz,    \
                         X, ldx, Y, ldy, nstrips, cps, nsl, epi, PART_);                                                       \
  } while (0)
  if (stats_ws) {
    SN_RING_LAUNCH(false, true, stats_ws);
    hipLaunchKernelGGL(spmm_stats_reduce_k, dim3(kSpmmStatsBlocks), dim3(kWG), 0, s, stats_ws, (int64_t)nstrips, stats_out);
  } else if (epi.e) {
    SN_RING_LAUNCH(true, false, (float *)nullptr);
  } else {
    SN_RING_LAUNCH(false, false, (float *)nullptr);
  }
#undef SN_RING_LAUNCH
  return launch_status();
}

int sn_spmm_csr_ring_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                         const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, void *stream) {
  return spmm_ring_launch(rowptr, colind, vals, M, K, nnz, X, ldx, N, Y, ldy, SpmmEpi{nullptr, 0, nullptr, 0}, stream);
}

int sn_spmm_csr_ring_elubwd_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K,
                                int64_t nnz, const float *X, int64_t ldx, int32_t N, const float *E, int64_t lde, const float *G,
                                int64_t ldg, float *Y, int64_t ldy, void *stream) {
  if (!E) return SN_E_NULL;
  return spmm_ring_launch(rowptr, colind, vals, M, K, nnz, X, ldx, N, Y, ldy, SpmmEpi{E, lde, G, ldg}, stream);
}

// the same launch, and max |Y| of every compute wave in y_absmax[sn_spmm_csr_ring_absmax_blocks(M, N)] (all entries written);
// an operator without entries is not taken (SN_E_UNSUPPORTED: that launch goes through the generic kernel)
int64_t sn_spmm_csr_ring_absmax_blocks(int64_t M, int32_t N) {
  if (M < 1 || (N != 64 && N != 128)) return 0;
  return (int64_t)ring_strips(M, N) * (N / kRingCS) * kRingNCW;
}
int sn_spmm_csr_ring_elubwd_absmax_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K,
                                       int64_t nnz, const float *X, int64_t ldx, int32_t N, const float *E, int64_t lde,
                                       const float *G, int64_t ldg, float *Y, int64_t ldy, float *y_absmax, void *stream) {
  if (!E || !y_absmax) return SN_E_NULL;
  if (M < 1) return SN_E_SHAPE;
  if (nnz == 0) return SN_E_UNSUPPORTED;
  return spmm_ring_launch(rowptr, colind, vals, M, K, nnz, X, ldx, N, Y, ldy, SpmmEpi{E, lde, G, ldg, y_absmax}, stream);
}

size_t sn_spmm_csr_ring_stats_workspace_bytes(int64_t M) {
  if (M < 1) return 0;
  // (the all-zero operator falls through to the generic kernel: the larger of the two)
  const size_t a = (size_t)ring_strips(M, 128) * 256 * sizeof(float), b = sn_spmm_csr_stats_workspace_bytes(M);
  return a > b ? a : b;
}

int sn_spmm_csr_ring_stats_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                               const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, double *stats_part, void *workspace,
                               size_t workspace_bytes, void *stream) {
  if (!stats_part || !workspace) return SN_E_NULL;
  if (M < 1) return SN_E_SHAPE;
  if (workspace_bytes < sn_spmm_csr_ring_stats_workspace_bytes(M)) return SN_E_WORKSPACE;
  return spmm_ring_launch(rowptr, colind, vals, M, K, nnz, X, ldx, N, Y, ldy, SpmmEpi{nullptr, 0, nullptr, 0}, stream,
                          static_cast<float *>(workspace), stats_part);
}

int32_t sn_spmm_csr_ring_half_window(void) { return kRingH; }

int sn_csr_band_i32(const int32_t *rowptr, const int32_t *colind, int64_t M, int64_t K, int32_t *band_longest_outside, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (M < 0 || K < 0) return SN_E_SHAPE;
  if (!fits_i32(M + 1) || !fits_i32(K)) return SN_E_RANGE;
  if (!band_longest_outside || (M > 0 && !rowptr)) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = sn_internal_fill(band_longest_outside, 0, 3 * sizeof(int32_t), s);
  if (e != hipSuccess) return (int)e;
  if (M == 0) return SN_OK;
  if (!colind) return SN_E_NULL;
  int64_t blocks = (M + kWG - 1) / kWG;
  blocks = blocks < 2048 ? blocks : 2048;
  hipLaunchKernelGGL(csr_band_k, dim3((unsigned)blocks), dim3(kWG), 0, s, rowptr, colind, M, band_longest_outside);
  return launch_status();
}

static int spmm_bsr4_launch(const int32_t *b_rowptr, const int32_t *b_colind, const float *b_vals, int64_t Mb, int64_t Kb,
                            int64_t nblocks, const float *X, int64_t ldx, int32_t x_group, int32_t N, float *Y,
                            int64_t ldy, int32_t y_group, SpmmEpi epi, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (Mb < 0 || Kb < 0 || nblocks < 0 || N < 1) return SN_E_SHAPE;
  if (!fits_i32(4 * Mb + 1) || !fits_i32(4 * Kb) || !fits_i32(nblocks)) return SN_E_RANGE;
  if (Mb == 0) return SN_OK;
  if (!b_rowptr || (nblocks > 0 && (!b_colind || !b_vals))) return SN_E_NULL;
  int st = check_dense(Y, ldy, y_group, N);
  if (st) return st;
  if (Kb > 0 || nblocks > 0) {
    st = check_dense(X, ldx, x_group, N);
    if (st) return st;
  }
  if (!(N == 16 || N == 32 || N == 64 || N == 128)) return SN_E_UNSUPPORTED;
  if (!aligned16(X) || !aligned16(Y) || !aligned16(b_vals) || ldx % 4 || ldy % 4) return SN_E_ALIGN;
  if (epi.e) {
    st = check_dense(epi.e, epi.lde, y_group, N);
    if (!st && epi.g) st = check_dense(epi.g, epi.ldg, y_group, N);
    if (st) return st;
    if (!aligned16(epi.e) || epi.lde % 4 || (epi.g && (!aligned16(epi.g) || epi.ldg % 4))) return SN_E_ALIGN;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t t_start, t_stop;
  timing_slot(1 | (epi.e ? 2 : 0) | (epi.g ? 4 : 0), 4 * Mb, 4 * Kb, nblocks, N, &t_start, &t_stop);
  const int rpb = kWG / (N / 4);
  const int64_t nchunks = (Mb + rpb - 1) / rpb;
  const unsigned grid = chunk_grid(nchunks);
  if (epi.e)
    SN_DISPATCH_N(spmm_bsr4_lds_epi, N, x_group, y_group, grid, s, b_rowptr, b_colind, b_vals, (int)Mb, X, ldx, Y, ldy, (int)nchunks, epi);
  else
    SN_DISPATCH_N(spmm_bsr4_lds, N, x_group, y_group, grid, s, b_rowptr, b_colind, b_vals, (int)Mb, X, ldx, Y, ldy, (int)nchunks);
  return launch_status();
}

int sn_spmm_bsr4_f32(const int32_t *b_rowptr, const int32_t *b_colind, const float *b_vals, int64_t Mb,
                     int64_t Kb, int64_t nblocks, const float *X, int64_t ldx, int32_t x_group,
                     int32_t N, float *Y, int64_t ldy, int32_t y_group, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  return spmm_bsr4_launch(b_rowptr, b_colind, b_vals, Mb, Kb, nblocks, X, ldx, x_group, N, Y, ldy, y_group,
                          SpmmEpi{nullptr, 0, nullptr, 0}, stream);
}

int sn_spmm_bsr4_elubwd_f32(const int32_t *b_rowptr, const int32_t *b_colind, const float *b_vals, int64_t Mb, int64_t Kb,
                            int64_t nblocks, const float *X, int64_t ldx, int32_t x_group, int32_t N, const float *E,
                            int64_t lde, const float *G, int64_t ldg, float *Y, int64_t ldy, int32_t y_group,
                            void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (!E) return SN_E_NULL;
  return spmm_bsr4_launch(b_rowptr, b_colind, b_vals, Mb, Kb, nblocks, X, ldx, x_group, N, Y, ldy, y_group,
                          SpmmEpi{E, lde, G, ldg}, stream);
}

static int spmm_q3_launch(const int32_t *b_rowptr, const float *q_blk, int64_t Mb, int64_t Kb, int64_t nblocks, const float *X,
                          int64_t ldx, int32_t x_group, int32_t N, float *Y, int64_t ldy, int32_t y_group, SpmmEpi epi,
                          void *stream, float *stats_part = nullptr, double *stats_out = nullptr) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (Mb < 0 || Kb < 0 || nblocks < 0 || N < 1) return SN_E_SHAPE;
  if (!fits_i32(4 * Mb + 1) || !fits_i32(4 * Kb) || !fits_i32(nblocks)) return SN_E_RANGE;
  if (Mb == 0) return SN_OK;
  if (!b_rowptr || (nblocks > 0 && !q_blk)) return SN_E_NULL;
  int st = check_dense(Y, ldy, y_group, N);
  if (st) return st;
  if (Kb > 0 || nblocks > 0) {
    st = check_dense(X, ldx, x_group, N);
    if (st) return st;
  }
  if (!(N == 16 || N == 32 || N == 64 || N == 128)) return SN_E_UNSUPPORTED;
  if (!aligned16(X) || !aligned16(Y) || !aligned16(q_blk) || ldx % 4 || ldy % 4) return SN_E_ALIGN;
  if (epi.e) {
    st = check_dense(epi.e, epi.lde, y_group, N);
    if (!st && epi.g) st = check_dense(epi.g, epi.ldg, y_group, N);
    if (st) return st;
    if (!aligned16(epi.e) || epi.lde % 4 || (epi.g && (!aligned16(epi.g) || epi.ldg % 4))) return SN_E_ALIGN;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t t_start, t_stop;
  timing_slot(1 | 8 | (epi.e ? 2 : 0) | (epi.g ? 4 : 0) | (stats_part ? 16 : 0), 4 * Mb, 4 * Kb, nblocks, N, &t_start, &t_stop);
  const int rpb = kWG / (N / 4);
  const int64_t nchunks = (Mb + rpb - 1) / rpb;
  const unsigned grid = chunk_grid(nchunks);
  const f4 *q = reinterpret_cast<const f4 *>(q_blk);
  // face-output products (3 blocks per block row) take the wide shape, vertex-output products (~6) the deep one
  // (a grid of fewer than two rounds of the deep shape's 4 workgroups per CU — the Mesh-MNIST batches — also runs wide: at
  // 1 200 workgroups the deep shape leaves a 17 % tail, measured 0.67 -> 0.77 on the config-2 vertex-output products)
  const bool wide = nblocks <= 4 * Mb || grid < 8u * kCUs;
  if (stats_part) {
    if (epi.e || (N != 32 && N != 16) || y_group != 4) return SN_E_UNSUPPORTED;
    if (!stats_out) return SN_E_NULL;
    if (N == 16) {          // 64-channel operands (the Mesh-MNIST models): partials [grid][2][64]
      if (wide) {
        if (x_group == 4) SN_KLAUNCH((spmm_q3_lds_stats_wide<16, 4, 4>), grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks, stats_part);
        else SN_KLAUNCH((spmm_q3_lds_stats_wide<16, 1, 4>), grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks, stats_part);
      } else {
        if (x_group == 4) SN_KLAUNCH((spmm_q3_lds_stats<16, 4, 4>), grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks, stats_part);
        else SN_KLAUNCH((spmm_q3_lds_stats<16, 1, 4>), grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks, stats_part);
      }
      hipLaunchKernelGGL(spmm_stats_reduce_k, dim3(kSpmmStatsBlocks), dim3(kWG), 0, s, stats_part, (int64_t)grid, stats_out, 128);
      return launch_status();
    }
    if (wide) {
      if (x_group == 4) SN_KLAUNCH((spmm_q3_lds_stats_wide<32, 4, 4>), grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks, stats_part);
      else SN_KLAUNCH((spmm_q3_lds_stats_wide<32, 1, 4>), grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks, stats_part);
    } else {
      if (x_group == 4) SN_KLAUNCH((spmm_q3_lds_stats<32, 4, 4>), grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks, stats_part);
      else SN_KLAUNCH((spmm_q3_lds_stats<32, 1, 4>), grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks, stats_part);
    }
    hipLaunchKernelGGL(spmm_stats_reduce_k, dim3(kSpmmStatsBlocks), dim3(kWG), 0, s, stats_part, (int64_t)grid, stats_out);
    return launch_status();
  }
  if (epi.e) {
    if (wide) SN_DISPATCH_N(spmm_q3_lds_epi_wide, N, x_group, y_group, grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks, epi);
    else SN_DISPATCH_N(spmm_q3_lds_epi, N, x_group, y_group, grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks, epi);
  } else {
    if (wide) SN_DISPATCH_N(spmm_q3_lds_wide, N, x_group, y_group, grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks);
    else SN_DISPATCH_N(spmm_q3_lds, N, x_group, y_group, grid, s, b_rowptr, q, (int)Mb, X, ldx, Y, ldy, (int)nchunks);
  }
  return launch_status();
}

int sn_spmm_q3_f32(const int32_t *b_rowptr, const float *q_blk, int64_t Mb, int64_t Kb, int64_t nblocks, const float *X,
                   int64_t ldx, int32_t x_group, int32_t N, float *Y, int64_t ldy, int32_t y_group, void *stream) {
  return spmm_q3_launch(b_rowptr, q_blk, Mb, Kb, nblocks, X, ldx, x_group, N, Y, ldy, y_group, SpmmEpi{nullptr, 0, nullptr, 0},
                        stream);
}

size_t sn_spmm_q3_stats_workspace_bytes(int64_t Mb) {
  if (Mb < 1) return 0;
  const int64_t nchunks = (Mb + 31) / 32;                 // N = 32: 32 block rows per workgroup
  return (size_t)chunk_grid(nchunks) * 256 * sizeof(float);
}

int sn_spmm_q3_stats_f32(const int32_t *b_rowptr, const float *q_blk, int64_t Mb, int64_t Kb, int64_t nblocks, const float *X,
                         int64_t ldx, int32_t x_group, int32_t N, float *Y, int64_t ldy, int32_t y_group, double *stats_part,
                         void *workspace, size_t workspace_bytes, void *stream) {
  if (!stats_part || !workspace) return SN_E_NULL;
  if (workspace_bytes < sn_spmm_q3_stats_workspace_bytes(Mb)) return SN_E_WORKSPACE;
  return spmm_q3_launch(b_rowptr, q_blk, Mb, Kb, nblocks, X, ldx, x_group, N, Y, ldy, y_group, SpmmEpi{nullptr, 0, nullptr, 0},
                        stream, static_cast<float *>(workspace), stats_part);
}

int32_t sn_spmm_q3_stats_blocks(void) { return kSpmmStatsBlocks; }

int sn_spmm_q3_elubwd_f32(const int32_t *b_rowptr, const float *q_blk, int64_t Mb, int64_t Kb, int64_t nblocks,
                          const float *X, int64_t ldx, int32_t x_group, int32_t N, const float *E, int64_t lde,
                          const float *G, int64_t ldg, float *Y, int64_t ldy, int32_t y_group, void *stream) {
  if (!E) return SN_E_NULL;
  return spmm_q3_launch(b_rowptr, q_blk, Mb, Kb, nblocks, X, ldx, x_group, N, Y, ldy, y_group, SpmmEpi{E, lde, G, ldg}, stream);
}

int64_t sn_spmm_q3_absmax_blocks(int64_t Mb, int32_t N) {
  if (Mb < 1 || !(N == 16 || N == 32 || N == 64 || N == 128)) return 0;
  const int rpb = kWG / (N / 4);
  return (int64_t)chunk_grid((Mb + rpb - 1) / rpb);
}

int sn_spmm_q3_elubwd_absmax_f32(const int32_t *b_rowptr, const float *q_blk, int64_t Mb, int64_t Kb, int64_t nblocks,
                                 const float *X, int64_t ldx, int32_t x_group, int32_t N, const float *E, int64_t lde,
                                 const float *G, int64_t ldg, float *Y, int64_t ldy, int32_t y_group, float *y_absmax,
                                 void *stream) {
  if (!E) return SN_E_NULL;
  return spmm_q3_launch(b_rowptr, q_blk, Mb, Kb, nblocks, X, ldx, x_group, N, Y, ldy, y_group, SpmmEpi{E, lde, G, ldg, y_absmax},
                        stream);
}

int sn_bsr4_to_q3_f32(const int32_t *b_colind, const float *b_vals, int64_t nblocks, float *q_blk, int32_t *not_quaternion,
                      void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nblocks < 0) return SN_E_SHAPE;
  if (!not_quaternion) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = sn_internal_fill(not_quaternion, 0, sizeof(int32_t), s);
  if (e != hipSuccess) return (int)e;
  if (nblocks == 0) return SN_OK;
  if (!b_colind || !b_vals || !q_blk) return SN_E_NULL;
  if (!aligned16(b_vals) || !aligned16(q_blk)) return SN_E_ALIGN;
  hipLaunchKernelGGL(bsr4_to_q3_k, dim3(grid_for(nblocks, kWG)), dim3(kWG), 0, s, b_colind, b_vals, nblocks,
                     reinterpret_cast<f4 *>(q_blk), not_quaternion);
  return launch_status();
}

int sn_coo_to_csr_i32(const int64_t *idx_batch, const int64_t *idx_row, const int64_t *idx_col,
                      int64_t nnz, int64_t B, int64_t R, int64_t Kb, int32_t *rowptr, int32_t *colind,
                      void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nnz < 0 || B < 1 || R < 0 || Kb < 0) return SN_E_SHAPE;
  const int64_t M = B * R;
  if (!fits_i32(M + 1) || !fits_i32(B * Kb) || !fits_i32(nnz)) return SN_E_RANGE;
  if (!rowptr || (nnz > 0 && (!idx_row || !idx_col || !colind))) return SN_E_NULL;
  const int64_t n = (M + 1 > nnz) ? M + 1 : nnz;
  hipLaunchKernelGGL(coo_to_csr_k, dim3(grid_for(n, kWG)), dim3(kWG), 0, static_cast<hipStream_t>(stream),
                     idx_batch, idx_row, idx_col, nnz, M, R, Kb, rowptr, colind);
  return launch_status();
}

size_t sn_scan_workspace_bytes(int64_t n) {
  if (n < 0) n = 0;
  return (size_t)((n + kScanTile - 1) / kScanTile + 1) * sizeof(int);
}

size_t sn_csr_transpose_workspace_bytes(int64_t M, int64_t K, int64_t nnz) {
  (void)M;
  (void)nnz;
  if (K < 0) K = 0;
  // cursor[K] (16-byte padded) + scan scratch for K+1 counters
  const size_t cursor = ((size_t)K * sizeof(int) + 15) & ~(size_t)15;
  return cursor + sn_scan_workspace_bytes(K + 1);
}

int sn_csr_transpose_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M,
                         int64_t K, int64_t nnz, int32_t *t_rowptr, int32_t *t_colind, float *t_vals,
                         void *workspace, size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (M < 0 || K < 0 || nnz < 0) return SN_E_SHAPE;
  if (!fits_i32(M + 1) || !fits_i32(K + 1) || !fits_i32(nnz)) return SN_E_RANGE;
  if (!t_rowptr || !rowptr) return SN_E_NULL;
  if (nnz > 0 && (!colind || !vals || !t_colind || !t_vals)) return SN_E_NULL;
  if (workspace_bytes < sn_csr_transpose_workspace_bytes(M, K, nnz) || (!workspace && K > 0))
    return SN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = sn_internal_fill(t_rowptr, 0, (size_t)(K + 1) * sizeof(int), s);
  if (e != hipSuccess) return (int)e;
  if (nnz == 0 || K == 0) return SN_OK;
  int *cursor = static_cast<int *>(workspace);
  const size_t cursor_bytes = ((size_t)K * sizeof(int) + 15) & ~(size_t)15;
  void *scan_ws = static_cast<char *>(workspace) + cursor_bytes;
  hipLaunchKernelGGL(histogram_cols, dim3(grid_for(nnz, kWG)), dim3(kWG), 0, s, colind, nnz, t_rowptr);
  int st = exclusive_scan_i32(t_rowptr, K + 1, t_rowptr, scan_ws, workspace_bytes - cursor_bytes, s);
  if (st) return st;
  e = sn_internal_copy2d(cursor, 0, t_rowptr, 0, (int64_t)((size_t)K * sizeof(int)), 1, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(transpose_scatter, dim3(grid_for(M, kWG)), dim3(kWG), 0, s, rowptr, colind, vals, M,
                     cursor, t_colind, t_vals);
  hipLaunchKernelGGL(sort_rows_by_col, dim3(grid_for(K, kWG)), dim3(kWG), 0, s, t_rowptr, K, t_colind,
                     t_vals);
  return launch_status();
}

int sn_bsr4_count(const int32_t *rowptr, const int32_t *colind, int64_t M, int64_t K, int32_t *b_rowptr,
                  void *workspace, size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (M < 0 || K < 0) return SN_E_SHAPE;
  if ((M & 3) || (K & 3)) return SN_E_UNSUPPORTED;
  if (!fits_i32(M + 1) || !fits_i32(K)) return SN_E_RANGE;
  if (!rowptr || !b_rowptr) return SN_E_NULL;
  const int64_t Mb = M / 4;
  if (workspace_bytes < sn_scan_workspace_bytes(Mb + 1) || !workspace) return SN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = sn_internal_fill(b_rowptr + Mb, 0, sizeof(int), s);
  if (e != hipSuccess) return (int)e;
  if (Mb > 0)
    hipLaunchKernelGGL((bsr4_merge<false>), dim3(grid_for(Mb, kWG)), dim3(kWG), 0, s, rowptr, colind,
                       (const float *)nullptr, Mb, b_rowptr, (int *)nullptr, (float *)nullptr);
  return exclusive_scan_i32(b_rowptr, Mb + 1, b_rowptr, workspace, workspace_bytes, s);
}

int sn_bsr4_fill(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K,
                 const int32_t *b_rowptr, int32_t *b_colind, float *b_vals, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (M < 0 || K < 0) return SN_E_SHAPE;
  if ((M & 3) || (K & 3)) return SN_E_UNSUPPORTED;
  if (!rowptr || !b_rowptr) return SN_E_NULL;
  const int64_t Mb = M / 4;
  if (Mb == 0) return SN_OK;
  if (!colind || !vals || !b_colind || !b_vals) return SN_E_NULL;
  if (!aligned16(b_vals)) return SN_E_ALIGN;
  hipLaunchKernelGGL((bsr4_merge<true>), dim3(grid_for(Mb, kWG)), dim3(kWG), 0,
                     static_cast<hipStream_t>(stream), rowptr, colind, vals, Mb,
                     const_cast<int *>(b_rowptr), b_colind, b_vals);
  return launch_status();
}

int sn_blockdiag_concat_i32(const int32_t *pool_rowptr, const int32_t *pool_colind,
                            const float *pool_vals, const int64_t *desc, int64_t B, int64_t size0,
                            int64_t size1, int64_t total, int32_t vals_per_entry, int32_t *out_rowptr,
                            int32_t *out_colind, float *out_vals, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (B < 0 || size0 < 0 || size1 < 0 || total < 0) return SN_E_SHAPE;
  if (vals_per_entry != 1 && vals_per_entry != 16 && vals_per_entry != 4) return SN_E_UNSUPPORTED;
  if (!fits_i32(B * size0 + 1) || !fits_i32(B * size1) || !fits_i32(total)) return SN_E_RANGE;
  if (!out_rowptr) return SN_E_NULL;
  if (B > 0 && (!desc || !pool_rowptr)) return SN_E_NULL;
  if (total > 0 && (!pool_vals || !out_vals || (vals_per_entry != 4 && (!pool_colind || !out_colind)))) return SN_E_NULL;
  if (vals_per_entry != 1 && total > 0 && (!aligned16(pool_vals) || !aligned16(out_vals))) return SN_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(blockdiag_rowptr, dim3(grid_for(B * size0 + 1, kWG)), dim3(kWG), 0, s, pool_rowptr,
                     desc, B, size0, total, out_rowptr);
  if (total > 0) {
    if (vals_per_entry == 1)
      hipLaunchKernelGGL((blockdiag_entries<1>), dim3(grid_for(total, kWG)), dim3(kWG), 0, s, pool_colind,
                         pool_vals, desc, B, size1, total, out_colind, out_vals);
    else if (vals_per_entry == 4)
      hipLaunchKernelGGL((blockdiag_entries<4>), dim3(grid_for(total, kWG)), dim3(kWG), 0, s, pool_colind,
                         pool_vals, desc, B, size1, total, out_colind, out_vals);
    else
      hipLaunchKernelGGL((blockdiag_entries<16>), dim3(grid_for(total, kWG)), dim3(kWG), 0, s, pool_colind,
                         pool_vals, desc, B, size1, total, out_colind, out_vals);
  }
  return launch_status();
}

int sn_blockdiag_concat_ragged_i32(const int32_t *pool_rowptr, const int32_t *pool_colind, const float *pool_vals,
                                   const int64_t *desc, int64_t B, int64_t total_rows, int64_t total_cols, int64_t total,
                                   int32_t vals_per_entry, int32_t *out_rowptr, int32_t *out_colind, float *out_vals,
                                   void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (B < 0 || total_rows < 0 || total_cols < 0 || total < 0) return SN_E_SHAPE;
  if (vals_per_entry != 1 && vals_per_entry != 16 && vals_per_entry != 4) return SN_E_UNSUPPORTED;
  if (!fits_i32(total_rows + 1) || !fits_i32(total_cols) || !fits_i32(total)) return SN_E_RANGE;
  if (!out_rowptr) return SN_E_NULL;
  if (B > 0 && (!desc || !pool_rowptr)) return SN_E_NULL;
  if (B == 0 && (total_rows > 0 || total > 0)) return SN_E_SHAPE;
  if (total > 0 && (!pool_vals || !out_vals || (vals_per_entry != 4 && (!pool_colind || !out_colind)))) return SN_E_NULL;
  if (vals_per_entry != 1 && total > 0 && (!aligned16(pool_vals) || !aligned16(out_vals))) return SN_E_ALIGN;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0) {
    hipError_t e = sn_internal_fill(out_rowptr, 0, sizeof(int32_t), s);
    return e == hipSuccess ? SN_OK : (int)e;
  }
  hipLaunchKernelGGL(blockdiag_rowptr_ragged, dim3(grid_for(total_rows + 1, kWG)), dim3(kWG), 0, s, pool_rowptr, desc, B,
                     total_rows, total, out_rowptr);
  if (total > 0) {
    const unsigned grid = grid_for(total, kWG);
    if (vals_per_entry == 1)
      hipLaunchKernelGGL((blockdiag_entries_ragged<1>), dim3(grid), dim3(kWG), 0, s, pool_colind, pool_vals, desc, B, total, out_colind, out_vals);
    else if (vals_per_entry == 4)
      hipLaunchKernelGGL((blockdiag_entries_ragged<4>), dim3(grid), dim3(kWG), 0, s, pool_colind, pool_vals, desc, B, total, out_colind, out_vals);
    else
      hipLaunchKernelGGL((blockdiag_entries_ragged<16>), dim3(grid), dim3(kWG), 0, s, pool_colind, pool_vals, desc, B, total, out_colind, out_vals);
  }
  return launch_status();
}

int sn_validate_csr_i32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                        int32_t *flags, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (M < 0 || K < 0 || nnz < 0) return SN_E_SHAPE;
  if (!fits_i32(M + 1) || !fits_i32(K) || !fits_i32(nnz)) return SN_E_RANGE;
  if (!flags) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = sn_internal_fill(flags, 0, sizeof(int32_t), s);
  if (e != hipSuccess) return (int)e;
  if (M == 0) return SN_OK;
  if (!rowptr || (nnz > 0 && !colind)) return SN_E_NULL;
  hipLaunchKernelGGL(validate_csr_k, dim3(grid_for(M, kWG)), dim3(kWG), 0, s, rowptr, colind, vals, M, K, nnz, flags);
  return launch_status();
}

size_t sn_segment_colsum_ragged_workspace_bytes(int64_t ntiles, int32_t C) {
  if (ntiles < 0 || C < 1) return 0;
  return (size_t)ntiles * (size_t)C * sizeof(double);
}

int sn_segment_colsum_ragged_f32(const float *x, int64_t ld, const int64_t *tiles, int64_t ntiles, const int64_t *seg_tile_ptr,
                                 int64_t nseg, int32_t C, const float *scale, float *out, void *workspace,
                                 size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (ntiles < 0 || nseg < 0 || C < 1 || ld < C) return SN_E_SHAPE;
  if (!fits_i32(ntiles) || !fits_i32(nseg * (int64_t)C)) return SN_E_RANGE;
  if (nseg == 0) return SN_OK;
  if (!out || !seg_tile_ptr || (ntiles > 0 && (!x || !tiles))) return SN_E_NULL;
  if (workspace_bytes < sn_segment_colsum_ragged_workspace_bytes(ntiles, C) || (ntiles > 0 && !workspace)) return SN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  double *partial = static_cast<double *>(workspace);
  if (ntiles > 0)
    hipLaunchKernelGGL(seg_colsum_ragged_tiles_k, dim3((unsigned)ntiles), dim3(kWG), 0, s, x, ld, tiles, (int)C, partial);
  hipLaunchKernelGGL(seg_colsum_ragged_final_k, dim3(grid_for(nseg * (int64_t)C, kWG)), dim3(kWG), 0, s, partial, seg_tile_ptr, nseg,
                     (int)C, scale, out);
  return launch_status();
}

int sn_bcast_rows_ragged_f32(const float *src, const int64_t *tiles, int64_t ntiles, float *dst, int64_t ldd, int32_t C,
                             void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (ntiles < 0 || C < 4 || (C & 3) || ldd < C || (ldd & 3)) return SN_E_SHAPE;
  if (!fits_i32(ntiles)) return SN_E_RANGE;
  if (ntiles == 0) return SN_OK;
  if (!src || !tiles || !dst) return SN_E_NULL;
  if (!aligned16(src) || !aligned16(dst)) return SN_E_ALIGN;
  hipLaunchKernelGGL(bcast_rows_ragged_k, dim3((unsigned)ntiles), dim3(kWG), 0, static_cast<hipStream_t>(stream), src, tiles, dst,
                     ldd, (int)C);
  return launch_status();
}

int sn_elu_into_f32(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t rows, int32_t C,
                    void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || C < 1 || lds < C || ldd < C) return SN_E_SHAPE;
  if (rows == 0) return SN_OK;
  if (!src || !dst) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (C % 4 == 0) && (lds % 4 == 0) && (ldd % 4 == 0) && aligned16(src) && aligned16(dst);
  if (vec)
    hipLaunchKernelGGL((elu_into_k<true>), dim3(grid_for(rows * (C / 4), kWG)), dim3(kWG), 0, s, src, lds,
                       dst, ldd, rows, (int)C, kStreamNT);
  else
    hipLaunchKernelGGL((elu_into_k<false>), dim3(grid_for(rows * (int64_t)C, kWG)), dim3(kWG), 0, s, src,
                       lds, dst, ldd, rows, (int)C, kStreamNT);
  return launch_status();
}

int sn_elu_bwd_acc_f32(const float *gdst, int64_t ldg, const float *gdst2, int64_t ldg2, const float *gadd, int64_t ldga,
                       const float *out, int64_t ldo, float *gsrc, int64_t ldgs, int64_t rows, int32_t C,
                       int32_t accumulate, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || C < 1 || ldg < C || ldo < C || ldgs < C || (gdst2 && ldg2 < C) || (gadd && ldga < C)) return SN_E_SHAPE;
  if (rows == 0) return SN_OK;
  if (!gdst || !out || !gsrc) return SN_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (C % 4 == 0) && (ldg % 4 == 0) && (ldo % 4 == 0) && (ldgs % 4 == 0) &&
                   aligned16(gdst) && aligned16(out) && aligned16(gsrc) && (!gdst2 || (aligned16(gdst2) && ldg2 % 4 == 0)) &&
                   (!gadd || (aligned16(gadd) && ldga % 4 == 0));
  const unsigned grid = grid_for(vec ? rows * (C / 4) : rows * (int64_t)C, kWG);
  if (vec) {
    if (accumulate) hipLaunchKernelGGL((elu_bwd_k<true, true>), dim3(grid), dim3(kWG), 0, s, gdst, ldg, gdst2, ldg2, gadd, ldga, out, ldo, gsrc, ldgs, rows, (int)C, kStreamNT);
    else hipLaunchKernelGGL((elu_bwd_k<true, false>), dim3(grid), dim3(kWG), 0, s, gdst, ldg, gdst2, ldg2, gadd, ldga, out, ldo, gsrc, ldgs, rows, (int)C, kStreamNT);
  } else {
    if (accumulate) hipLaunchKernelGGL((elu_bwd_k<false, true>), dim3(grid), dim3(kWG), 0, s, gdst, ldg, gdst2, ldg2, gadd, ldga, out, ldo, gsrc, ldgs, rows, (int)C, kStreamNT);
    else hipLaunchKernelGGL((elu_bwd_k<false, false>), dim3(grid), dim3(kWG), 0, s, gdst, ldg, gdst2, ldg2, gadd, ldga, out, ldo, gsrc, ldgs, rows, (int)C, kStreamNT);
  }
  return launch_status();
}

int sn_timing_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_timing_mu);
  g_timing_on = on != 0;
  g_timing_linear = on != 2;
  return SN_OK;
}

int64_t sn_timing_count(void) {
  std::lock_guard<std::mutex> lk(g_timing_mu);
  return (int64_t)g_timing.size();
}

int sn_timing_drain(double *ms, int64_t *meta, int64_t capacity, int64_t *written) {
  if (capacity < 0 || !written || (capacity > 0 && (!ms || !meta))) return SN_E_NULL;
  std::vector<TimedLaunch> recs;
  {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    recs.swap(g_timing);
  }
  int64_t n = 0;
  int status = SN_OK;
  for (const TimedLaunch &t : recs) {
    float el = 0.f;
    hipError_t e = hipEventSynchronize(t.stop);
    if (e == hipSuccess) e = hipEventElapsedTime(&el, t.start, t.stop);
    if (e != hipSuccess && status == SN_OK) status = (int)e;
    if (e == hipSuccess && n < capacity) {
      ms[n] = el;
      for (int i = 0; i < 5; ++i) meta[5 * n + i] = t.meta[i];
      ++n;
    }
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  *written = n;
  return status;
}

}  // extern "C"
