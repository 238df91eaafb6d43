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
#include <hip/hip_ext.h>
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

inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SN_OK : (int)e;
}
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// the split kernels keep a lane's byte offset inside a 32-row tile in 32 bits: leading dimensions below 2^24 elements
inline bool ld32(int64_t a, int64_t b = 0, int64_t c = 0, int64_t d = 0, int64_t e = 0) {
  const int64_t lim = (int64_t)1 << 24;
  return a < lim && b < lim && c < lim && d < lim && e < lim;
}
// ELU in the GEMM epilogue runs while the matrix pipe of this SIMD idles (one wave per SIMD), so it uses the hardware
// exponential (v_exp_f32) instead of the ~30-instruction expm1f: |error| <= 1.2e-7 absolute, far below the 1e-5 parity bar.
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_DGRAD_ELU = 2 };

struct EpiArgs {
  // forward  : v0 bias[Nout], v1 residual (rows x Nout, ld1) | NULL, o2 y_elu (rows x Nout, ld2) | NULL
  // dgrad    : v0 x (rows x Nout, ld1), v1 / v2 / v3 center / B / Cc [Nout] (B == NULL: plain product)
  // dgrad+elu: as dgrad (B required); columns c < half leave as  o2[r][c] = dx·elu'(x[r][c]) + v4[r][c]  (o2: ld2, v4: ld3,
  //            v4 may be NULL) instead of Out[r][c]; columns >= half go to Out as usual
  const float *v0, *v1, *v2, *v3;
  float *o2;
  int64_t ld1, ld2;
  const float *v4;
  int64_t ld3;
  int half;
  // per-mesh row vectors (period = rows per mesh > 0): forward: segv[mesh(r)][c] REPLACES the bias; dgrad+elu: rowmask[r] *
  // segv[mesh(r)][c] is added before the activation derivative (rowmask may be NULL = 1)
  const float *segv;
  int64_t period, ldseg;
  const float *rowmask;
  // forward with an ELU copy: per-workgroup column sums / sums of squares of the ACTIVATED output, [gridDim.x][2][128] fp64
  // (NULL: not wanted) — the BatchNorm statistics of the first half of the next stage's concat buffer, so that the
  // statistics pass of that stage reads only the propagated half
  double *stats;
  int stats_blocks;      // blocks the consumer of `stats` reads (sn_linear_fwd_stats_blocks): those past the grid are zeroed
  // narrow layers (J < 128, a multiple of 4 — the models' last layer has 120 outputs): forward: output columns that exist
  // (the weights, bias, residual of the others read as 0, nothing is stored to them); input gradient: columns of dy = rows of
  // W that exist (the others read as 0).  128: full width.
  int jv;
  // ragged meshes: segoff[0 .. nseg] = first row of every mesh (and the row count), each mesh at least 32 rows; takes the
  // place of `period` (then 0) for the per-mesh vectors.  NULL: equal meshes of `period` rows.
  const int64_t *segoff;
  int nseg;
  // input gradient through the activation: absmax[blockIdx.x] <- max |gact| over the rows this workgroup wrote (entries from
  // the grid's size up to kAbsmaxBlocks are zeroed) — the bound the two-piece weight gradient of the layer BELOW needs for its
  // dy operand (sn_wgrad_bounded_f32).  NULL: not wanted.
  float *absmax;
  // tile assignment: 0 — workgroup b owns a contiguous range of tiles; 1 — tiles b, b + grid, b + 2 grid, ... (all workgroups
  // walk ONE contiguous window of every operand together instead of 256 separate streams each: same-box A/B of the config-3
  // step on three boxes, round 4: 19.99 -> 19.46, 20.38 -> 19.80, 19.84 -> 19.77 ms).
  int interleave;
  // forward with an ELU copy (one-tile kernels): tile_sums[tile][128] <- column sums of the ACTIVATED output over the tile's
  // valid rows (fp32; every tile written exactly once).  What the half-width global-average stage needs of its operand beyond
  // the BatchNorm sums: per-mesh column sums follow from the tiles inside a mesh (sn_avg_stats_from_tiles_f32), so the
  // statistics pass over the operand (segstats_k: 165 MB per stage at the ARAP batch) disappears.  NULL: not wanted.
  float *tile_sums;
};
constexpr int kAbsmaxBlocks = 512;      // >= the largest grid of any input-gradient launch (2 workgroups per CU)

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

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// The Linear layers' products on the 16-bit matrix pipe (16x the fp32 MFMA rate per instruction) from an EXACT two-piece split.
//
// fp16 carries 11 significant bits, so TWO round-to-nearest pieces hold an fp32 value to 2^-23: x = h + l, h = rn16(x),
// l = rn16(x - h) (the remainder is an exact fp32 subtraction), and x·w needs three partial products
//     xh·wh + (xh·wl + xl·wh)                  (the dropped xl·wl is <= 2^-24 |x·w|)
// each exact in the fp32 accumulator (11 x 11 bits) — the result is as accurate as the fp32 MFMA / an fmaf chain
// (tests/test_dense_gpu.py bounds both against fp64) while the kernel turns from MFMA-bound into HBM-bound.  (Round 1-3 used
// three bf16 pieces, six partial products: LABNOTES k20.)  What fp16 lacks is RANGE
// (5 exponent bits: activations behind a cotangent Laplacian reach 1e5, gradients sit at 1e-6), so both operands are
// scaled by exact powers of two that factor out of the contraction over k: every data ROW by 2^(14 - E_row) from its own
// absolute maximum (found by the loader wave with four DPP steps, a few VALU operations per 1 KiB load), every weight
// COLUMN by 2^(14 - E_col), and the low pieces by a further 2^11 so that they sit in the normal range whenever the high
// piece does (their two products go to separate accumulators, folded in with 2^-11 at the end).  The inverse scales are
// applied to the fp32 result in the epilogue.  Elements more than 2^28 below their row's maximum lose low-order bits — an
// error below 2^-37 of the row's scale.  Non-finite inputs give NaN, as in the bf16 form.
// v_mfma_f32_32x32x16_f16 has the operand layout and rate of the bf16 instruction.
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16v mfma_f16(const u4 &a, const u4 &b, const f16v &c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
constexpr float kLowUp = 2048.f, kLowDown = 1.f / 2048.f;      // 2^11: the low pieces' own scale

// scale_up = 2^(14 - E), scale_down = 2^(E - 14) for a row / column whose absolute maximum is m = f·2^E, f in [0.5, 1)
__device__ __forceinline__ void pow2_scales(float m, float &up, float &down) {
  int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 126;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);              // zero / denormal / non-finite rows: any finite scale will do
  up = __uint_as_float((unsigned)(127 + 14 - e) << 23);
  down = __uint_as_float((unsigned)(127 - 14 + e) << 23);
}
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
// x·up -> fp16 pieces (round to nearest): H = rn16(x·up), L = rn16((x·up - H)·2^11); x·up and the remainder are exact in fp32
__device__ __forceinline__ void split4_h2(const f4 &x, float up, u2 &H, u2 &L) {
  const f4 xs = x * up;
  const h2v h01 = __builtin_convertvector(f2v{xs.x, xs.y}, h2v), h23 = __builtin_convertvector(f2v{xs.z, xs.w}, h2v);
  const f2v l01 = f2v{xs.x - (float)h01.x, xs.y - (float)h01.y} * kLowUp;
  const f2v l23 = f2v{xs.z - (float)h23.x, xs.w - (float)h23.y} * kLowUp;
  H = u2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
  L = u2{__builtin_bit_cast(unsigned, __builtin_convertvector(l01, h2v)),
         __builtin_bit_cast(unsigned, __builtin_convertvector(l23, h2v))};
}
__device__ __forceinline__ void split8_h2(const f4 &p, const f4 &q, float up, u4 &H, u4 &L) {
  u2 h0, l0, h1, l1;
  split4_h2(p, up, h0, l0);
  split4_h2(q, up, h1, l1);
  H = u4{h0.x, h0.y, h1.x, h1.y};
  L = u4{l0.x, l0.y, l1.x, l1.y};
}
__device__ __forceinline__ float absmax4(const f4 &v) {
  return fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
}
// maximum of the bit patterns v (non-negative floats order like unsigned integers) over the 16 lanes of a DPP row, in every
// lane: xor 1, xor 2 (quad permutes), half-row and row mirrors — four v_max_u32_dpp
template <int CTRL>
__device__ __forceinline__ unsigned dpp_umax(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
  return v > o ? v : o;
}
__device__ __forceinline__ unsigned row16_umax(unsigned v) {
  v = dpp_umax<0xB1>(v);        // quad_perm [1,0,3,2]
  v = dpp_umax<0x4E>(v);        // quad_perm [2,3,0,1]
  v = dpp_umax<0x141>(v);       // row_half_mirror
  v = dpp_umax<0x140>(v);       // row_mirror
  return v;
}

// Operand path.  A lane's MFMA fragment is 32 bytes of ITS row, so fragment-shaped global loads touch 32 cache lines per
// wave instruction and the four waves of a workgroup would repeat both the loads and the split — the texture addresser and
// the vector ALU, not HBM or the matrix pipe, then bound the kernel (measured: the fp32-MFMA kernel above and a
// register-fed split-bf16 kernel run at the same speed).  Instead each wave loads 8 of the tile's 32 rows with full-line
// loads (one wave instruction = 1 KiB contiguous), splits them ONCE and writes the three bf16 images of the tile to LDS
// (row stride 2K+16 bytes: conflict-free ds_read_b128); all four waves then read their fragments of every image.  The
// conversion of tile t+1 (and the global loads of tile t+2 into the registers it frees) is spread over the k-steps of tile
// t, so the vector ALU works in the shadow of the matrix pipe; one workgroup barrier per tile.
template <int I>
struct IC {
  static constexpr int value = I;
};
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {          // f(IC<I>{}) for I in [I, N): the index is a constant expression
  if constexpr (I < N) {
    f(IC<I>{});
    static_for<I + 1, N>(f);
  }
}

// LDS writes of this wave done, then the workgroup barrier — without the vmcnt(0) a __syncthreads() fence would add
// (it would drain the global prefetch and wait for the previous tile's stores to be acknowledged).
#define SN_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- global accesses of the streaming kernel: raw buffer resources ----
// With ONE wave per SIMD (the weights own the register file) every instruction of the wave — scalar, s_nop, address
// arithmetic — costs an issue slot (about four cycles), and the matrix pipe runs in the shadow of that stream: the kernel is
// bound by its instruction count.  Plain 64-bit pointers spent a quarter of it on per-access row clamps and address
// products (r2 listing: 914 instructions per tile and wave, 229 of them scalar, 3.0 us per tile against 1.9 us of HBM time).
// Every streamed matrix is therefore addressed through a raw buffer resource whose base is the first row of the CURRENT
// tile and whose extent ends with the matrix: a lane's byte offset inside a tile is a kernel constant (32 bits), rows past
// the end read as 0 and are not written (no clamps, no predicates), and moving to the next tile is six scalar operations.
// Cache hints (A/B runs on config 3, r1): outputs NON-TEMPORAL — the 0.3-0.6 GB a launch writes would otherwise sit dirty
// in L2 / Infinity Cache and be written back under the NEXT kernel's reads (the transposed SpMM that consumes dx: 83 %
// instead of 79 % of the HBM roofline, 0.4 ms per step); operands plain loads (non-temporal measured neutral to worse).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
struct RowWindow {
  const char *p;       // first row of the current tile
  int step;            // bytes per 32-row tile (leading dimensions stay below 2^24 elements)
  int n_full, n_last;  // bytes from the first row of a tile to the end of its last addressed row: full tile | the matrix's last
  // a row-major fp32 matrix of `rows` rows, leading dimension ld >= its width; `span` = columns addressed from `base` (a row
  // past the end then starts at or beyond the extent for every addressed column); the window starts at row row0
  __device__ __forceinline__ void init(const float *base, int64_t ld, int64_t row0, int last_rows, int span) {
    step = 128 * (int)ld;
    p = reinterpret_cast<const char *>(base + row0 * ld);
    n_full = 4 * (31 * (int)ld + span);
    n_last = 4 * ((last_rows - 1) * (int)ld + span);
  }
  __device__ __forceinline__ void init_empty() {             // a window nothing is read from or written to
    step = 0;
    p = nullptr;
    n_full = n_last = 0;
  }
  __device__ __forceinline__ void next() { p += step; }
  __device__ __forceinline__ void stride(int g) { step *= g; }      // advance g tiles per next()
  // to_last = tiles between the window's tile and the matrix's last tile (0: it is the last, < 0: past the matrix)
  __device__ __forceinline__ rsrc_t rsrc(int to_last) const {
    const int n = to_last > 0 ? n_full : (to_last == 0 ? n_last : 0);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p), 0, n, 0x00020000);      // raw buffer: stride 0, bytes
  }
};
__device__ __forceinline__ f4 bld4(rsrc_t r, int voff) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void bst4(rsrc_t r, int voff, const f4 &v) {          // non-temporal
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, voff, 0, 2);
}

template <int PC, int K, int NT, bool TRANSW, int EPI, bool SIDE, bool ELU, bool SMALL = false, int WV = 4>
__global__ __launch_bounds__(64 * WV, (PC == 2 && K == 128 && NT == 1 && WV == 4) ? 2 : 1) void gemm_rows_split_k(const float *__restrict__ In, int64_t ldi,
                                                         const float *__restrict__ W, int64_t ldw,
                                                         float *__restrict__ Out, int64_t ldo, int64_t rows, EpiArgs ep) {
  constexpr int KS = K / 16;                 // MFMA k-steps per output tile
  constexpr int RS = 2 * K + 16;             // bytes per row of one 16-bit image
  constexpr int PART = 32 * RS;              // bytes per image
  constexpr int SEG = K / 64;                // 256-byte segments per data row (2 | 4)
  constexpr int NCH = 8 / WV;                // 4-row chunks a wave stages per tile: the workgroup's WV waves (4 | 8) share the 32 rows
  constexpr int NL = NCH * SEG;              // load instructions per wave per tile: chunk p (4 of the wave's rows) x segment
  constexpr int CSTEP = KS / 2;              // one conversion chunk every CSTEP k-steps (WV = 8: the one chunk at k-step 0)
  constexpr int NOUT = 32 * WV * NT;         // output columns: a wave owns 32·NT of them
  constexpr int CPR = 8 * NT;                // 16-byte chunks per row of a wave's output slab (32·NT columns)
  constexpr int SROW = 16 * CPR + 16;        // bytes per staged output row (+16: conflict-free transposition)
  constexpr int RPI = 64 / CPR;              // output rows per store instruction (8 | 4)
  constexpr int NST = 32 / RPI;              // store instructions per slab (4 | 8)
  static_assert(PC == 2, "two fp16 pieces");
  __shared__ __attribute__((aligned(16))) unsigned char img[2][PC][PART];
  __shared__ __attribute__((aligned(16))) unsigned char stg[WV][32 * SROW];
  __shared__ float s_rs[2][32];                         // inverse row scales of the tile held by each image
  __shared__ float s_amax[WV];                          // dgrad+elu: per-wave max |gact| (EpiArgs::absmax)
  __shared__ __attribute__((aligned(16))) float s_cs[NOUT];          // inverse column scales of the weights
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = lane & 31, h = lane >> 5;
  // ---- stationary weights, split once: w?[t][ks] = pieces of Wmat[col = 32(wave·NT + t) + n][k = 16ks + 8h .. +7] ----
  u4 wh[NT][KS], wl[NT][KS];
  constexpr int kOob = 0x7fffff00;          // a byte offset past every buffer extent: loads return 0, stores are dropped
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load_w = [&](int col, int ks, f4 &p, f4 &q) {
    if constexpr (!TRANSW) {                  // forward: W is (J x K), row = output column
      p = q = zero4;
      if (col < ep.jv) {
        p = *reinterpret_cast<const f4 *>(W + (int64_t)col * ldw + 16 * ks + 8 * h);
        q = *reinterpret_cast<const f4 *>(W + (int64_t)col * ldw + 16 * ks + 8 * h + 4);
      }
    } else {                                  // input gradient: W is (J x C), row = k (jv is a multiple of 4)
      const int k0w = 16 * ks + 8 * h;
      const float *w0 = W + (int64_t)k0w * ldw + col;
      p = q = zero4;
      if (k0w < ep.jv) p = f4{w0[0], w0[ldw], w0[2 * ldw], w0[3 * ldw]};
      if (k0w + 4 < ep.jv) q = f4{w0[4 * ldw], w0[5 * ldw], w0[6 * ldw], w0[7 * ldw]};
    }
  };
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = 32 * (wave * NT + t) + n;
    float cup = 1.f;
    // SMALL (operands of at most kSmallRows rows: a FAUST tower, a Mesh-MNIST batch): ONE pass over the weights, a lane's
    // 16·KS values waiting in registers for the column maximum.  The prologue is 5-9 us of such a launch's 7-12 us
    // (fragment-shaped loads touch 32 lines per instruction, so the second pass costs as much as the first): 11.3 -> 8.7 us
    // for a K = 256 forward launch of 32 rows.  Large operands keep the two passes: the held values cost the tile loop of
    // the register-bound kernels 5-10 % (input gradient through the activation, forward with residual; measured), and
    // there the prologue is 2 % of the launch.  The two K = 128 forward kernels without a side operand run two
    // workgroups per CU (256 registers) and would spill: two passes always.
    constexpr bool ONEPASS = SMALL && !(PC == 2 && K == 128 && NT == 1 && !TRANSW && !SIDE);
    f4 wp[ONEPASS ? KS : 1], wq[ONEPASS ? KS : 1];
    if constexpr (ONEPASS) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) load_w(col, ks, wp[ks], wq[ks]);
    }
    {                                        // column scale from the column's absolute maximum (my half of k, then my partner's)
      float cm = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f4 p, q;
        if constexpr (ONEPASS) p = wp[ks], q = wq[ks];
        else load_w(col, ks, p, q);
        cm = fmaxf(cm, fmaxf(absmax4(p), absmax4(q)));
      }
      cm = fmaxf(cm, __shfl_xor(cm, 32));
      float cdown;
      pow2_scales(cm, cup, cdown);
      if (h == 0) s_cs[col] = cdown;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f4 p, q;
      if constexpr (ONEPASS) p = wp[ks], q = wq[ks];
      else load_w(col, ks, p, q);
      split8_h2(p, q, cup, wh[t][ks], wl[t][ks]);
    }
  }
  __syncthreads();                          // s_cs complete (read once below, by other lanes)
  const int64_t ntiles = (rows + 31) / 32;
  const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int G = ep.interleave ? (int)gridDim.x : 1;                 // tiles between two of mine
  int64_t tile = G > 1 ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * per;
  const int64_t tend = G > 1 ? ntiles : (tile + per < ntiles ? tile + per : ntiles);
  constexpr bool STATS = (EPI == EPI_FWD) && ELU && NT == 1;
  if (tile >= tend) {
    if constexpr (STATS)
      if (ep.stats)                                                                  // a workgroup without tiles adds nothing
        for (int64_t b = blockIdx.x; b < ep.stats_blocks; b += gridDim.x) ep.stats[b * 256 + threadIdx.x] = 0.0;
    if constexpr (EPI == EPI_DGRAD_ELU)
      if (ep.absmax && threadIdx.x == 0)
        for (int b = blockIdx.x; b < kAbsmaxBlocks; b += gridDim.x) ep.absmax[b] = 0.f;
    return;
  }
  float amax = 0.f;                                     // dgrad+elu: max |gact| over my share of the stored rows
  double ssum[STATS ? 4 : 1], ssq[STATS ? 4 : 1];
#pragma unroll
  for (int i = 0; i < (STATS ? 4 : 1); ++i) ssum[i] = ssq[i] = 0.0;
  // ---- epilogue geometry: the slab goes through LDS so that global accesses are full lines — lane l handles the 16 bytes
  // at chunk (l % CPR) of rows (l / CPR) + RPI·j; its columns, hence its epilogue constants, are fixed for the kernel ----
  const int erow = lane / CPR;
  const int ecol = 32 * NT * wave + 4 * (lane % CPR);
  f4 k0 = {0.f, 0.f, 0.f, 0.f}, k1 = k0, k2 = k0;
  constexpr bool DGE = (EPI == EPI_DGRAD_ELU);
  static_assert(!DGE || SIDE, "the ELU variant needs the BatchNorm tail operand");
  const bool lowhalf = DGE && 32 * NT * wave < ep.half;          // a wave's 32·NT columns lie on one side (scalar condition)
  const bool colv = EPI != EPI_FWD || ecol < ep.jv;          // my 4 output columns exist
  if constexpr (EPI == EPI_FWD) {
    if (colv) k0 = *reinterpret_cast<const f4 *>(ep.v0 + ecol);                      // bias
  } else if constexpr (SIDE) {
    if (ep.v1) k0 = *reinterpret_cast<const f4 *>(ep.v1 + ecol);                     // center (optional)
    k1 = *reinterpret_cast<const f4 *>(ep.v2 + ecol);                                // B
    k2 = *reinterpret_cast<const f4 *>(ep.v3 + ecol);                                // Cc
  }
  const f4 kcs = *reinterpret_cast<const f4 *>(s_cs + ecol);                           // inverse scales of my 4 columns
  unsigned char *const sw = &stg[wave][0] + n * SROW + 16 * h;                       // where my accumulators go (+128t + 32g)
  const unsigned char *const sr = &stg[wave][0] + erow * SROW + 16 * (lane % CPR);   // what I read back (+ RPI·j·SROW)
  // my byte offset inside a tile of each epilogue matrix (row erow, + RPI rows per store instruction), and the windows
  const float *side_p = (EPI == EPI_FWD) ? ep.v1 : ep.v0;   // SIDE: forward: the residual; dgrad: x of the BatchNorm tail
  const bool has_out = (EPI != EPI_FWD && !DGE) || Out != nullptr;
  const bool has_ga = DGE && lowhalf && ep.v4 != nullptr;
  const int vo_side = colv ? 4 * (erow * (int)ep.ld1 + ecol) : kOob, js_side = 4 * RPI * (int)ep.ld1;
  const int vo_out = colv ? 4 * (erow * (int)ldo + ecol) : kOob, js_out = 4 * RPI * (int)ldo;
  const int vo_o2 = colv ? 4 * (erow * (int)ep.ld2 + ecol) : kOob, js_o2 = 4 * RPI * (int)ep.ld2;
  const int vo_ga = 4 * (erow * (int)ep.ld3 + ecol), js_ga = 4 * RPI * (int)ep.ld3;
  RowWindow w_in, w_side, w_out, w_o2, w_ga;
  const int64_t row0 = tile * 32;
  // valid rows of a tile: 32 except in the matrix's last tile
  const int last_nrt = (int)(rows - (ntiles - 1) * 32);
  int to_last = (int)(ntiles - 1 - tile);                    // tiles from the current one to the matrix's last
  const int ispan = EPI == EPI_FWD ? K : ep.jv;              // columns the operands really have (a window's span must not
  const int ospan = EPI == EPI_FWD ? ep.jv : NOUT;           // exceed its leading dimension, or the row past the end is in range)
  w_in.init(In, ldi, row0, last_nrt, ispan);
  if constexpr (SIDE) w_side.init(side_p, ep.ld1, row0, last_nrt, ospan);
  if (has_out) w_out.init(Out, ldo, row0, last_nrt, ospan);
  else w_out.init_empty();                                   // stores through it are dropped
  if constexpr (DGE || (EPI == EPI_FWD && ELU)) w_o2.init(ep.o2, ep.ld2, row0, last_nrt, DGE ? ep.half : ospan);
  if constexpr (DGE) {
    if (has_ga) w_ga.init(ep.v4, ep.ld3, row0, last_nrt, ep.half);
    else w_ga.init_empty();
  }
  if (G > 1) {
    w_in.stride(G);
    if constexpr (SIDE) w_side.stride(G);
    w_out.stride(G);
    if constexpr (DGE || (EPI == EPI_FWD && ELU)) w_o2.stride(G);
    if constexpr (DGE) w_ga.stride(G);
  }
  // per-mesh vectors: the mesh of the tile's first row and the row where the next mesh starts, kept up as tiles advance
  const bool useseg = (ep.period > 0 || ep.segoff != nullptr) && (EPI == EPI_FWD || (DGE && lowhalf));      // wave-uniform
  int64_t seg_m = 0;
  int seg_left = 0;                                          // rows from the tile's first row to the next mesh (< 2^31)
  if (useseg) {
    if (ep.segoff) {                                         // the mesh that holds row0: last m with segoff[m] <= row0
      int lo = 0, hi = ep.nseg - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ep.segoff[mid] <= row0) lo = mid;
        else hi = mid - 1;
      }
      seg_m = lo;
      seg_left = (int)(ep.segoff[lo + 1] - row0);
    } else {
      seg_m = row0 / ep.period;
      seg_left = (int)((seg_m + 1) * ep.period - row0);
    }
  }

  // ---- loader: this wave stages rows 8·wave .. +7 of a tile, four rows per load instruction (16 lanes x 16 bytes = one
  // 256-byte segment of a row): chunk p = rows 4p .. 4p+3, all SEG segments — a row's absolute maximum is then one
  // 16-lane (single DPP row) reduction for four rows at once ----
  const int lr = lane >> 4, lc = lane & 15;
  const int lrow = 4 * NCH * wave + lr;                        // + 4p
  int lvo[SEG];                                                // per segment: + 16·ldi·p; columns past jv (dy of a narrow layer): out
#pragma unroll
  for (int s = 0; s < SEG; ++s)
    lvo[s] = (EPI == EPI_FWD || 64 * s + 4 * lc < ep.jv) ? 4 * (lrow * (int)ldi + 4 * lc) + 256 * s : kOob;
  const int lps = 16 * (int)ldi;
  f4 raw[NL];
  auto load_chunk = [&](rsrc_t r, int p) {
#pragma unroll
    for (int s = 0; s < SEG; ++s) raw[p * SEG + s] = bld4(r, lvo[s] + p * lps);
  };
  auto convert_chunk = [&](int buf, int p) {
    unsigned char *d = &img[buf][0][0] + (lrow + 4 * p) * RS + 8 * lc;          // segment s at + 128·s
    float m = absmax4(raw[p * SEG]);
#pragma unroll
    for (int s = 1; s < SEG; ++s) m = fmaxf(m, absmax4(raw[p * SEG + s]));
    float up, down;
    pow2_scales(__uint_as_float(row16_umax(__float_as_uint(m))), up, down);
#pragma unroll
    for (int s = 0; s < SEG; ++s) {
      u2 H, L;
      split4_h2(raw[p * SEG + s], up, H, L);
      *reinterpret_cast<u2 *>(d + 128 * s) = H;
      *reinterpret_cast<u2 *>(d + 128 * s + PART) = L;
    }
    if (lc == 0) s_rs[buf][lrow + 4 * p] = down;
  };
  // prologue: tile 0 converted into image 0, tile 1 in flight in the registers; the input window then runs two tiles ahead
  {
    const rsrc_t r0 = w_in.rsrc(to_last);
#pragma unroll
    for (int p = 0; p < NCH; ++p) load_chunk(r0, p);
    w_in.next();
    const rsrc_t r1 = w_in.rsrc(to_last - G);
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      convert_chunk(0, p);
      load_chunk(r1, p);
    }
    w_in.next();
  }

  auto do_tile = [&](auto bufc, int64_t tl) {
    constexpr int buf = decltype(bufc)::value;
    SN_LDS_BARRIER();                  // image `buf` complete and visible; every wave is done reading image buf^1
    const rsrc_t r_in = w_in.rsrc(to_last - 2 * G);      // my tile after next (past the matrix: reads 0, no traffic)
    const int nrt = to_last == 0 ? last_nrt : 32;        // valid rows of this tile
    // side operand in the epilogue layout (full lines), requested now, consumed after the k loop
    f4 sd[NST];
    if constexpr (SIDE) {
      const rsrc_t r_side = w_side.rsrc(to_last);
#pragma unroll
      for (int j = 0; j < NST; ++j) sd[j] = bld4(r_side, vo_side + j * js_side);
    }
    // per-mesh vector of my rows (forward: the bias; dgrad+elu: added before elu'): REQUESTED here, combined in the epilogue —
    // anything computed from these loads before the k loop would wait for them, i.e. drain the row prefetch, once per tile.
    // A 32-row tile meets at most two meshes (period >= 32): the mesh of its first row, and the next one from row `nb` on.
    f4 sg0 = {0.f, 0.f, 0.f, 0.f}, sg1 = sg0;
    float smk[NST];
    int nb = 32;
    const bool usemask = DGE && useseg && ep.rowmask != nullptr;
    if (useseg) {
      while (seg_left <= 0) {          // (one step with contiguous tiles; a stride of G tiles may pass several meshes)
        ++seg_m;
        seg_left += ep.segoff ? (int)(ep.segoff[seg_m + 1] - ep.segoff[seg_m]) : (int)ep.period;
      }
      nb = seg_left < nrt ? seg_left : nrt;                            // rows >= nb (if any) belong to the next mesh
      const float *s0 = ep.segv + seg_m * ep.ldseg + (colv ? ecol : 0);
      const float *s1 = nb < nrt ? s0 + ep.ldseg : s0;                  // (no next mesh in this tile: the same vector again)
      sg0 = *reinterpret_cast<const f4 *>(s0);
      sg1 = *reinterpret_cast<const f4 *>(s1);
      if constexpr (DGE) {
        if (usemask) {
          const float *mrow = ep.rowmask + tl * 32;
#pragma unroll
          for (int j = 0; j < NST; ++j) {
            const int rt = erow + RPI * j;
            smk[j] = mrow[rt < nrt ? rt : nrt - 1];
          }
        }
      }
    }
    f4 ga[NST];                        // dgrad+elu, low half: the gradient added after the activation derivative
    if constexpr (DGE) {
      if (lowhalf) {                   // (without the operand: an empty window — the loads return zeros, no traffic)
        const rsrc_t r_ga = w_ga.rsrc(to_last);
#pragma unroll
        for (int j = 0; j < NST; ++j) ga[j] = bld4(r_ga, vo_ga + j * js_ga);
      }
    }
    // accumulators per output tile: leading products | the two cross products; the first k-step starts them from a zero operand
    f16v acc0[NT], acc1[NT];
    const f16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned char *fp = &img[buf][0][0] + n * RS + 16 * h;
    constexpr int LOWP = PART;                  // offset of the low piece's image
    const float rs = s_rs[buf][n];              // inverse scale of MY data row (accumulator layout: lane (n, h) holds row n)
    u4 dh = *reinterpret_cast<const u4 *>(fp), dl = *reinterpret_cast<const u4 *>(fp + LOWP);
    static_for<0, KS>([&](auto ic) {
      constexpr int ks = decltype(ic)::value;
      __builtin_amdgcn_sched_barrier(0);
      u4 nh, nl;                                  // fragments of the next k-step: read while this one is multiplied
      if constexpr (ks + 1 < KS) {
        nh = *reinterpret_cast<const u4 *>(fp + 32 * (ks + 1));
        nl = *reinterpret_cast<const u4 *>(fp + LOWP + 32 * (ks + 1));
      }
      if constexpr (ks % CSTEP == 0 && ks / CSTEP < NCH) {      // conversion of the next tile, one chunk at a time, under the MFMAs
        constexpr int p = ks / CSTEP;
        convert_chunk(buf ^ 1, p);
        load_chunk(r_in, p);
      }
      // three exact products: leading | the two cross terms (one accumulator)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc1[t] = mfma_f16(wl[t][ks], dh, ks == 0 ? zero : acc1[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc0[t] = mfma_f16(wh[t][ks], dh, ks == 0 ? zero : acc0[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc1[t] = mfma_f16(wh[t][ks], dl, acc1[t]);
      if constexpr (ks + 1 < KS) {
        dh = nh; dl = nl;
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue.  Accumulator layout: lane (n, h) holds row n, columns 32t + 8g + 4h .. +3 of the wave's slab.  Through
    // this wave's LDS staging area (same-wave LDS operations complete in order: no barrier) into the full-line layout ----
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        // leading products + 2^-11 x cross products, then the row's inverse scale (all exact factors)
        *reinterpret_cast<f4 *>(sw + 128 * t + 32 * g) =
            f4{__builtin_fmaf(acc1[t][4 * g], kLowDown, acc0[t][4 * g]) * rs,
               __builtin_fmaf(acc1[t][4 * g + 1], kLowDown, acc0[t][4 * g + 1]) * rs,
               __builtin_fmaf(acc1[t][4 * g + 2], kLowDown, acc0[t][4 * g + 2]) * rs,
               __builtin_fmaf(acc1[t][4 * g + 3], kLowDown, acc0[t][4 * g + 3]) * rs};
    const rsrc_t r_out = w_out.rsrc(to_last);
    rsrc_t r_o2 = r_out;                          // (placeholder unless the kernel has the second output)
    if constexpr (DGE || (EPI == EPI_FWD && ELU)) r_o2 = w_o2.rsrc(to_last);
    // output rows RPI·j + erow, my 4 columns: the product (inverse column scales applied) + the epilogue's linear part
    auto out_row = [&](int j) {
      f4 v = *reinterpret_cast<const f4 *>(sr + RPI * j * SROW);
      v *= kcs;
      if constexpr (EPI == EPI_FWD) {
        if (useseg) v += (erow + RPI * j < nb ? sg0 : sg1);          // (scalar condition: a branch, not a select)
        else v += k0;
        if constexpr (SIDE) v += sd[j];
      } else if constexpr (SIDE) {            // dgrad (both forms): BatchNorm tail
        const f4 xv = sd[j] - k0;
        v.x += __builtin_fmaf(xv.x, k1.x, k2.x);
        v.y += __builtin_fmaf(xv.y, k1.y, k2.y);
        v.z += __builtin_fmaf(xv.z, k1.z, k2.z);
        v.w += __builtin_fmaf(xv.w, k1.w, k2.w);
      }
      return v;
    };
    if constexpr (DGE) {
      if (lowhalf) {                     // through the activation: elu'(.) from the activation OUTPUT held in the side operand
#pragma unroll
        for (int j = 0; j < NST; ++j) {
          f4 v = out_row(j);
          const f4 o = sd[j];
          if (useseg) {
            f4 sv = erow + RPI * j < nb ? sg0 : sg1;
            if (usemask) sv *= smk[j];
            v += sv;
          }
          // elu'(x) from elu(x) = o:  1 for o > 0, o + 1 otherwise  =  min(o, 0) + 1;  v·elu' as one fma
          v = f4{__builtin_fmaf(v.x, fminf(o.x, 0.f), v.x), __builtin_fmaf(v.y, fminf(o.y, 0.f), v.y),
                 __builtin_fmaf(v.z, fminf(o.z, 0.f), v.z), __builtin_fmaf(v.w, fminf(o.w, 0.f), v.w)};
          v += ga[j];                    // (no such operand: the empty window read zeros)
          bst4(r_o2, vo_o2 + j * js_o2, v);
          // (rows past the matrix's end are computed — from zeros — but not stored: they must not enter the bound)
          const float m4 = fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
          amax = (erow + RPI * j < nrt) ? fmaxf(amax, m4) : amax;
        }
      } else {
#pragma unroll
        for (int j = 0; j < NST; ++j) bst4(r_out, vo_out + j * js_out, out_row(j));
      }
    } else {
      f4 tsum = {0.f, 0.f, 0.f, 0.f};                // STATS: this tile's column sums of elu(y) over my rows (EpiArgs::tile_sums)
#pragma unroll
      for (int j = 0; j < NST; ++j) {
        const f4 v = out_row(j);
        bst4(r_out, vo_out + j * js_out, v);
        if constexpr (EPI == EPI_FWD && ELU) {
          const f4 ev = f4{elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
          bst4(r_o2, vo_o2 + j * js_o2, ev);
          if constexpr (STATS) {
            if (erow + RPI * j < nrt) {              // rows past the end hold elu(bias): not part of the statistics
              const double e0 = ev.x, e1 = ev.y, e2 = ev.z, e3 = ev.w;
              ssum[0] += e0; ssum[1] += e1; ssum[2] += e2; ssum[3] += e3;
              ssq[0] = __builtin_fma(e0, e0, ssq[0]); ssq[1] = __builtin_fma(e1, e1, ssq[1]);
              ssq[2] = __builtin_fma(e2, e2, ssq[2]); ssq[3] = __builtin_fma(e3, e3, ssq[3]);
              tsum += ev;
            }
          }
        }
      }
      if constexpr (STATS) {
        if (ep.tile_sums) {                          // add the 8 row lanes of a chunk (lanes chunk + 8 erow), fixed butterfly
#pragma unroll
          for (int o = CPR; o < 64; o <<= 1) {
            tsum.x += __shfl_xor(tsum.x, o); tsum.y += __shfl_xor(tsum.y, o);
            tsum.z += __shfl_xor(tsum.z, o); tsum.w += __shfl_xor(tsum.w, o);
          }
          if (erow == 0) *reinterpret_cast<f4 *>(ep.tile_sums + tl * 128 + ecol) = tsum;
        }
      }
    }
    w_in.next();
    to_last -= G;
    seg_left -= 32 * G;
    if constexpr (SIDE) w_side.next();
    w_out.next();
    if constexpr (DGE || (EPI == EPI_FWD && ELU)) w_o2.next();
    if constexpr (DGE) w_ga.next();
  };
  while (true) {
    do_tile(IC<0>{}, tile);
    if ((tile += G) >= tend) break;
    do_tile(IC<1>{}, tile);
    if ((tile += G) >= tend) break;
  }
  if constexpr (EPI == EPI_DGRAD_ELU) {
    if (ep.absmax) {                       // workgroup maximum of |gact| (non-negative floats order like their bit patterns)
      unsigned mb = __float_as_uint(amax);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)mb, o);
        mb = other > mb ? other : mb;
      }
      if (lane == 0) s_amax[wave] = __uint_as_float(mb);
      __syncthreads();
      if (threadIdx.x == 0) {
        float m = s_amax[0];
#pragma unroll
        for (int w = 1; w < WV; ++w) m = fmaxf(m, s_amax[w]);
        ep.absmax[blockIdx.x] = m;
        for (int b = blockIdx.x + gridDim.x; b < kAbsmaxBlocks; b += gridDim.x) ep.absmax[b] = 0.f;
      }
    }
  }
  if constexpr (STATS) {
    if (ep.stats) {
      // lane (erow, chunk) holds the sums of its 4 columns over its rows of every tile: combine the 8 row lanes of a
      // chunk through this wave's staging area (same-wave LDS operations complete in order), in a fixed order
      static_assert(CPR == 8 && 32 * SROW >= 64 * 64, "statistics staging assumes the NT == 1 epilogue geometry");
      double *sl = reinterpret_cast<double *>(&stg[wave][0]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sl[lane * 8 + i] = ssum[i];
        sl[lane * 8 + 4 + i] = ssq[i];
      }
      const int chunk = lane >> 3, val = lane & 7;
      double tot = 0.0;
#pragma unroll
      for (int er = 0; er < 8; ++er) tot += sl[(er * 8 + chunk) * 8 + val];
      ep.stats[(int64_t)blockIdx.x * 256 + (val >> 2) * 128 + 32 * wave + 4 * chunk + (val & 3)] = tot;
      for (int64_t b = blockIdx.x + gridDim.x; b < ep.stats_blocks; b += gridDim.x) ep.stats[b * 256 + threadIdx.x] = 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// gemm_fwd_w8_k — the forward kernel as EIGHT waves of 16 output columns (two waves per SIMD).
//
// gemm_rows_split_k keeps a wave's 32 output columns of the split weights in registers: 128 VGPRs at K = 256, and with the
// accumulators, the rows in flight and the epilogue that is one wave per SIMD (272-325 registers) — every instruction of the
// tile loop then costs an issue slot nobody else can fill, and the K = 256 forward launches ran at 4.6-4.7 TB/s where the
// K = 128 ones (two workgroups per CU) reach 5.1-5.5.  Here a wave owns SIXTEEN columns (v_mfma_f32_16x16x32_f16: i = output
// column, n = data row, 32 k per instruction), 64 weight registers, and a workgroup of eight such waves covers the 128
// outputs: two waves per SIMD, whose scalar / vector / LDS / matrix instructions issue side by side.  The price is that every
// wave reads all of the tile's fragments (256 KB of LDS reads per tile instead of 128: 1 k cycles of a 4 k-cycle tile).
// No accumulator exchange, no second barrier: the tile loop is gemm_rows_split_k's (image t+1 converted and tile t+2
// requested under the k loop of tile t — one 4-row chunk per wave —, one barrier per tile, the slab through a per-wave LDS
// transposition so that a quad of lanes stores one row's 64 bytes).  Forward only, two fp16 pieces, J = 128, plain bias.
// ------------------------------------------------------------------------------------------------------------------
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mfma16_f16(const u4 &a, const u4 &b, const f4 &c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}

template <int K, bool SIDE, bool ELU>
__global__ __launch_bounds__(512, 1) void gemm_fwd_w8_k(const float *__restrict__ In, int64_t ldi, const float *__restrict__ W,
                                                        int64_t ldw, float *__restrict__ Out, int64_t ldo, int64_t rows,
                                                        EpiArgs ep) {
  constexpr int WV = 8;
  constexpr int KS = K / 32;                 // MFMA k-steps per tile
  constexpr int RS = 2 * K + 16;             // bytes per row of one fp16 image
  constexpr int PART = 32 * RS;              // bytes per image
  constexpr int SEG = K / 64;                // 256-byte segments per data row
  constexpr int SROW = 64 + 16;              // bytes per staged output row of a wave's 16-column slab
  constexpr int NST = 2;                     // store instructions per slab: 16 rows x 64 bytes each
  __shared__ __attribute__((aligned(16))) unsigned char img[2][2][PART];
  __shared__ __attribute__((aligned(16))) unsigned char stg[WV][32 * SROW];
  __shared__ float s_rs[2][32];
  __shared__ __attribute__((aligned(16))) float s_cs[128];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n16 = lane & 15, kg = lane >> 4;
  // ---- stationary weights: pieces of Wmat[col = 16 wave + n16][k = 32 ks + 8 kg .. +7], scaled by the column's power of two ----
  u4 wh[KS], wl[KS];
  {
    const float *wr = W + (int64_t)(16 * wave + n16) * ldw + 8 * kg;
    float cm = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f4 p = *reinterpret_cast<const f4 *>(wr + 32 * ks), q = *reinterpret_cast<const f4 *>(wr + 32 * ks + 4);
      cm = fmaxf(cm, fmaxf(absmax4(p), absmax4(q)));
    }
    cm = fmaxf(cm, __shfl_xor(cm, 16));
    cm = fmaxf(cm, __shfl_xor(cm, 32));
    float cup, cdown;
    pow2_scales(cm, cup, cdown);
    if (kg == 0) s_cs[16 * wave + n16] = cdown;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f4 p = *reinterpret_cast<const f4 *>(wr + 32 * ks), q = *reinterpret_cast<const f4 *>(wr + 32 * ks + 4);
      split8_h2(p, q, cup, wh[ks], wl[ks]);
    }
  }
  __syncthreads();
  const int64_t ntiles = (rows + 31) / 32;
  const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int G = ep.interleave ? (int)gridDim.x : 1;                 // tiles between two of mine
  int64_t tile = G > 1 ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * per;
  const int64_t tend = G > 1 ? ntiles : (tile + per < ntiles ? tile + per : ntiles);
  constexpr bool STATS = ELU;
  if (tile >= tend) {
    if constexpr (STATS)
      if (ep.stats && threadIdx.x < 256)
        for (int64_t b = blockIdx.x; b < ep.stats_blocks; b += gridDim.x) ep.stats[b * 256 + threadIdx.x] = 0.0;
    return;
  }
  double ssum[STATS ? 4 : 1], ssq[STATS ? 4 : 1];
#pragma unroll
  for (int i = 0; i < (STATS ? 4 : 1); ++i) ssum[i] = ssq[i] = 0.0;
  // ---- epilogue geometry: lane l stores the 16 bytes at chunk (l % 4) of rows (l / 4) + 16 j of its wave's slab ----
  const int erow = lane >> 2, echunk = lane & 3;
  const int ecol = 16 * wave + 4 * echunk;
  const f4 k0 = *reinterpret_cast<const f4 *>(ep.v0 + ecol);                         // bias
  const f4 kcs = *reinterpret_cast<const f4 *>(s_cs + ecol);                         // inverse column scales
  unsigned char *const sw = &stg[wave][0] + n16 * SROW + 16 * kg;                    // my accumulators: row n16 (+16), columns 4 kg ..
  const unsigned char *const sr = &stg[wave][0] + erow * SROW + 16 * echunk;         // what I read back (+ 16 SROW j)
  const bool has_out = Out != nullptr;
  const int vo_side = 4 * (erow * (int)ep.ld1 + ecol), js_side = 64 * (int)ep.ld1;
  const int vo_out = 4 * (erow * (int)ldo + ecol), js_out = 64 * (int)ldo;
  const int vo_o2 = 4 * (erow * (int)ep.ld2 + ecol), js_o2 = 64 * (int)ep.ld2;
  RowWindow w_in, w_side, w_out, w_o2;
  const int64_t row0 = tile * 32;
  const int last_nrt = (int)(rows - (ntiles - 1) * 32);
  int to_last = (int)(ntiles - 1 - tile);
  w_in.init(In, ldi, row0, last_nrt, K);
  if constexpr (SIDE) w_side.init(ep.v1, ep.ld1, row0, last_nrt, 128);
  if (has_out) w_out.init(Out, ldo, row0, last_nrt, 128);
  else w_out.init_empty();
  if constexpr (ELU) w_o2.init(ep.o2, ep.ld2, row0, last_nrt, 128);
  if (G > 1) {
    w_in.stride(G);
    if constexpr (SIDE) w_side.stride(G);
    w_out.stride(G);
    if constexpr (ELU) w_o2.stride(G);
  }

  // ---- loader: this wave stages rows 4 wave .. +3 of a tile (16 lanes x 16 bytes = one 256-byte segment of a row) ----
  const int lr = lane >> 4, lc = lane & 15;
  const int lrow = 4 * wave + lr;
  int lvo[SEG];
#pragma unroll
  for (int s = 0; s < SEG; ++s) lvo[s] = 4 * (lrow * (int)ldi + 4 * lc) + 256 * s;
  f4 raw[SEG];
  auto load_rows = [&](rsrc_t r) {
#pragma unroll
    for (int s = 0; s < SEG; ++s) raw[s] = bld4(r, lvo[s]);
  };
  auto convert_rows = [&](int buf) {
    unsigned char *d = &img[buf][0][0] + lrow * RS + 8 * lc;
    float m = absmax4(raw[0]);
#pragma unroll
    for (int s = 1; s < SEG; ++s) m = fmaxf(m, absmax4(raw[s]));
    float up, down;
    pow2_scales(__uint_as_float(row16_umax(__float_as_uint(m))), up, down);
#pragma unroll
    for (int s = 0; s < SEG; ++s) {
      u2 H, L;
      split4_h2(raw[s], up, H, L);
      *reinterpret_cast<u2 *>(d + 128 * s) = H;
      *reinterpret_cast<u2 *>(d + 128 * s + PART) = L;
    }
    if (lc == 0) s_rs[buf][lrow] = down;
  };
  {
    load_rows(w_in.rsrc(to_last));
    w_in.next();
    const rsrc_t r1 = w_in.rsrc(to_last - G);
    convert_rows(0);
    load_rows(r1);
    w_in.next();
  }

  auto do_tile = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    SN_LDS_BARRIER();
    const rsrc_t r_in = w_in.rsrc(to_last - 2 * G);
    const int nrt = to_last == 0 ? last_nrt : 32;
    f4 sd[NST];
    if constexpr (SIDE) {
      const rsrc_t r_side = w_side.rsrc(to_last);
#pragma unroll
      for (int j = 0; j < NST; ++j) sd[j] = bld4(r_side, vo_side + j * js_side);
    }
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    f4 acc0[2], acc1[2];
    const unsigned char *fp = &img[buf][0][0] + n16 * RS + 16 * kg;          // my fragments: rows n16 | n16 + 16, + 64 ks
    const float rs0 = s_rs[buf][n16], rs1 = s_rs[buf][n16 + 16];
    u4 dh0 = *reinterpret_cast<const u4 *>(fp), dl0 = *reinterpret_cast<const u4 *>(fp + PART),
       dh1 = *reinterpret_cast<const u4 *>(fp + 16 * RS), dl1 = *reinterpret_cast<const u4 *>(fp + 16 * RS + PART);
    static_for<0, KS>([&](auto ic) {
      constexpr int ks = decltype(ic)::value;
      __builtin_amdgcn_sched_barrier(0);
      u4 nh0, nl0, nh1, nl1;
      if constexpr (ks + 1 < KS) {
        nh0 = *reinterpret_cast<const u4 *>(fp + 64 * (ks + 1));
        nl0 = *reinterpret_cast<const u4 *>(fp + PART + 64 * (ks + 1));
        nh1 = *reinterpret_cast<const u4 *>(fp + 16 * RS + 64 * (ks + 1));
        nl1 = *reinterpret_cast<const u4 *>(fp + 16 * RS + PART + 64 * (ks + 1));
      }
      if constexpr (ks == 0) {                 // my four rows of the next tile, then the same rows of the tile after it
        convert_rows(buf ^ 1);
        load_rows(r_in);
      }
      acc1[0] = mfma16_f16(wl[ks], dh0, ks == 0 ? zero : acc1[0]);
      acc1[1] = mfma16_f16(wl[ks], dh1, ks == 0 ? zero : acc1[1]);
      acc0[0] = mfma16_f16(wh[ks], dh0, ks == 0 ? zero : acc0[0]);
      acc0[1] = mfma16_f16(wh[ks], dh1, ks == 0 ? zero : acc0[1]);
      acc1[0] = mfma16_f16(wh[ks], dl0, acc1[0]);
      acc1[1] = mfma16_f16(wh[ks], dl1, acc1[1]);
      if constexpr (ks + 1 < KS) {
        dh0 = nh0; dl0 = nl0; dh1 = nh1; dl1 = nl1;
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    // accumulator layout: lane (n16, kg) holds rows n16 / n16 + 16, columns 4 kg .. +3 of the wave's slab
    *reinterpret_cast<f4 *>(sw) = f4{__builtin_fmaf(acc1[0].x, kLowDown, acc0[0].x) * rs0, __builtin_fmaf(acc1[0].y, kLowDown, acc0[0].y) * rs0,
                                     __builtin_fmaf(acc1[0].z, kLowDown, acc0[0].z) * rs0, __builtin_fmaf(acc1[0].w, kLowDown, acc0[0].w) * rs0};
    *reinterpret_cast<f4 *>(sw + 16 * SROW) =
        f4{__builtin_fmaf(acc1[1].x, kLowDown, acc0[1].x) * rs1, __builtin_fmaf(acc1[1].y, kLowDown, acc0[1].y) * rs1,
           __builtin_fmaf(acc1[1].z, kLowDown, acc0[1].z) * rs1, __builtin_fmaf(acc1[1].w, kLowDown, acc0[1].w) * rs1};
    const rsrc_t r_out = w_out.rsrc(to_last);
    rsrc_t r_o2 = r_out;
    if constexpr (ELU) r_o2 = w_o2.rsrc(to_last);
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      f4 v = *reinterpret_cast<const f4 *>(sr + 16 * j * SROW);
      v = v * kcs + k0;
      if constexpr (SIDE) v += sd[j];
      bst4(r_out, vo_out + j * js_out, v);
      if constexpr (ELU) {
        const f4 ev = f4{elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w)};
        bst4(r_o2, vo_o2 + j * js_o2, ev);
        if (erow + 16 * j < nrt) {               // rows past the end hold elu(bias): not part of the statistics
          const double e0 = ev.x, e1 = ev.y, e2 = ev.z, e3 = ev.w;      // fp64 from the first addition on (as gemm_rows_split_k)
          ssum[0] += e0; ssum[1] += e1; ssum[2] += e2; ssum[3] += e3;
          ssq[0] = __builtin_fma(e0, e0, ssq[0]); ssq[1] = __builtin_fma(e1, e1, ssq[1]);
          ssq[2] = __builtin_fma(e2, e2, ssq[2]); ssq[3] = __builtin_fma(e3, e3, ssq[3]);
        }
      }
    }
    w_in.next();
    to_last -= G;
    if constexpr (SIDE) w_side.next();
    w_out.next();
    if constexpr (ELU) w_o2.next();
  };
  while (true) {
    do_tile(IC<0>{});
    if ((tile += G) >= tend) break;
    do_tile(IC<1>{});
    if ((tile += G) >= tend) break;
  }
  if constexpr (STATS) {
    if (ep.stats) {
      // lane (erow, chunk) holds the sums of its 4 columns over its rows of every tile: add the 16 row lanes of a chunk (lanes
      // 4 erow + chunk) in a fixed butterfly order
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {
          ssum[i] += __shfl_xor(ssum[i], o);
          ssq[i] += __shfl_xor(ssq[i], o);
        }
      if (erow == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ep.stats[(int64_t)blockIdx.x * 256 + ecol + i] = ssum[i];
          ep.stats[(int64_t)blockIdx.x * 256 + 128 + ecol + i] = ssq[i];
        }
      }
      if (threadIdx.x < 256)
        for (int64_t b = blockIdx.x + gridDim.x; b < ep.stats_blocks; b += gridDim.x) ep.stats[b * 256 + threadIdx.x] = 0.0;
    }
  }
}

// SN_GEMM_VARIANT: default (unset / anything but 0) the two-piece fp16 kernels; 0 = the fp32-MFMA kernel above, the ONE A/B
// baseline of the Linear kernels (exact fp32 products; no fused epilogues).  Read once per process.
inline int gemm_variant() {
  static const int v = [] {
    const char *e = getenv("SN_GEMM_VARIANT");
    return (e && atoi(e) == 0) ? 0 : 2;
  }();
  return v;
}

#define SN_UNPAREN(...) __VA_ARGS__
constexpr int64_t kSmallRows = 131072;      // operands up to this many rows take the one-pass weight prologue (gemm_rows_split_k<…, SMALL>)
// launch gemm_rows_split_k<2, TARGS...> (SMALL form for operands of at most kSmallRows rows); with the timing facility on
// (sn_timing_enable: t_start / t_stop of the enclosing entry point) the kernel's own start / stop go into two events
#define SN_SPLIT_LAUNCH(TARGS, ...)                                                                                    \
  do {                                                                                                                 \
    if (rows <= kSmallRows) {                                                                                          \
      if (t_start) hipExtLaunchKernelGGL((gemm_rows_split_k<2, SN_UNPAREN TARGS, true>), dim3(grid), dim3(kWG), 0, s, t_start, t_stop, 0, __VA_ARGS__); \
      else hipLaunchKernelGGL((gemm_rows_split_k<2, SN_UNPAREN TARGS, true>), dim3(grid), dim3(kWG), 0, s, __VA_ARGS__); \
    } else {                                                                                                           \
      if (t_start) hipExtLaunchKernelGGL((gemm_rows_split_k<2, SN_UNPAREN TARGS>), dim3(grid), dim3(kWG), 0, s, t_start, t_stop, 0, __VA_ARGS__); \
      else hipLaunchKernelGGL((gemm_rows_split_k<2, SN_UNPAREN TARGS>), dim3(grid), dim3(kWG), 0, s, __VA_ARGS__);     \
    }                                                                                                                  \
  } while (0)
#define SN_SPLIT_LAUNCH_NT2(K_, EPI_, SIDE_, ...) SN_SPLIT_LAUNCH((K_, 2, true, EPI_, SIDE_, false), __VA_ARGS__)

// Workgroups per CU.  One 4-wave workgroup per CU is a single wave per SIMD that owns the register file; the K = 128, one-tile
// kernels of the fp16 form need <= 256 registers and run TWO per CU (2 waves per SIMD): -14 % on the plain forward, -10 % on
// the input gradient through the activation, neutral with a residual (same box, r2); the others lose 3-7 % when their grid is
// doubled (the second round of workgroups re-loads and re-splits the weights).
inline int gemm_wgs(int K, int NT) { return (gemm_variant() == 2 && K == 128 && NT == 1) ? 2 : 1; }
inline unsigned gemm_grid(int64_t rows, int wgs) {
  const int64_t ntiles = (rows + 31) / 32;
  int64_t b = (int64_t)kCUs * wgs;
  if (b > ntiles) b = ntiles;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

// 0: the process runs the fp32-MFMA A/B baseline of the Linear kernels (SN_GEMM_VARIANT=0), 2: the shipped 16-bit matrix-pipe
// kernels — what the Python launchers ask instead of reading the environment themselves
int32_t sn_gemm_variant(void) { return gemm_variant(); }

// (the largest grid any forward kernel uses: a kernel with a smaller one zeroes the blocks past its own)
int32_t sn_linear_fwd_stats_blocks(int64_t rows) { return rows > 0 ? (int32_t)gemm_grid(rows, 2) : 0; }

int sn_linear_fwd_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *bias,
                      const float *residual, int64_t ldr, float *y, int64_t ldy, float *y_elu, int64_t lde,
                      int64_t rows, int32_t K, int32_t J, double *elu_stats_part, void *stream) {
  return sn_linear_fwd_tiles_f32(x, ldx, W, ldw, bias, residual, ldr, y, ldy, y_elu, lde, rows, K, J, elu_stats_part, nullptr, stream);
}

int sn_linear_fwd_tiles_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *bias,
                            const float *residual, int64_t ldr, float *y, int64_t ldy, float *y_elu, int64_t lde,
                            int64_t rows, int32_t K, int32_t J, double *elu_stats_part, float *tile_sums, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (tile_sums && (!elu_stats_part || J != 128)) return SN_E_UNSUPPORTED;      // (rides on the statistics epilogue, full width)
  if (rows < 0 || K < 1 || J < 1 || ldx < K || ldw < K || (y && ldy < J)) return SN_E_SHAPE;
  if (J > 128 || (J % 4) || (J != 128 && gemm_variant() == 0) || (K != 128 && K != 256) || !ld32(ldx, ldy, ldr, lde))
    return SN_E_UNSUPPORTED;
  if (rows == 0) return SN_OK;
  if (!x || !W || !bias || (!y && !(y_elu && gemm_variant() != 0))) return SN_E_NULL;     // y may be NULL when only elu(y) is wanted
  if (!aligned16(x) || !aligned16(W) || !aligned16(bias) || (y && (!aligned16(y) || (ldy % 4))) || (ldx % 4) || (ldw % 4) ||
      (residual && (!aligned16(residual) || (ldr % 4) || ldr < J)) || (y_elu && (!aligned16(y_elu) || (lde % 4) || lde < J)))
    return SN_E_ALIGN;
  if (elu_stats_part && (!y_elu || gemm_variant() == 0)) return SN_E_UNSUPPORTED;
  EpiArgs ep{bias, residual, nullptr, nullptr, y_elu, ldr, lde, nullptr, 0, 0, nullptr, 0, 0, nullptr, elu_stats_part,
             sn_linear_fwd_stats_blocks(rows), (int)J, nullptr, 0, nullptr, 1, tile_sums};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = gemm_grid(rows, gemm_wgs(K, 1));
  const bool x3 = gemm_variant() != 0;
  hipEvent_t t_start = nullptr, t_stop = nullptr;
  if (x3)
    sn_internal_timing_slot(0x100 | (residual ? 2 : 0) | (y_elu ? 1 : 0) | (y ? 4 : 0), rows, K,
                            rows * 4 * ((int64_t)K + (y ? J : 0) + (residual ? J : 0) + (y_elu ? J : 0)), J, &t_start, &t_stop);
#define SN_X3_FWD(KK, RES, EL) SN_SPLIT_LAUNCH((KK, 1, false, EPI_FWD, RES, EL), x, ldx, W, ldw, y, ldy, rows, ep)
#define SN_W8_FWD(KK, RES, EL)                                                                                         \
  do {                                                                                                                 \
    if (t_start) hipExtLaunchKernelGGL((gemm_fwd_w8_k<KK, RES, EL>), dim3(grid8), dim3(512), 0, s, t_start, t_stop, 0, x, ldx, W, ldw, y, ldy, rows, ep); \
    else hipLaunchKernelGGL((gemm_fwd_w8_k<KK, RES, EL>), dim3(grid8), dim3(512), 0, s, x, ldx, W, ldw, y, ldy, rows, ep); \
  } while (0)
  // eight waves of 16 columns (gemm_fwd_w8_k), large operands: the K = 256 launches that write only the activated copy (the one
  // shape it wins: one output stream — a wave's rows are 64-byte segments, and with a residual and two outputs the three
  // half-line streams cost more than the second wave per SIMD gains; LABNOTES r4w8)
  if (gemm_variant() == 2 && J == 128 && rows > kSmallRows && !tile_sums && K == 256 && !residual && y_elu && !y) {
    const unsigned grid8 = gemm_grid(rows, 1);
    SN_W8_FWD(256, false, true);
    return launch_status();
  }
  if (x3) {
    const int sel = (K == 256 ? 4 : 0) + (residual ? 2 : 0) + (y_elu ? 1 : 0);
    switch (sel) {
      case 0: SN_X3_FWD(128, false, false); break;
      case 1: SN_X3_FWD(128, false, true); break;
      case 2: SN_X3_FWD(128, true, false); break;
      case 3: SN_X3_FWD(128, true, true); break;
      case 4: SN_X3_FWD(256, false, false); break;
      case 5: SN_X3_FWD(256, false, true); break;
      case 6: SN_X3_FWD(256, true, false); break;
      default: SN_X3_FWD(256, true, true); break;
    }
  }
  else if (K == 256)
    hipLaunchKernelGGL((gemm_rows_k<256, 1, false, EPI_FWD>), dim3(grid), dim3(kWG), 0, s, x, ldx, W, ldw, y, ldy, rows, ep);
  else
    hipLaunchKernelGGL((gemm_rows_k<128, 1, false, EPI_FWD>), dim3(grid), dim3(kWG), 0, s, x, ldx, W, ldw, y, ldy, rows, ep);
  return launch_status();
}

int sn_linear_dgrad_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                        const float *center, const float *B, const float *Cc, float *dx, int64_t lddx, int64_t rows,
                        int32_t J, int32_t C, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || J < 1 || C < 1 || lddy < J || ldw < C || lddx < C) return SN_E_SHAPE;
  if (J > 128 || (J % 4) || (J != 128 && gemm_variant() == 0) || (C != 128 && C != 256) || !ld32(lddy, ldx, lddx))
    return SN_E_UNSUPPORTED;
  if (rows == 0) return SN_OK;
  if (!dy || !W || !dx) return SN_E_NULL;
  if (B && (!x || !Cc)) return SN_E_NULL;
  if (!aligned16(dy) || !aligned16(W) || !aligned16(dx) || (lddy % 4) || (ldw % 4) || (lddx % 4)) return SN_E_ALIGN;
  if (B && (!aligned16(x) || !aligned16(B) || !aligned16(Cc) || (center && !aligned16(center)) || (ldx % 4) || ldx < C))
    return SN_E_ALIGN;
  EpiArgs ep{x, center, B, Cc, nullptr, ldx, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, 0, (int)J, nullptr, 0, nullptr, 1, nullptr};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = gemm_grid(rows, gemm_wgs(128, C / 128));
  const bool x3 = gemm_variant() != 0;
  hipEvent_t t_start = nullptr, t_stop = nullptr;
  if (x3) sn_internal_timing_slot(0x200 | (B ? 1 : 0), rows, C, rows * 4 * ((int64_t)J + C + (B ? C : 0)), J, &t_start, &t_stop);
  if (C == 256 && x3 && B)
    SN_SPLIT_LAUNCH_NT2(128, EPI_DGRAD, true, dy, lddy, W, ldw, dx, lddx, rows, ep);
  else if (C == 256 && x3)
    SN_SPLIT_LAUNCH_NT2(128, EPI_DGRAD, false, dy, lddy, W, ldw, dx, lddx, rows, ep);
  else if (C == 256)
    hipLaunchKernelGGL((gemm_rows_k<128, 2, true, EPI_DGRAD>), dim3(grid), dim3(kWG), 0, s, dy, lddy, W, ldw, dx, lddx, rows, ep);
  else if (x3 && B)
    SN_SPLIT_LAUNCH((128, 1, true, EPI_DGRAD, true, false), dy, lddy, W, ldw, dx, lddx, rows, ep);
  else if (x3)
    SN_SPLIT_LAUNCH((128, 1, true, EPI_DGRAD, false, false), dy, lddy, W, ldw, dx, lddx, rows, ep);
  else
    hipLaunchKernelGGL((gemm_rows_k<128, 1, true, EPI_DGRAD>), dim3(grid), dim3(kWG), 0, s, dy, lddy, W, ldw, dx, lddx, rows, ep);
  return launch_status();
}

int32_t sn_linear_dgrad_absmax_blocks(void) { return kAbsmaxBlocks; }

int sn_linear_dgrad_elu_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                            const float *center, const float *B, const float *Cc, float *dx_hi, int64_t lddx, float *gact,
                            int64_t ldga, const float *gadd, int64_t ldgadd, int64_t rows, int32_t J, int32_t C,
                            void *stream) {
  return sn_linear_dgrad_elu_absmax_f32(dy, lddy, W, ldw, x, ldx, center, B, Cc, dx_hi, lddx, gact, ldga, gadd, ldgadd, rows, J, C,
                                        nullptr, stream);
}

int sn_linear_dgrad_elu_absmax_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                                   const float *center, const float *B, const float *Cc, float *dx_hi, int64_t lddx, float *gact,
                                   int64_t ldga, const float *gadd, int64_t ldgadd, int64_t rows, int32_t J, int32_t C,
                                   float *gact_absmax, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || J < 1 || C < 2 || lddy < J || ldw < C || lddx < C / 2 || ldga < C / 2 || ldx < C) return SN_E_SHAPE;
  if (J > 128 || (J % 4) || (C != 128 && C != 256) || gemm_variant() == 0 || !ld32(lddy, ldx, lddx, ldga, ldgadd))
    return SN_E_UNSUPPORTED;
  if (rows == 0) {                     // nothing to launch; the bound of an empty operand is 0
    if (gact_absmax && sn_internal_fill(gact_absmax, 0, kAbsmaxBlocks * sizeof(float), static_cast<hipStream_t>(stream)) != hipSuccess)
      return (int)hipGetLastError();
    return SN_OK;
  }
  if (!dy || !W || !dx_hi || !gact || !x || !B || !Cc) return SN_E_NULL;
  if (!aligned16(dy) || !aligned16(W) || !aligned16(dx_hi) || !aligned16(gact) || !aligned16(x) || !aligned16(B) ||
      !aligned16(Cc) || (center && !aligned16(center)) || (gadd && (!aligned16(gadd) || (ldgadd % 4) || ldgadd < C / 2)) ||
      (lddy % 4) || (ldw % 4) || (lddx % 4) || (ldga % 4) || (ldx % 4))
    return SN_E_ALIGN;
  const int half = C / 2;
  EpiArgs ep{x, center, B, Cc, gact, ldx, ldga, gadd, ldgadd, half, nullptr, 0, 0, nullptr, nullptr, 0, (int)J, nullptr, 0, gact_absmax, 1, nullptr};
  float *out = dx_hi - half;           // the kernel indexes absolute columns; only columns >= half are written through `out`
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = gemm_grid(rows, gemm_wgs(128, C / 128));
  hipEvent_t t_start = nullptr, t_stop = nullptr;
  sn_internal_timing_slot(0x200 | 2 | (gadd ? 4 : 0), rows, C, rows * 4 * ((int64_t)J + C + C + (gadd ? half : 0)), J, &t_start, &t_stop);
  if (C == 256)
    SN_SPLIT_LAUNCH_NT2(128, EPI_DGRAD_ELU, true, dy, lddy, W, ldw, out,
                       lddx, rows, ep);
  else
    SN_SPLIT_LAUNCH((128, 1, true, EPI_DGRAD_ELU, true, false), dy, lddy, W, ldw, out,
                       lddx, rows, ep);
  return launch_status();
}

// segoff != NULL: ragged meshes (rows_per_seg ignored); the caller guarantees segoff[0] = 0 < ... < segoff[nseg] = rows with
// at least 32 rows per mesh (device array: not checked here)
static int fwd_segbias_launch(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *segbias,
                              int64_t rows_per_seg, const int64_t *segoff, int32_t nseg, const float *residual, int64_t ldr,
                              float *y, int64_t ldy, float *y_elu, int64_t lde, int64_t rows, int32_t K, int32_t J,
                              double *elu_stats_part, void *stream, float *tile_sums = nullptr) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || K < 1 || J < 1 || ldx < K || ldw < K || (y && ldy < J) || (!segoff && rows_per_seg < 1) || (segoff && nseg < 1))
    return SN_E_SHAPE;
  if (J > 128 || (J % 4) || (K != 128 && K != 256) || gemm_variant() == 0 || !ld32(ldx, ldy, ldr, lde) ||
      (!segoff && (rows_per_seg < 32 || rows_per_seg > 0x7fffffffLL)))
    return SN_E_UNSUPPORTED;
  if (rows == 0) return SN_OK;
  if (!x || !W || !segbias || (!y && !y_elu)) return SN_E_NULL;
  if (!aligned16(x) || !aligned16(W) || !aligned16(segbias) || (y && (!aligned16(y) || (ldy % 4))) || (ldx % 4) || (ldw % 4) ||
      (residual && (!aligned16(residual) || (ldr % 4) || ldr < J)) || (y_elu && (!aligned16(y_elu) || (lde % 4) || lde < J)))
    return SN_E_ALIGN;
  if (elu_stats_part && !y_elu) return SN_E_UNSUPPORTED;
  if (tile_sums && (!elu_stats_part || J != 128)) return SN_E_UNSUPPORTED;
  EpiArgs ep{segbias, residual, nullptr, nullptr, y_elu, ldr, lde, nullptr, 0, 0, segbias, segoff ? 0 : rows_per_seg, J, nullptr,
             elu_stats_part, sn_linear_fwd_stats_blocks(rows), (int)J, segoff, nseg, nullptr, 1, tile_sums};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = gemm_grid(rows, gemm_wgs(K, 1));
  hipEvent_t t_start = nullptr, t_stop = nullptr;
  sn_internal_timing_slot(0x100 | 8 | (residual ? 2 : 0) | (y_elu ? 1 : 0) | (y ? 4 : 0), rows, K,
                          rows * 4 * ((int64_t)K + (y ? J : 0) + (residual ? J : 0) + (y_elu ? J : 0)), J, &t_start, &t_stop);
  const int sel = (K == 256 ? 4 : 0) + (residual ? 2 : 0) + (y_elu ? 1 : 0);
  switch (sel) {
    case 0: SN_X3_FWD(128, false, false); break;
    case 1: SN_X3_FWD(128, false, true); break;
    case 2: SN_X3_FWD(128, true, false); break;
    case 3: SN_X3_FWD(128, true, true); break;
    case 4: SN_X3_FWD(256, false, false); break;
    case 5: SN_X3_FWD(256, false, true); break;
    case 6: SN_X3_FWD(256, true, false); break;
    default: SN_X3_FWD(256, true, true); break;
  }
  return launch_status();
}

int sn_linear_fwd_segbias_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *segbias,
                              int64_t rows_per_seg, const float *residual, int64_t ldr, float *y, int64_t ldy, float *y_elu,
                              int64_t lde, int64_t rows, int32_t K, int32_t J, double *elu_stats_part, void *stream) {
  return fwd_segbias_launch(x, ldx, W, ldw, segbias, rows_per_seg, nullptr, 0, residual, ldr, y, ldy, y_elu, lde, rows, K, J,
                            elu_stats_part, stream);
}

int sn_linear_fwd_segbias_tiles_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *segbias,
                                    int64_t rows_per_seg, const float *residual, int64_t ldr, float *y, int64_t ldy, float *y_elu,
                                    int64_t lde, int64_t rows, int32_t K, int32_t J, double *elu_stats_part, float *tile_sums,
                                    void *stream) {
  return fwd_segbias_launch(x, ldx, W, ldw, segbias, rows_per_seg, nullptr, 0, residual, ldr, y, ldy, y_elu, lde, rows, K, J,
                            elu_stats_part, stream, tile_sums);
}

int sn_linear_fwd_segbias_ragged_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *segbias,
                                     const int64_t *segoff, int32_t nseg, const float *residual, int64_t ldr, float *y,
                                     int64_t ldy, float *y_elu, int64_t lde, int64_t rows, int32_t K, int32_t J,
                                     double *elu_stats_part, void *stream) {
  if (!segoff) return SN_E_NULL;
  return fwd_segbias_launch(x, ldx, W, ldw, segbias, 0, segoff, nseg, residual, ldr, y, ldy, y_elu, lde, rows, K, J,
                            elu_stats_part, stream);
}

int sn_linear_fwd_segbias_ragged_tiles_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *segbias,
                                           const int64_t *segoff, int32_t nseg, const float *residual, int64_t ldr, float *y,
                                           int64_t ldy, float *y_elu, int64_t lde, int64_t rows, int32_t K, int32_t J,
                                           double *elu_stats_part, float *tile_sums, void *stream) {
  if (!segoff) return SN_E_NULL;
  return fwd_segbias_launch(x, ldx, W, ldw, segbias, 0, segoff, nseg, residual, ldr, y, ldy, y_elu, lde, rows, K, J,
                            elu_stats_part, stream, tile_sums);
}

static int dgrad_eluseg_launch(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                               const float *center, const float *B, const float *Cc, const float *segvec,
                               int64_t rows_per_seg, const int64_t *segoff, int32_t nseg, const float *rowmask, float *gact,
                               int64_t ldga, const float *gadd, int64_t ldgadd, int64_t rows, int32_t J, int32_t C,
                               void *stream, float *gact_absmax = nullptr) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (rows < 0 || J < 1 || C < 1 || lddy < J || ldw < C || ldga < C || ldx < C || (segvec && !segoff && rows_per_seg < 1) ||
      (segoff && (nseg < 1 || !segvec)))
    return SN_E_SHAPE;
  if (J > 128 || (J % 4) || (C != 128 && C != 256) || gemm_variant() == 0 || !ld32(lddy, ldx, ldga, ldgadd) ||
      (segvec && !segoff && (rows_per_seg < 32 || rows_per_seg > 0x7fffffffLL)))
    return SN_E_UNSUPPORTED;
  if (rows == 0) {                     // nothing to launch; the bound of an empty operand is 0
    if (gact_absmax && sn_internal_fill(gact_absmax, 0, kAbsmaxBlocks * sizeof(float), static_cast<hipStream_t>(stream)) != hipSuccess)
      return (int)hipGetLastError();
    return SN_OK;
  }
  if (!dy || !W || !gact || !x || !B || !Cc || (rowmask && !segvec)) return SN_E_NULL;      // segvec may be NULL: no per-mesh vector
  if (!aligned16(dy) || !aligned16(W) || !aligned16(gact) || !aligned16(x) || !aligned16(B) || !aligned16(Cc) ||
      (segvec && !aligned16(segvec)) || (center && !aligned16(center)) ||
      (gadd && (!aligned16(gadd) || (ldgadd % 4) || ldgadd < C)) || (lddy % 4) || (ldw % 4) || (ldga % 4) || (ldx % 4))
    return SN_E_ALIGN;
  EpiArgs ep{x, center, B, Cc, gact, ldx, ldga, gadd, ldgadd, (int)C, segvec, (segvec && !segoff) ? rows_per_seg : 0, C, rowmask,
             nullptr, 0, (int)J, segoff, nseg, gact_absmax, 1, nullptr};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = gemm_grid(rows, gemm_wgs(128, C / 128));
  hipEvent_t t_start = nullptr, t_stop = nullptr;
  sn_internal_timing_slot(0x200 | 8 | (gadd ? 4 : 0), rows, C, rows * 4 * ((int64_t)J + C + C + (gadd ? C : 0)), J, &t_start, &t_stop);
  float *none = nullptr;               // every column leaves through gact: nothing is written through Out
  if (C == 256)
    SN_SPLIT_LAUNCH_NT2(128, EPI_DGRAD_ELU, true, dy, lddy, W, ldw, none,
                       (int64_t)0, rows, ep);
  else
    SN_SPLIT_LAUNCH((128, 1, true, EPI_DGRAD_ELU, true, false), dy, lddy, W, ldw, none,
                       (int64_t)0, rows, ep);
  return launch_status();
}

int sn_linear_dgrad_eluseg_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                               const float *center, const float *B, const float *Cc, const float *segvec,
                               int64_t rows_per_seg, const float *rowmask, float *gact, int64_t ldga, const float *gadd,
                               int64_t ldgadd, int64_t rows, int32_t J, int32_t C, void *stream) {
  return dgrad_eluseg_launch(dy, lddy, W, ldw, x, ldx, center, B, Cc, segvec, rows_per_seg, nullptr, 0, rowmask, gact, ldga, gadd,
                             ldgadd, rows, J, C, stream);
}

int sn_linear_dgrad_eluseg_absmax_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                                      const float *center, const float *B, const float *Cc, const float *segvec,
                                      int64_t rows_per_seg, const float *rowmask, float *gact, int64_t ldga, const float *gadd,
                                      int64_t ldgadd, int64_t rows, int32_t J, int32_t C, float *gact_absmax, void *stream) {
  return dgrad_eluseg_launch(dy, lddy, W, ldw, x, ldx, center, B, Cc, segvec, rows_per_seg, nullptr, 0, rowmask, gact, ldga, gadd,
                             ldgadd, rows, J, C, stream, gact_absmax);
}

int sn_linear_dgrad_eluseg_ragged_absmax_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                                             const float *center, const float *B, const float *Cc, const float *segvec,
                                             const int64_t *segoff, int32_t nseg, float *gact, int64_t ldga, const float *gadd,
                                             int64_t ldgadd, int64_t rows, int32_t J, int32_t C, float *gact_absmax,
                                             void *stream) {
  if (!segoff) return SN_E_NULL;
  return dgrad_eluseg_launch(dy, lddy, W, ldw, x, ldx, center, B, Cc, segvec, 0, segoff, nseg, nullptr, gact, ldga, gadd, ldgadd,
                             rows, J, C, stream, gact_absmax);
}

int sn_linear_dgrad_eluseg_ragged_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                                      const float *center, const float *B, const float *Cc, const float *segvec,
                                      const int64_t *segoff, int32_t nseg, float *gact, int64_t ldga, const float *gadd,
                                      int64_t ldgadd, int64_t rows, int32_t J, int32_t C, void *stream) {
  if (!segoff) return SN_E_NULL;
  return dgrad_eluseg_launch(dy, lddy, W, ldw, x, ldx, center, B, Cc, segvec, 0, segoff, nseg, nullptr, gact, ldga, gadd, ldgadd,
                             rows, J, C, stream);
}

}  // extern "C"
