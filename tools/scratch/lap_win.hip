// Prototype (scratch): Laplacian-type SpMM over row tiles with a CONTIGUOUS window of X staged in LDS by LDS-DMA.
//
// A workgroup owns R consecutive output rows and one 32-column slice (one 128-byte line per X row).  Its window
// [ws, ws + wlen) of X rows (the tile's column range, capped) arrives in LDS by global_load_lds (1 KiB per wave instruction,
// fully coalesced), together with the tile's CSR entries and row pointers; after ONE barrier the product runs out of LDS.
// Entries whose column lies outside the window (closed meshes: the wrap-around rows) gather from global memory.
// Several workgroups per CU overlap one tile's DMA with another's arithmetic.  Same k-ascending FMA chain as the CSR oracle.
// Built: hipcc -O3 --offload-arch=gfx950 -shared -fPIC lap_win.hip -o liblapwin.so
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kWG = 256;

__global__ __launch_bounds__(kWG) void win_table_k(const int *__restrict__ rowptr, const int *__restrict__ colind, int M, int K,
                                                   int R, int ntiles, int wcap, int2 *__restrict__ twin) {
  // one wave per tile
  const int tile = blockIdx.x * (kWG / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (tile >= ntiles) return;
  const int r0 = tile * R;
  const int nr = (M - r0) < R ? (M - r0) : R;
  int cmin = INT_MAX, cmax = -1;
  for (int lr = lane; lr < nr; lr += 64) {
    const int kb = rowptr[r0 + lr], ke = rowptr[r0 + lr + 1];
    if (ke > kb) {
      const int a = colind[kb], b = colind[ke - 1];
      cmin = a < cmin ? a : cmin;
      cmax = b > cmax ? b : cmax;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const int a = __shfl_xor(cmin, o), b = __shfl_xor(cmax, o);
    cmin = a < cmin ? a : cmin;
    cmax = b > cmax ? b : cmax;
  }
  if (lane == 0) {
    int ws = 0, wl = 0;
    if (cmax >= 0) {
      ws = cmin;
      wl = cmax + 1 - cmin;
      if (wl > wcap) {                    // centre the capped window on the tile's diagonal; the rest gathers from global
        int c = r0 - (wcap - nr) / 2;
        const int hi = cmax + 1 - wcap;
        c = c < hi ? c : hi;
        ws = c > cmin ? c : cmin;
        wl = wcap;
      }
    }
    twin[tile] = make_int2(ws, wl);
  }
}

template <bool EPI, bool STATS, int KB>
__global__ __launch_bounds__(kWG) void spmm_win_k(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                  const float *__restrict__ vals, const int2 *__restrict__ twin, int M, int K,
                                                  const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy,
                                                  int R, int ntiles, int nsl, int wmax, int emax, const float *__restrict__ E,
                                                  int64_t lde, const float *__restrict__ G, int64_t ldg,
                                                  float *__restrict__ stats_part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *xs = reinterpret_cast<float *>(smem);             // wmax x 32 floats
  int *sc = reinterpret_cast<int *>(xs + (size_t)wmax * 32);   // emax (multiple of 64)
  float *sv = reinterpret_cast<float *>(sc + emax);         // emax
  int *rp = reinterpret_cast<int *>(sv + emax);             // R + 64

  const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
  const int tpx = (ntiles + 7) >> 3;
  const int tl = slot / nsl, sl = slot - tl * nsl;
  const int tile = xcd * tpx + tl;
  if (tile >= ntiles) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = tile * R;
  const int nr = (M - r0) < R ? (M - r0) : R;
  const int2 tw = twin[tile];
  const int ws = tw.x, wlen = tw.y;
  const int k0 = rowptr[r0], k1 = rowptr[r0 + nr];
  const int ne = k1 - k0;
  const bool ent_lds = ne <= emax;
  const int c0 = sl * 32;
  {
    const int nrow8 = (wlen + 7) >> 3;
    const float *xg = X + c0 + (lane & 7) * 4;
    for (int i = wave; i < nrow8; i += kWG / 64) {
      int row = ws + 8 * i + (lane >> 3);
      row = row < K ? row : K - 1;
      __builtin_amdgcn_global_load_lds(xg + (int64_t)row * ldx, xs + i * 256, 16, 0, 0);
    }
    if (ent_lds)
      for (int p0 = wave * 64; p0 < ne; p0 += kWG) {
        int p = p0 + lane;
        p = p < ne ? p : ne - 1;
        __builtin_amdgcn_global_load_lds(colind + k0 + p, sc + p0, 4, 0, 0);
        __builtin_amdgcn_global_load_lds(vals + k0 + p, sv + p0, 4, 0, 0);
      }
    for (int p0 = wave * 64; p0 < nr + 1; p0 += kWG) {
      int p = p0 + lane;
      p = p < nr + 1 ? p : nr;
      __builtin_amdgcn_global_load_lds(rowptr + r0 + p, rp + p0, 4, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int g = lane >> 3, sub = lane & 7;
  const float *xl = xs + sub * 4;
  const float *xgl = X + c0 + sub * 4;
  f4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = ssum;
  for (int lr = wave * 8 + g; lr < nr; lr += 32) {
    const int r = r0 + lr;
    const int kb = rp[lr] - k0, ke = rp[lr + 1] - k0;
    f4 ev, gv;
    if constexpr (EPI) {
      ev = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(E + (int64_t)r * lde + c0 + sub * 4));
      if (G) gv = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(G + (int64_t)r * ldg + c0 + sub * 4));
    }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = kb; k < ke; k += KB) {
      int c[KB];
      float a[KB];
      f4 x[KB];
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int o = (k + j < ke) ? k + j : ke - 1;
        if (ent_lds) {
          c[j] = sc[o];
          a[j] = sv[o];
        } else {
          c[j] = colind[k0 + o];
          a[j] = vals[k0 + o];
        }
      }
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int u = c[j] - ws;
        const bool in = (unsigned)u < (unsigned)wlen;
        x[j] = *reinterpret_cast<const f4 *>(xl + (in ? u : 0) * 32);
        if (!in) x[j] = *reinterpret_cast<const f4 *>(xgl + (int64_t)c[j] * ldx);
      }
#pragma unroll
      for (int j = 0; j < KB; ++j)
        if (k + j < ke) {
          acc.x = __builtin_fmaf(a[j], x[j].x, acc.x);
          acc.y = __builtin_fmaf(a[j], x[j].y, acc.y);
          acc.z = __builtin_fmaf(a[j], x[j].z, acc.z);
          acc.w = __builtin_fmaf(a[j], x[j].w, acc.w);
        }
    }
    if constexpr (EPI) {
      acc = f4{acc.x * (ev.x > 0.f ? 1.f : ev.x + 1.f), acc.y * (ev.y > 0.f ? 1.f : ev.y + 1.f),
               acc.z * (ev.z > 0.f ? 1.f : ev.z + 1.f), acc.w * (ev.w > 0.f ? 1.f : ev.w + 1.f)};
      if (G) acc += gv;
    }
    __builtin_nontemporal_store(acc, reinterpret_cast<f4 *>(Y + (int64_t)r * ldy + c0 + sub * 4));
    if constexpr (STATS) {
      ssum += acc;
      ssq.x = __builtin_fmaf(acc.x, acc.x, ssq.x); ssq.y = __builtin_fmaf(acc.y, acc.y, ssq.y);
      ssq.z = __builtin_fmaf(acc.z, acc.z, ssq.z); ssq.w = __builtin_fmaf(acc.w, acc.w, ssq.w);
    }
  }
  if constexpr (STATS) {
    // [wave*8+g][sum | squares][32] through the (now free) window area; one (2 x 32) partial per workgroup and slice
    __syncthreads();
    float *st = xs + (wave * 8 + g) * 64;
    *reinterpret_cast<f4 *>(st + sub * 4) = ssum;
    *reinterpret_cast<f4 *>(st + 32 + sub * 4) = ssq;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 64) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 32; ++w) tot += xs[w * 64 + t];
      stats_part[(int64_t)tile * (2 * 32 * nsl) + (t >> 5) * (32 * nsl) + sl * 32 + (t & 31)] = tot;
    }
  }
}

extern "C" int lw_table(const int *rowptr, const int *colind, int M, int K, int R, int wcap, void *twin, void *stream) {
  const int ntiles = (M + R - 1) / R;
  if (ntiles == 0) return 0;
  hipLaunchKernelGGL(win_table_k, dim3((ntiles + 3) / 4), dim3(kWG), 0, static_cast<hipStream_t>(stream), rowptr, colind, M, K, R,
                     ntiles, wcap, static_cast<int2 *>(twin));
  return (int)hipGetLastError();
}

extern "C" int lw_lds_bytes(int R, int wmax, int emax) { return wmax * 128 + emax * 8 + (R + 64) * 4; }

extern "C" int lw_spmm(const int *rowptr, const int *colind, const float *vals, const void *twin, int M, int K, const float *X,
                       int64_t ldx, float *Y, int64_t ldy, int N, int R, int wmax, int emax, const float *E, int64_t lde,
                       const float *G, int64_t ldg, float *stats_part, int kb, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ntiles = (M + R - 1) / R;
  const int nsl = N / 32;
  const unsigned grid = (unsigned)(((ntiles + 7) / 8) * 8 * nsl);
  const size_t shm = (size_t)lw_lds_bytes(R, wmax, emax);
  if (shm > 160 * 1024) return -2;
#define LW(EPI_, ST_, KB_)                                                                                                        \
  do {                                                                                                                            \
    hipFuncSetAttribute((const void *)spmm_win_k<EPI_, ST_, KB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);        \
    hipLaunchKernelGGL((spmm_win_k<EPI_, ST_, KB_>), dim3(grid), dim3(kWG), shm, s, rowptr, colind, vals,                          \
                       static_cast<const int2 *>(twin), M, K, X, ldx, Y, ldy, R, ntiles, nsl, wmax, emax, E, lde, G, ldg,        \
                       stats_part);                                                                                               \
    return (int)hipGetLastError();                                                                                                \
  } while (0)
  if (stats_part) LW(false, true, 8);
  if (E) LW(true, false, 8);
  if (kb == 4) LW(false, false, 4);
  if (kb == 16) LW(false, false, 16);
  LW(false, false, 8);
}
