// Prototype (scratch): Laplacian-type SpMM as a SLIDING WINDOW over the rows of X held in an LDS ring.
//
// A persistent workgroup owns a strip of consecutive row chunks (R rows each) and one 32-column slice (one 128-byte line per
// X row).  The ring holds W = 4R rows of X addressed by (row mod W): while chunk t is multiplied out of LDS (its window is the
// X chunks t-1, t, t+1), the DMA of X chunk t+2 and of the CSR entries of chunk t+1 is in flight (global_load_lds, 1 KiB per
// wave instruction).  Every X line and every entry is requested ONCE per slice; columns outside the window (|c - r| > R:
// wrap-around rows of closed meshes, arbitrary orderings) gather from global memory.  k-ascending FMA chain as the CSR oracle.
// Built: hipcc -O3 --offload-arch=gfx950 -shared -fPIC lap_ring.hip -o liblapring.so
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int W, int NT, int CS, bool EPI, bool STATS>
__global__ __launch_bounds__(NT) void spmm_ring_k(const int *__restrict__ rowptr, const int *__restrict__ colind,
                                                  const float *__restrict__ vals, int M, int K, const float *__restrict__ X,
                                                  int64_t ldx, float *__restrict__ Y, int64_t ldy, int nstrips, int cps, int nsl,
                                                  int ecap, const float *__restrict__ E, int64_t lde,
                                                  const float *__restrict__ G, int64_t ldg, float *__restrict__ stats_part, int mode) {
  constexpr int R = W / 4;
  constexpr int NW = NT / 64;
  constexpr int RPS = R + 64;                   // row-pointer slots per buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *xs = reinterpret_cast<float *>(smem);                       // W x CS floats
  int *sc = reinterpret_cast<int *>(xs + W * CS);                     // [2][ecap]
  float *sv = reinterpret_cast<float *>(sc + 2 * ecap);               // [2][ecap]
  int *rp = reinterpret_cast<int *>(sv + 2 * ecap);                   // [2][RPS]

  const int b = blockIdx.x, xcd = b & 7, li = b >> 3;
  const int spx = nstrips >> 3;                                       // strips per XCD (nstrips is a multiple of 8)
  const int strip = xcd * spx + li / nsl, sl = li % nsl;
  const int nchunks = (M + R - 1) / R;
  const int t0 = strip * cps;
  const int t1 = (t0 + cps) < nchunks ? (t0 + cps) : nchunks;
  if (li / nsl >= spx || t0 >= t1) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = sl * CS;
  constexpr int LPX = CS / 4;                  // lanes per X row in a DMA instruction
  constexpr int RPI = 64 / LPX;                // rows per DMA instruction
  const float *xg = X + c0 + (lane % LPX) * 4;

  auto issue_x = [&](int j) {                                         // X chunk j -> ring
    if (j < 0 || (int64_t)j * R >= K) return;
    for (int i = wave; i < R / RPI; i += NW) {
      int row = j * R + RPI * i + lane / LPX;
      row = row < K ? row : K - 1;
      __builtin_amdgcn_global_load_lds(xg + (int64_t)row * ldx, xs + ((j * R + RPI * i) & (W - 1)) * CS, 16, 0, 0);
    }
  };
  auto issue_e = [&](int t, int k0, int k1) {                         // entries + row pointers of chunk t -> buffer t & 1
    const int buf = t & 1;
    const int ne = (k1 - k0) < ecap ? (k1 - k0) : ecap;               // entries past the buffer are read from global memory
    const int r0 = t * R;
    const int nr = (M - r0) < R ? (M - r0) : R;
    for (int p0 = wave * 64; p0 < ne; p0 += NT) {
      int p = p0 + lane;
      p = p < ne ? p : ne - 1;
      __builtin_amdgcn_global_load_lds(colind + k0 + p, sc + buf * ecap + p0, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(vals + k0 + p, sv + buf * ecap + p0, 4, 0, 0);
    }
    for (int p0 = wave * 64; p0 < nr + 1; p0 += NT) {
      int p = p0 + lane;
      p = p < nr + 1 ? p : nr;
      __builtin_amdgcn_global_load_lds(rowptr + r0 + p, rp + buf * RPS + p0, 4, 0, 0);
    }
  };
  auto kof = [&](int t) {                                             // first entry of chunk t (uniform: scalar load)
    const int64_t r = (int64_t)t * R;
    return rowptr[r < M ? r : M];
  };

  int k0 = kof(t0), k1 = kof(t0 + 1), k2 = kof(t0 + 2);
  issue_x(t0 - 1);
  issue_x(t0);
  issue_x(t0 + 1);
  issue_e(t0, k0, k1);

  const int g = lane >> 3, sub = lane & 7;
  const float *xl = xs + sub * 4;
  const float *xgl = X + c0 + sub * 4;
  f4 ssum[CS / 32], ssq[CS / 32];
#pragma unroll
  for (int v = 0; v < CS / 32; ++v) ssum[v] = ssq[v] = f4{0.f, 0.f, 0.f, 0.f};
  for (int t = t0; t < t1; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                  // chunk t landed; everyone is done with chunk t-1
    issue_x(t + 2);
    if (t + 1 < t1) issue_e(t + 1, k1, k2);
    const int k3 = kof(t + 3);
    const int buf = t & 1;
    const int *scb = sc + buf * ecap;
    const float *svb = sv + buf * ecap;
    const int *rpb = rp + buf * RPS;
    const int r0 = t * R;
    const int nr = (M - r0) < R ? (M - r0) : R;
    const int wlo = (t - 1) * R;
    constexpr int NV = CS / 32;                                       // float4 pieces per lane
    if (mode & 2) { k0 = k1; k1 = k2; k2 = k3; continue; }            // ablation: DMA + barriers only
    for (int lr0 = wave * 8; lr0 < nr; lr0 += NW * 8) {               // (wave-uniform trip count)
      const int lr = lr0 + g;
      const bool live = lr < nr;
      const int r = r0 + lr;
      int kb = 0, ke = 0;
      if (live) {
        kb = rpb[lr] - k0;
        ke = rpb[lr + 1] - k0;
      }
      f4 ev[NV], gv[NV];
      if constexpr (EPI) {
        if (live) {
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            ev[v] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(E + (int64_t)r * lde + c0 + v * 32 + sub * 4));
            if (G) gv[v] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(G + (int64_t)r * ldg + c0 + v * 32 + sub * 4));
          }
        }
      }
      // fast path (the whole wave): rows of at most 8 entries, all of them in the LDS buffer, all columns in the window.
      // The 8 (column, coefficient) slots of a row are read from kb onwards whatever the row's length: slots past its end
      // hold the next rows' entries (or spare buffer words) and are replaced by (first column, 0) — fma(0, x, acc) == acc.
      const int len = ke - kb;
      bool ok = ke <= ecap && len <= 8;
      int c[8];
      float a[8];
      {
        const int *cp = scb + kb;
        const float *ap = svb + kb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          c[j] = cp[j];
          a[j] = ap[j];
        }
        const int cl = scb[ke > 0 ? ke - 1 : 0];                      // (columns ascend within a row: first and last bound the rest)
        ok = ok && (len <= 0 || ((unsigned)(c[0] - wlo) < (unsigned)(3 * R) && (unsigned)(cl - wlo) < (unsigned)(3 * R)));
      }
      f4 acc[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = f4{0.f, 0.f, 0.f, 0.f};
      if (__builtin_amdgcn_ballot_w64(!ok) == 0) {
        f4 x[8][NV];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cj = j < len ? c[j] : c[0];
          a[j] = j < len ? a[j] : 0.f;
          const float *xp = xl + (cj & (W - 1)) * CS;
#pragma unroll
          for (int v = 0; v < NV; ++v) x[j][v] = *reinterpret_cast<const f4 *>(xp + v * 32);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            acc[v].x = __builtin_fmaf(a[j], x[j][v].x, acc[v].x);
            acc[v].y = __builtin_fmaf(a[j], x[j][v].y, acc[v].y);
            acc[v].z = __builtin_fmaf(a[j], x[j][v].z, acc[v].z);
            acc[v].w = __builtin_fmaf(a[j], x[j][v].w, acc[v].w);
          }
        }
      } else {
        for (int k = kb; k < ke; ++k) {                               // slow path: entry by entry, LDS or global
          int c;
          float a;
          if (k < ecap) {
            c = scb[k];
            a = svb[k];
          } else {
            c = colind[k0 + k];
            a = vals[k0 + k];
          }
          const bool in = (unsigned)(c - wlo) < (unsigned)(3 * R);
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            f4 x;
            if (in) x = *reinterpret_cast<const f4 *>(xl + (c & (W - 1)) * CS + v * 32);
            else x = *reinterpret_cast<const f4 *>(xgl + (int64_t)c * ldx + v * 32);
            acc[v].x = __builtin_fmaf(a, x.x, acc[v].x);
            acc[v].y = __builtin_fmaf(a, x.y, acc[v].y);
            acc[v].z = __builtin_fmaf(a, x.z, acc[v].z);
            acc[v].w = __builtin_fmaf(a, x.w, acc[v].w);
          }
        }
      }
      if (live) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          f4 o = acc[v];
          if constexpr (EPI) {
            o = f4{o.x * (ev[v].x > 0.f ? 1.f : ev[v].x + 1.f), o.y * (ev[v].y > 0.f ? 1.f : ev[v].y + 1.f),
                   o.z * (ev[v].z > 0.f ? 1.f : ev[v].z + 1.f), o.w * (ev[v].w > 0.f ? 1.f : ev[v].w + 1.f)};
            if (G) o += gv[v];
          }
          if (!(mode & 1) || o.x == 12345.678f)                     // ablation: no stores
            __builtin_nontemporal_store(o, reinterpret_cast<f4 *>(Y + (int64_t)r * ldy + c0 + v * 32 + sub * 4));
          if constexpr (STATS) {
            ssum[v] += o;
            ssq[v].x = __builtin_fmaf(o.x, o.x, ssq[v].x); ssq[v].y = __builtin_fmaf(o.y, o.y, ssq[v].y);
            ssq[v].z = __builtin_fmaf(o.z, o.z, ssq[v].z); ssq[v].w = __builtin_fmaf(o.w, o.w, ssq[v].w);
          }
        }
      }
    }
    k0 = k1; k1 = k2; k2 = k3;
  }
  if constexpr (STATS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float *st = xs + (wave * 8 + g) * (2 * CS);                       // [NW*8][sum | squares][CS]
#pragma unroll
    for (int v = 0; v < CS / 32; ++v) {
      *reinterpret_cast<f4 *>(st + v * 32 + sub * 4) = ssum[v];
      *reinterpret_cast<f4 *>(st + CS + v * 32 + sub * 4) = ssq[v];
    }
    __syncthreads();
    const int tt = threadIdx.x;
    if (tt < 2 * CS) {
      float tot = 0.f;
      for (int w = 0; w < NW * 8; ++w) tot += xs[w * (2 * CS) + tt];
      stats_part[(int64_t)strip * (2 * CS * nsl) + (tt / CS) * (CS * nsl) + sl * CS + (tt % CS)] = tot;
    }
  }
}

extern "C" int lr_lds_bytes(int W, int CS, int ecap) { return W * CS * 4 + 2 * ecap * 8 + 2 * (W / 4 + 64) * 4; }

extern "C" int lr_spmm(const int *rowptr, const int *colind, const float *vals, int M, int K, const float *X, int64_t ldx, float *Y,
                       int64_t ldy, int N, int W, int NT, int CS, int nstrips, int ecap, const float *E, int64_t lde, const float *G,
                       int64_t ldg, float *stats_part, int mode, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int R = W / 4;
  const int nchunks = (M + R - 1) / R;
  const int cps = (nchunks + nstrips - 1) / nstrips;
  const int nsl = N / CS;
  const unsigned grid = (unsigned)(nstrips * nsl);
  const size_t shm = (size_t)lr_lds_bytes(W, CS, ecap);
  if (shm > 160 * 1024 || nstrips % 8) return -2;
#define LR(W_, NT_, CS_, EPI_, ST_)                                                                                               \
  do {                                                                                                                            \
    hipFuncSetAttribute((const void *)spmm_ring_k<W_, NT_, CS_, EPI_, ST_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL((spmm_ring_k<W_, NT_, CS_, EPI_, ST_>), dim3(grid), dim3(NT_), shm, s, rowptr, colind, vals, M, K, X, ldx, \
                       Y, ldy, nstrips, cps, nsl, ecap, E, lde, G, ldg, stats_part, mode);                                              \
    return (int)hipGetLastError();                                                                                                \
  } while (0)
#define LR_V(W_, NT_, CS_)                                                                                                        \
  if (W == W_ && NT == NT_ && CS == CS_) {                                                                                        \
    if (stats_part) LR(W_, NT_, CS_, false, true);                                                                                \
    if (E) LR(W_, NT_, CS_, true, false);                                                                                         \
    LR(W_, NT_, CS_, false, false);                                                                                               \
  }
  LR_V(512, 256, 32) LR_V(512, 512, 32) LR_V(1024, 512, 32) LR_V(1024, 1024, 32) LR_V(512, 512, 64) LR_V(512, 1024, 64)
  LR_V(256, 256, 64) LR_V(256, 512, 64)
  return -1;
}
