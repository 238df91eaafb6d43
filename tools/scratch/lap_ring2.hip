// Prototype (scratch), second form: sliding-window Laplacian SpMM with LOADER / COMPUTE wave specialisation.
//
// As lap_ring.hip (a persistent workgroup walks a strip of rows; X rows live in an LDS ring addressed by row mod W), with
//   * the window decoupled from the step: step t covers rows [tR, (t+1)R) and needs X rows [tR - H, (t+1)R + H);
//   * NLW loader waves that only issue LDS-DMA (X piece, CSR entries and row pointers of step t + D) and wait for it with
//     a partial s_waitcnt vmcnt — they never store, so their counter counts DMA only and D steps stay in flight;
//   * R/8 compute waves (one 8-row round per step) that read LDS, multiply and store, and never wait for their stores;
//   * one s_barrier per step (no implicit vmcnt(0)).
// Built: hipcc -O3 --offload-arch=gfx950 -shared -fPIC lap_ring2.hip -o liblapring2.so
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

template <int CS, int W, int R, int H, int D, int NLW, bool Q16, bool EPI, bool STATS>
__global__ __launch_bounds__((R / 8 + NLW) * 64) void spmm_ring2_k(
    const int *__restrict__ rowptr, const int *__restrict__ colind, const float *__restrict__ vals, int M, int K, int nnz,
    const float *__restrict__ X, int64_t ldx, float *__restrict__ Y, int64_t ldy, int nstrips, int cps, int nsl,
    const float *__restrict__ E, int64_t lde, const float *__restrict__ G, int64_t ldg, float *__restrict__ stats_part, int mode) {
  constexpr int NCW = R / 8;                    // compute waves: one 8-row round each per step
  constexpr int ECAP = R * 8;                   // entry slots per step buffer (a multiple of 256)
  constexpr int NB = D + 1;                     // entry / row-pointer buffers
  constexpr int RPS = R + 64;
  constexpr int LPX = CS / 4, RPI = 64 / LPX;   // lanes per X row in a DMA instruction, rows per instruction
  constexpr int NV = CS / 32;                   // float4 pieces per lane
  static_assert(W >= (D + 1) * R + 2 * H, "ring too small");
  static_assert((W & (W - 1)) == 0 && R % RPI == 0 && H % RPI == 0 && ECAP % 256 == 0, "shape");
  // DMA instructions of ONE step per loader wave (constant, so that the partial wait is an immediate)
  constexpr int NX = (R / RPI + NLW - 1) / NLW;
  constexpr int EPI_ = Q16 ? 256 : 64;          // entries per DMA instruction
  constexpr int NE = 2 * ((ECAP / EPI_ + NLW - 1) / NLW);
  constexpr int NR = ((R + 1 + 63) / 64 + NLW - 1) / NLW;
  constexpr int NSTEP = NX + NE + NR;
  static_assert((D - 1) * NSTEP <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *xs = reinterpret_cast<float *>(smem);                        // W x CS floats
  int *sc = reinterpret_cast<int *>(xs + W * CS);                     // [NB][ECAP]
  float *sv = reinterpret_cast<float *>(sc + NB * ECAP);              // [NB][ECAP]
  int *rp = reinterpret_cast<int *>(sv + NB * ECAP);                  // [NB][RPS]

  const int b = blockIdx.x, xcd = b & 7, li = b >> 3;
  const int spx = nstrips >> 3;
  const int strip = xcd * spx + li / nsl, sl = li % nsl;
  const int nsteps = (M + R - 1) / R;
  const int t0 = strip * cps;
  const int t1 = (t0 + cps) < nsteps ? (t0 + cps) : nsteps;
  if (li / nsl >= spx || t0 >= t1) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = sl * CS;

  if (wave >= NCW) {
    // ------------------------------------------------------------------ loader
    const int lw = wave - NCW;
    const float *xg = X + c0 + (lane % LPX) * 4;
    auto issue_rows = [&](int row0, int i) {                          // RPI rows starting at row0 + RPI*i -> ring
      int row = row0 + RPI * i + lane / LPX;
      row = row < 0 ? 0 : (row < K ? row : K - 1);
      __builtin_amdgcn_global_load_lds(xg + (int64_t)row * ldx, xs + ((row0 + RPI * i) & (W - 1)) * CS, 16, 0, 0);
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
      if constexpr (Q16) {
        // 16 bytes per lane from the 16-byte aligned quad that holds entry k0: slot j of the buffer is entry (k0 & ~3) + j.
        // (reads up to 3 entries past k1: the arrays must be readable up to the next multiple of 4 entries)
        const int q0 = k0 >> 2;
        int nq = ((k1 + 3) >> 2) - q0;                                // quads that hold entries of this step
        nq = nq < ECAP / 4 ? nq : ECAP / 4;
#pragma unroll
        for (int q = 0; q < NE / 2; ++q) {
          int p0 = (lw + q * NLW) * 64;                               // first quad of this instruction
          p0 = p0 < ECAP / 4 ? p0 : ECAP / 4 - 64;
          int p = p0 + lane;
          p = p < nq ? p : (nq > 0 ? nq - 1 : 0);
          int64_t kq = (int64_t)(q0 + p) * 4;
          const int64_t last = nnz > 4 ? (int64_t)((nnz - 1) >> 2) * 4 : 0;
          kq = kq < last ? kq : last;
          __builtin_amdgcn_global_load_lds(colind + kq, sc + buf * ECAP + p0 * 4, 16, 0, 0);
          __builtin_amdgcn_global_load_lds(vals + kq, sv + buf * ECAP + p0 * 4, 16, 0, 0);
        }
      } else {
        int ne = k1 - k0;
        ne = ne < ECAP ? ne : ECAP;
#pragma unroll
        for (int q = 0; q < NE / 2; ++q) {
          int p0 = (lw + q * NLW) * 64;
          p0 = p0 < ECAP ? p0 : ECAP - 64;
          int p = p0 + lane;
          p = p < ne ? p : (ne > 0 ? ne - 1 : 0);
          int k = k0 + p;
          k = k < nnz ? k : nnz - 1;
          __builtin_amdgcn_global_load_lds(colind + k, sc + buf * ECAP + p0, 4, 0, 0);
          __builtin_amdgcn_global_load_lds(vals + k, sv + buf * ECAP + p0, 4, 0, 0);
        }
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
    // prologue: the first window [t0 R - H, t0 R + H) and the steps t0 .. t0 + D - 1
    for (int i = lw; i < 2 * H / RPI; i += NLW) issue_rows(t0 * R - H, i);
#pragma unroll
    for (int d = 0; d < D; ++d) issue_step(t0 + d);                   // (steps past t1 are loaded too: harmless, clamped)
    wait_vmcnt<(D - 1) * NSTEP>();
    __builtin_amdgcn_s_barrier();
    for (int t = t0; t < t1; ++t) {
      issue_step(t + D);                                              // (steps past t1 too: clamped, harmless — uniform counts)
      wait_vmcnt<(D - 1) * NSTEP>();                                  // step t + 1 has landed
      __builtin_amdgcn_s_barrier();
    }
    wait_vmcnt<0>();                                                  // nothing may land in LDS after the workgroup has gone
    return;
  }

  // -------------------------------------------------------------------- compute
  const int g = lane >> 3, sub = lane & 7;
  const float *xl = xs + sub * 4;
  const float *xgl = X + c0 + sub * 4;
  f4 ssum[NV], ssq[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) ssum[v] = ssq[v] = f4{0.f, 0.f, 0.f, 0.f};
  f4 evn[NV], gvn[NV];
  if constexpr (EPI) {
    const int rf = t0 * R + wave * 8 + g;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      evn[v] = gvn[v] = f4{0.f, 0.f, 0.f, 0.f};
      if (rf < M) {
        evn[v] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(E + (int64_t)rf * lde + c0 + v * 32 + sub * 4));
        if (G) gvn[v] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(G + (int64_t)rf * ldg + c0 + v * 32 + sub * 4));
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
    if (!(mode & 2)) {
      const int k0 = Q16 ? (rpb[0] & ~3) : rpb[0];                    // entry held by slot 0 of the buffer
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
        const int rn = r + R;                                         // this lane group's row of the next step
        if (rn < M && t + 1 < t1) {
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            evn[v] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(E + (int64_t)rn * lde + c0 + v * 32 + sub * 4));
            if (G) gvn[v] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(G + (int64_t)rn * ldg + c0 + v * 32 + sub * 4));
          }
        }
      }
      const int len = ke - kb;
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
      }
      const int cl = scb[ke > 0 ? ke - 1 : 0];                        // (columns ascend within a row)
      const bool fits = ke <= ECAP && len <= 8;
      const bool inwin = len <= 0 || ((unsigned)(c[0] - wlo) < (unsigned)(R + 2 * H) && (unsigned)(cl - wlo) < (unsigned)(R + 2 * H));
      f4 acc[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = f4{0.f, 0.f, 0.f, 0.f};
      if (__builtin_amdgcn_ballot_w64(!(fits && inwin)) == 0) {
        // fast path (whole wave): slots past a row's end become (first column, 0): fma(0, x, acc) == acc
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
      } else if (__builtin_amdgcn_ballot_w64(!fits) == 0) {
        // mixed path (some column outside the window: wrap-around rows of closed meshes): the same batch, every slot read
        // from the ring AND, where its column is outside, from global memory (lanes inside are masked off), then selected
        f4 x[8][NV];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cj = j < len ? c[j] : c[0];
          a[j] = j < len ? a[j] : 0.f;
          const bool in = (unsigned)(cj - wlo) < (unsigned)(R + 2 * H);
          const float *xp = xl + (cj & (W - 1)) * CS;
          const float *gp = xgl + (int64_t)cj * ldx;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            x[j][v] = *reinterpret_cast<const f4 *>(xp + v * 32);
            if (!in) x[j][v] = *reinterpret_cast<const f4 *>(gp + v * 32);
          }
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
        for (int k = kb; k < ke; ++k) {                               // slow path (rows longer than 8 entries / past the buffer)
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
            else x = *reinterpret_cast<const f4 *>(xgl + (int64_t)cc * ldx + v * 32);
            acc[v].x = __builtin_fmaf(aa, x.x, acc[v].x);
            acc[v].y = __builtin_fmaf(aa, x.y, acc[v].y);
            acc[v].z = __builtin_fmaf(aa, x.z, acc[v].z);
            acc[v].w = __builtin_fmaf(aa, x.w, acc[v].w);
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
          if (!(mode & 1) || o.x == 12345.678f)
            __builtin_nontemporal_store(o, reinterpret_cast<f4 *>(Y + (int64_t)r * ldy + c0 + v * 32 + sub * 4));
          if constexpr (STATS) {
            ssum[v] += o;
            ssq[v].x = __builtin_fmaf(o.x, o.x, ssq[v].x); ssq[v].y = __builtin_fmaf(o.y, o.y, ssq[v].y);
            ssq[v].z = __builtin_fmaf(o.z, o.z, ssq[v].z); ssq[v].w = __builtin_fmaf(o.w, o.w, ssq[v].w);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // every LDS read of this step has returned
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (STATS) {
    // (the loader waves have left: a finished wave no longer counts at s_barrier)
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
      stats_part[(int64_t)strip * (2 * CS * nsl) + (tt / CS) * (CS * nsl) + sl * CS + (tt % CS)] = tot;
    }
  }
}

extern "C" int lr2_lds_bytes(int CS, int W, int R, int D) { return W * CS * 4 + (D + 1) * (R * 8) * 8 + (D + 1) * (R + 64) * 4; }

extern "C" int lr2_spmm(const int *rowptr, const int *colind, const float *vals, int M, int K, int nnz, const float *X, int64_t ldx,
                        float *Y, int64_t ldy, int N, int variant, int nstrips, const float *E, int64_t lde, const float *G,
                        int64_t ldg, float *stats_part, int mode, void *stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (nstrips % 8 || nnz < 1) return -2;
#define LR2K(CS_, W_, R_, H_, D_, NLW_, Q_, EPI_, ST_)                                                                          \
  do {                                                                                                                            \
    hipFuncSetAttribute((const void *)spmm_ring2_k<CS_, W_, R_, H_, D_, NLW_, Q_, EPI_, ST_>,                                   \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                                                    \
    hipLaunchKernelGGL((spmm_ring2_k<CS_, W_, R_, H_, D_, NLW_, Q_, EPI_, ST_>), dim3(grid), dim3((R_ / 8 + NLW_) * 64), shm, s,  \
                       rowptr, colind, vals, M, K, nnz, X, ldx, Y, ldy, nstrips, cps, nsl, E, lde, G, ldg, stats_part, mode);     \
  } while (0)
#define LR2(CS_, W_, R_, H_, D_, NLW_, Q_)                                                                                        \
  do {                                                                                                                            \
    const int nsteps = (M + R_ - 1) / R_;                                                                                         \
    const int cps = (nsteps + nstrips - 1) / nstrips;                                                                             \
    const int nsl = N / CS_;                                                                                                      \
    const unsigned grid = (unsigned)(nstrips * nsl);                                                                              \
    const size_t shm = (size_t)lr2_lds_bytes(CS_, W_, R_, D_);                                                                    \
    if (shm > 160 * 1024) return -3;                                                                                              \
    if (stats_part) LR2K(CS_, W_, R_, H_, D_, NLW_, Q_, false, true);                                                             \
    else if (E) LR2K(CS_, W_, R_, H_, D_, NLW_, Q_, true, false);                                                                 \
    else LR2K(CS_, W_, R_, H_, D_, NLW_, Q_, false, false);                                                                       \
    return (int)hipGetLastError();                                                                                                \
  } while (0)
  switch (variant) {
    case 0: LR2(32, 512, 64, 160, 2, 2, false);     // 2 workgroups per CU (80 KB class), 4-byte entry DMA, 2 loaders (= round-1 v6 with H 160)
    case 1: LR2(32, 512, 64, 160, 2, 4, false);     //   4 loader waves
    case 2: LR2(32, 512, 64, 160, 2, 2, true);      //   16-byte entry DMA, 2 loaders
    case 3: LR2(32, 512, 64, 160, 2, 4, true);      //   16-byte entry DMA, 4 loaders
    case 4: LR2(64, 512, 64, 160, 2, 4, true);      // 64-column slices, 1 workgroup per CU, 4 loaders
    case 5: LR2(64, 512, 64, 160, 2, 8, true);      //   8 loaders
    case 6: LR2(64, 512, 64, 160, 2, 4, false);
    case 7: LR2(64, 512, 64, 128, 3, 4, true);
    case 8: LR2(32, 1024, 64, 160, 3, 4, true);     // 1 workgroup per CU, deep ring
    default: break;
  }
  return -1;
}
