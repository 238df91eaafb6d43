// sn_meshops.hip — device-side construction of the Dirac operators from (V, F)  (gfx950).
// fp64 geometry in the reference's operation order; FMA contraction is switched off for this file so that the
// results round to the same fp32 values as the numpy pipeline (src/utils/mesh.py:17-64).
#pragma clang fp contract(off)

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "sn_spmm.h"

// fills / device copies as kernels of this library (sn_kernels.hip says why not hipMemsetAsync)
hipError_t sn_internal_fill(void *dst, int value, size_t bytes, hipStream_t s);
hipError_t sn_internal_copy2d(void *dst, int64_t dpitch, const void *src, int64_t spitch, int64_t width, int64_t rows, hipStream_t s);

namespace {

constexpr int kWG = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kWG * kScanItems;

inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SN_OK : (int)e;
}
inline unsigned grid_for(int64_t n) {
  int64_t b = (n + kWG - 1) / kWG;
  if (b < 1) b = 1;
  if (b > 65535 * 16) b = 65535 * 16;
  return (unsigned)b;
}

// mesh.dist entry: sqrt(((V[a]-V[b])**2).sum())  — (dx² + dy²) + dz², as numpy sums three terms
__device__ __forceinline__ double edge_len(const float *__restrict__ V, int a, int b) {
  const double dx = (double)V[3 * a] - (double)V[3 * b];
  const double dy = (double)V[3 * a + 1] - (double)V[3 * b + 1];
  const double dz = (double)V[3 * a + 2] - (double)V[3 * b + 2];
  return sqrt((dx * dx + dy * dy) + dz * dz);
}

// mesh.area (Heron, 1e-6 floor), src/utils/mesh.py:67-80; also counts incident faces per vertex
__global__ __launch_bounds__(kWG) void face_area_k(const float *__restrict__ V, const int *__restrict__ F, int64_t nF,
                                                   double *__restrict__ Af, int *__restrict__ vcount) {
  for (int64_t f = (int64_t)blockIdx.x * kWG + threadIdx.x; f < nF; f += (int64_t)gridDim.x * kWG) {
    const int i = F[3 * f], j = F[3 * f + 1], k = F[3 * f + 2];
    const double lij = edge_len(V, i, j), ljk = edge_len(V, j, k), lki = edge_len(V, k, i);
    const double s = ((lij + ljk) + lki) / 2;
    const double q = ((s * (s - lij)) * (s - ljk)) * (s - lki);
    Af[f] = q > 0 ? sqrt(q) : 1e-6;
    atomicAdd(&vcount[i], 1);
    atomicAdd(&vcount[j], 1);
    atomicAdd(&vcount[k], 1);
  }
}

// ---- int32 exclusive scan (same 3-launch scheme as sn_kernels.hip) ---------------------------------------
__device__ __forceinline__ int block_excl_scan(int v, int *total) {
  __shared__ int wsum[kWG / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
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
__global__ __launch_bounds__(kWG) void scan_sums_k(const int *__restrict__ in, int64_t n, int *__restrict__ sums) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  int s = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) s += in[base + i];
  int tot;
  block_excl_scan(s, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(kWG) void scan_top_k(int *__restrict__ sums, int nblk) {
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
__global__ __launch_bounds__(kWG) void scan_apply_k(const int *__restrict__ in, int64_t n, const int *__restrict__ sums,
                                                    int *__restrict__ out) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  int v[kScanItems], s = 0;
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

// incidence lists: entry = 4*face + corner, scattered with an atomic cursor then sorted per vertex (= by face)
__global__ __launch_bounds__(kWG) void incidence_scatter_k(const int *__restrict__ F, int64_t nF, int *__restrict__ cursor,
                                                           int *__restrict__ inc) {
  for (int64_t f = (int64_t)blockIdx.x * kWG + threadIdx.x; f < nF; f += (int64_t)gridDim.x * kWG)
#pragma unroll
    for (int c = 0; c < 3; ++c) inc[atomicAdd(&cursor[F[3 * f + c]], 1)] = (int)(4 * f + c);
}

// sort each vertex's list, write DiA's block columns, and Av_j = sum_{incident faces, ascending} Af/3 (mesh.py:43-45)
__global__ __launch_bounds__(kWG) void vertex_lists_k(const int *__restrict__ vptr, int64_t nV, int *__restrict__ inc,
                                                      const double *__restrict__ Af, double *__restrict__ Av,
                                                      int *__restrict__ dia_colind) {
  for (int64_t v = (int64_t)blockIdx.x * kWG + threadIdx.x; v < nV; v += (int64_t)gridDim.x * kWG) {
    const int b = vptr[v], e = vptr[v + 1];
    for (int i = b + 1; i < e; ++i) {
      const int key = inc[i];
      int j = i - 1;
      while (j >= b && inc[j] > key) {
        inc[j + 1] = inc[j];
        --j;
      }
      inc[j + 1] = key;
    }
    double a = 0;
    for (int i = b; i < e; ++i) {
      a += Af[inc[i] >> 2] / 3;
      dia_colind[i] = inc[i] >> 2;
    }
    Av[v] = a;
  }
}

__global__ __launch_bounds__(kWG) void vertex_sort_k(const int *__restrict__ vptr, int64_t nV, int *__restrict__ inc) {
  for (int64_t v = (int64_t)blockIdx.x * kWG + threadIdx.x; v < nV; v += (int64_t)gridDim.x * kWG) {
    const int b = vptr[v], e = vptr[v + 1];
    for (int i = b + 1; i < e; ++i) {
      const int key = inc[i];
      int j = i - 1;
      while (j >= b && inc[j] > key) {
        inc[j + 1] = inc[j];
        --j;
      }
      inc[j + 1] = key;
    }
  }
}

// the 4x4 block  -Q(0,e)/(2Af)  (mesh.py:28-33,55-58), row-major, as doubles
__device__ __forceinline__ void dirac_block(const float *__restrict__ V, const int *__restrict__ F, int64_t f, int c,
                                            double Af, double *m /*16*/) {
  const int a = F[3 * f + (c + 1) % 3], b = F[3 * f + (c + 2) % 3];
  const double ex = (double)V[3 * a] - (double)V[3 * b];
  const double ey = (double)V[3 * a + 1] - (double)V[3 * b + 1];
  const double ez = (double)V[3 * a + 2] - (double)V[3 * b + 2];
  const double sc = 2 * Af;
  // Q(0,b,c,d) = [[0,-b,-c,-d],[b,0,-d,c],[c,d,0,-b],[d,-c,b,0]];  entry = -(q)/(2Af)
  const double q[16] = {0.0, -ex, -ey, -ez, ex, 0.0, -ez, ey, ey, ez, 0.0, -ex, ez, -ey, ex, 0.0};
#pragma unroll
  for (int i = 0; i < 16; ++i) m[i] = -q[i] / sc;
}

// Di (block row = face) and DiAT (same structure, blocks = (DiA block)^T = D_block * Af/Av)
__global__ __launch_bounds__(kWG) void di_fill_k(const float *__restrict__ V, const int *__restrict__ F, int64_t nF,
                                                 const double *__restrict__ Af, const double *__restrict__ Av,
                                                 int *__restrict__ di_rowptr, int *__restrict__ di_colind,
                                                 float *__restrict__ di_vals, float *__restrict__ diat_vals) {
  for (int64_t f = (int64_t)blockIdx.x * kWG + threadIdx.x; f <= nF; f += (int64_t)gridDim.x * kWG) {
    di_rowptr[f] = (int)(3 * f);
    if (f == nF) break;
    int c0 = 0, c1 = 1, c2 = 2;                        // corners ordered by vertex id (block columns ascending)
    int v0 = F[3 * f], v1 = F[3 * f + 1], v2 = F[3 * f + 2];
#define SN_SWAP(a, b, x, y) if (a > b) { int t = a; a = b; b = t; t = x; x = y; y = t; }
    SN_SWAP(v0, v1, c0, c1) SN_SWAP(v1, v2, c1, c2) SN_SWAP(v0, v1, c0, c1)
#undef SN_SWAP
    const int vs[3] = {v0, v1, v2}, cs[3] = {c0, c1, c2};
    const double af = Af[f];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      double m[16];
      dirac_block(V, F, f, cs[s], af, m);
      const int64_t o = 3 * f + s;
      di_colind[o] = vs[s];
      const double av = Av[vs[s]];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        di_vals[16 * o + i] = (float)m[i];
        diat_vals[16 * o + i] = (float)((m[i] * af) / av);      // (mat.T * Af / Av) transposed back
      }
    }
  }
}

// DiA (block row = vertex) = D_block^T * Af/Av, and DiT = D_block^T (same structure)
__global__ __launch_bounds__(kWG) void dia_fill_k(const float *__restrict__ V, const int *__restrict__ F, int64_t nV,
                                                  const int *__restrict__ vptr, const int *__restrict__ inc,
                                                  const double *__restrict__ Af, const double *__restrict__ Av,
                                                  int *__restrict__ dia_rowptr, float *__restrict__ dia_vals,
                                                  float *__restrict__ dit_vals) {
  for (int64_t v = (int64_t)blockIdx.x * kWG + threadIdx.x; v <= nV; v += (int64_t)gridDim.x * kWG) {
    dia_rowptr[v] = vptr[v];
    if (v == nV) break;
    const double av = Av[v];
    for (int o = vptr[v]; o < vptr[v + 1]; ++o) {
      const int64_t f = inc[o] >> 2;
      const int c = inc[o] & 3;
      const double af = Af[f];
      double m[16];
      dirac_block(V, F, f, c, af, m);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double t = m[4 * k + r];                          // transpose
          dit_vals[16 * (int64_t)o + 4 * r + k] = (float)t;
          dia_vals[16 * (int64_t)o + 4 * r + k] = (float)((t * af) / av);
        }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Cotangent Laplacian, vertex-centric.  Thread i walks its incident faces (ascending face id) and builds, for every
// neighbour j, W[i,j] (row i's value) and W[j,i] (row j's value, needed for the column sum d_i) from the same face data,
// exactly as the reference's permutation loop does:  (-l_pq^2 + l_qr^2 + l_rp^2) / (8 a + 1e-6)  for (p,q,r) = (i,j,k).
// ------------------------------------------------------------------------------------------------
constexpr int kMaxDeg = SN_LAP_MAX_DEGREE;

template <bool FILL>
__global__ __launch_bounds__(kWG) void laplacian_rows_k(const float *__restrict__ V, const int *__restrict__ F, int64_t nV,
                                                        const int *__restrict__ vptr, const int *__restrict__ inc,
                                                        const double *__restrict__ Af, int *__restrict__ rowptr,
                                                        int *__restrict__ colind, float *__restrict__ vals,
                                                        int *__restrict__ status_flag) {
  for (int64_t i = (int64_t)blockIdx.x * kWG + threadIdx.x; i < nV; i += (int64_t)gridDim.x * kWG) {
    const int b = vptr[i], e = vptr[i + 1];
    if (e - b > kMaxDeg) {
      if (status_flag) atomicExch(status_flag, 1);
      if constexpr (!FILL) rowptr[i] = 1;
      continue;
    }
    int nb[2 * kMaxDeg];          // neighbour ids (with duplicates), in face order
    double wij[2 * kMaxDeg];      // contribution to W[i, nb]
    double wji[2 * kMaxDeg];      // contribution to W[nb, i]
    int n = 0;
    double A = 0;
    for (int o = b; o < e; ++o) {
      const int64_t f = inc[o] >> 2;
      const int c = inc[o] & 3;
      const int j = F[3 * f + (c + 1) % 3], k = F[3 * f + (c + 2) % 3];
      const double lij = edge_len(V, (int)i, j), ljk = edge_len(V, j, k), lki = edge_len(V, k, (int)i);
      const double a2 = lij * lij, b2 = ljk * ljk, c2 = lki * lki;
      const double den = 8 * Af[f] + 1e-6;
      // permutations (i,j,k): W[i,j];  (i,k,j): W[i,k];  (j,i,k): W[j,i];  (k,i,j): W[k,i]
      nb[n] = j; wij[n] = ((-a2 + b2) + c2) / den; wji[n] = ((-a2 + c2) + b2) / den; ++n;
      nb[n] = k; wij[n] = ((-c2 + b2) + a2) / den; wji[n] = ((-c2 + a2) + b2) / den; ++n;
      const double t = Af[f] / 3 / 4;
      A += t;
      A += t;
    }
    // stable insertion sort by neighbour id (keeps face order among duplicates)
    for (int x = 1; x < n; ++x) {
      const int kn = nb[x];
      const double u = wij[x], v = wji[x];
      int y = x - 1;
      while (y >= 0 && nb[y] > kn) {
        nb[y + 1] = nb[y]; wij[y + 1] = wij[y]; wji[y + 1] = wji[y];
        --y;
      }
      nb[y + 1] = kn; wij[y + 1] = u; wji[y + 1] = v;
    }
    // merge duplicates; d_i = sum over neighbours (ascending) of W[nb, i]; entries with W[i,nb] == 0 are dropped
    const double ainv = 1 / (A + 1e-9);
    double d = 0;
    int cnt = 0;
    int out = FILL ? rowptr[i] : 0;
    bool diag_done = false;
    for (int x = 0; x < n;) {
      const int kn = nb[x];
      double w = 0, wt = 0;
      while (x < n && nb[x] == kn) { w += wij[x]; wt += wji[x]; ++x; }
      if (wt != 0) d += wt;
      if (w != 0) {
        if constexpr (FILL) {
          if (!diag_done && kn > i) { ++out; diag_done = true; }          // leave the diagonal slot, filled below
          colind[out] = kn;
          vals[out] = (float)(ainv * (0 - w));
          ++out;
        }
        ++cnt;
      }
    }
    if constexpr (FILL) {
      // diagonal position: after the neighbours smaller than i
      int pos = rowptr[i];
      for (int x = 0, seen = -1; x < n; ++x) {
        if (nb[x] == seen) continue;
        seen = nb[x];
        double w = 0;
        for (int y = x; y < n && nb[y] == seen; ++y) w += wij[y];
        if (w != 0 && seen < i) ++pos;
      }
      colind[pos] = (int)i;
      vals[pos] = (float)(ainv * d);
    } else {
      rowptr[i] = cnt + 1;
    }
  }
}

inline size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }

}  // namespace

extern "C" {

size_t sn_dirac_workspace_bytes(int64_t nV, int64_t nF) {
  if (nV < 0) nV = 0;
  if (nF < 0) nF = 0;
  const size_t scan = (size_t)((nV + 1 + kScanTile - 1) / kScanTile + 1) * sizeof(int);
  return align16((size_t)nF * sizeof(double)) + align16((size_t)nV * sizeof(double)) + align16((size_t)(nV + 1) * sizeof(int)) * 2 +
         align16((size_t)3 * nF * sizeof(int)) + align16(scan);
}

int sn_dirac_bsr4_from_mesh(const float *V, const int32_t *F, int64_t nV, int64_t nF, int32_t *di_rowptr,
                            int32_t *di_colind, float *di_vals, float *diat_vals, int32_t *dia_rowptr,
                            int32_t *dia_colind, float *dia_vals, float *dit_vals, void *workspace,
                            size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nV < 0 || nF < 0) return SN_E_SHAPE;
  if (4 * nV + 1 > INT_MAX || 12 * nF > INT_MAX) return SN_E_RANGE;
  if (!di_rowptr || !dia_rowptr) return SN_E_NULL;
  if (nF > 0 && (!V || !F || !di_colind || !di_vals || !diat_vals || !dia_colind || !dia_vals || !dit_vals)) return SN_E_NULL;
  if (workspace_bytes < sn_dirac_workspace_bytes(nV, nF) || !workspace) return SN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  char *w = static_cast<char *>(workspace);
  double *Af = reinterpret_cast<double *>(w); w += align16((size_t)nF * sizeof(double));
  double *Av = reinterpret_cast<double *>(w); w += align16((size_t)nV * sizeof(double));
  int *vptr = reinterpret_cast<int *>(w); w += align16((size_t)(nV + 1) * sizeof(int));
  int *cursor = reinterpret_cast<int *>(w); w += align16((size_t)(nV + 1) * sizeof(int));
  int *inc = reinterpret_cast<int *>(w); w += align16((size_t)3 * nF * sizeof(int));
  int *sums = reinterpret_cast<int *>(w);
  hipError_t e = sn_internal_fill(vptr, 0, (size_t)(nV + 1) * sizeof(int), s);
  if (e != hipSuccess) return (int)e;
  if (nF > 0) hipLaunchKernelGGL(face_area_k, dim3(grid_for(nF)), dim3(kWG), 0, s, V, F, nF, Af, vptr);
  const int64_t n = nV + 1;
  const int nblk = (int)((n + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(scan_sums_k, dim3(nblk), dim3(kWG), 0, s, vptr, n, sums);
  hipLaunchKernelGGL(scan_top_k, dim3(1), dim3(kWG), 0, s, sums, nblk);
  hipLaunchKernelGGL(scan_apply_k, dim3(nblk), dim3(kWG), 0, s, vptr, n, sums, vptr);
  e = sn_internal_copy2d(cursor, 0, vptr, 0, (int64_t)((size_t)(nV + 1) * sizeof(int)), 1, s);
  if (e != hipSuccess) return (int)e;
  if (nF > 0) hipLaunchKernelGGL(incidence_scatter_k, dim3(grid_for(nF)), dim3(kWG), 0, s, F, nF, cursor, inc);
  if (nV > 0) hipLaunchKernelGGL(vertex_lists_k, dim3(grid_for(nV)), dim3(kWG), 0, s, vptr, nV, inc, Af, Av, dia_colind);
  hipLaunchKernelGGL(di_fill_k, dim3(grid_for(nF + 1)), dim3(kWG), 0, s, V, F, nF, Af, Av, di_rowptr, di_colind, di_vals, diat_vals);
  hipLaunchKernelGGL(dia_fill_k, dim3(grid_for(nV + 1)), dim3(kWG), 0, s, V, F, nV, vptr, inc, Af, Av, dia_rowptr, dia_vals, dit_vals);
  return launch_status();
}

size_t sn_laplacian_workspace_bytes(int64_t nV, int64_t nF) { return sn_dirac_workspace_bytes(nV, nF); }

int sn_laplacian_csr_from_mesh(const float *V, const int32_t *F, int64_t nV, int64_t nF, int32_t phase,
                               int32_t *rowptr, int32_t *colind, float *vals, int32_t *status_flag,
                               void *workspace, size_t workspace_bytes, void *stream) {
  (void)hipGetLastError();      // a stale error left by an earlier runtime call of this thread is not ours to report
  if (nV < 0 || nF < 0 || (phase != 0 && phase != 1)) return SN_E_SHAPE;
  if (nV + 1 > INT_MAX || 3 * nF > INT_MAX) return SN_E_RANGE;
  if (!rowptr) return SN_E_NULL;
  if (nF > 0 && (!V || !F)) return SN_E_NULL;
  if (phase == 1 && nV > 0 && (!colind || !vals)) return SN_E_NULL;
  if (workspace_bytes < sn_laplacian_workspace_bytes(nV, nF) || !workspace) return SN_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  char *w = static_cast<char *>(workspace);
  double *Af = reinterpret_cast<double *>(w); w += align16((size_t)nF * sizeof(double));
  w += align16((size_t)nV * sizeof(double));                                       // (Av slot of the Dirac layout, unused)
  int *vptr = reinterpret_cast<int *>(w); w += align16((size_t)(nV + 1) * sizeof(int));
  int *cursor = reinterpret_cast<int *>(w); w += align16((size_t)(nV + 1) * sizeof(int));
  int *inc = reinterpret_cast<int *>(w); w += align16((size_t)3 * nF * sizeof(int));
  int *sums = reinterpret_cast<int *>(w);
  const int64_t n = nV + 1;
  const int nblk = (int)((n + kScanTile - 1) / kScanTile);
  if (phase == 0) {
    hipError_t e = sn_internal_fill(vptr, 0, (size_t)(nV + 1) * sizeof(int), s);
    if (e != hipSuccess) return (int)e;
    if (status_flag) {
      e = sn_internal_fill(status_flag, 0, sizeof(int), s);
      if (e != hipSuccess) return (int)e;
    }
    if (nF > 0) hipLaunchKernelGGL(face_area_k, dim3(grid_for(nF)), dim3(kWG), 0, s, V, F, nF, Af, vptr);
    hipLaunchKernelGGL(scan_sums_k, dim3(nblk), dim3(kWG), 0, s, vptr, n, sums);
    hipLaunchKernelGGL(scan_top_k, dim3(1), dim3(kWG), 0, s, sums, nblk);
    hipLaunchKernelGGL(scan_apply_k, dim3(nblk), dim3(kWG), 0, s, vptr, n, sums, vptr);
    e = sn_internal_copy2d(cursor, 0, vptr, 0, (int64_t)((size_t)(nV + 1) * sizeof(int)), 1, s);
    if (e != hipSuccess) return (int)e;
    if (nF > 0) hipLaunchKernelGGL(incidence_scatter_k, dim3(grid_for(nF)), dim3(kWG), 0, s, F, nF, cursor, inc);
    if (nV > 0) hipLaunchKernelGGL(vertex_sort_k, dim3(grid_for(nV)), dim3(kWG), 0, s, vptr, nV, inc);
    e = sn_internal_fill(rowptr + nV, 0, sizeof(int), s);
    if (e != hipSuccess) return (int)e;
    if (nV > 0)
      hipLaunchKernelGGL((laplacian_rows_k<false>), dim3(grid_for(nV)), dim3(kWG), 0, s, V, F, nV, vptr, inc, Af, rowptr,
                         (int *)nullptr, (float *)nullptr, status_flag);
    hipLaunchKernelGGL(scan_sums_k, dim3(nblk), dim3(kWG), 0, s, rowptr, n, sums);
    hipLaunchKernelGGL(scan_top_k, dim3(1), dim3(kWG), 0, s, sums, nblk);
    hipLaunchKernelGGL(scan_apply_k, dim3(nblk), dim3(kWG), 0, s, rowptr, n, sums, rowptr);
  } else if (nV > 0) {
    hipLaunchKernelGGL((laplacian_rows_k<true>), dim3(grid_for(nV)), dim3(kWG), 0, s, V, F, nV, vptr, inc, Af, rowptr, colind,
                       vals, (int *)nullptr);
  }
  return launch_status();
}

}  // extern "C"
