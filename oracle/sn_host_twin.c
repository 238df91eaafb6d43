/* sn_host_twin.c — HOST-POINTER TWINS of the C-ABI entry points of include/sn_spmm.h ("plus a host-pointer CPU
 * twin of each for GPU-less tests").  TEST INFRASTRUCTURE ONLY: built into oracle/libsn_oracle.so, used by tests/ to exercise
 * the ABI contract (argument validation order, status codes, layouts, "every output element is written") in a container
 * without a GPU, and — in the -m gpu suite — to show that the device library returns the same status for the same invalid
 * call.  The product (surfacenetworks_amd/) never loads this file: it has no CPU path.
 *
 * Every sn_host_X has the signature of sn_X (the `stream` argument is accepted and ignored; pointers are host pointers) and
 * the same checks in the same order as the device entry point in surfacenetworks_amd/csrc/sn_kernels.hip.  The arithmetic
 * is the oracle's (k-ascending fmaf chain, sn_oracle.c), so results are also bit-identical to the device kernels. */
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sn_spmm.h"

static int fits_i32(int64_t v) { return v >= 0 && v <= (int64_t)INT_MAX; }
static int aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

static int check_dense(const float *P, int64_t ld, int group, int N) {
  if (!P) return SN_E_NULL;
  if (group != 1 && group != 4) return SN_E_LD;
  if (group == 1 && ld < N) return SN_E_LD;
  if (group == 4 && ld < 4 * (int64_t)N) return SN_E_LD;
  return SN_OK;
}
static int64_t row_off(int64_t r, int64_t ld, int group, int N) {
  return group == 1 ? r * ld : (r / 4) * ld + (r % 4) * (int64_t)N;
}

int sn_host_spmm_csr_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                         const float *X, int64_t ldx, int32_t x_group, int32_t N, float *Y, int64_t ldy, int32_t y_group,
                         void *stream) {
  (void)stream;
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
  for (int64_t r = 0; r < M; ++r) {
    float *y = Y + row_off(r, ldy, y_group, N);
    for (int j = 0; j < N; ++j) {
      float acc = 0.f;
      for (int k = rowptr[r]; k < rowptr[r + 1]; ++k)
        acc = fmaf(vals[k], X[row_off(colind[k], ldx, x_group, N) + j], acc);
      y[j] = acc;
    }
  }
  return SN_OK;
}

/* sn_spmm_csr_ring_f32 (the sliding-window Laplacian product): the same k-ordered FMA chain as sn_spmm_csr_f32 on a square
 * operator at N = 64 | 128, contiguous rows; the window is an implementation detail of the device kernel, the result is not. */
#define SN_HOST_RING_R 64
#define SN_HOST_RING_H 160
int sn_host_spmm_csr_ring_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                              const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, void *stream) {
  (void)stream;
  if (M < 0 || K < 0 || nnz < 0 || N < 1) return SN_E_SHAPE;
  if (!fits_i32(M + SN_HOST_RING_R + 1) || !fits_i32(K) || !fits_i32(nnz)) return SN_E_RANGE;
  if (M != K) return SN_E_UNSUPPORTED;
  if (N != 64 && N != 128) return SN_E_UNSUPPORTED;
  if (M == 0) return SN_OK;
  if (!rowptr || (nnz > 0 && (!colind || !vals))) return SN_E_NULL;
  int st = check_dense(Y, ldy, 1, N);
  if (!st) st = check_dense(X, ldx, 1, N);
  if (st) return st;
  if (!aligned16(X) || !aligned16(Y) || ldx % 4 || ldy % 4) return SN_E_ALIGN;
  for (int64_t r = 0; r < M; ++r)
    for (int j = 0; j < N; ++j) {
      float acc = 0.f;
      for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) acc = fmaf(vals[k], X[(int64_t)colind[k] * ldx + j], acc);
      Y[r * ldy + j] = acc;
    }
  return SN_OK;
}

/* sn_csr_band_i32: out = { max |column - row|, longest row (INT32_MAX when a row's columns do not ascend), rows reaching past the half window } */
int sn_host_csr_band_i32(const int32_t *rowptr, const int32_t *colind, int64_t M, int64_t K, int32_t *out, void *stream) {
  (void)stream;
  if (M < 0 || K < 0) return SN_E_SHAPE;
  if (!fits_i32(M + 1) || !fits_i32(K)) return SN_E_RANGE;
  if (!out || (M > 0 && !rowptr)) return SN_E_NULL;
  out[0] = out[1] = out[2] = 0;
  if (M == 0) return SN_OK;
  if (!colind) return SN_E_NULL;
  for (int64_t r = 0; r < M; ++r) {
    const int kb = rowptr[r], ke = rowptr[r + 1];
    if (ke > kb) {
      int cmin = colind[kb], cmax = cmin, asc = 1;
      for (int k = kb + 1; k < ke; ++k) {          /* a row whose columns do not ascend strictly reports INT32_MAX as its length */
        if (colind[k] <= colind[k - 1]) asc = 0;
        if (colind[k] < cmin) cmin = colind[k];
        if (colind[k] > cmax) cmax = colind[k];
      }
      const int lo = (int)r - cmin, hi = cmax - (int)r;
      const int far = lo > hi ? lo : hi;
      const int len = asc ? ke - kb : 0x7fffffff;
      if (far > out[0]) out[0] = far;
      if (len > out[1]) out[1] = len;
      if (far > SN_HOST_RING_H) out[2] += 1;
    }
  }
  return SN_OK;
}

int sn_host_spmm_bsr4_f32(const int32_t *b_rowptr, const int32_t *b_colind, const float *b_vals, int64_t Mb, int64_t Kb,
                          int64_t nblocks, const float *X, int64_t ldx, int32_t x_group, int32_t N, float *Y, int64_t ldy,
                          int32_t y_group, void *stream) {
  (void)stream;
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
  for (int64_t br = 0; br < Mb; ++br)
    for (int q = 0; q < 4; ++q) {
      float *y = Y + row_off(4 * br + q, ldy, y_group, N);
      for (int j = 0; j < N; ++j) {
        float acc = 0.f;
        for (int k = b_rowptr[br]; k < b_rowptr[br + 1]; ++k)
          for (int c = 0; c < 4; ++c)
            acc = fmaf(b_vals[16 * (int64_t)k + 4 * q + c], X[row_off(4 * (int64_t)b_colind[k] + c, ldx, x_group, N) + j], acc);
        y[j] = acc;
      }
    }
  return SN_OK;
}

int sn_host_spmm_q3_f32(const int32_t *b_rowptr, const float *q_blk, int64_t Mb, int64_t Kb, int64_t nblocks, const float *X,
                        int64_t ldx, int32_t x_group, int32_t N, float *Y, int64_t ldy, int32_t y_group, void *stream) {
  (void)stream;
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
  for (int64_t br = 0; br < Mb; ++br)
    for (int j = 0; j < N; ++j) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int k = b_rowptr[br]; k < b_rowptr[br + 1]; ++k) {
        const float p1 = q_blk[4 * (int64_t)k], p2 = q_blk[4 * (int64_t)k + 1], p3 = q_blk[4 * (int64_t)k + 2];
        int32_t bc;
        memcpy(&bc, &q_blk[4 * (int64_t)k + 3], 4);
        const float x0 = X[row_off(4 * (int64_t)bc, ldx, x_group, N) + j], x1 = X[row_off(4 * (int64_t)bc + 1, ldx, x_group, N) + j],
                    x2 = X[row_off(4 * (int64_t)bc + 2, ldx, x_group, N) + j], x3 = X[row_off(4 * (int64_t)bc + 3, ldx, x_group, N) + j];
        a0 = fmaf(p1, x1, a0);  a0 = fmaf(p2, x2, a0);  a0 = fmaf(p3, x3, a0);       /* M(p), see sn_spmm.h */
        a1 = fmaf(-p1, x0, a1); a1 = fmaf(p3, x2, a1);  a1 = fmaf(-p2, x3, a1);
        a2 = fmaf(-p2, x0, a2); a2 = fmaf(-p3, x1, a2); a2 = fmaf(p1, x3, a2);
        a3 = fmaf(-p3, x0, a3); a3 = fmaf(p2, x1, a3);  a3 = fmaf(-p1, x2, a3);
      }
      Y[row_off(4 * br, ldy, y_group, N) + j] = a0;
      Y[row_off(4 * br + 1, ldy, y_group, N) + j] = a1;
      Y[row_off(4 * br + 2, ldy, y_group, N) + j] = a2;
      Y[row_off(4 * br + 3, ldy, y_group, N) + j] = a3;
    }
  return SN_OK;
}

int sn_host_coo_to_csr_i32(const int64_t *idx_batch, const int64_t *idx_row, const int64_t *idx_col, int64_t nnz, int64_t B,
                           int64_t R, int64_t Kb, int32_t *rowptr, int32_t *colind, void *stream) {
  (void)stream;
  if (nnz < 0 || B < 1 || R < 0 || Kb < 0) return SN_E_SHAPE;
  const int64_t M = B * R;
  if (!fits_i32(M + 1) || !fits_i32(B * Kb) || !fits_i32(nnz)) return SN_E_RANGE;
  if (!rowptr || (nnz > 0 && (!idx_row || !idx_col || !colind))) return SN_E_NULL;
  int64_t k = 0;
  for (int64_t r = 0; r <= M; ++r) {                 /* rowptr[r] = first k with key(k) >= r: interior empty rows come out right */
    while (k < nnz && (idx_batch ? idx_batch[k] * R : 0) + idx_row[k] < r) ++k;
    rowptr[r] = (int32_t)k;
  }
  for (int64_t t = 0; t < nnz; ++t) colind[t] = (int32_t)((idx_batch ? idx_batch[t] * Kb : 0) + idx_col[t]);
  return SN_OK;
}

size_t sn_host_csr_transpose_workspace_bytes(int64_t M, int64_t K, int64_t nnz) {
  (void)M; (void)nnz;
  if (K < 0) K = 0;
  return ((size_t)K * sizeof(int) + 15) & ~(size_t)15;
}

int sn_host_csr_transpose_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K,
                              int64_t nnz, int32_t *t_rowptr, int32_t *t_colind, float *t_vals, void *workspace,
                              size_t workspace_bytes, void *stream) {
  (void)stream;
  if (M < 0 || K < 0 || nnz < 0) return SN_E_SHAPE;
  if (!fits_i32(M + 1) || !fits_i32(K + 1) || !fits_i32(nnz)) return SN_E_RANGE;
  if (!t_rowptr || !rowptr) return SN_E_NULL;
  if (nnz > 0 && (!colind || !vals || !t_colind || !t_vals)) return SN_E_NULL;
  if (workspace_bytes < sn_host_csr_transpose_workspace_bytes(M, K, nnz) || (!workspace && K > 0)) return SN_E_WORKSPACE;
  memset(t_rowptr, 0, (size_t)(K + 1) * sizeof(int32_t));
  if (nnz == 0 || K == 0) return SN_OK;
  for (int64_t k = 0; k < nnz; ++k) t_rowptr[colind[k] + 1]++;
  for (int64_t c = 0; c < K; ++c) t_rowptr[c + 1] += t_rowptr[c];
  int32_t *cursor = (int32_t *)workspace;
  memcpy(cursor, t_rowptr, (size_t)K * sizeof(int32_t));
  for (int64_t r = 0; r < M; ++r)                     /* row-major sweep: every output row comes out sorted by column */
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int pos = cursor[colind[k]]++;
      t_colind[pos] = (int32_t)r;
      t_vals[pos] = vals[k];
    }
  return SN_OK;
}

static int concat_common(const int32_t *pool_rowptr, const int32_t *pool_colind, const float *pool_vals, const int64_t *desc,
                         int64_t B, int stride, int64_t total_rows, int64_t total, int32_t vpe, int32_t *out_rowptr,
                         int32_t *out_colind, float *out_vals, int64_t size0, int64_t size1) {
  for (int64_t b = 0; b < B; ++b) {
    const int64_t *d = desc + stride * b;
    const int64_t row0 = stride == 6 ? d[4] : b * size0;
    const int64_t row1 = stride == 6 ? (b + 1 < B ? desc[stride * (b + 1) + 4] : total_rows) : (b + 1) * size0;
    const int64_t shift = stride == 6 ? d[5] : b * size1;
    for (int64_t i = row0; i < row1; ++i) {
      const int64_t r = i - row0;
      out_rowptr[i] = (int32_t)(d[3] + pool_rowptr[d[0] + (r < d[2] ? r : d[2])]);
    }
    const int64_t cnt = pool_rowptr[d[0] + d[2]];
    for (int64_t k = 0; k < cnt; ++k) {
      const int64_t src = d[1] + k, dst = d[3] + k;
      if (vpe == 4) {
        memcpy(out_vals + 4 * dst, pool_vals + 4 * src, 16);
        int32_t c;
        memcpy(&c, pool_vals + 4 * src + 3, 4);
        c += (int32_t)shift;
        memcpy(out_vals + 4 * dst + 3, &c, 4);
      } else {
        out_colind[dst] = pool_colind[src] + (int32_t)shift;
        memcpy(out_vals + (int64_t)vpe * dst, pool_vals + (int64_t)vpe * src, (size_t)vpe * 4);
      }
    }
  }
  out_rowptr[total_rows] = (int32_t)total;
  return SN_OK;
}

int sn_host_blockdiag_concat_i32(const int32_t *pool_rowptr, const int32_t *pool_colind, const float *pool_vals,
                                 const int64_t *desc, int64_t B, int64_t size0, int64_t size1, int64_t total,
                                 int32_t vals_per_entry, int32_t *out_rowptr, int32_t *out_colind, float *out_vals, void *stream) {
  (void)stream;
  if (B < 0 || size0 < 0 || size1 < 0 || total < 0) return SN_E_SHAPE;
  if (vals_per_entry != 1 && vals_per_entry != 16 && vals_per_entry != 4) return SN_E_UNSUPPORTED;
  if (!fits_i32(B * size0 + 1) || !fits_i32(B * size1) || !fits_i32(total)) return SN_E_RANGE;
  if (!out_rowptr) return SN_E_NULL;
  if (B > 0 && (!desc || !pool_rowptr)) return SN_E_NULL;
  if (total > 0 && (!pool_vals || !out_vals || (vals_per_entry != 4 && (!pool_colind || !out_colind)))) return SN_E_NULL;
  if (vals_per_entry != 1 && total > 0 && (!aligned16(pool_vals) || !aligned16(out_vals))) return SN_E_ALIGN;
  return concat_common(pool_rowptr, pool_colind, pool_vals, desc, B, 4, B * size0, total, vals_per_entry, out_rowptr, out_colind,
                       out_vals, size0, size1);
}

int sn_host_blockdiag_concat_ragged_i32(const int32_t *pool_rowptr, const int32_t *pool_colind, const float *pool_vals,
                                        const int64_t *desc, int64_t B, int64_t total_rows, int64_t total_cols, int64_t total,
                                        int32_t vals_per_entry, int32_t *out_rowptr, int32_t *out_colind, float *out_vals,
                                        void *stream) {
  (void)stream;
  if (B < 0 || total_rows < 0 || total_cols < 0 || total < 0) return SN_E_SHAPE;
  if (vals_per_entry != 1 && vals_per_entry != 16 && vals_per_entry != 4) return SN_E_UNSUPPORTED;
  if (!fits_i32(total_rows + 1) || !fits_i32(total_cols) || !fits_i32(total)) return SN_E_RANGE;
  if (!out_rowptr) return SN_E_NULL;
  if (B > 0 && (!desc || !pool_rowptr)) return SN_E_NULL;
  if (B == 0 && (total_rows > 0 || total > 0)) return SN_E_SHAPE;
  if (total > 0 && (!pool_vals || !out_vals || (vals_per_entry != 4 && (!pool_colind || !out_colind)))) return SN_E_NULL;
  if (vals_per_entry != 1 && total > 0 && (!aligned16(pool_vals) || !aligned16(out_vals))) return SN_E_ALIGN;
  if (B == 0) {
    out_rowptr[0] = 0;
    return SN_OK;
  }
  return concat_common(pool_rowptr, pool_colind, pool_vals, desc, B, 6, total_rows, total, vals_per_entry, out_rowptr, out_colind,
                       out_vals, 0, 0);
}

int sn_host_validate_csr_i32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                             int32_t *flags, void *stream) {
  (void)stream;
  if (M < 0 || K < 0 || nnz < 0) return SN_E_SHAPE;
  if (!fits_i32(M + 1) || !fits_i32(K) || !fits_i32(nnz)) return SN_E_RANGE;
  if (!flags) return SN_E_NULL;
  *flags = 0;
  if (M == 0) return SN_OK;
  if (!rowptr || (nnz > 0 && !colind)) return SN_E_NULL;
  int bad = 0;
  for (int64_t r = 0; r < M; ++r) {
    const int b = rowptr[r], e = rowptr[r + 1];
    if (r == 0 && b != 0) bad |= 1;
    if (e < b) bad |= 2;
    if (r == M - 1 && e != (int)nnz) bad |= 4;
    if (b < 0 || e > nnz || e < b) continue;
    int prev = -1;
    for (int k = b; k < e; ++k) {
      const int c = colind[k];
      if (c < 0 || c >= K) bad |= 8;
      if (c <= prev) bad |= 16;
      prev = c;
      if (vals && !isfinite(vals[k])) bad |= 32;
    }
  }
  *flags = bad;
  return SN_OK;
}

int sn_host_rb4_count(const int32_t *rowptr, const int32_t *colind, int64_t M, int64_t K, int32_t *b_ptr, void *workspace,
                      size_t workspace_bytes, void *stream) {
  (void)stream; (void)workspace; (void)workspace_bytes;
  if (M < 0 || K < 0) return SN_E_SHAPE;
  if (!fits_i32(M + 4) || !fits_i32(K)) return SN_E_RANGE;
  if (!rowptr || !b_ptr) return SN_E_NULL;
  const int64_t Mb = (M + 3) / 4;
  b_ptr[0] = 0;
  for (int64_t br = 0; br < Mb; ++br) {
    int p[4], e[4], n = 0;
    for (int q = 0; q < 4; ++q) {
      const int64_t r = 4 * br + q;
      p[q] = r < M ? rowptr[r] : 0;
      e[q] = r < M ? rowptr[r + 1] : 0;
    }
    for (;;) {
      int cur = INT_MAX;
      for (int q = 0; q < 4; ++q)
        if (p[q] < e[q] && colind[p[q]] < cur) cur = colind[p[q]];
      if (cur == INT_MAX) break;
      for (int q = 0; q < 4; ++q)
        if (p[q] < e[q] && colind[p[q]] == cur) ++p[q];
      ++n;
    }
    b_ptr[br + 1] = b_ptr[br] + n;
  }
  return SN_OK;
}

int sn_host_rb4_fill(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, const int32_t *b_ptr,
                     int32_t *b_col, float *b_val, void *stream) {
  (void)stream;
  if (M < 0 || K < 0) return SN_E_SHAPE;
  if (!rowptr || !b_ptr) return SN_E_NULL;
  const int64_t Mb = (M + 3) / 4;
  if (Mb == 0) return SN_OK;
  if (!colind || !vals || !b_col || !b_val) return SN_E_NULL;
  if (!aligned16(b_val)) return SN_E_ALIGN;
  for (int64_t br = 0; br < Mb; ++br) {
    int p[4], e[4], out = b_ptr[br];
    for (int q = 0; q < 4; ++q) {
      const int64_t r = 4 * br + q;
      p[q] = r < M ? rowptr[r] : 0;
      e[q] = r < M ? rowptr[r + 1] : 0;
    }
    for (;;) {
      int cur = INT_MAX;
      for (int q = 0; q < 4; ++q)
        if (p[q] < e[q] && colind[p[q]] < cur) cur = colind[p[q]];
      if (cur == INT_MAX) break;
      b_col[out] = cur;
      for (int q = 0; q < 4; ++q) {
        b_val[4 * (int64_t)out + q] = 0.f;
        if (p[q] < e[q] && colind[p[q]] == cur) b_val[4 * (int64_t)out + q] = vals[p[q]++];
      }
      ++out;
    }
  }
  return SN_OK;
}

int sn_host_spmm_rb4_f32(const int32_t *b_ptr, const int32_t *b_col, const float *b_val, int64_t M, int64_t K, int64_t capacity,
                         const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, void *stream) {
  (void)stream;
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
  for (int64_t r = 0; r < M; ++r)
    for (int j = 0; j < N; ++j) {
      float acc = 0.f;
      for (int k = b_ptr[r / 4]; k < b_ptr[r / 4 + 1]; ++k) acc = fmaf(b_val[4 * (int64_t)k + (r & 3)], X[(int64_t)b_col[k] * ldx + j], acc);
      Y[r * ldy + j] = acc;
    }
  return SN_OK;
}

int sn_host_elu_into_f32(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t rows, int32_t C, void *stream) {
  (void)stream;
  if (rows < 0 || C < 1 || lds < C || ldd < C) return SN_E_SHAPE;
  if (rows == 0) return SN_OK;
  if (!src || !dst) return SN_E_NULL;
  for (int64_t r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) {
      const float x = src[r * lds + c];
      dst[r * ldd + c] = x > 0.f ? x : expm1f(x);
    }
  return SN_OK;
}
