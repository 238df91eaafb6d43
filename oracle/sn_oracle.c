/*
 * sn_oracle.c — TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, host pointers) of the
 * reference algorithms on the Surface-Network SpMM hot path, used as the parity checker for the
 * HIP kernels.  Nothing under surfacenetworks_amd/ may import, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Pinning: the reference ships no golden vectors for this path.  This restatement is
 * pinned against outputs of the reference itself, generated in the build container by importing
 * /root/reference/src (tests/golden/make_golden.py -> the .npz fixtures under tests/golden): torch.mm(sparse, dense)
 * forward/backward, sparse_diag_cat, sparse_cat.  The two CUDA kernels cannot run here (cupy +
 * pynvrtc + NVIDIA driver); oracle_sparse_bmm / oracle_batch_csr restate them line by line and are
 * cross-checked against the torch.mm results on the same operators.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* --------------------------------------------------------------------------------------------
 * src/utils/cuda/sparse_bmm.cu:16-61
 *   C[b,i,j] = sum_{k = col_ptr[b,i]}^{col_ptr[b,i+1]-1} values[k] * dense[b, col_ind[k], j]
 * col_ptr is (B, R+1) int64 with GLOBAL nnz offsets; dense is (B,K,N) row-major; C is (B,R,N).
 * The kernel accumulates `value += s_values * dense` in k order (sparse_bmm.cu:49-55); nvcc
 * contracts that to an FMA, restated here with fmaf so that an FMA-based GPU kernel can be
 * compared bit for bit.
 * -------------------------------------------------------------------------------------------- */
void oracle_sparse_bmm(float *C, const float *values, const int64_t *col_ind, const int64_t *col_ptr,
                       const float *dense, int64_t B, int64_t R, int64_t K, int64_t N) {
  for (int64_t b = 0; b < B; ++b)
    for (int64_t i = 0; i < R; ++i) {
      const int64_t start = col_ptr[b * (R + 1) + i];      /* sparse_bmm.cu:36 */
      const int64_t end = col_ptr[b * (R + 1) + i + 1];    /* sparse_bmm.cu:37 */
      for (int64_t j = 0; j < N; ++j) {
        float value = 0.0f;                                 /* sparse_bmm.cu:35 */
        for (int64_t k = start; k < end; ++k)               /* sparse_bmm.cu:42-56 */
          value = fmaf(values[k], dense[b * K * N + col_ind[k] * N + j], value);
        C[b * R * N + i * N + j] = value;                   /* sparse_bmm.cu:58-60 */
      }
    }
}

/* --------------------------------------------------------------------------------------------
 * src/utils/cuda/batch_csr.cu:13-47, restated literally INCLUDING its defect: a row with no
 * entries in the interior of a batch keeps col_ptr == 0 (the launcher pre-fills zeros,
 * batch_csr.py:46-49), which makes the preceding row's end pointer 0.  Only valid when every
 * row up to the last non-empty one has at least one entry.  indices is (3, nnz) row-major.
 * -------------------------------------------------------------------------------------------- */
void oracle_batch_csr(const int64_t *indices, int64_t nnz, int64_t B, int64_t R, int64_t *col_ind,
                      int64_t *col_ptr) {
  memset(col_ptr, 0, (size_t)(B * (R + 1)) * sizeof(int64_t)); /* batch_csr.py:46-49 */
  for (int64_t ind = 0; ind < nnz; ++ind) {
    const int64_t batch_id = indices[ind];                   /* :18 */
    const int64_t row_id = indices[ind + nnz];               /* :19 */
    const int64_t col_id = indices[ind + 2 * nnz];           /* :20 */
    const int64_t prev_batch_id = ind > 0 ? indices[ind - 1] : -1;        /* :22-25 */
    const int64_t prev_row_id = ind > 0 ? indices[ind - 1 + nnz] : -1;    /* :27-30 */
    col_ind[ind] = col_id;                                   /* :32-34 */
    if (batch_id != prev_batch_id || row_id != prev_row_id) {             /* :36 */
      col_ptr[batch_id * (R + 1) + row_id] = ind;            /* :37 */
      if (batch_id > 0 && row_id == 0)                       /* :39-41 */
        col_ptr[prev_batch_id * (R + 1) + prev_row_id + 1] = ind;
    }
    if (ind + 1 == nnz) col_ptr[batch_id * (R + 1) + row_id + 1] = ind + 1; /* :44-46 */
  }
}

/* --------------------------------------------------------------------------------------------
 * The SPECIFIED behaviour of the replacement (include/sn_spmm.h: sn_coo_to_csr_i32): sorted COO
 * (batch,row,col) -> block-diagonal CSR, global row = b*R + r, global col = b*Kb + c, correct for
 * empty rows anywhere.  Equals what ATen's COO->CSR does inside torch.mm(sparse, dense)
 * (call sites src/utils/utils_pt.py:167,176,202,214) and equals oracle_batch_csr wherever that one
 * is valid.  idx_batch may be NULL.
 * -------------------------------------------------------------------------------------------- */
void oracle_coo_to_csr_i32(const int64_t *idx_batch, const int64_t *idx_row, const int64_t *idx_col,
                           int64_t nnz, int64_t B, int64_t R, int64_t Kb, int32_t *rowptr,
                           int32_t *colind) {
  const int64_t M = B * R;
  memset(rowptr, 0, (size_t)(M + 1) * sizeof(int32_t));
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t b = idx_batch ? idx_batch[k] : 0;
    rowptr[b * R + idx_row[k] + 1] += 1;
    colind[k] = (int32_t)(b * Kb + idx_col[k]);
  }
  for (int64_t i = 0; i < M; ++i) rowptr[i + 1] += rowptr[i];
}

/* dense-row addressing of include/sn_spmm.h ("group" layout) */
static inline int64_t row_off(int64_t r, int64_t ld, int group, int N) {
  return (r / group) * ld + (r % group) * (int64_t)N;
}

/* --------------------------------------------------------------------------------------------
 * Y = A·X on the block-diagonal CSR (int32), same arithmetic as oracle_sparse_bmm
 * (sparse_bmm.cu:42-56: k-ascending FMA chain), with the (ld, group) addressing of the C-ABI so
 * the quaternion view x.view(B*V*4, C/4) (src/utils/utils_pt.py:201,213) can live inside a concat
 * buffer.  This is the function the HIP kernels are compared against bit for bit.
 * -------------------------------------------------------------------------------------------- */
void oracle_spmm_csr_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M,
                         const float *X, int64_t ldx, int xg, int N, float *Y, int64_t ldy, int yg) {
  for (int64_t r = 0; r < M; ++r) {
    float *y = Y + row_off(r, ldy, yg, N);
    for (int j = 0; j < N; ++j) {
      float acc = 0.0f;
      for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
        acc = fmaf(vals[k], X[row_off(colind[k], ldx, xg, N) + j], acc);
      y[j] = acc;
    }
  }
}

/* fp64 ground truth of the same product (products and sums in double, result left in double). */
void oracle_spmm_csr_f64(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M,
                         const float *X, int64_t ldx, int xg, int N, double *Y /* M x N contiguous */) {
  for (int64_t r = 0; r < M; ++r)
    for (int j = 0; j < N; ++j) {
      double acc = 0.0;
      for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
        acc += (double)vals[k] * (double)X[row_off(colind[k], ldx, xg, N) + j];
      Y[r * (int64_t)N + j] = acc;
    }
}

/* --------------------------------------------------------------------------------------------
 * CSR of A^T, entries of each output row ordered by column — the coalesced form that
 * matrix1.transpose(2,1).coalesce() + batch_csr produce on every backward
 * (src/utils/cuda/sparse_bmm_func.py:66-67).
 * -------------------------------------------------------------------------------------------- */
void oracle_csr_transpose_f32(const int32_t *rowptr, const int32_t *colind, const float *vals,
                              int64_t M, int64_t K, int32_t *t_rowptr, int32_t *t_colind,
                              float *t_vals) {
  memset(t_rowptr, 0, (size_t)(K + 1) * sizeof(int32_t));
  const int64_t nnz = rowptr[M];
  for (int64_t k = 0; k < nnz; ++k) t_rowptr[colind[k] + 1] += 1;
  for (int64_t c = 0; c < K; ++c) t_rowptr[c + 1] += t_rowptr[c];
  int32_t *cursor = (int32_t *)malloc((size_t)(K > 0 ? K : 1) * sizeof(int32_t));
  memcpy(cursor, t_rowptr, (size_t)K * sizeof(int32_t));
  for (int64_t r = 0; r < M; ++r) /* row-ascending scan => each output row sorted by column */
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const int32_t pos = cursor[colind[k]]++;
      t_colind[pos] = (int32_t)r;
      t_vals[pos] = vals[k];
    }
  free(cursor);
}

/* --------------------------------------------------------------------------------------------
 * sparse_diag_cat (src/utils/utils_pt.py:41-53): block i gets indices + i*[size0; size1], blocks are
 * concatenated and coalesced; every block is padded to (size0,size1).  Here on per-mesh CSR inputs
 * with the descriptor table of include/sn_spmm.h (sn_blockdiag_concat_i32).
 * -------------------------------------------------------------------------------------------- */
void oracle_blockdiag_concat_i32(const int32_t *pool_rowptr, const int32_t *pool_colind,
                                 const float *pool_vals, const int64_t *desc, int64_t B, int64_t size0,
                                 int64_t size1, int64_t total, int vpe, int32_t *out_rowptr,
                                 int32_t *out_colind, float *out_vals) {
  for (int64_t b = 0; b < B; ++b) {
    const int64_t *d = desc + 4 * b;
    const int64_t rows = d[2], base = d[3];
    for (int64_t r = 0; r < size0; ++r)
      out_rowptr[b * size0 + r] = (int32_t)(base + pool_rowptr[d[0] + (r < rows ? r : rows)]);
    const int64_t cnt = pool_rowptr[d[0] + rows];
    for (int64_t k = 0; k < cnt; ++k) {
      out_colind[base + k] = pool_colind[d[1] + k] + (int32_t)(b * size1);
      memcpy(out_vals + (base + k) * vpe, pool_vals + (d[1] + k) * vpe, (size_t)vpe * sizeof(float));
    }
  }
  out_rowptr[B * size0] = (int32_t)total;
}

/* --------------------------------------------------------------------------------------------
 * CSR -> 4x4-block BSR (packed Dirac form; every block of Di is -Q(0,e)/(2Af), src/utils/mesh.py:28-33,55-58).
 * Two passes like the C-ABI: count (returns nblocks, fills b_rowptr) then fill.
 * -------------------------------------------------------------------------------------------- */
int64_t oracle_bsr4_count(const int32_t *rowptr, const int32_t *colind, int64_t M, int32_t *b_rowptr) {
  const int64_t Mb = M / 4;
  int64_t total = 0;
  for (int64_t br = 0; br < Mb; ++br) {
    b_rowptr[br] = (int32_t)total;
    int32_t p[4], e[4];
    for (int q = 0; q < 4; ++q) { p[q] = rowptr[4 * br + q]; e[q] = rowptr[4 * br + q + 1]; }
    for (;;) {
      int32_t cur = INT32_MAX;
      for (int q = 0; q < 4; ++q)
        if (p[q] < e[q] && (colind[p[q]] >> 2) < cur) cur = colind[p[q]] >> 2;
      if (cur == INT32_MAX) break;
      for (int q = 0; q < 4; ++q)
        while (p[q] < e[q] && (colind[p[q]] >> 2) == cur) ++p[q];
      ++total;
    }
  }
  b_rowptr[Mb] = (int32_t)total;
  return total;
}

void oracle_bsr4_fill(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M,
                      const int32_t *b_rowptr, int32_t *b_colind, float *b_vals) {
  const int64_t Mb = M / 4;
  for (int64_t br = 0; br < Mb; ++br) {
    int32_t out = b_rowptr[br];
    int32_t p[4], e[4];
    for (int q = 0; q < 4; ++q) { p[q] = rowptr[4 * br + q]; e[q] = rowptr[4 * br + q + 1]; }
    for (;;) {
      int32_t cur = INT32_MAX;
      for (int q = 0; q < 4; ++q)
        if (p[q] < e[q] && (colind[p[q]] >> 2) < cur) cur = colind[p[q]] >> 2;
      if (cur == INT32_MAX) break;
      b_colind[out] = cur;
      float *blk = b_vals + 16 * (int64_t)out;
      memset(blk, 0, 16 * sizeof(float));
      for (int q = 0; q < 4; ++q)
        while (p[q] < e[q] && (colind[p[q]] >> 2) == cur) {
          blk[q * 4 + (colind[p[q]] & 3)] = vals[p[q]];
          ++p[q];
        }
      ++out;
    }
  }
}

/* F.elu with alpha = 1 (src/utils/utils_pt.py:161,171,195,208) and its derivative written in terms
 * of the output, as autograd does: d/dx = 1 (x > 0) | out + 1 (x <= 0). */
void oracle_elu(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t rows, int C) {
  for (int64_t r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) {
      const float x = src[r * lds + c];
      dst[r * ldd + c] = x > 0.0f ? x : expm1f(x);
    }
}

void oracle_elu_bwd(const float *gdst, int64_t ldg, const float *gdst2, int64_t ldg2, const float *gadd, int64_t ldga,
                    const float *out, int64_t ldo, float *gsrc, int64_t ldgs, int64_t rows, int C, int accumulate) {
  for (int64_t r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) {
      const float o = out[r * ldo + c];
      const float d = (gdst[r * ldg + c] + (gdst2 ? gdst2[r * ldg2 + c] : 0.0f)) * (o > 0.0f ? 1.0f : o + 1.0f) +
                      (gadd ? gadd[r * ldga + c] : 0.0f);
      float *p = gsrc + r * ldgs + c;
      *p = accumulate ? *p + d : d;
    }
}

/* --------------------------------------------------------------------------------------------
 * Checkers for the BatchNorm+Linear helpers (include/sn_spmm.h): the reductions nn.BatchNorm1d / nn.Linear perform
 * inside GraphConv1x1 (src/utils/utils_pt.py:83-99), in double precision.
 * -------------------------------------------------------------------------------------------- */
void oracle_colstats(const float *x, int64_t ld, int64_t rows, int C, double *out /* [2*C] */) {
  for (int c = 0; c < 2 * C; ++c) out[c] = 0.0;
  for (int64_t r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) {
      const double v = x[r * ld + c];
      out[c] += v;
      out[C + c] += v * v;
    }
}

void oracle_wgrad(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows, int J,
                  int C, double *G /* J x C */) {
  for (int64_t i = 0; i < (int64_t)J * C; ++i) G[i] = 0.0;
  for (int64_t r = 0; r < rows; ++r)
    for (int j = 0; j < J; ++j) {
      const double d = dy[r * lddy + j];
      for (int c = 0; c < C; ++c) G[(int64_t)j * C + c] += d * (double)(x[r * ldx + c] - (center ? center[c] : 0.0f));
    }
}

void oracle_affine_cols_acc(float *dx, int64_t lddx, const float *x, int64_t ldx, const float *center, const float *B,
                            const float *Cc, int64_t rows, int C) {
  for (int64_t r = 0; r < rows; ++r)
    for (int c = 0; c < C; ++c) dx[r * lddx + c] += fmaf(x[r * ldx + c] - (center ? center[c] : 0.0f), B[c], Cc[c]);
}
