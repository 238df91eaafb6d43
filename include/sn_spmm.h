/*
 * sn_spmm.h — C-ABI of the MI355X-native Surface-Network sparse-operator layer.
 *
 * This is the drop-in boundary for the ONE hot path of jiangzhongshi/SurfaceNetworks:
 * the sparse-operator x dense-feature product inside the Lap/Dirac ResNet block and the
 * operator-format plumbing around it.  Every entry point below names the reference
 * interface it replaces (paths relative to the reference checkout, file:line).
 *
 * Conventions (all entry points)
 *   - Every pointer is a DEVICE pointer (HBM) unless its name ends in `_host`.
 *   - The library never allocates, frees or synchronises: the caller owns inputs, outputs
 *     and workspaces (reference: launchers allocate with values.new(), sparse_bmm.py:53).
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous and stream-ordered
 *     (reference: launch on torch.cuda.current_stream(), sparse_bmm.py:59, batch_csr.py:56).
 *   - No per-operator or per-shape caches (the reference keeps module-level kernel / handle caches, sparse_bmm_func.py:20-21,
 *     sparse_bmm.py:26,63 — deliberately not kept); re-entrant; safe from several host threads on different streams.
 *     The ONLY process-global state: (a) environment switches, read ONCE per process — the complete list, asserted against the
 *     sources by tests/test_boundary.py::test_environment_switches_are_the_documented_ones:
 *       library  SN_GEMM_VARIANT      0 = the fp32-MFMA kernels (gemm_rows_k, wgrad_mfma_k): the ONE A/B baseline of the Linear
 *                                     kernels (exact fp32 products, no fused epilogues); anything else / unset = the shipped kernels
 *       package  SN_PAIR_FUSED        0 = dense-correspondence loss on the materialised score matrix (baseline of sn_pair_fused_*)
 *                SN_STRICT            1 = raise where a shape falls back to a library GEMM / an unfused composition
 *                SN_DEBUG_VALIDATE    1 = sn_validate_csr_i32 on every operator built from CSR arrays
 *                SN_RESIDENT          0 = utils_pt's batching functions take the reference's host path (no resident cache)
 *                SN_RESIDENT_MAX_GB   HBM budget of that cache (default 96)
 *                SN_DP_FORCE_CPU      1 = dp.init_distributed ignores the GPU (CPU gloo tests)
 *                SN_PLANS             0 = every block launches its kernels one by one from Python (no launch plans, plans.py)
 *                SN_PLAN_GRAPHS       0 = a plan run always walks its launch list (no graph launch at repeating addresses)
 *     SWITCHES: SN_GEMM_VARIANT SN_PAIR_FUSED SN_STRICT SN_DEBUG_VALIDATE SN_RESIDENT SN_RESIDENT_MAX_GB SN_DP_FORCE_CPU SN_PLANS SN_PLAN_GRAPHS
 *     (b) the opt-in per-launch timing facility sn_timing_* below (a mutex-guarded list, off by default).  Neither affects
 *     results.  The Python layer adds process-wide DEFAULTS with the same property: functional.set_dirac_format /
 *     set_laplacian_format (kernel form; one operator can choose for itself, SparseOperator.format) and set_bn_sync (opt-in
 *     global BatchNorm statistics).
 *   - Return value: 0 = success; negative = SN_E_* invalid argument; positive = hipError_t of the
 *     failed launch.  sn_status_string() renders either.
 *   - Indices are int32 (the reference uses int64, utils_pt.py:62, sparse_bmm.cu:17); an operator
 *     whose nnz or row/col count does not fit int32 is rejected with SN_E_RANGE.
 *   - Every output element is written (no reliance on zero-initialised outputs), including rows
 *     with no entries — the reference kernel silently mis-handles interior empty rows
 *     (batch_csr.cu:35-41) and that defect is NOT reproduced.
 *
 * Row addressing of the dense operands ("group" layout)
 *   A dense operand with N columns is addressed as
 *        row r  ->  base + (r / group) * ld + (r % group) * N          (floats)
 *   group = 1 : ordinary row-major matrix with leading dimension ld >= N.
 *   group = 4 : the quaternion view of the Dirac path.  The reference views v:(B,V,C) as
 *               (B*V*4, C/4) (utils_pt.py:201,213): the 4 rows of one vertex/face are C = 4N
 *               contiguous floats.  With ld = 4N this is the same contiguous matrix; with ld = 2C
 *               the operand lives inside one half of the (B,Nodes,2C) concat buffer that feeds
 *               BatchNorm+Linear (utils_pt.py:204,216), so SpMM can read from / write into that
 *               buffer directly and torch.cat disappears.
 */
#ifndef SN_SPMM_H_
#define SN_SPMM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SN_ABI_VERSION 1

/* negative status codes (positive codes are hipError_t values) */
#define SN_OK            0
#define SN_E_NULL       -1   /* a required pointer is NULL                                   */
#define SN_E_SHAPE      -2   /* negative / inconsistent dimension                            */
#define SN_E_RANGE      -3   /* dimension or nnz does not fit the int32 index type           */
#define SN_E_LD         -4   /* leading dimension / group layout invalid for N               */
#define SN_E_ALIGN      -5   /* reserved: pointer/ld alignment required by a kernel variant  */
#define SN_E_WORKSPACE  -6   /* workspace too small                                          */
#define SN_E_UNSUPPORTED -7  /* e.g. BSR4 requested for M or K not a multiple of 4           */

int         sn_abi_version(void);
const char *sn_status_string(int status);

/* ------------------------------------------------------------------------------------------
 * Y = A · X      A: M x K in CSR (int32 rowptr[M+1], colind[nnz], fp32 vals[nnz]), fp32 throughout.
 *
 * Replaces: SparseBMM.__call__(values, col_ind, col_ptr, size, dense)   src/utils/cuda/sparse_bmm.py:28-61
 *           and its kernel                                               src/utils/cuda/sparse_bmm.cu:16-61
 *           and the ATen call sites  torch.mm(L, x) / torch.mm(Di, x) / torch.mm(DiA, x)
 *                                                                         src/utils/utils_pt.py:167,176,202,214
 * The batched (B,R,K) operator of the reference is the block-diagonal 2-D operator with
 * M = B*R, K = B*Kb (col_ptr there already holds global nnz offsets, sparse_bmm.cu:36-37).
 * The same entry point computes the backward  grad_X = Aᵀ · grad_Y  when given the CSR of Aᵀ
 * (sparse_bmm_func.py:66-70); no gradient w.r.t. the operator exists (sparse_bmm_func.py:72).
 *
 * X: K rows, Y: M rows, both N columns, addressed with (ld, group) as described above.
 * Any N >= 1 is accepted; N in {16,32,64,128} with 16-byte aligned bases/ld take the
 * vectorised wave64 kernels.  X and Y must not alias.
 * ------------------------------------------------------------------------------------------ */
int sn_spmm_csr_f32(const int32_t *rowptr, const int32_t *colind, const float *vals,
                    int64_t M, int64_t K, int64_t nnz,
                    const float *X, int64_t ldx, int32_t x_group,
                    int32_t N,
                    float *Y, int64_t ldy, int32_t y_group,
                    void *stream);

/* sn_spmm_csr_stats_f32: sn_spmm_csr_f32 for N = 128, y_group = 1 (the Laplacian product on 128-channel rows,
 * torch.mm(L, x) at src/utils/utils_pt.py:167,176) that ALSO leaves the BatchNorm statistics of its output:
 * stats_part[sn_spmm_q3_stats_blocks()][2][128] fp64 partial column sums / sums of squares, to be combined by
 * sn_colstats_merge_f64 — the statistics pass over the propagated half of [x | L x] (utils_pt.py:168-169,177-178:
 * torch.cat + BatchNorm1d) disappears.  workspace: sn_spmm_csr_stats_workspace_bytes(M). Y is bit-identical to sn_spmm_csr_f32. */
size_t sn_spmm_csr_stats_workspace_bytes(int64_t M);
int sn_spmm_csr_stats_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                          const float *X, int64_t ldx, int32_t x_group, int32_t N, float *Y, int64_t ldy, int32_t y_group,
                          double *stats_part, void *workspace, size_t workspace_bytes, void *stream);

/* Same product with A stored as 4x4-block BSR (block rows Mb = M/4, block cols Kb = K/4,
 * bvals holds 16 floats per block, row-major inside the block).  This is the packed form of the
 * quaternionic Dirac operators: every 4x4 block of Di is -Q(0,e)/(2 Af) (src/utils/mesh.py:28-33,55-58).
 * Results are bit-identical to sn_spmm_csr_f32 on the same operator for finite X (the explicit
 * zeros of a block contribute fma(0,x,acc) == acc). N must be a multiple of 4. */
int sn_spmm_bsr4_f32(const int32_t *b_rowptr, const int32_t *b_colind, const float *b_vals,
                     int64_t Mb, int64_t Kb, int64_t nblocks,
                     const float *X, int64_t ldx, int32_t x_group,
                     int32_t N,
                     float *Y, int64_t ldy, int32_t y_group,
                     void *stream);

/* Quaternion-packed Dirac operators ("Q3").  Every 4x4 block of Di, DiA and their transposes is the matrix of a
 * multiplication by a pure quaternion (src/utils/mesh.py:28-33: Q(0,e); :55-58: -Q/(2 Af) and its transpose times Af/Av),
 *     M(p) = [[0, p1, p2, p3], [-p1, 0, p3, -p2], [-p2, -p3, 0, p1], [-p3, p2, -p1, 0]],
 * i.e. three floats.  q_blk holds one 16-byte record (p1, p2, p3, block column as int32 bits) per block, b_rowptr is the
 * BSR4 block-row pointer.  The operator stream is 16 instead of 68 bytes per block; results are bit-identical to
 * sn_spmm_bsr4_f32 / sn_spmm_csr_f32 for finite X (same k-ascending FMA order; zeros contribute fma(0,x,acc) = acc).
 * sn_bsr4_to_q3_f32 packs a BSR4 operator and sets *not_quaternion (device int32) to 1 if any block is not exactly M(p):
 * the caller then keeps the BSR4 form.  sn_blockdiag_concat_i32 assembles pooled Q3 operators with vals_per_entry = 4
 * (the column offset is applied inside the record; colind arguments are unused). */
int sn_spmm_q3_f32(const int32_t *b_rowptr, const float *q_blk, int64_t Mb, int64_t Kb, int64_t nblocks,
                   const float *X, int64_t ldx, int32_t x_group, int32_t N,
                   float *Y, int64_t ldy, int32_t y_group, void *stream);
/* sn_spmm_q3_stats_f32: sn_spmm_q3_f32 for N = 32 or 16, y_group = 4 (the (rows/4, 4N) view of a 128- / 64-channel tensor) that ALSO
 * leaves the BatchNorm statistics of its output: stats_part[sn_spmm_q3_stats_blocks()][2][4N] fp64 partial column sums /
 * sums of squares (per channel c = N*component + column), to be combined by sn_colstats_merge_f64 — the statistics pass over
 * the propagated half of a stage's concat buffer (utils_pt.py:204-205,216-217: torch.cat + BatchNorm1d) disappears.
 * Each workgroup's fp32 partial of its 32 output rows goes through `workspace` (sn_spmm_q3_stats_workspace_bytes(Mb)) and is
 * added up in fp64 in a fixed order (deterministic).  Y is bit-identical to sn_spmm_q3_f32. */
size_t sn_spmm_q3_stats_workspace_bytes(int64_t Mb);
int32_t sn_spmm_q3_stats_blocks(void);
int sn_spmm_q3_stats_f32(const int32_t *b_rowptr, const float *q_blk, int64_t Mb, int64_t Kb, int64_t nblocks, const float *X,
                         int64_t ldx, int32_t x_group, int32_t N, float *Y, int64_t ldy, int32_t y_group, double *stats_part,
                         void *workspace, size_t workspace_bytes, void *stream);
int sn_spmm_q3_elubwd_f32(const int32_t *b_rowptr, const float *q_blk, int64_t Mb, int64_t Kb, int64_t nblocks,
                          const float *X, int64_t ldx, int32_t x_group, int32_t N,
                          const float *E, int64_t lde, const float *G, int64_t ldg,
                          float *Y, int64_t ldy, int32_t y_group, void *stream);
/* The same product, which also leaves an upper bound of max |Y| for the two-piece weight gradient that reads Y as its dy
 * operand (sn_wgrad_*_bounded_f32): y_absmax — device, sn_spmm_q3_absmax_blocks(Mb, N) floats, one maximum per workgroup;
 * the consumer takes the maximum of all of them.  No atomics, nothing to zero beforehand; deterministic. */
int64_t sn_spmm_q3_absmax_blocks(int64_t Mb, int32_t N);
int sn_spmm_q3_elubwd_absmax_f32(const int32_t *b_rowptr, const float *q_blk, int64_t Mb, int64_t Kb, int64_t nblocks,
                                 const float *X, int64_t ldx, int32_t x_group, int32_t N, const float *E, int64_t lde,
                                 const float *G, int64_t ldg, float *Y, int64_t ldy, int32_t y_group, float *y_absmax,
                                 void *stream);
int sn_bsr4_to_q3_f32(const int32_t *b_colind, const float *b_vals, int64_t nblocks, float *q_blk,
                      int32_t *not_quaternion, void *stream);

/* The same two products with the backward of the ELU that precedes the propagation fused into the store:
 *     Y = (A·X) ∘ elu'(E) + G,      elu'(·) taken from the activation OUTPUT E: 1 where E > 0, E + 1 elsewhere
 * — with A = Lᵀ / Diᵀ / DiAᵀ this is the gradient that autograd assembles from the sparse product's backward
 * (src/utils/cuda/sparse_bmm_func.py:60-71), ELUBackward (F.elu at src/utils/utils_pt.py:161,171,193,208) and the sum of
 * the other branches' gradients (G), without the two intermediate arrays.  E and G have Y's shape and row grouping
 * (y_group) with their own leading dimensions; G may be NULL.  N in {16,32,64,128}, 16-byte aligned operands. */
int sn_spmm_csr_elubwd_f32(const int32_t *rowptr, const int32_t *colind, const float *vals,
                           int64_t M, int64_t K, int64_t nnz,
                           const float *X, int64_t ldx, int32_t x_group,
                           int32_t N,
                           const float *E, int64_t lde, const float *G, int64_t ldg,
                           float *Y, int64_t ldy, int32_t y_group,
                           void *stream);
int sn_spmm_bsr4_elubwd_f32(const int32_t *b_rowptr, const int32_t *b_colind, const float *b_vals,
                            int64_t Mb, int64_t Kb, int64_t nblocks,
                            const float *X, int64_t ldx, int32_t x_group,
                            int32_t N,
                            const float *E, int64_t lde, const float *G, int64_t ldg,
                            float *Y, int64_t ldy, int32_t y_group,
                            void *stream);

/* ------------------------------------------------------------------------------------------
 * Sorted COO -> CSR.
 *
 * Replaces: BatchCSR.__call__(indices, size) -> (col_ind, col_ptr)      src/utils/cuda/batch_csr.py:28-59
 *           and its kernel                                               src/utils/cuda/batch_csr.cu:13-47
 *           plus the implicit COO->CSR inside ATen's sparse addmm behind utils_pt.py:167,176,202,214.
 *
 * idx_batch may be NULL (2-D operator, B = 1).  For a 3-D batched operator (B, R, Kb) — the output
 * of sparse_cat (utils_pt.py:21-39) — the result is the block-diagonal CSR with
 *      global row = b*R + r  (M = B*R rows),   global col = b*Kb + c.
 * Indices are the int64 rows of a torch COO `_indices()` tensor; entries must be sorted by
 * (batch,row[,col]) — i.e. coalesced, as the reference requires (batch_csr.cu comment, :13).
 * rowptr (M+1 entries) is computed by binary search per row, so interior empty rows are correct.
 * ------------------------------------------------------------------------------------------ */
int sn_coo_to_csr_i32(const int64_t *idx_batch, const int64_t *idx_row, const int64_t *idx_col,
                      int64_t nnz, int64_t B, int64_t R, int64_t Kb,
                      int32_t *rowptr, int32_t *colind,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * CSR of Aᵀ from CSR of A (deterministic: entries of each output row ordered by column).
 *
 * Replaces: matrix1.transpose(2,1).coalesce() + batch_csr on every backward
 *                                                                         src/utils/cuda/sparse_bmm_func.py:66-67
 * Done ONCE per operator here and kept resident next to A.
 * workspace: sn_csr_transpose_workspace_bytes(M, K, nnz) bytes of device scratch.
 * ------------------------------------------------------------------------------------------ */
size_t sn_csr_transpose_workspace_bytes(int64_t M, int64_t K, int64_t nnz);
int sn_csr_transpose_f32(const int32_t *rowptr, const int32_t *colind, const float *vals,
                         int64_t M, int64_t K, int64_t nnz,
                         int32_t *t_rowptr, int32_t *t_colind, float *t_vals,
                         void *workspace, size_t workspace_bytes,
                         void *stream);

/* ------------------------------------------------------------------------------------------
 * CSR -> BSR4 (4x4 blocks).  Two phases because the caller owns the allocation:
 *   1. sn_bsr4_count: b_rowptr[Mb+1] <- exclusive scan of the number of distinct 4-wide block
 *      columns touched by each group of 4 rows; the caller reads b_rowptr[Mb] (= nblocks).
 *   2. sn_bsr4_fill : b_colind[nblocks], b_vals[16*nblocks] (zero-filled where A has no entry).
 * Column indices inside each CSR row must be sorted ascending (coalesced operator).
 * workspace for count: sn_scan_workspace_bytes(Mb + 1).
 * ------------------------------------------------------------------------------------------ */
size_t sn_scan_workspace_bytes(int64_t n);
int sn_bsr4_count(const int32_t *rowptr, const int32_t *colind, int64_t M, int64_t K,
                  int32_t *b_rowptr, void *workspace, size_t workspace_bytes, void *stream);
int sn_bsr4_fill(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K,
                 const int32_t *b_rowptr, int32_t *b_colind, float *b_vals, void *stream);

/* ------------------------------------------------------------------------------------------
 * CSR -> RB4 ("row-blocked", 4x1 blocks) and the RB4 product — the packed form of the Laplacian-type operators
 * (torch.mm(L, x) with ~7 entries per row, src/utils/utils_pt.py:167,176; L from src/utils/mesh.py:102-112 and
 * src/utils/graph.py:40-66).  Rows 4b .. 4b+3 share one sorted list of the columns any of them touches; every listed
 * column carries four coefficients (zero where a row lacks it).  Four consecutive mesh rows reach ~16 distinct columns with
 * ~28 entries, so the product gathers each X row once per 4-row group instead of once per row.
 *   sn_rb4_count: b_ptr[Mb+1] <- exclusive scan of the listed-column counts, Mb = ceil(M/4) (rows past M are empty);
 *                 workspace sn_scan_workspace_bytes(Mb + 1).  The total b_ptr[Mb] never exceeds nnz, so the caller can
 *                 size b_col / b_val by nnz without reading it back.
 *   sn_rb4_fill : b_col[total] (ascending inside a group), b_val[4*total] (16-byte aligned).
 *   sn_spmm_rb4_f32 / _elubwd_f32 / _stats_f32: Y = A X on plain row-major operands (group = 1), N in {64, 128}; same
 *                 epilogue / statistics semantics as the CSR entry points; `capacity` = allocated length of b_col.
 * Results are bit-identical to sn_spmm_csr_f32 for finite X (same k-ascending FMA order per row; a zero coefficient
 * contributes fma(0, x, acc) = acc; a non-finite X entry reaches all 4 rows of a group that lists its column).
 * ------------------------------------------------------------------------------------------ */
int sn_rb4_count(const int32_t *rowptr, const int32_t *colind, int64_t M, int64_t K, int32_t *b_ptr,
                 void *workspace, size_t workspace_bytes, void *stream);
int sn_rb4_fill(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K,
                const int32_t *b_ptr, int32_t *b_col, float *b_val, void *stream);
int sn_spmm_rb4_f32(const int32_t *b_ptr, const int32_t *b_col, const float *b_val, int64_t M, int64_t K, int64_t capacity,
                    const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, void *stream);
int sn_spmm_rb4_elubwd_f32(const int32_t *b_ptr, const int32_t *b_col, const float *b_val, int64_t M, int64_t K,
                           int64_t capacity, const float *X, int64_t ldx, int32_t N, const float *E, int64_t lde,
                           const float *G, int64_t ldg, float *Y, int64_t ldy, void *stream);
/* ... and max |Y| per wave into y_absmax[sn_spmm_rb4_absmax_blocks(M, N)] floats (every entry written): the bound the two-piece
 * weight gradient of the layer below needs for its dy operand (sn_wgrad_bounded_f32) — the fused backward of a Laplacian stage
 * (SparseBMMFunc.backward + ELUBackward, sparse_bmm_func.py:60-71) writes exactly that operand. */
int64_t sn_spmm_rb4_absmax_blocks(int64_t M, int32_t N);
int sn_spmm_rb4_elubwd_absmax_f32(const int32_t *b_ptr, const int32_t *b_col, const float *b_val, int64_t M, int64_t K,
                                  int64_t capacity, const float *X, int64_t ldx, int32_t N, const float *E, int64_t lde,
                                  const float *G, int64_t ldg, float *Y, int64_t ldy, float *y_absmax, void *stream);
size_t sn_spmm_rb4_stats_workspace_bytes(int64_t M);
int sn_spmm_rb4_stats_f32(const int32_t *b_ptr, const int32_t *b_col, const float *b_val, int64_t M, int64_t K,
                          int64_t capacity, const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy,
                          double *stats_part, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Sliding-window ("ring") product for BANDED SQUARE CSR operators — the Laplacian products torch.mm(L, x) at 64 / 128
 * channels (src/utils/utils_pt.py:167,176) on batches that fill the chip; same arguments, epilogue and statistics semantics
 * as sn_spmm_csr_f32 / _elubwd_f32 / _stats_f32 with group = 1, straight from the CSR arrays (no derived form).
 * A persistent workgroup walks a strip of rows and keeps the X rows within +-sn_spmm_csr_ring_half_window() of the current
 * rows in an LDS ring: every X line and every entry is requested once per 64-column slice, by LDS-DMA.  Columns outside
 * the window are gathered from global memory (correct for ANY operator, fast for banded ones): callers decide with
 * sn_csr_band_i32, which leaves max |column - row|, the longest row and the number of rows with an entry outside the window
 * in band_longest_outside[0..2] (device memory); an operator with a row whose columns do not ascend strictly reports
 * INT32_MAX as its longest row (the ring kernel bounds a row by its first and last entry: such an operator must not take it).
 * Requirements: M == K, N in {64, 128}, columns ascending inside each row; SN_E_UNSUPPORTED otherwise.
 * Results are bit-identical to sn_spmm_csr_f32 (same k-ascending FMA order per row).
 * ------------------------------------------------------------------------------------------ */
int sn_spmm_csr_ring_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                         const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, void *stream);
int sn_spmm_csr_ring_elubwd_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K,
                                int64_t nnz, const float *X, int64_t ldx, int32_t N, const float *E, int64_t lde, const float *G,
                                int64_t ldg, float *Y, int64_t ldy, void *stream);
/* ... and max |Y| per compute wave into y_absmax[sn_spmm_csr_ring_absmax_blocks(M, N)] floats (every entry written; as
 * sn_spmm_rb4_elubwd_absmax_f32).  An operator without entries is not taken (SN_E_UNSUPPORTED). */
int64_t sn_spmm_csr_ring_absmax_blocks(int64_t M, int32_t N);
int sn_spmm_csr_ring_elubwd_absmax_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K,
                                       int64_t nnz, const float *X, int64_t ldx, int32_t N, const float *E, int64_t lde,
                                       const float *G, int64_t ldg, float *Y, int64_t ldy, float *y_absmax, void *stream);
size_t sn_spmm_csr_ring_stats_workspace_bytes(int64_t M);
int sn_spmm_csr_ring_stats_f32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                               const float *X, int64_t ldx, int32_t N, float *Y, int64_t ldy, double *stats_part, void *workspace,
                               size_t workspace_bytes, void *stream);
int32_t sn_spmm_csr_ring_half_window(void);
int sn_csr_band_i32(const int32_t *rowptr, const int32_t *colind, int64_t M, int64_t K, int32_t *band_longest_outside, void *stream);

/* ------------------------------------------------------------------------------------------
 * Block-diagonal batch assembly from a device-resident pool of per-mesh operators.
 *
 * Replaces: sparse_diag_cat(tensors, size0, size1)                       src/utils/utils_pt.py:41-53
 *           (index shift i*[size0;size1], concat, coalesce() sort on the host every step) and the
 *           per-sample sp_sparse_to_pt_sparse conversions               src/utils/utils_pt.py:56-69,
 *           src/as_rigid_as_possible/main.py:161-162,174-175.
 *
 * The pool holds every mesh's operator once (CSR with vals_per_entry = 1, or BSR4 with 16), rowptr
 * local to the mesh (starting at 0).  desc is a (B x 4) int64 table, one row per selected mesh:
 *      desc[b] = { offset of the mesh's rowptr in pool_rowptr,
 *                  offset of the mesh's first entry in pool_colind (vals: x vals_per_entry),
 *                  number of rows of the mesh (<= size0),
 *                  offset of the mesh's first entry in the OUTPUT (exclusive prefix of entry counts) }
 * Output: out_rowptr[B*size0 + 1], out_colind[total], out_vals[total*vals_per_entry] with
 * row = b*size0 + r and col = b*size1 + c; rows beyond a mesh's own count are empty (the padding to
 * the batch maximum of the reference).  `total` = number of entries of the whole batch.
 * ------------------------------------------------------------------------------------------ */
int sn_blockdiag_concat_i32(const int32_t *pool_rowptr, const int32_t *pool_colind, const float *pool_vals,
                            const int64_t *desc, int64_t B, int64_t size0, int64_t size1,
                            int64_t total, int32_t vals_per_entry,
                            int32_t *out_rowptr, int32_t *out_colind, float *out_vals,
                            void *stream);

/* Ragged (packed) form of the same assembly — what the reference cannot express: sparse_diag_cat pads every block to
 * (size0, size1) = the batch maximum (src/utils/utils_pt.py:48-52; src/as_rigid_as_possible/main.py:172-185 computes the
 * maxima), so half of a 1k..20k-vertex batch is padding.  Here mesh b owns the output rows [desc[b][4], desc[b+1][4])
 * (the last mesh up to total_rows), the first desc[b][2] of which carry its entries, and its column indices are shifted by
 * desc[b][5].  With both offsets = exclusive prefix sums of the meshes' own row / column counts the batch has no padding
 * at all: the dense operands are the concatenation of the meshes' rows.  (desc[b][4] = b*size0, desc[b][5] = b*size1
 * reproduces sn_blockdiag_concat_i32.)
 *      desc[b] = { rowptr offset in the pool, entry offset in the pool, rows with entries,
 *                  entry offset in the output, first output row, column shift }          (B x 6) int64
 * desc[b][4] must be ascending with desc[0][4] = 0.  Output: out_rowptr[total_rows + 1], out_colind[total],
 * out_vals[total * vals_per_entry]; total_cols only enters the int32 range check. */
int sn_blockdiag_concat_ragged_i32(const int32_t *pool_rowptr, const int32_t *pool_colind, const float *pool_vals,
                                   const int64_t *desc, int64_t B, int64_t total_rows, int64_t total_cols,
                                   int64_t total, int32_t vals_per_entry,
                                   int32_t *out_rowptr, int32_t *out_colind, float *out_vals,
                                   void *stream);

/* ------------------------------------------------------------------------------------------
 * Debug validation of a CSR operator on the device.  The reference kernels index without any bounds check
 * (src/utils/cuda/sparse_bmm.cu:16-61, batch_csr.cu:13-47) and so do the product kernels here (an out-of-range column is
 * an out-of-bounds gather); this entry point is the opt-in check.  *flags (device int32) receives a bit set:
 *   1 rowptr[0] != 0 | 2 rowptr not non-decreasing | 4 rowptr[M] != nnz | 8 column index outside [0, K) |
 *   16 column indices of a row not strictly ascending (operator not coalesced) | 32 non-finite value (vals may be NULL).
 * 0 = the operator is well formed.  The Python layer runs it on every operator it builds when SN_DEBUG_VALIDATE=1 and
 * raises (status SN_E_RANGE semantics) — see surfacenetworks_amd/operators.py.
 * ------------------------------------------------------------------------------------------ */
int sn_validate_csr_i32(const int32_t *rowptr, const int32_t *colind, const float *vals, int64_t M, int64_t K, int64_t nnz,
                        int32_t *flags, void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused elementwise helpers of the residual blocks (each replaces separate ATen passes).
 *
 * sn_elu_into_f32: dst[r, 0:C] = elu(src[r, 0:C]) for `rows` rows; src stride lds, dst stride ldd.
 *   Replaces F.elu (utils_pt.py:161,171,195,208) + the first operand copy of torch.cat
 *   (utils_pt.py:168,177,204,216): ELU is written straight into the first half of the concat buffer.
 * sn_elu_bwd_acc_f32: gsrc[r,c] (+)= (gdst[r,c] + gdst2[r,c]) * (out[r,c] > 0 ? 1 : out[r,c] + 1) + gadd[r,c], out = elu
 *   value.  gdst2 and gadd may be NULL.  The activated tensor feeds both the concat buffer and the SpMM (two gradients
 *   meet before the ELU derivative) and the un-activated tensor also feeds the residual path (gadd, after it);
 *   accumulate != 0 additionally adds into gsrc.
 * ------------------------------------------------------------------------------------------ */
int sn_elu_into_f32(const float *src, int64_t lds, float *dst, int64_t ldd,
                    int64_t rows, int32_t C, void *stream);
int sn_elu_bwd_acc_f32(const float *gdst, int64_t ldg, const float *gdst2, int64_t ldg2, const float *gadd, int64_t ldga,
                       const float *out, int64_t ldo, float *gsrc, int64_t ldgs, int64_t rows, int32_t C,
                       int32_t accumulate, void *stream);


/* ------------------------------------------------------------------------------------------
 * BatchNorm("pre") + Linear of GraphConv1x1 (src/utils/utils_pt.py:83-99) without materialising the
 * normalised tensor.  With s = gamma*invstd, t = beta - mean*s the layer is  y = x·(W·diag(s))ᵀ + (b + W·t),
 * so the forward needs only per-channel statistics of x; the backward needs G = dyᵀ·x and colsum(dy), from
 * which every BatchNorm reduction follows algebraically (sum dz = colsum(dy)·W, sum dz*x = sum_j W∘G).
 *
 * sn_colstats_f32 : out[0:C] = column sums, out[C:2C] = column sums of squares of x (rows x C, row stride ld),
 *                   accumulated in fp64, two deterministic stages.  Replaces the statistics pass of
 *                   nn.BatchNorm1d (train mode) and the bias-gradient reduction.
 * sn_wgrad_f32    : G (J x C, row-major, fp32) = dyᵀ·(x - center) with dy (rows x J, stride lddy), x (rows x C, stride
 *                   ldx), center[C] optional (NULL = 0; BatchNorm passes the batch mean so that no cancellation
 *                   between dyᵀ·x and mean·colsum(dy) is left to fp32):
 *                   split-K over row slabs on the fp32 MFMA (v_mfma_f32_32x32x2_f32), partial tiles reduced in a
 *                   fixed order.  J <= 128 and a multiple of 4, C in {128, 256}.  dysum (optional, J doubles) receives the
 *                   column sums of dy — the bias gradient and the input of sn_bn_bwd_coeffs_f32 — accumulated by the
 *                   same pass over dy, so no separate reduction kernel is needed.
 *                   Replaces the weight-gradient GEMM of nn.Linear for tall-skinny operands (K = rows ~ 1e5..1e6).
 * sn_affine_cols_acc_f32 : dx[r,c] += (x[r,c] - center[c])*B[c] + Cc[c]  (the elementwise tail of BatchNorm
 *                   backward, fused; center may be NULL).
 * ------------------------------------------------------------------------------------------ */
size_t sn_colstats_workspace_bytes(int64_t rows, int32_t C);
int sn_colstats_f32(const float *x, int64_t ld, int64_t rows, int32_t C, double *out,
                    void *workspace, size_t workspace_bytes, void *stream);
/* sn_colstats_into_f32 : the same statistics written as ONE HALF of a wider vector: sums at out[out_off + c], sums of
 *                   squares at out[out_ld + out_off + c] (out_ld = full width).
 * sn_colstats_merge_f64 : the other half, from the per-workgroup partials ([nblk][2][C] fp64, nblk =
 *                   sn_linear_fwd_stats_blocks(rows)) that sn_linear_fwd_f32 / sn_linear_fwd_segbias_f32 leave when asked for
 *                   the column statistics of their ELU output (elu_stats_part): the statistics pass of the next stage then
 *                   reads only the propagated half of its concat buffer. */
int sn_colstats_into_f32(const float *x, int64_t ld, int64_t rows, int32_t C, double *out, int64_t out_ld, int64_t out_off,
                         void *workspace, size_t workspace_bytes, void *stream);
int sn_colstats_merge_f64(const double *part, int32_t nblk, int32_t C, double *out, int64_t out_ld, int64_t out_off,
                          void *stream);
/* sn_colstats_merge2_f64 : both halves at once — out (2 x (C_lo + C_hi) fp64, [sums | squares]) from the partials of the two
 * producers ([nblk_lo][2][ld_lo] and [nblk_hi][2][ld_hi], of which the first C_lo / C_hi channels count: ld = C for a
 * producer's own layout, 128 for the forward GEMM of a 64-output layer); the same sums, bit for bit, as two
 * sn_colstats_merge_f64 calls on packed partials. */
int sn_colstats_merge2_f64(const double *part_lo, int32_t nblk_lo, int32_t C_lo, int32_t ld_lo, const double *part_hi,
                           int32_t nblk_hi, int32_t C_hi, int32_t ld_hi, double *out, void *stream);
int32_t sn_linear_fwd_stats_blocks(int64_t rows);
size_t sn_wgrad_workspace_bytes(int64_t rows, int32_t J, int32_t C);
int sn_wgrad_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                 int32_t J, int32_t C, float *G, double *dysum, void *workspace, size_t workspace_bytes, void *stream);
/* sn_wgrad_seg_f32: the same weight gradient for a batch of rows / rows_per_seg meshes, with the row slabs aligned to mesh
 * boundaries so that the pass also yields the PER-MESH column sums of dy, seg_dysum (nseg x J, fp32) — what the half-width
 * global-average stage needs (it replaces a sn_segment_colsum_f32 pass over dy).  dysum and seg_dysum are required; rows
 * must be a multiple of rows_per_seg.  Split-bf16 kernels only (SN_E_UNSUPPORTED with SN_GEMM_VARIANT=0). */
size_t sn_wgrad_seg_workspace_bytes(int64_t rows, int64_t rows_per_seg, int32_t J, int32_t C);
int sn_wgrad_seg_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                     int64_t rows_per_seg, int32_t J, int32_t C, float *G, double *dysum, float *seg_dysum, void *workspace,
                     size_t workspace_bytes, void *stream);
/* sn_wgrad_slabs_f32: the same for RAGGED meshes (packed batches, no padding rows — the reference pads every mesh to the
 * batch maximum, src/as_rigid_as_possible/main.py:172-185): the caller supplies the row slabs, slab b = rows
 * [slab_off[b], slab_off[b+1]) (device int64[nslab + 1], consecutive, none crossing a mesh boundary) and the slabs of every
 * mesh, [seg_slab_ptr[m], seg_slab_ptr[m+1]) (device int64[nseg + 1]); seg_dysum is (nseg x J).  Workspace:
 * nslab * 128 * (C + 1) floats.  Uniform-wave kernel only (SN_E_UNSUPPORTED with SN_GEMM_VARIANT=0 / SN_WGRAD_VARIANT=1). */
int sn_wgrad_slabs_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                       const int64_t *slab_off, int32_t nslab, const int64_t *seg_slab_ptr, int32_t nseg, int32_t J, int32_t C,
                       float *G, double *dysum, float *seg_dysum, void *workspace, size_t workspace_bytes, void *stream);
/* sn_wgrad_*_bounded_f32: the three weight gradients above on TWO fp16 pieces per operand (three exact partial products
 * instead of the six of the three-piece bf16 split: half the matrix work; measured -20 % per launch, round 4).  fp16 has
 * five exponent bits, and the contraction runs over the rows, so the powers of two that bring the operands into range must be
 * constant along a column and known before the first row is read.  The caller supplies the bounds they follow from:
 *   dybound   device, n_dybound floats whose maximum is >= max |dy| over the whole operand — the per-workgroup maxima left by
 *             the kernel that produced dy (sn_linear_dgrad_elu_absmax_f32, sn_linear_dgrad_eluseg_absmax_f32,
 *             sn_spmm_q3_elubwd_absmax_f32), or one number from any other source; the kernel takes their maximum itself;
 *   xinvstd   device, C floats: BatchNorm's inverse standard deviations 1 / sqrt(var + eps) of x's columns about `center`
 *             (the biased batch variance, src/utils/utils_pt.py:83-99 nn.BatchNorm1d), over stat_rows >= rows rows that include
 *             every row of x:  |x[r][c] - center[c]| <= sqrt(stat_rows) / xinvstd[c]  holds for every row, rigorously.
 * Both operands are scaled so that their bound lands in [2^14, 2^15) (nothing can overflow); an element keeps 22 significant
 * bits down to 2^-17 of its operand's bound and 2^-25 of the scaled unit (2^-39 of the bound) in absolute terms below that
 * (SN_WGRAD_H=1: a second accumulator for low pieces scaled by a further 2^11 — 2^-28 / 2^-50 — at the cost of the second row
 * block in flight); results leave multiplied by the exact inverse scales.  Eval-mode BatchNorm (running statistics) offers no such bound: use the unbounded forms.
 * Same arguments, workspace and status codes as the forms above; SN_E_SHAPE when stat_rows < rows, SN_E_NULL without bounds. */
int sn_wgrad_bounded_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                         int32_t J, int32_t C, float *G, double *dysum, void *workspace, size_t workspace_bytes,
                         const float *dybound, int64_t n_dybound, const float *xinvstd, int64_t stat_rows, void *stream);
int sn_wgrad_seg_bounded_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                             int64_t rows_per_seg, int32_t J, int32_t C, float *G, double *dysum, float *seg_dysum,
                             void *workspace, size_t workspace_bytes, const float *dybound, int64_t n_dybound,
                             const float *xinvstd, int64_t stat_rows, void *stream);
int sn_wgrad_slabs_bounded_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *center, int64_t rows,
                               const int64_t *slab_off, int32_t nslab, const int64_t *seg_slab_ptr, int32_t nseg, int32_t J,
                               int32_t C, float *G, double *dysum, float *seg_dysum, void *workspace, size_t workspace_bytes,
                               const float *dybound, int64_t n_dybound, const float *xinvstd, int64_t stat_rows, void *stream);
/* sn_wgrad_thin_f32: weight and bias gradient of a Linear with 1..8 input channels — the models' first layer,
 * GraphConv1x1(6 | 3 -> C, batch_norm=None) (src/as_rigid_as_possible/models.py:113, src/utils/utils_pt.py:99) — on
 * rows ~ 1e5..1e6:  G (J x C, row-major, fp32) = dy^T x,  db (J, optional) = colsum(dy).  One pass over dy; fp32
 * accumulation over <= 64 rows per thread, fp64 above; two deterministic stages.  J % 4 == 0 and J/4 must divide 256
 * (SN_E_UNSUPPORTED otherwise).  Replaces the weight-gradient GEMM + the bias reduction of nn.Linear's backward. */
size_t sn_wgrad_thin_workspace_bytes(int64_t rows, int32_t J, int32_t C);
int sn_wgrad_thin_f32(const float *dy, int64_t lddy, const float *x, int64_t ldx, int64_t rows, int32_t J, int32_t C,
                      float *G, float *db, void *workspace, size_t workspace_bytes, void *stream);
/* Masked smooth-L1 training loss of the ARAP harness (src/as_rigid_as_possible/main.py:225-226:
 * `outputs * mask`, F.smooth_l1_loss(size_average=False) / batch_size), forward and backward as one pass each:
 *   loss  = scale * sum_{r,c} l(out[r,c]*rowmask[r] - target[r,c]),  l(d) = d*d/2 if |d| < 1 else |d| - 1/2   (fp64 sums)
 *   gout  = gloss[0]*scale * rowmask[r] * clamp(out[r,c]*rowmask[r] - target[r,c], -1, 1)
 * rowmask (rows floats) may be NULL (= all ones); gloss is a DEVICE scalar (the gradient of the loss value). */
size_t sn_masked_smooth_l1_workspace_bytes(int64_t rows, int32_t C);
int sn_masked_smooth_l1_fwd_f32(const float *out, int64_t ldo, const float *target, int64_t ldt, const float *rowmask,
                                int64_t rows, int32_t C, double scale, float *loss, void *workspace, size_t workspace_bytes,
                                void *stream);
int sn_masked_smooth_l1_bwd_f32(const float *out, int64_t ldo, const float *target, int64_t ldt, const float *rowmask,
                                int64_t rows, int32_t C, double scale, const float *gloss, float *gout, int64_t ldg,
                                void *stream);
/* sn_gather_segments_f32: batch assembly of dense per-sample windows (the ARAP sampler's inputs / targets,
 * src/as_rigid_as_possible/main.py:126-152): out[(i*rows_per_item + r)*len + c] = src[base[i] + r*row_stride + c],
 * i < nitems, r < rows_per_item, c < len; base is a DEVICE array of element offsets into the resident dataset. */
int sn_gather_segments_f32(const float *src, const int64_t *base, int64_t nitems, int64_t rows_per_item, int64_t row_stride,
                           int32_t len, float *out, void *stream);
/* ... straight into a PACKED batch (no padded intermediate, no boolean-mask gather): item i supplies its first
 * item_off[i+1] - item_off[i] rows, written to output rows item_off[i] ..; item_off: int64[nitems + 1] on the device,
 * total_rows = item_off[nitems] (known to the caller, who built the table). */
int sn_gather_segments_ragged_f32(const float *src, const int64_t *base, const int64_t *item_off, int64_t nitems,
                                  int64_t total_rows, int64_t row_stride, int32_t len, float *out, void *stream);
/* sn_pair_argmin_f32: target of the dense-correspondence loss (src/dense_correspondence/main.py:236-237,
 * `_, GAB = torch.min(GA[:, liA[lB]] + GB[liB[lA], :], dim=1)`):  out[r] = argmin_j ( GA[r*ldA + pa[j]] + GB[pb[r]*ldB + j] ),
 * r < NA, j < NB, with pa = liA[lB] (NB entries) and pb = liB[lA] (NA entries) as DEVICE int64 arrays; the two gathered
 * NA x NB matrices are never materialised.  fp32 sum as in the reference; ties go to the lowest j, a NaN wins (torch.min).
 * GA has colsA valid columns per row (rows up to 60 KB are staged in LDS, wider ones gathered from global memory); the
 * caller guarantees 0 <= pa[j] < colsA and that pb[r] indexes a row of GB. */
int sn_pair_argmin_f32(const float *GA, int64_t ldA, int64_t colsA, const int64_t *pa, const float *GB, int64_t ldB,
                       const int64_t *pb, int64_t NA, int64_t NB, int64_t *out, void *stream);
/* sn_pair_ce_fwd_f32 / sn_pair_ce_bwd_f32: the cross entropy of the dense-correspondence loss on the score matrix
 * (src/dense_correspondence/main.py:238-239, `F.cross_entropy(outputs[0, :NA, :NB], GAB)`; replaces log_softmax, nll_loss,
 * their backward passes and the zero padding of the slice's gradient).  Forward: lse[r] = log sum_j<NB exp(S[r][j]) and
 * rowloss[r] = lse[r] - S[r][target[r]] for r < NA (the loss is their mean; the caller sums NA floats).  Backward:
 * dS[r][j] = (*gloss / NA) * (softmax(S[r][:NB])[j] - [j == target[r]]) for r < NA, j < NB and 0 for the rest of the
 * rows x cols matrix (the padding of the batch); gloss is a DEVICE scalar.  fp32 throughout, max-subtracted like torch's.
 * 0 <= target[r] < NB is the caller's guarantee (not checked on the device). */
int sn_pair_ce_fwd_f32(const float *S, int64_t ld, const int64_t *target, int64_t NA, int64_t NB, float *lse, float *rowloss,
                       void *stream);
int sn_pair_ce_bwd_f32(const float *S, int64_t ld, const int64_t *target, const float *lse, const float *gloss, int64_t NA,
                       int64_t NB, int64_t rows, int64_t cols, float *dS, int64_t ldd, void *stream);
/* sn_pair_fused_fwd_f32 / sn_pair_fused_bwd_f32: the same loss computed FROM THE TOWER FEATURES, the score matrix never
 * written (replaces `torch.bmm(FA, FB.transpose(1, 2))`, src/dense_correspondence/models.py:203, together with the cross
 * entropy of main.py:238-239 and both their backward passes).  FA: rowsA x K, FB: rowsB x K row-major fp32 (K <= 128; the
 * towers emit 120), the first NA / NB rows are scored, the rest is the padding of the batch.  Scores are formed tile by tile
 * on the bf16 matrix pipe from three-piece truncation splits of the features (six exact partial products per term, fp32
 * accumulation: fp32-accurate, not bit-equal to an fp32 FMA chain) and reduced on the spot.
 *   forward : lse[r], rowloss[r] (r < NA) as sn_pair_ce_fwd_f32; the split features are left in `workspace`
 *             (sn_pair_fused_workspace_bytes(rowsA, rowsB) bytes, 16-byte aligned) for the backward pass.
 *   backward: dFA[r][k] = (*gloss / NA) sum_j (softmax(S[r])[j] - [j == target[r]]) FB[j][k] and
 *             dFB[j][k] = (*gloss / NA) sum_r (...) FA[r][k]; rows >= NA / NB of dFA / dFB are set to 0.  Reads the workspace
 *             the forward call filled (same rowsA, rowsB, K) and its lse.  Fixed summation order: run-to-run identical.
 * 0 <= target[r] < NB is the caller's guarantee. */
size_t sn_pair_fused_workspace_bytes(int64_t rowsA, int64_t rowsB);
int sn_pair_fused_fwd_f32(const float *FA, int64_t lda, const float *FB, int64_t ldb, const int64_t *target, int64_t NA,
                          int64_t NB, int64_t rowsA, int64_t rowsB, int32_t K, float *lse, float *rowloss, void *workspace,
                          size_t workspace_bytes, void *stream);
int sn_pair_fused_bwd_f32(const int64_t *target, const float *lse, const float *gloss, int64_t NA, int64_t NB, int64_t rowsA,
                          int64_t rowsB, int32_t K, float *dFA, int64_t ldda, float *dFB, int64_t lddb, void *workspace,
                          size_t workspace_bytes, void *stream);
/* sn_linear_thin_fwd_f32: forward of that first layer, y = x·W^T + bias (x: rows x C, C <= 8; W: J x C), and optionally
 * elu(y) into y_elu (the first half of the next block's concat buffer; replaces the F.elu of utils_pt.py:161,195).  y or
 * y_elu may be NULL (not both).  Ascending-k fp32 FMA chain on top of the bias. */
int sn_linear_thin_fwd_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *bias, int64_t rows,
                           int32_t C, int32_t J, float *y, int64_t ldy, float *y_elu, int64_t lde, void *stream);
int sn_affine_cols_acc_f32(float *dx, int64_t lddx, const float *x, int64_t ldx, const float *center, const float *B,
                           const float *Cc, int64_t rows, int32_t C, void *stream);
/* sn_affine_cols_elu_bwd_f32: the same tail for a layer whose operand x is itself the output of an ELU (the models' last
 * layer, `conv2(F.elu(v))`, src/as_rigid_as_possible/models.py:149-150), continued through that activation in the same
 * pass:  dx[r,c] = (dx[r,c] + (x[r,c]-center[c])*B[c] + Cc[c]) * (x[r,c] > 0 ? 1 : x[r,c] + 1).  B = Cc = NULL: no
 * BatchNorm tail (eval mode).  Replaces the tail pass + ELUBackward. */
int sn_affine_cols_elu_bwd_f32(float *dx, int64_t lddx, const float *x, int64_t ldx, const float *center, const float *B,
                               const float *Cc, int64_t rows, int32_t C, void *stream);

/* Small coefficient kernels of the folded BatchNorm+Linear (they replace ~40 tiny elementwise launches per layer):
 * sn_bn_fold_f32 : from the column statistics (stats = [sums | sums of squares], fp64, `rows` rows) or, when
 *   training == 0, from running_mean / running_var, compute mean[C], invstd[C], s = gamma*invstd, t = beta - mean*s,
 *   Wf = W·diag(s) (J x C), bf = b + W·t; in training mode also update running_mean / running_var in place with
 *   `momentum` and the unbiased variance, as nn.BatchNorm1d does, and add 1 to *num_batches_tracked (may be NULL).
 *   b may be NULL.
 * sn_bn_bwd_coeffs_f32 : from Gc = dyᵀ·(x - mean) (J x C), the column sums of dy (fp64, first J entries of `dystats`),
 *   W, s, invstd, mean, beta: dW = s∘Gc + colsum(dy) ⊗ beta, db = colsum(dy), dgamma = invstd ∘ sum_j W∘Gc,
 *   dbeta = colsum(dy)·W, and the coefficients of the elementwise tail Bc = -s∘invstd∘dgamma/rows, Cc = -s∘dbeta/rows. */
int sn_bn_fold_f32(const double *stats, int64_t rows, const float *gamma, const float *beta, const float *W,
                   const float *b, int32_t J, int32_t C, double eps, double momentum, int32_t training,
                   float *running_mean, float *running_var, float *mean, float *invstd, float *s, float *t,
                   float *Wf, float *bf, int64_t *num_batches_tracked, void *stream);
/* sn_colstats_partial_f32 : the statistics pass over x WITHOUT its final reduction — partial[sn_colstats_blocks(rows)][2][C]
 * fp64, the producer format of sn_colstats_merge_f64.
 * sn_wgrad_bounded_enabled / sn_gemm_variant : what the process-wide baseline switch SN_GEMM_VARIANT resolved to (read once):
 * 1 / 2 with the shipped 16-bit matrix-pipe kernels, 0 / 0 with the fp32-MFMA A/B baseline — launchers ask the library
 * instead of parsing the environment themselves. */
int32_t sn_wgrad_bounded_enabled(void);
int32_t sn_gemm_variant(void);
int32_t sn_colstats_blocks(int64_t rows);
int sn_colstats_partial_f32(const float *x, int64_t ld, int64_t rows, int32_t C, double *partial, void *stream);
int sn_bn_bwd_coeffs_f32(const float *Gc, const double *dystats, const float *W, const float *s, const float *invstd,
                         const float *beta, int64_t rows, int32_t J, int32_t C, float *dW, float *db, float *dgamma,
                         float *dbeta, float *Bc, float *Cc, void *stream);

/* ------------------------------------------------------------------------------------------
 * Global-average block helpers (AvgResNet2, src/utils/utils_pt.py:222-243; global_average :120-122).
 * A batch is nseg meshes of rows_per_seg (padded) rows each; mask[r] in {0,1} marks real vertices (may be NULL = all).
 *
 * sn_segment_colsum_f32 : out[seg, c] = sum over the rows r of mesh seg of mask[r] * x[r, c]   (fp64 accumulation,
 *                         two deterministic stages).  Replaces (x * mask).sum(1).
 * sn_bcast_rows_f32     : dst[r, c] = src[seg(r), c]  — writes the per-mesh mean into the second half of the concat
 *                         buffer (expand_as + contiguous + torch.cat in the reference).
 * sn_elu_bwd_bcast_f32  : gsrc[r,c] = (gdst[r,c] + mask[r] * bias[seg(r), c]) * elu'(out[r,c]) + gadd[r,c]: ELU backward
 *                         with the gradient of the mean path folded in (gadd: optional residual-path gradient).
 *
 * Half-width form of a global-average stage.  The second half of its concat buffer [e | mean(e) broadcast] is a per-mesh
 * constant, so the stage needs only e: the BatchNorm statistics of the second half, its share of the Linear product (a
 * per-mesh bias), of the weight gradient and of the input gradient are nseg x C algebra on the per-mesh mean m:
 * sn_avg_fwd_prep_f32   : m = segsum * inv_count;  stats (2 x 2C fp64, the layout sn_bn_fold_f32 reads) = [stats1 |
 *                         rows_per_seg * sum_mesh m, rows_per_seg * sum_mesh m^2]  (stats1 = sn_colstats_f32 of e).
 * sn_avg_stats_f32      : the two above in ONE pass over e (per-mesh masked sums, and sums / sums of squares of all rows,
 *                         fp64): m and stats as sn_avg_fwd_prep_f32 would produce them from sn_segment_colsum_f32 +
 *                         sn_colstats_f32.
 * sn_seg_affine_f32     : out[g, j] = bias[j] + sum_c A[g, c] * W[j, c]  — the per-mesh bias  m·Wf[:, C:]^T + bf  consumed
 *                         by sn_linear_fwd_segbias_f32.
 * sn_avg_bwd_gc_f32     : Gc (J x 2C) = [ G1 | sum_mesh seg_dy[mesh]^T (m[mesh] - mu2) ]  (G1 = sn_wgrad_f32 of the first
 *                         half, seg_dy = sn_segment_colsum_f32 of dy): the operand of sn_bn_bwd_coeffs_f32.
 * sn_avg_bwd_segvec_f32 : v[mesh, c] = inv_count[mesh] * ( seg_dy[mesh]·Wf2[:, c] + rows_per_seg * ((m[mesh,c] - mu2[c]) *
 *                         B2[c] + C2[c]) ): gradient of the mean path, added per row (times the row mask) inside
 *                         sn_linear_dgrad_eluseg_f32.
 * sn_bn_fold_seg_f32    : sn_bn_fold_f32 (training mode) of the stage's 2C-wide layer AND sn_seg_affine_f32 in one launch: the
 *                         workgroup that folds output row j also forms segbias[g, j] = bf[j] + m[g]·Wf[j, C:] (same operands,
 *                         same order: the same numbers).  C = the layer's full width (2 x channels of e), <= 256.
 * sn_avg_bn_bwd_f32     : sn_avg_bwd_gc_f32 + sn_bn_bwd_coeffs_f32 + sn_avg_bwd_segvec_f32 in one launch — all three are per
 *                         channel, so the workgroup of 32 channels runs them for its own; bit-identical.  G1 (J x C), dystats,
 *                         seg_dy (nseg x J) from sn_wgrad_seg_f32 / sn_wgrad_slabs_f32; segoff != NULL: ragged meshes (their
 *                         row counts from the offsets) instead of rows_per_seg.  J <= 128, C % 32 == 0.
 * ------------------------------------------------------------------------------------------ */
int sn_bn_fold_seg_f32(const double *stats, int64_t rows, const float *gamma, const float *beta, const float *W, const float *b,
                       int32_t J, int32_t C, double eps, double momentum, float *running_mean, float *running_var, float *mean,
                       float *invstd, float *s, float *t, float *Wf, float *bf, int64_t *num_batches_tracked, const float *seg_mean,
                       int64_t nseg, float *segbias, void *stream);
int sn_avg_bn_bwd_f32(const float *G1, const double *dystats, const float *seg_dy, const float *seg_mean, const float *mu2,
                      const float *W, const float *s, const float *invstd, const float *beta, int64_t rows, int32_t J, int32_t C,
                      int64_t nseg, const float *Wf2, int64_t ldw, const float *inv_count, int64_t rows_per_seg, const int64_t *segoff,
                      float *dW, float *db, float *dgamma, float *dbeta, float *Bc, float *Cc, float *segvec, void *stream);
/* Ragged forms for PACKED batches (meshes of different sizes, no padding): the rows of mesh g are cut into tiles of at most
 * 256 rows; tiles is an (ntiles x 3) int64 device table {mesh, first row, rows}, tiles of one mesh consecutive,
 * seg_tile_ptr[nseg + 1] the first tile of every mesh.
 * sn_segment_colsum_ragged_f32 : out[g, c] = scale[g] * sum over the rows r of mesh g of x[r, c]  (fp64, two deterministic
 *                                stages; scale may be NULL).  With scale = 1 / vertex count: global_average on a packed batch.
 * sn_bcast_rows_ragged_f32     : dst[r, :] = src[mesh(r), :]. */
/* sn_avg_prep_ragged_f32        : BatchNorm statistics (2 x 2C fp64, the layout sn_bn_fold_f32 reads) of [e | per-mesh mean
 *                                broadcast] on a packed batch: first half from the statistics partials ([nblk][2][C] fp64) of the
 *                                kernel that wrote e, second half  sum_g len_g m_g,  sum_g len_g m_g^2  from the means and the
 *                                row offsets segoff[nseg + 1]. */
int sn_avg_prep_ragged_f32(const float *seg_mean, const int64_t *segoff, int64_t nseg, int32_t C, const double *stats_part,
                           int32_t nblk, double *stats, void *stream);
size_t sn_segment_colsum_ragged_workspace_bytes(int64_t ntiles, int32_t C);
int sn_segment_colsum_ragged_f32(const float *x, int64_t ld, const int64_t *tiles, int64_t ntiles, const int64_t *seg_tile_ptr,
                                 int64_t nseg, int32_t C, const float *scale, float *out, void *workspace,
                                 size_t workspace_bytes, void *stream);
int sn_bcast_rows_ragged_f32(const float *src, const int64_t *tiles, int64_t ntiles, float *dst, int64_t ldd, int32_t C,
                             void *stream);
size_t sn_segment_colsum_workspace_bytes(int64_t rows_per_seg, int64_t nseg, int32_t C);
int sn_segment_colsum_f32(const float *x, int64_t ld, const float *mask, int64_t rows_per_seg, int64_t nseg, int32_t C,
                          float *out, void *workspace, size_t workspace_bytes, void *stream);
int sn_bcast_rows_f32(const float *src, float *dst, int64_t ldd, int64_t rows_per_seg, int64_t nseg, int32_t C,
                      void *stream);
int sn_elu_bwd_bcast_f32(const float *gdst, int64_t ldg, const float *out, int64_t ldo, const float *bias,
                         const float *mask, const float *gadd, int64_t ldga, float *gsrc, int64_t ldgs,
                         int64_t rows_per_seg, int64_t nseg, int32_t C, void *stream);
int sn_avg_fwd_prep_f32(const float *segsum, const float *inv_count, int64_t nseg, int32_t C, int64_t rows_per_seg,
                        const double *stats1, float *m, double *stats, void *stream);
size_t sn_avg_stats_workspace_bytes(int64_t rows_per_seg, int64_t nseg, int32_t C);
int sn_avg_stats_f32(const float *e, int64_t ld, const float *mask, const float *inv_count, int64_t rows_per_seg, int64_t nseg,
                     int32_t C, float *m, double *stats, void *workspace, size_t workspace_bytes, void *stream);
int sn_seg_affine_f32(const float *A, int64_t nseg, int32_t K, const float *W, int64_t ldw, const float *bias, int32_t J,
                      float *out, void *stream);
int sn_avg_bwd_gc_f32(const float *G1, const float *seg_dy, const float *m, const float *mu2, int64_t nseg, int32_t J,
                      int32_t C, float *Gc, void *stream);
int sn_avg_bwd_segvec_f32(const float *seg_dy, const float *Wf2, int64_t ldw, const float *m, const float *mu2,
                          const float *B2, const float *C2, const float *inv_count, int64_t rows_per_seg, int64_t nseg,
                          int32_t J, int32_t C, float *out, void *stream);
/* ... for ragged meshes: mesh g has segoff[g+1] - segoff[g] rows (device int64[nseg + 1]) in place of rows_per_seg */
int sn_avg_bwd_segvec_ragged_f32(const float *seg_dy, const float *Wf2, int64_t ldw, const float *m, const float *mu2,
                                 const float *B2, const float *C2, const float *inv_count, const int64_t *segoff, int64_t nseg,
                                 int32_t J, int32_t C, float *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Device-side construction of the quaternionic Dirac operators from a triangle mesh.
 *
 * Replaces: mesh.dirac(V, F)                      src/utils/mesh.py:35-64  (dense O(F·V) numpy builder, 8 s at V=1000)
 *           with mesh.dist / mesh.area            src/utils/mesh.py:17-26, 67-80
 *           and Q = quaternion_matrix             src/utils/mesh.py:28-33
 * All geometry is evaluated in fp64 with the reference's operation order (no FMA contraction) and rounded to fp32 at
 * the end, exactly like `D.astype('float32')` in the reference's preprocessing (add_laplacian.py:61-65).
 *
 * Inputs : V (nV x 3, fp32 — the dtype the reference stores in its datasets), F (nF x 3, int32).
 * Outputs: four operators in BSR4 (4x4 blocks, explicit zeros):
 *            Di   (4nF x 4nV): block row = face, 3 blocks, block columns ascending
 *            DiA  (4nV x 4nF): block row = vertex, one block per incident face, block columns ascending
 *            DiT = Diᵀ (structure of DiA), DiAT = DiAᵀ (structure of Di)   — for the backward products
 *          di_rowptr[nF+1], di_colind[3nF], di_vals[48nF], diat_vals[48nF];
 *          dia_rowptr[nV+1], dia_colind[3nF], dia_vals[48nF], dit_vals[48nF]   (DiT shares DiA's index arrays,
 *          DiAT shares Di's).  Faces must reference three distinct vertices.
 * workspace: sn_dirac_workspace_bytes(nV, nF).
 * ------------------------------------------------------------------------------------------ */
size_t sn_dirac_workspace_bytes(int64_t nV, int64_t nF);
int sn_dirac_bsr4_from_mesh(const float *V, const int32_t *F, int64_t nV, int64_t nF,
                            int32_t *di_rowptr, int32_t *di_colind, float *di_vals, float *diat_vals,
                            int32_t *dia_rowptr, int32_t *dia_colind, float *dia_vals, float *dit_vals,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Profiling aid (off by default; the library's only global state, mutex-guarded).  While enabled every SpMM launch
 * (sn_spmm_csr_f32 / sn_spmm_bsr4_f32 and their _elubwd forms) is issued with hipExtLaunchKernelGGL so that the KERNEL's own start and stop are
 * stamped into two events: durations carry no marker / kernel-boundary overhead and agree with rocprofv3's kernel trace.
 * sn_timing_drain waits for the recorded launches, writes up to `capacity` durations (ms) and 5 int64 per record
 * {kind (bit 0: 0 csr, 1 blocked; bit 5: RB4 (4x1 row blocks); bit 3: the blocked form is Q3; bit 4: the launch also left column statistics (sn_spmm_q3_stats_f32); bit 1: fused ELU-backward epilogue, E read; bit 2: G read
 *  too), M, K,
 *  nnz (csr) | nblocks (bsr4), N}, and clears the list.  The Linear-layer launchers (forward, input gradient, weight gradient)
 * record themselves the same way: kind 0x100 / 0x200 / 0x400 + a variant number, then rows, input width, operand bytes, output
 * width.  sn_timing_enable(1): everything; (2): the sparse products only (an event pair costs a launch ~2 us of GPU time); (0): off.
 * ------------------------------------------------------------------------------------------ */
int     sn_timing_enable(int32_t on);
int64_t sn_timing_count(void);
int     sn_timing_drain(double *ms, int64_t *meta, int64_t capacity, int64_t *written);

/* ------------------------------------------------------------------------------------------
 * Device-side construction of the mass-normalised cotangent Laplacian  L = A^-1 (D - W)  in CSR.
 *
 * Replaces: mesh.dist / mesh.area / mesh.cotangent_weights     src/utils/mesh.py:17-26, 67-80, 102-112
 *           graph.laplacian(W, normalized=False)               src/utils/graph.py:40-49
 *           L = A * L                                           src/mesh_mnist/add_laplacian.py:47-48
 * fp64 in the reference's operation order (W[i,j] += (-l_ij^2 + l_jk^2 + l_ki^2)/(8a + 1e-6) per face, A[i] += a/3/4
 * twice per incident face in face order, d = column sums of W in ascending row order), rounded to fp32 at the end.
 * Two phases (the caller owns the allocation): phase 0 writes rowptr[nV+1] (row i holds its distinct neighbours with a
 * non-zero weight plus the diagonal); the caller reads rowptr[nV] = nnz, allocates colind/vals, and calls phase 1.
 * A vertex may have at most SN_LAP_MAX_DEGREE incident faces (else SN_E_UNSUPPORTED is reported through *status_flag).
 * workspace: sn_laplacian_workspace_bytes(nV, nF) — must be the same buffer for both phases.
 * ------------------------------------------------------------------------------------------ */
#define SN_LAP_MAX_DEGREE 24
size_t sn_laplacian_workspace_bytes(int64_t nV, int64_t nF);
int sn_laplacian_csr_from_mesh(const float *V, const int32_t *F, int64_t nV, int64_t nF, int32_t phase,
                               int32_t *rowptr, int32_t *colind, float *vals, int32_t *status_flag,
                               void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Row-streaming fp32 GEMMs of the per-node Linear layers with the weights held in registers
 * (v_mfma_f32_32x32x2_f32; the operands are tall-skinny: rows ~ 1e5..1e6, K and N in {128, 256}).
 *
 * sn_linear_fwd_f32 : y[r, j] = sum_k x[r,k] * W[j,k] + bias[j] (+ residual[r,j]);  optionally also
 *                     y_elu[r, j] = elu(y[r, j]) written to a second destination (e.g. the concat buffer of the next
 *                     stage).  W is (J x K) row-major — the BN-folded weight W·diag(s) of sn_bn_fold_f32.
 *                     Replaces nn.Linear (src/utils/utils_pt.py:89,99) + the residual add (utils_pt.py:180,220) + the
 *                     F.elu that follows (utils_pt.py:171,208).
 * sn_linear_dgrad_f32 : dx[r, c] = sum_j dy[r,j] * W[j,c]  (+ (x[r,c] - center[c]) * B[c] + Cc[c]  when B != NULL):
 *                     input gradient of the folded BatchNorm+Linear with the BatchNorm tail fused in the epilogue
 *                     (replaces the dgrad GEMM + sn_affine_cols_acc_f32).  W is (J x C) row-major.
 * sn_linear_dgrad_elu_f32 : the same input gradient when x is the concat buffer [elu(u) | P·elu(u)] of a residual stage
 *                     (C = 2·C/2 columns): the first C/2 columns continue through the activation in the epilogue,
 *                         gact[r, c] = dx[r, c] * elu'(x[r, c]) + gadd[r, c]      (c < C/2; gadd may be NULL),
 *                     and only the last C/2 columns are written, to dx_hi (rows x C/2) — the operand of the transposed
 *                     sparse product.  B (the BatchNorm tail) is required.  Split-bf16 kernels only
 *                     (SN_E_UNSUPPORTED with SN_GEMM_VARIANT=0: the caller composes the unfused calls).
 * sn_linear_fwd_segbias_f32 : the forward with a PER-MESH bias, y[r, :] = x[r]·W^T + segbias[r / rows_per_seg, :]
 *                     (+ residual) — the half-width global-average stage; y may be NULL when only y_elu is wanted (also
 *                     accepted by sn_linear_fwd_f32).  rows_per_seg >= 32.  Split-bf16 kernels only.
 * sn_linear_dgrad_eluseg_f32 : input gradient through the activation for ALL C columns with a per-mesh vector added before
 *                     the derivative:  gact[r, c] = (dy[r]·W[:, c] + (x[r,c] - center[c]) B[c] + Cc[c] + rowmask[r] *
 *                     segvec[r / rows_per_seg, c]) * elu'(x[r, c]) + gadd[r, c]   (rowmask, gadd may be NULL).
 * sn_linear_dgrad_eluseg_f32 also takes segvec = NULL (then rowmask must be NULL and rows_per_seg is ignored): every column
 * through the activation without a per-mesh vector — the backward of conv(F.elu(v)), the models' last layer.
 * Supported: K in {128, 256} for the forward; C in {128, 256} for the input gradient; J = 128 or, with the split kernels
 * (SN_GEMM_VARIANT != 0), any multiple of 4 up to 128 — the last layer has 120 outputs (models.py:148-150): weights,
 * bias, residual and dy columns past J are never read and nothing is stored past column J of y / y_elu; all leading
 * dimensions multiples of 4 floats, below 2^24 floats (a lane's byte offset inside a 32-row tile is kept in 32 bits),
 * and 16-byte aligned bases (else SN_E_UNSUPPORTED / SN_E_ALIGN: the caller falls back to a library GEMM).
 * The kernels address every streamed matrix through a raw buffer window that ends with its last row: nothing before the
 * first or past the last row of a view, and no column outside it, is read or written (strided views inside larger
 * buffers are safe; tests/test_dense_gpu.py::test_linear_kernels_stay_inside_their_views).
 * elu_stats_part holds sn_linear_fwd_stats_blocks(rows) blocks — the largest grid any forward kernel uses; a kernel with a
 * smaller grid writes zeros into the blocks past its own, so the merge may always read all of them.
 * ------------------------------------------------------------------------------------------ */
int sn_linear_fwd_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *bias,
                      const float *residual, int64_t ldr, float *y, int64_t ldy, float *y_elu, int64_t lde,
                      int64_t rows, int32_t K, int32_t J, double *elu_stats_part /* NULL | [stats_blocks][2][128] */,
                      void *stream);
int sn_linear_dgrad_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                        const float *center, const float *B, const float *Cc, float *dx, int64_t lddx,
                        int64_t rows, int32_t J, int32_t C, void *stream);
int sn_linear_dgrad_elu_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                            const float *center, const float *B, const float *Cc, float *dx_hi, int64_t lddx,
                            float *gact, int64_t ldga, const float *gadd, int64_t ldgadd,
                            int64_t rows, int32_t J, int32_t C, void *stream);
/* The same launch, which also leaves an upper bound of max |gact| — gact is the dy operand of the layer below, and the
 * two-piece weight gradient (sn_wgrad_*_bounded_f32) needs that bound before it reads the first row.  gact_absmax: device,
 * sn_linear_dgrad_absmax_blocks() floats, one maximum per workgroup (entries past the grid zeroed): the consumer takes the
 * maximum of all of them.  No atomics, no zero-fill before the launch; deterministic.  NULL: as the plain entry point. */
int32_t sn_linear_dgrad_absmax_blocks(void);
int sn_linear_dgrad_elu_absmax_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                                   const float *center, const float *B, const float *Cc, float *dx_hi, int64_t lddx,
                                   float *gact, int64_t ldga, const float *gadd, int64_t ldgadd,
                                   int64_t rows, int32_t J, int32_t C, float *gact_absmax, void *stream);
int sn_linear_fwd_segbias_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *segbias,
                              int64_t rows_per_seg, const float *residual, int64_t ldr, float *y, int64_t ldy,
                              float *y_elu, int64_t lde, int64_t rows, int32_t K, int32_t J, double *elu_stats_part,
                              void *stream);
int sn_linear_dgrad_eluseg_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                               const float *center, const float *B, const float *Cc, const float *segvec,
                               int64_t rows_per_seg, const float *rowmask, float *gact, int64_t ldga, const float *gadd,
                               int64_t ldgadd, int64_t rows, int32_t J, int32_t C, void *stream);
/* sn_linear_fwd_tiles_f32 / sn_linear_fwd_segbias_tiles_f32: the two forward launches above, which also leave the column sums of
 * the ACTIVATED output per 32-row tile: tile_sums[(rows + 31) / 32][128] floats (every tile written; J = 128, with y_elu and
 * elu_stats_part).  The half-width global-average stage that consumes y_elu (AvgResNet2, src/utils/utils_pt.py:230-243) then needs
 * no statistics pass over it: sn_avg_stats_from_tiles_f32 forms the per-mesh masked sums from the tiles inside each mesh (rows
 * of tiles shared by two meshes, and of tiles that hold a masked-out row, are read from e) and the BatchNorm sums from the
 * statistics partials — the outputs of sn_avg_stats_f32 (m: nseg x C, stats: 2 x 2C fp64) without reading e
 * (165 MB per stage at the ARAP batch).  workspace: nseg * C floats.  C = 128. */
int sn_linear_fwd_tiles_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *bias, const float *residual,
                            int64_t ldr, float *y, int64_t ldy, float *y_elu, int64_t lde, int64_t rows, int32_t K, int32_t J,
                            double *elu_stats_part, float *tile_sums, void *stream);
int sn_linear_fwd_segbias_tiles_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *segbias,
                                    int64_t rows_per_seg, const float *residual, int64_t ldr, float *y, int64_t ldy, float *y_elu,
                                    int64_t lde, int64_t rows, int32_t K, int32_t J, double *elu_stats_part, float *tile_sums,
                                    void *stream);
int sn_avg_stats_from_tiles_f32(const float *tile_sums, const double *stats_part, int32_t nblk, const float *e, int64_t ld,
                                const float *mask, const float *inv_count, int64_t rows_per_seg, int64_t nseg, int32_t C,
                                float *m, double *stats, float *workspace, void *stream);
/* ... and for RAGGED meshes (mesh g = rows [segoff[g], segoff[g+1]), inv_count[g] = 1 / its row count, no mask): the per-mesh means
 * and the statistics of [e | mean broadcast] without a pass over e; sn_linear_fwd_segbias_ragged_tiles_f32 is the ragged
 * per-mesh-bias launch that leaves the tile sums. */
int sn_avg_stats_from_tiles_ragged_f32(const float *tile_sums, const double *stats_part, int32_t nblk, const float *e, int64_t ld,
                                       const int64_t *segoff, const float *inv_count, int64_t nseg, int32_t C, float *m,
                                       double *stats, float *workspace, void *stream);
int sn_linear_fwd_segbias_ragged_tiles_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *segbias,
                                           const int64_t *segoff, int32_t nseg, const float *residual, int64_t ldr, float *y,
                                           int64_t ldy, float *y_elu, int64_t lde, int64_t rows, int32_t K, int32_t J,
                                           double *elu_stats_part, float *tile_sums, void *stream);
/* The two per-mesh-vector kernels for RAGGED meshes (packed batches): mesh g owns rows [segoff[g], segoff[g+1]) (device
 * int64[nseg + 1], segoff[0] = 0, segoff[nseg] = rows, every mesh at least 32 rows — the caller's responsibility: a device
 * array is not validated here); no row mask (a packed batch has no padding rows). */
int sn_linear_fwd_segbias_ragged_f32(const float *x, int64_t ldx, const float *W, int64_t ldw, const float *segbias,
                                     const int64_t *segoff, int32_t nseg, const float *residual, int64_t ldr, float *y,
                                     int64_t ldy, float *y_elu, int64_t lde, int64_t rows, int32_t K, int32_t J,
                                     double *elu_stats_part, void *stream);
int sn_linear_dgrad_eluseg_ragged_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                                      const float *center, const float *B, const float *Cc, const float *segvec,
                                      const int64_t *segoff, int32_t nseg, float *gact, int64_t ldga, const float *gadd,
                                      int64_t ldgadd, int64_t rows, int32_t J, int32_t C, void *stream);
/* ... and with the bound of |gact| for the two-piece weight gradient of the layer below (see sn_linear_dgrad_elu_absmax_f32:
 * gact_absmax holds sn_linear_dgrad_absmax_blocks() floats, one maximum per workgroup). */
int sn_linear_dgrad_eluseg_absmax_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x, int64_t ldx,
                                      const float *center, const float *B, const float *Cc, const float *segvec,
                                      int64_t rows_per_seg, const float *rowmask, float *gact, int64_t ldga, const float *gadd,
                                      int64_t ldgadd, int64_t rows, int32_t J, int32_t C, float *gact_absmax, void *stream);
int sn_linear_dgrad_eluseg_ragged_absmax_f32(const float *dy, int64_t lddy, const float *W, int64_t ldw, const float *x,
                                             int64_t ldx, const float *center, const float *B, const float *Cc,
                                             const float *segvec, const int64_t *segoff, int32_t nseg, float *gact, int64_t ldga,
                                             const float *gadd, int64_t ldgadd, int64_t rows, int32_t J, int32_t C,
                                             float *gact_absmax, void *stream);

/* ---- launch plans: one host call enqueues a whole residual block ------------------------------------------------------------
 * Replaces, per block, the Python dispatch the reference pays per op: LapResNet2 / DirResNet2 / AvgResNet2.forward
 * (src/utils/utils_pt.py:159-180, 191-220, 230-243) and their autograd backward are sequences of torch ops launched one by one
 * from the interpreter inside the training loops (src/as_rigid_as_possible/main.py:217-232, src/mesh_mnist/main.py:151-167,
 * src/dense_correspondence/main.py:310-327), which cannot be graph-captured by an unmodified driver.
 *
 * A plan is a HOST object holding the launch list of one block direction: entries of this header (by index into the table
 * sn_plan_lookup searches) with their scalar arguments, pointer arguments as (slot, byte offset).  The caller supplies the
 * slots' base addresses per run — slot 0/1 by convention the block's two workspace arenas (ONE allocation each, sized once per
 * shape by the caller), the rest the tensors the block reads or writes — and sn_plan_run calls the recorded entry points in
 * order on `stream`: the same launchers, kernels and grids as calling them one by one, bit-identical results.  The library
 * still never allocates device memory, synchronises or keeps per-shape state: plans are created, owned and destroyed by the
 * caller, immutable while running, usable from several threads / streams at once.
 *   kind[i]: 0 integer (ival), 1 floating point (dval), 2 pointer = slot_base[slot[i]] + ival[i], 3 NULL pointer, 4 the stream
 *   (last argument of every entry point).  sn_plan_add_call checks every kind against the parameter type of the entry point
 *   (SN_E_UNSUPPORTED on a mismatch, SN_E_SHAPE on a wrong argument count).
 *   sn_plan_add_memset / sn_plan_add_copy: a byte fill / a device-to-device copy of a 2-D region (rows x width_bytes, pitches in
 *   bytes; rows = 1: flat) — what torch.zeros / Tensor.copy_ do between the reference's ops.
 *   sn_plan_run: 0, or the status of the first failing entry (its index in *failed_node, else -1); SN_E_NULL when a slot a
 *   recorded pointer refers to is handed over as 0. */
typedef struct sn_plan sn_plan;
int sn_plan_create(sn_plan **out);
int sn_plan_destroy(sn_plan *plan);
int32_t sn_plan_lookup(const char *entry_point_name);
int32_t sn_plan_entry_count(void);
const char *sn_plan_entry_name(int32_t fn);
const char *sn_plan_entry_signature(int32_t fn);        /* one char per parameter: p pointer, i integer, d floating point */
int64_t sn_plan_length(const sn_plan *plan);
int sn_plan_add_call(sn_plan *plan, int32_t fn, int32_t nargs, const int32_t *kind, const int32_t *slot, const int64_t *ival,
                     const double *dval);
int sn_plan_add_memset(sn_plan *plan, int32_t slot, int64_t offset, int32_t byte_value, int64_t pitch, int64_t width_bytes,
                       int64_t rows);
int sn_plan_add_copy(sn_plan *plan, int32_t dst_slot, int64_t dst_offset, int64_t dst_pitch, int32_t src_slot, int64_t src_offset,
                     int64_t src_pitch, int64_t width_bytes, int64_t rows);
int sn_plan_run(const sn_plan *plan, const uint64_t *slot_base, int32_t nslots, void *stream, int32_t *failed_node);

/* A plan at FIXED slot addresses as one graph launch.  In a training loop the addresses of a plan run repeat from step to step (the
 * framework's caching allocator hands the same blocks to the same sequence of requests); sn_plan_instantiate captures the launch
 * list at those addresses into an executable hipGraph (kernel nodes only; on a private stream, thread-local capture mode, nothing
 * runs), sn_plan_exec_launch enqueues it on `stream` with ONE hipGraphLaunch — or walks the list like sn_plan_run when `stream` is
 * itself being captured by the caller or the per-launch timer is on (plan and addresses are passed for that).  Same kernels, same
 * arguments, same order: bit-identical to sn_plan_run.  The object is the caller's (keyed by the addresses it was made for, valid
 * while the memory behind them is), holds no device memory; SN_E_UNSUPPORTED while the timer is on. */
typedef struct sn_plan_exec sn_plan_exec;
int sn_plan_instantiate(const sn_plan *plan, const uint64_t *slot_base, int32_t nslots, sn_plan_exec **out, int32_t *failed_node);
int sn_plan_exec_launch(const sn_plan_exec *exec, const sn_plan *plan, const uint64_t *slot_base, int32_t nslots, void *stream,
                        int32_t *failed_node);
int sn_plan_exec_destroy(sn_plan_exec *exec);

#ifdef __cplusplus
}
#endif
#endif /* SN_SPMM_H_ */
