"""Tensor-level launchers of the C-ABI (include/sn_spmm.h).  PyTorch is plumbing here: it owns the HBM
allocations and the HIP stream; every byte of the hot path is moved by the kernels in csrc/sn_kernels.hip.

All functions require device ("cuda" = HIP under PyTorch-ROCm) tensors and raise otherwise — there is no
CPU path in the product.  Launches go to torch's current stream and never synchronise, except
csr_to_bsr4 which must read one integer (the block count) back to size its outputs.
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = [
    "note_absmax", "take_absmax", "absmax_wanted", "clear_absmax", "tile_sums_supported", "new_tile_sums", "avg_stats_from_tiles", "spmm_csr", "spmm_bsr4", "spmm_csr_elubwd", "spmm_bsr4_elubwd", "spmm_q3", "spmm_q3_stats", "spmm_q3_stats_supported", "spmm_csr_stats", "spmm_csr_stats_supported", "csr_to_rb4", "spmm_rb4", "spmm_rb4_stats", "spmm_rb4_supported", "spmm_ring", "spmm_ring_stats", "spmm_ring_supported", "ring_half_window", "csr_band", "bsr4_to_q3", "coo_to_csr", "csr_transpose", "csr_to_bsr4", "blockdiag_concat", "blockdiag_concat_ragged", "validate_csr",
    "elu_into", "elu_bwd", "colstats", "wgrad", "wgrad_supported", "affine_cols_acc", "affine_cols_elu_bwd",
    "bn_fold", "bn_bwd_coeffs", "segment_colsum", "bcast_rows", "segment_colsum_ragged", "bcast_rows_ragged", "elu_bwd_bcast", "dirac_from_mesh", "laplacian_from_mesh", "linear_fwd", "linear_fwd_supported", "linear_dgrad",
    "linear_dgrad_supported", "linear_dgrad_elu", "linear_dgrad_elu_supported",
    "avg_stage_supported", "avg_fwd_prep", "avg_stats", "seg_affine", "avg_bwd_gc", "avg_bwd_segvec", "linear_fwd_segbias", "linear_dgrad_eluseg", "wgrad_seg", "wgrad_slabs", "avg_bwd_segvec_ragged", "linear_fwd_segbias_ragged", "linear_dgrad_eluseg_ragged", "avg_stage_ragged_supported", "wgrad_thin", "wgrad_thin_supported", "linear_thin_fwd", "masked_smooth_l1_fwd", "masked_smooth_l1_bwd", "gather_segments", "pair_argmin", "pair_ce_fwd", "pair_ce_bwd", "pair_fused_fwd", "pair_fused_bwd", "colstats_partial", "linear_fwd_stats_blocks", "elu_stats_supported", "new_elu_stats_part", "colstats_halves", "colstats_into", "colstats_from_part", "colstats_merge_into",
    "avg_merged_supported", "avg_stats_ragged", "avg_stats_from_tiles_ragged", "gather_segments_ragged",
]


def _dev(*tensors) -> None:
    if _lib._recorder is not None and _lib.recorder() is not None and _lib._recorder.allow_cpu:
        return                                # (tests record — never run — launch plans on CPU tensors)
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "surfacenetworks_amd kernels run on the MI355X only: got a CPU tensor. There is no CPU/eager "
                "fallback in this package (the reference's CPU torch.sparse path lives in oracle/ for tests).")


def _p(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  Through the two C entry points behind
    torch.cuda.current_stream() when this torch has them: the Stream-object route costs ~8 us per call on the host, as
    much as everything else around a launch."""
    if _lib._recorder is not None and _lib.recorder() is not None:      # a launch plan is being recorded: the stream is an
        return 0                                                         # argument of sn_plan_run
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _ld(t: torch.Tensor) -> int:
    """Leading dimension of a 2-D view whose rows are contiguous."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"expected a 2-D tensor with unit column stride, got shape {tuple(t.shape)} stride {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def _check_dense(x: torch.Tensor, rows: int, group: int, N: int, what: str) -> int:
    if x.dtype != torch.float32:
        raise TypeError(f"{what} must be float32")
    if rows % group or x.shape[0] != rows // group or x.shape[1] != group * N:
        raise ValueError(f"{what}: expected ({rows // group}, {group * N}) for {rows} operator rows in groups of {group}, got {tuple(x.shape)}")
    return _ld(x)


def spmm_csr(rowptr, colind, vals, M: int, K: int, x, y, group: int = 1) -> None:
    """y <- A·x.  x: (K/group, group*N) and y: (M/group, group*N) 2-D views with contiguous rows
    (any row stride), e.g. halves of a (rows, 2C) concat buffer.  group=4 is the quaternion view."""
    _dev(rowptr, colind, vals, x, y)
    N = y.shape[1] // group
    ldx = _check_dense(x, K, group, N, "x")
    ldy = _check_dense(y, M, group, N, "y")
    _lib.call("sn_spmm_csr_f32", _p(rowptr), _p(colind), _p(vals), M, K, int(colind.numel()),
              _p(x), ldx, group, N, _p(y), ldy, group, _stream())


def spmm_csr_stats_supported(N: int, group: int) -> bool:
    return N == 128 and group == 1


def spmm_csr_stats(rowptr, colind, vals, M: int, K: int, x, y):
    """y <- A·x as spmm_csr (N = 128, plain row-major operands) and the partial column statistics of y: returns the
    (blocks, 2, 128) float64 partials that colstats_halves merges (sn_spmm_csr_stats_f32)."""
    _dev(rowptr, colind, vals, x, y)
    N = y.shape[1]
    if not spmm_csr_stats_supported(N, 1):
        raise ValueError("spmm_csr_stats: 128-column operands only")
    ldx = _check_dense(x, K, 1, N, "x")
    ldy = _check_dense(y, M, 1, N, "y")
    lib = _lib.load()
    part = torch.empty((int(lib.sn_spmm_q3_stats_blocks()), 2, 128), dtype=torch.float64, device=y.device)
    ws_bytes = int(lib.sn_spmm_csr_stats_workspace_bytes(M))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=y.device)
    _lib.call("sn_spmm_csr_stats_f32", _p(rowptr), _p(colind), _p(vals), M, K, int(colind.numel()), _p(x), ldx, 1, N, _p(y), ldy, 1,
              _p(part), _p(ws), ws_bytes, _stream())
    return part


def csr_to_rb4(rowptr, colind, vals, M: int, K: int):
    """CSR -> RB4 (4x1 row blocks): (b_ptr [ceil(M/4)+1], b_col [nnz], b_val [nnz, 4]).  No synchronisation: the arrays are
    sized by nnz, an upper bound of the listed-column total (the tail past b_ptr[-1] is never read)."""
    _dev(rowptr, colind, vals)
    dev = rowptr.device
    Mb = (M + 3) // 4
    nnz = int(colind.numel())
    b_ptr = torch.empty(Mb + 1, dtype=torch.int32, device=dev)
    ws_bytes = int(_lib.load().sn_scan_workspace_bytes(Mb + 1))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _lib.call("sn_rb4_count", _p(rowptr), _p(colind), M, K, _p(b_ptr), _p(ws), ws_bytes, _stream())
    b_col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)[:nnz]
    b_val = torch.empty((max(nnz, 1), 4), dtype=torch.float32, device=dev)[:nnz]
    if nnz:
        _lib.call("sn_rb4_fill", _p(rowptr), _p(colind), _p(vals), M, K, _p(b_ptr), _p(b_col), _p(b_val), _stream())
    return b_ptr, b_col, b_val


def spmm_rb4_supported(N: int, group: int) -> bool:
    return group == 1 and N in (64, 128)


def spmm_rb4(b_ptr, b_col, b_val, M: int, K: int, x, y, e=None, g=None, want_absmax: bool = False):
    """y <- A·x for an RB4 operator (sn_spmm_rb4_f32); with e: (A·x) * elu'(e) + g (sn_spmm_rb4_elubwd_f32).
    want_absmax (with e): also returns the per-wave maxima of |y| (sn_spmm_rb4_elubwd_absmax_f32), else None."""
    _dev(b_ptr, b_col, b_val, x, y, e, g)
    N = y.shape[1]
    ldx = _check_dense(x, K, 1, N, "x")
    ldy = _check_dense(y, M, 1, N, "y")
    cap = int(b_col.numel())
    if e is None:
        _lib.call("sn_spmm_rb4_f32", _p(b_ptr), _p(b_col), _p(b_val), M, K, cap, _p(x), ldx, N, _p(y), ldy, _stream())
    else:
        lde = _check_dense(e, M, 1, N, "e")
        ldg = _check_dense(g, M, 1, N, "g") if g is not None else 0
        if want_absmax and M > 0:
            am = torch.empty(int(_lib.load().sn_spmm_rb4_absmax_blocks(M, N)), dtype=torch.float32, device=y.device)
            _lib.call("sn_spmm_rb4_elubwd_absmax_f32", _p(b_ptr), _p(b_col), _p(b_val), M, K, cap, _p(x), ldx, N, _p(e), lde, _p(g),
                      ldg, _p(y), ldy, _p(am), _stream())
            return am
        _lib.call("sn_spmm_rb4_elubwd_f32", _p(b_ptr), _p(b_col), _p(b_val), M, K, cap, _p(x), ldx, N, _p(e), lde, _p(g), ldg,
                  _p(y), ldy, _stream())
    return None


def spmm_rb4_stats(b_ptr, b_col, b_val, M: int, K: int, x, y):
    """spmm_rb4 (N = 128) that also returns the (blocks, 2, 128) float64 partial column statistics of y."""
    _dev(b_ptr, b_col, b_val, x, y)
    N = y.shape[1]
    if N != 128:
        raise ValueError("spmm_rb4_stats: 128-column operands only")
    ldx = _check_dense(x, K, 1, N, "x")
    ldy = _check_dense(y, M, 1, N, "y")
    lib = _lib.load()
    part = torch.empty((int(lib.sn_spmm_q3_stats_blocks()), 2, 128), dtype=torch.float64, device=y.device)
    ws_bytes = int(lib.sn_spmm_rb4_stats_workspace_bytes(M))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=y.device)
    _lib.call("sn_spmm_rb4_stats_f32", _p(b_ptr), _p(b_col), _p(b_val), M, K, int(b_col.numel()), _p(x), ldx, N, _p(y), ldy,
              _p(part), _p(ws), ws_bytes, _stream())
    return part


# Rows below which the ring kernel's persistent strips are mostly prologue (one workgroup per CU, >= 8 steps of 64 rows each)
RING_MIN_ROWS = 131072
# Largest share of rows with an entry outside the window (they gather from global memory inside the kernel: closed meshes'
# wrap-around rows) for which the ring kernel is still the faster one (a 65 x 106 torus batch has 3 %: 0.67-0.83 of the
# roofline against the row-blocked kernel's 0.59-0.69)
RING_MAX_OUTSIDE = 0.05


def spmm_ring_supported(N: int, group: int, M: int, K: int) -> bool:
    from . import kernels as _self      # (RING_MIN_ROWS is read through the module so that tests / tools can lower it)

    return group == 1 and N in (64, 128) and M == K and M >= _self.RING_MIN_ROWS


def ring_half_window() -> int:
    return int(_lib.load().sn_spmm_csr_ring_half_window())


def csr_band(rowptr, colind, M: int, K: int):
    """(max |column - row|, longest row, rows with an entry outside the ring kernel's window) of a CSR operator — reads
    three ints back from the device (once per operator)."""
    _dev(rowptr, colind)
    out = torch.empty(3, dtype=torch.int32, device=rowptr.device)
    _lib.call("sn_csr_band_i32", _p(rowptr), _p(colind), M, K, _p(out), _stream())
    band, longest, outside = out.tolist()
    return int(band), int(longest), int(outside)


def spmm_ring(rowptr, colind, vals, M: int, K: int, x, y, e=None, g=None, want_absmax: bool = False):
    """y <- A·x for a banded square CSR operator through the sliding-window kernel (sn_spmm_csr_ring_f32); with e:
    (A·x) * elu'(e) + g (sn_spmm_csr_ring_elubwd_f32).  Bit-identical to spmm_csr.
    want_absmax (with e): also returns the per-wave maxima of |y| (sn_spmm_csr_ring_elubwd_absmax_f32), else None."""
    _dev(rowptr, colind, vals, x, y, e, g)
    N = y.shape[1]
    ldx = _check_dense(x, K, 1, N, "x")
    ldy = _check_dense(y, M, 1, N, "y")
    nnz = int(colind.numel())
    if e is None:
        _lib.call("sn_spmm_csr_ring_f32", _p(rowptr), _p(colind), _p(vals), M, K, nnz, _p(x), ldx, N, _p(y), ldy, _stream())
    else:
        lde = _check_dense(e, M, 1, N, "e")
        ldg = _check_dense(g, M, 1, N, "g") if g is not None else 0
        if want_absmax and M > 0 and nnz > 0:
            am = torch.empty(int(_lib.load().sn_spmm_csr_ring_absmax_blocks(M, N)), dtype=torch.float32, device=y.device)
            _lib.call("sn_spmm_csr_ring_elubwd_absmax_f32", _p(rowptr), _p(colind), _p(vals), M, K, nnz, _p(x), ldx, N, _p(e), lde,
                      _p(g), ldg, _p(y), ldy, _p(am), _stream())
            return am
        _lib.call("sn_spmm_csr_ring_elubwd_f32", _p(rowptr), _p(colind), _p(vals), M, K, nnz, _p(x), ldx, N, _p(e), lde, _p(g), ldg,
                  _p(y), ldy, _stream())
    return None


def spmm_ring_stats(rowptr, colind, vals, M: int, K: int, x, y):
    """spmm_ring (N = 128) that also returns the (blocks, 2, 128) float64 partial column statistics of y."""
    _dev(rowptr, colind, vals, x, y)
    N = y.shape[1]
    if N != 128:
        raise ValueError("spmm_ring_stats: 128-column operands only")
    ldx = _check_dense(x, K, 1, N, "x")
    ldy = _check_dense(y, M, 1, N, "y")
    lib = _lib.load()
    part = torch.empty((int(lib.sn_spmm_q3_stats_blocks()), 2, 128), dtype=torch.float64, device=y.device)
    ws_bytes = int(lib.sn_spmm_csr_ring_stats_workspace_bytes(M))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=y.device)
    _lib.call("sn_spmm_csr_ring_stats_f32", _p(rowptr), _p(colind), _p(vals), M, K, int(colind.numel()), _p(x), ldx, N, _p(y), ldy,
              _p(part), _p(ws), ws_bytes, _stream())
    return part


def spmm_bsr4(b_rowptr, b_colind, b_vals, Mb: int, Kb: int, x, y, group: int = 1) -> None:
    _dev(b_rowptr, b_colind, b_vals, x, y)
    N = y.shape[1] // group
    ldx = _check_dense(x, 4 * Kb, group, N, "x")
    ldy = _check_dense(y, 4 * Mb, group, N, "y")
    _lib.call("sn_spmm_bsr4_f32", _p(b_rowptr), _p(b_colind), _p(b_vals), Mb, Kb, int(b_colind.numel()),
              _p(x), ldx, group, N, _p(y), ldy, group, _stream())


def spmm_csr_elubwd(rowptr, colind, vals, M: int, K: int, x, e, g, y, group: int = 1) -> None:
    """y <- (A·x) * elu'(e) + g with elu' taken from the activation output e (sn_spmm_csr_elubwd_f32); g may be None."""
    _dev(rowptr, colind, vals, x, e, g, y)
    N = y.shape[1] // group
    ldx = _check_dense(x, K, group, N, "x")
    ldy = _check_dense(y, M, group, N, "y")
    lde = _check_dense(e, M, group, N, "e")
    ldg = _check_dense(g, M, group, N, "g") if g is not None else 0
    _lib.call("sn_spmm_csr_elubwd_f32", _p(rowptr), _p(colind), _p(vals), M, K, int(colind.numel()),
              _p(x), ldx, group, N, _p(e), lde, _p(g), ldg, _p(y), ldy, group, _stream())


def spmm_bsr4_elubwd(b_rowptr, b_colind, b_vals, Mb: int, Kb: int, x, e, g, y, group: int = 1) -> None:
    _dev(b_rowptr, b_colind, b_vals, x, e, g, y)
    N = y.shape[1] // group
    ldx = _check_dense(x, 4 * Kb, group, N, "x")
    ldy = _check_dense(y, 4 * Mb, group, N, "y")
    lde = _check_dense(e, 4 * Mb, group, N, "e")
    ldg = _check_dense(g, 4 * Mb, group, N, "g") if g is not None else 0
    _lib.call("sn_spmm_bsr4_elubwd_f32", _p(b_rowptr), _p(b_colind), _p(b_vals), Mb, Kb, int(b_colind.numel()),
              _p(x), ldx, group, N, _p(e), lde, _p(g), ldg, _p(y), ldy, group, _stream())


def spmm_q3(b_rowptr, q_blk, Mb: int, Kb: int, x, y, group: int = 1, e=None, g=None, want_absmax: bool = False):
    """y <- A·x for a quaternion-packed Dirac operator (sn_spmm_q3_f32); with e: (A·x) * elu'(e) + g (sn_spmm_q3_elubwd_f32).
    want_absmax (with e): also returns the per-workgroup maxima of |y| (sn_spmm_q3_elubwd_absmax_f32), else None."""
    _dev(b_rowptr, q_blk, x, y, e, g)
    N = y.shape[1] // group
    ldx = _check_dense(x, 4 * Kb, group, N, "x")
    ldy = _check_dense(y, 4 * Mb, group, N, "y")
    nblk = int(q_blk.shape[0])
    if e is None:
        _lib.call("sn_spmm_q3_f32", _p(b_rowptr), _p(q_blk), Mb, Kb, nblk, _p(x), ldx, group, N, _p(y), ldy, group, _stream())
    else:
        lde = _check_dense(e, 4 * Mb, group, N, "e")
        ldg = _check_dense(g, 4 * Mb, group, N, "g") if g is not None else 0
        if want_absmax and Mb > 0:
            am = torch.empty(int(_lib.load().sn_spmm_q3_absmax_blocks(Mb, N)), dtype=torch.float32, device=y.device)
            _lib.call("sn_spmm_q3_elubwd_absmax_f32", _p(b_rowptr), _p(q_blk), Mb, Kb, nblk, _p(x), ldx, group, N, _p(e), lde, _p(g),
                      ldg, _p(y), ldy, group, _p(am), _stream())
            return am
        _lib.call("sn_spmm_q3_elubwd_f32", _p(b_rowptr), _p(q_blk), Mb, Kb, nblk, _p(x), ldx, group, N, _p(e), lde, _p(g), ldg,
                  _p(y), ldy, group, _stream())
    return None


def spmm_q3_stats_supported(N: int, group: int) -> bool:
    return N in (16, 32) and group == 4


def spmm_q3_stats(b_rowptr, q_blk, Mb: int, Kb: int, x, y, group: int = 4):
    """y <- A·x as spmm_q3, and the partial column statistics of y viewed as (Mb, 128): returns the (blocks, 2, 128) float64
    partials that colstats_halves / colstats_from_part-style merges combine (sn_spmm_q3_stats_f32)."""
    _dev(b_rowptr, q_blk, x, y)
    N = y.shape[1] // group
    if not spmm_q3_stats_supported(N, group):
        raise ValueError("spmm_q3_stats: 128- or 64-channel operands in the group-4 layout only")
    ldx = _check_dense(x, 4 * Kb, group, N, "x")
    ldy = _check_dense(y, 4 * Mb, group, N, "y")
    lib = _lib.load()
    part = torch.empty((int(lib.sn_spmm_q3_stats_blocks()), 2, 4 * N), dtype=torch.float64, device=y.device)
    ws_bytes = int(lib.sn_spmm_q3_stats_workspace_bytes(Mb))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=y.device)
    _lib.call("sn_spmm_q3_stats_f32", _p(b_rowptr), _p(q_blk), Mb, Kb, int(q_blk.shape[0]), _p(x), ldx, group, N, _p(y), ldy,
              group, _p(part), _p(ws), ws_bytes, _stream())
    return part


def bsr4_to_q3(b_colind, b_vals):
    """(q_blk (nblocks, 4) fp32 with the block column in the 4th word's bits, flag (1,) int32 device tensor: 1 if some block
    is not a pure-quaternion matrix) — sn_bsr4_to_q3_f32.  The flag is left on the device: the caller decides when to read it."""
    _dev(b_colind, b_vals)
    nblk = int(b_colind.numel())
    q = torch.empty((nblk, 4), dtype=torch.float32, device=b_colind.device)
    flag = torch.empty(1, dtype=torch.int32, device=b_colind.device)
    _lib.call("sn_bsr4_to_q3_f32", _p(b_colind), _p(b_vals), nblk, _p(q), _p(flag), _stream())
    return q, flag


def coo_to_csr(idx_batch, idx_row, idx_col, B: int, R: int, Kb: int):
    """Sorted int64 COO index rows -> (rowptr int32 [B*R+1], colind int32 [nnz]) of the block-diagonal operator."""
    _dev(idx_batch, idx_row, idx_col)
    nnz = int(idx_row.numel())
    dev = idx_row.device
    rowptr = torch.empty(B * R + 1, dtype=torch.int32, device=dev)
    colind = torch.empty(nnz, dtype=torch.int32, device=dev)
    ib = None if idx_batch is None else idx_batch.contiguous()
    _lib.call("sn_coo_to_csr_i32", _p(ib), _p(idx_row.contiguous()), _p(idx_col.contiguous()), nnz, B, R, Kb,
              _p(rowptr), _p(colind), _stream())
    return rowptr, colind


def csr_transpose(rowptr, colind, vals, M: int, K: int):
    _dev(rowptr, colind, vals)
    nnz = int(colind.numel())
    dev = rowptr.device
    t_rowptr = torch.empty(K + 1, dtype=torch.int32, device=dev)
    t_colind = torch.empty(nnz, dtype=torch.int32, device=dev)
    t_vals = torch.empty(nnz, dtype=torch.float32, device=dev)
    ws_bytes = int(_lib.load().sn_csr_transpose_workspace_bytes(M, K, nnz))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _lib.call("sn_csr_transpose_f32", _p(rowptr), _p(colind), _p(vals), M, K, nnz, _p(t_rowptr), _p(t_colind),
              _p(t_vals), _p(ws), ws_bytes, _stream())
    return t_rowptr, t_colind, t_vals


def csr_to_bsr4(rowptr, colind, vals, M: int, K: int):
    """CSR -> 4x4-block BSR.  Synchronises once (reads the block count)."""
    _dev(rowptr, colind, vals)
    if M % 4 or K % 4:
        raise ValueError("BSR4 needs M and K to be multiples of 4")
    dev = rowptr.device
    Mb = M // 4
    b_rowptr = torch.empty(Mb + 1, dtype=torch.int32, device=dev)
    ws_bytes = int(_lib.load().sn_scan_workspace_bytes(Mb + 1))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _lib.call("sn_bsr4_count", _p(rowptr), _p(colind), M, K, _p(b_rowptr), _p(ws), ws_bytes, _stream())
    nblocks = int(b_rowptr[-1].item())
    b_colind = torch.empty(nblocks, dtype=torch.int32, device=dev)
    b_vals = torch.empty(max(nblocks, 1) * 16, dtype=torch.float32, device=dev)[: nblocks * 16]
    _lib.call("sn_bsr4_fill", _p(rowptr), _p(colind), _p(vals), M, K, _p(b_rowptr), _p(b_colind), _p(b_vals), _stream())
    return b_rowptr, b_colind, b_vals


def blockdiag_concat(pool_rowptr, pool_colind, pool_vals, desc, size0: int, size1: int, total: int, vpe: int = 1):
    """Batch assembly from the resident pool; `desc` is the (B,4) int64 table of include/sn_spmm.h (device)."""
    _dev(pool_rowptr, pool_colind, pool_vals, desc)
    B = int(desc.shape[0])
    dev = pool_rowptr.device
    out_rowptr = torch.empty(B * size0 + 1, dtype=torch.int32, device=dev)
    out_colind = torch.empty(total, dtype=torch.int32, device=dev) if vpe != 4 else None      # Q3 records carry the column
    out_vals = torch.empty(total * vpe, dtype=torch.float32, device=dev)
    _lib.call("sn_blockdiag_concat_i32", _p(pool_rowptr), _p(pool_colind), _p(pool_vals), _p(desc), B, size0, size1,
              total, vpe, _p(out_rowptr), _p(out_colind), _p(out_vals), _stream())
    return out_rowptr, out_colind, out_vals


def blockdiag_concat_ragged(pool_rowptr, pool_colind, pool_vals, desc, total_rows: int, total_cols: int, total: int, vpe: int = 1):
    """Packed (unpadded) batch assembly; `desc` is the (B,6) int64 table of sn_blockdiag_concat_ragged_i32 (device)."""
    _dev(pool_rowptr, pool_colind, pool_vals, desc)
    B = int(desc.shape[0])
    dev = pool_rowptr.device
    out_rowptr = torch.empty(total_rows + 1, dtype=torch.int32, device=dev)
    out_colind = torch.empty(total, dtype=torch.int32, device=dev) if vpe != 4 else None      # Q3 records carry the column
    out_vals = torch.empty(total * vpe, dtype=torch.float32, device=dev)
    _lib.call("sn_blockdiag_concat_ragged_i32", _p(pool_rowptr), _p(pool_colind), _p(pool_vals), _p(desc), B, total_rows,
              total_cols, total, vpe, _p(out_rowptr), _p(out_colind), _p(out_vals), _stream())
    return out_rowptr, out_colind, out_vals


def validate_csr(rowptr, colind, vals, M: int, K: int) -> int:
    """Bit set of defects of a CSR operator (0 = well formed), see sn_validate_csr_i32.  Synchronises (reads the flags)."""
    _dev(rowptr, colind, vals)
    flags = torch.empty(1, dtype=torch.int32, device=rowptr.device)
    _lib.call("sn_validate_csr_i32", _p(rowptr), _p(colind), _p(vals), M, K, int(colind.numel()), _p(flags), _stream())
    return int(flags.item())


def elu_into(src, dst) -> None:
    """dst <- elu(src); both 2-D (rows, C) views with contiguous rows (dst may be half of a concat buffer)."""
    _dev(src, dst)
    if src.shape != dst.shape:
        raise ValueError("elu_into: shape mismatch")
    _lib.call("sn_elu_into_f32", _p(src), _ld(src), _p(dst), _ld(dst), src.shape[0], src.shape[1], _stream())


def elu_bwd(gdst, out, gsrc, accumulate: bool, gdst2=None, gadd=None) -> None:
    """gsrc (+)= (gdst + gdst2) * elu'(.) + gadd, the derivative expressed through the activation output `out`."""
    _dev(gdst, out, gsrc, gdst2, gadd)
    for t in (gdst2, gadd):
        if t is not None and t.shape != out.shape:
            raise ValueError("elu_bwd: shape mismatch")
    if not (gdst.shape == out.shape == gsrc.shape):
        raise ValueError("elu_bwd: shape mismatch")
    _lib.call("sn_elu_bwd_acc_f32", _p(gdst), _ld(gdst), _p(gdst2), _ld(gdst2) if gdst2 is not None else 0, _p(gadd),
              _ld(gadd) if gadd is not None else 0, _p(out), _ld(out), _p(gsrc), _ld(gsrc), out.shape[0], out.shape[1],
              1 if accumulate else 0, _stream())


def colstats(x):
    """(2, C) float64: column sums and column sums of squares of the 2-D view x (contiguous rows, any row stride)."""
    _dev(x)
    rows, C = x.shape
    out = torch.empty((2, C), dtype=torch.float64, device=x.device)
    ws_bytes = int(_lib.load().sn_colstats_workspace_bytes(rows, C))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    _lib.call("sn_colstats_f32", _p(x), _ld(x), rows, C, _p(out), _p(ws), ws_bytes, _stream())
    return out


def wgrad_supported(J: int, C: int) -> bool:
    return C in (128, 256) and J <= 128 and J % 4 == 0


# ---- bounds of gradient tensors (max |dy|) for the two-piece weight gradient ------------------------------------------------
# A kernel that PRODUCES a gradient tensor (the input-gradient GEMM through the activation, the transposed quaternion product
# with the fused ELU backward) also leaves the per-workgroup maxima of what it wrote; the weight gradient of the layer below
# reads that tensor as its dy operand and needs the bound before its first row (sn_wgrad_*_bounded_f32).  Between the two
# lie autograd's view nodes (a block returns (rows, C), the next one receives (B, V, C).reshape(...): new tensor objects, same
# memory), so the bound travels in a small table keyed by the data pointer.  An entry holds a strong reference to the tensor it
# describes — its memory cannot be handed to another tensor while the entry lives — and the tensor's version counter (an
# in-place edit invalidates it); only the last few entries are kept (a gradient is consumed within a few launches).
_ABSMAX_KEEP = 8
_absmax_table = {}          # data_ptr -> (tensor, version, maxima)
_absmax_enabled = None


def absmax_wanted() -> bool:
    """Whether producers of gradient tensors should leave their maxima: asked of the library once (sn_wgrad_bounded_enabled —
    the one place that decides whether the bounded two-piece weight gradient runs)."""
    global _absmax_enabled
    if _absmax_enabled is None:
        _absmax_enabled = bool(_lib.load().sn_wgrad_bounded_enabled())
    return _absmax_enabled


def note_absmax(t, maxima) -> None:
    """Record that max |t| <= max(maxima) (a device tensor of per-workgroup maxima left by the kernel that wrote t)."""
    if maxima is None:
        return
    if len(_absmax_table) >= _ABSMAX_KEEP:
        for k in list(_absmax_table)[: len(_absmax_table) - _ABSMAX_KEEP + 1]:      # dicts keep insertion order: oldest first
            del _absmax_table[k]
    _absmax_table.pop(t.data_ptr(), None)
    _absmax_table[t.data_ptr()] = (t, t._version if not t.is_inference() else None, maxima)
    rec = _lib.recorder() if _lib._recorder is not None else None
    if rec is not None:                     # (a launch plan re-attaches what is still noted when its dry run ends)
        rec.notes.append((t, maxima))


def clear_absmax() -> None:
    """Drop every recorded bound (and with it the references to the gradient tensors they describe): called by the train steps
    once a backward pass has run — bounds nobody took (a gradient consumed only as an added term, the gradient into the first
    layer) would otherwise keep up to _ABSMAX_KEEP gradient tensors alive across steps."""
    _absmax_table.clear()


def take_absmax(t):
    """The maxima recorded for exactly this memory (same first element, element count and contents), or None."""
    ent = _absmax_table.pop(t.data_ptr(), None)
    if ent is None:
        return None
    src, version, maxima = ent
    rec = _lib.recorder() if _lib._recorder is not None else None
    if rec is not None:
        rec.notes[:] = [n for n in rec.notes if n[1] is not maxima]
    if src.numel() != t.numel() or not t.is_contiguous() or src.dtype != t.dtype or \
            (version is not None and src._version != version):
        return None
    return maxima


def _bounds_args(bounds, C: int):
    """(dybound (n,) fp32, invstd (C,) fp32, stat_rows) -> the trailing arguments of the sn_wgrad_*_bounded_f32 calls."""
    dyb, invstd, stat_rows = bounds
    _dev(dyb, invstd)
    if dyb.dtype != torch.float32 or not dyb.is_contiguous() or invstd.dtype != torch.float32 or invstd.numel() != C or \
            not invstd.is_contiguous():
        raise ValueError("wgrad bounds: fp32 maxima of |dy| (contiguous) and the (C,) fp32 inverse standard deviations of x")
    return _p(dyb), int(dyb.numel()), _p(invstd), int(stat_rows)


def wgrad(dy, x, center=None, want_colsum: bool = False, bounds=None):
    """G = dy^T · (x - center) (J x C, fp32) for tall-skinny operands on the 16-bit matrix pipe (exact splits); see sn_wgrad_f32.
    want_colsum: also return colsum(dy) as a (J,) float64 tensor, accumulated by the same pass.
    bounds = (dybound, invstd, stat_rows): an upper bound of max |dy| (device, one float), BatchNorm's inverse standard
    deviations of x's columns about `center` and the row count behind them — the product then runs on two fp16 pieces
    (sn_wgrad_bounded_f32: half the matrix work of the three-piece bf16 form)."""
    _dev(dy, x, center)
    rows, J = dy.shape
    C = x.shape[1]
    if x.shape[0] != rows:
        raise ValueError("wgrad: row mismatch")
    G = torch.empty((J, C), dtype=torch.float32, device=x.device)
    dysum = torch.empty(J, dtype=torch.float64, device=x.device) if want_colsum else None
    ws_bytes = int(_lib.load().sn_wgrad_workspace_bytes(rows, J, C))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    if bounds is not None:
        _lib.call("sn_wgrad_bounded_f32", _p(dy), _ld(dy), _p(x), _ld(x), _p(center), rows, J, C, _p(G), _p(dysum), _p(ws), ws_bytes,
                  *_bounds_args(bounds, C), _stream())
    else:
        _lib.call("sn_wgrad_f32", _p(dy), _ld(dy), _p(x), _ld(x), _p(center), rows, J, C, _p(G), _p(dysum), _p(ws), ws_bytes,
                  _stream())
    return (G, dysum) if want_colsum else G


def affine_cols_acc(dx, x, B, Cc, center=None) -> None:
    """dx[r,c] += (x[r,c] - center[c])*B[c] + Cc[c] in place."""
    _dev(dx, x, B, Cc, center)
    if dx.shape != x.shape or B.numel() != x.shape[1] or Cc.numel() != x.shape[1]:
        raise ValueError("affine_cols_acc: shape mismatch")
    _lib.call("sn_affine_cols_acc_f32", _p(dx), _ld(dx), _p(x), _ld(x), _p(center), _p(B), _p(Cc), x.shape[0], x.shape[1], _stream())


def affine_cols_elu_bwd(dx, x, B=None, Cc=None, center=None) -> None:
    """dx[r,c] = (dx[r,c] + (x[r,c] - center[c])*B[c] + Cc[c]) * elu'(.) in place, x = the ELU OUTPUT that fed the layer
    (sn_affine_cols_elu_bwd_f32); B = Cc = None: only the activation derivative."""
    _dev(dx, x, B, Cc, center)
    if dx.shape != x.shape or (B is not None and (B.numel() != x.shape[1] or Cc is None or Cc.numel() != x.shape[1])):
        raise ValueError("affine_cols_elu_bwd: shape mismatch")
    _lib.call("sn_affine_cols_elu_bwd_f32", _p(dx), _ld(dx), _p(x), _ld(x), _p(center), _p(B), _p(Cc), x.shape[0], x.shape[1],
              _stream())


def bn_fold(stats, rows: int, gamma, beta, W, b, eps: float, momentum: float, training: bool, running_mean, running_var,
            num_batches_tracked=None):
    """Fold BatchNorm into the Linear weights (see sn_bn_fold_f32). Returns (mean, invstd, s, t, Wf, bf); updates the
    running statistics (and the int64 batch counter, when given) in place when training."""
    _dev(stats, gamma, beta, W, b, running_mean, running_var, num_batches_tracked)
    if num_batches_tracked is not None and num_batches_tracked.dtype != torch.int64:
        raise TypeError("num_batches_tracked must be int64")
    J, C = W.shape
    dev = W.device
    vec = torch.empty((4, C), dtype=torch.float32, device=dev)
    Wf = torch.empty((J, C), dtype=torch.float32, device=dev)
    bf = torch.empty(J, dtype=torch.float32, device=dev)
    _lib.call("sn_bn_fold_f32", _p(stats), rows, _p(gamma), _p(beta), _p(W.contiguous()), _p(b), J, C, float(eps),
              float(momentum), 1 if training else 0, _p(running_mean), _p(running_var), _p(vec[0]), _p(vec[1]),
              _p(vec[2]), _p(vec[3]), _p(Wf), _p(bf), _p(num_batches_tracked), _stream())
    return vec[0], vec[1], vec[2], vec[3], Wf, bf


def linear_fwd_stats_blocks(rows: int) -> int:
    """Partial rows the ELU-statistics epilogue of a forward GEMM over `rows` rows leaves (sn_linear_fwd_stats_blocks)."""
    return int(_lib.load().sn_linear_fwd_stats_blocks(rows))


def colstats_partial(x):
    """(partials (nblk, 2, C) float64, nblk): the statistics pass over the 2-D view x without its final reduction
    (sn_colstats_partial_f32)."""
    _dev(x)
    rows, C = x.shape
    nblk = int(_lib.load().sn_colstats_blocks(rows))
    part = torch.empty((max(nblk, 1), 2, C), dtype=torch.float64, device=x.device)
    _lib.call("sn_colstats_partial_f32", _p(x), _ld(x), rows, C, _p(part), _stream())
    return part, nblk


def bn_bwd_coeffs(Gc, dystats, W, s, invstd, beta, rows: int, has_bias: bool):
    """(dW, db, dgamma, dbeta, Bc, Cc) of the folded BatchNorm+Linear backward (see sn_bn_bwd_coeffs_f32)."""
    _dev(Gc, dystats, W, s, invstd, beta)
    J, C = W.shape
    dev = W.device
    dW = torch.empty((J, C), dtype=torch.float32, device=dev)
    db = torch.empty(J, dtype=torch.float32, device=dev) if has_bias else None
    vec = torch.empty((4, C), dtype=torch.float32, device=dev)
    _lib.call("sn_bn_bwd_coeffs_f32", _p(Gc), _p(dystats), _p(W.contiguous()), _p(s), _p(invstd), _p(beta.contiguous()),
              rows, J, C, _p(dW), _p(db), _p(vec[0]), _p(vec[1]), _p(vec[2]), _p(vec[3]), _stream())
    return dW, db, vec[0], vec[1], vec[2], vec[3]


def segment_colsum(x, mask, rows_per_seg: int, nseg: int):
    """(nseg, C) fp32: per-mesh masked column sums of the 2-D view x (nseg*rows_per_seg rows); mask is (rows,) or None."""
    _dev(x, mask)
    C = x.shape[1]
    if x.shape[0] != rows_per_seg * nseg:
        raise ValueError("segment_colsum: row count mismatch")
    out = torch.empty((nseg, C), dtype=torch.float32, device=x.device)
    ws_bytes = int(_lib.load().sn_segment_colsum_workspace_bytes(rows_per_seg, nseg, C))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    _lib.call("sn_segment_colsum_f32", _p(x), _ld(x), _p(mask), rows_per_seg, nseg, C, _p(out), _p(ws), ws_bytes, _stream())
    return out


def segment_colsum_ragged(x, tiles, seg_tile_ptr, nseg: int, scale=None):
    """(nseg, C) fp32: per-mesh column sums of the 2-D view x of a PACKED batch, times the optional per-mesh `scale`
    (sn_segment_colsum_ragged_f32); tiles / seg_tile_ptr: the int64 device tables of operators.PackedSegments."""
    _dev(x, tiles, seg_tile_ptr, scale)
    C = x.shape[1]
    nt = int(tiles.shape[0])
    out = torch.empty((nseg, C), dtype=torch.float32, device=x.device)
    ws_bytes = int(_lib.load().sn_segment_colsum_ragged_workspace_bytes(nt, C))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    _lib.call("sn_segment_colsum_ragged_f32", _p(x), _ld(x), _p(tiles), nt, _p(seg_tile_ptr), nseg, C, _p(scale), _p(out), _p(ws),
              ws_bytes, _stream())
    return out


def bcast_rows_ragged(src, tiles, dst) -> None:
    """dst[r, :] = src[mesh(r), :] for a PACKED batch (sn_bcast_rows_ragged_f32)."""
    _dev(src, tiles, dst)
    if dst.shape[1] != src.shape[1]:
        raise ValueError("bcast_rows_ragged: shape mismatch")
    _lib.call("sn_bcast_rows_ragged_f32", _p(src.contiguous()), _p(tiles), int(tiles.shape[0]), _p(dst), _ld(dst), src.shape[1], _stream())


def bcast_rows(src, dst, rows_per_seg: int) -> None:
    """dst[r, :] = src[r // rows_per_seg, :]."""
    _dev(src, dst)
    nseg, C = src.shape
    if dst.shape[0] != nseg * rows_per_seg or dst.shape[1] != C:
        raise ValueError("bcast_rows: shape mismatch")
    _lib.call("sn_bcast_rows_f32", _p(src), _p(dst), _ld(dst), rows_per_seg, nseg, C, _stream())


def elu_bwd_bcast(gdst, out, bias, mask, gsrc, rows_per_seg: int, gadd=None) -> None:
    """gsrc = (gdst + mask[r] * bias[mesh(r)]) * elu'(out) + gadd."""
    _dev(gdst, out, bias, mask, gsrc, gadd)
    nseg, C = bias.shape
    _lib.call("sn_elu_bwd_bcast_f32", _p(gdst), _ld(gdst), _p(out), _ld(out), _p(bias), _p(mask), _p(gadd),
              _ld(gadd) if gadd is not None else 0, _p(gsrc), _ld(gsrc), rows_per_seg, nseg, C, _stream())


def dirac_from_mesh(V, F):
    """Dirac operators of one triangle mesh (or of a batch laid out as one disjoint mesh) built on the device.
    V: (nV, 3) fp32, F: (nF, 3) int32.  Returns four BSR4 triples (rowptr, colind, vals):
    Di (4nF x 4nV), DiAT (= DiA^T, Di's structure), DiA (4nV x 4nF), DiT (= Di^T, DiA's structure)."""
    _dev(V, F)
    if V.dtype != torch.float32 or F.dtype != torch.int32 or V.shape[1] != 3 or F.shape[1] != 3:
        raise TypeError("dirac_from_mesh wants V (nV,3) float32 and F (nF,3) int32")
    V, F = V.contiguous(), F.contiguous()
    nV, nF = V.shape[0], F.shape[0]
    dev = V.device
    i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
    f32 = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
    di_rp, di_ci, di_v, diat_v = i32(nF + 1), i32(3 * nF), f32(48 * nF), f32(48 * nF)
    dia_rp, dia_ci, dia_v, dit_v = i32(nV + 1), i32(3 * nF), f32(48 * nF), f32(48 * nF)
    ws_bytes = int(_lib.load().sn_dirac_workspace_bytes(nV, nF))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _lib.call("sn_dirac_bsr4_from_mesh", _p(V), _p(F), nV, nF, _p(di_rp), _p(di_ci), _p(di_v), _p(diat_v), _p(dia_rp),
              _p(dia_ci), _p(dia_v), _p(dit_v), _p(ws), ws_bytes, _stream())
    return (di_rp, di_ci, di_v), (di_rp, di_ci, diat_v), (dia_rp, dia_ci, dia_v), (dia_rp, dia_ci, dit_v)


def laplacian_from_mesh(V, F):
    """Mass-normalised cotangent Laplacian of one mesh (or a batch laid out as one disjoint mesh) built on the device:
    (rowptr, colind, vals) CSR.  Synchronises once (reads nnz).  V: (nV,3) fp32, F: (nF,3) int32."""
    _dev(V, F)
    if V.dtype != torch.float32 or F.dtype != torch.int32:
        raise TypeError("laplacian_from_mesh wants V float32 and F int32")
    V, F = V.contiguous(), F.contiguous()
    nV, nF = V.shape[0], F.shape[0]
    dev = V.device
    rowptr = torch.empty(nV + 1, dtype=torch.int32, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = int(_lib.load().sn_laplacian_workspace_bytes(nV, nF))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _lib.call("sn_laplacian_csr_from_mesh", _p(V), _p(F), nV, nF, 0, _p(rowptr), None, None, _p(flag), _p(ws), ws_bytes, _stream())
    nnz, bad = int(rowptr[-1].item()), int(flag.item())
    if bad:
        raise _lib.SnError("laplacian_from_mesh: a vertex has more incident faces than SN_LAP_MAX_DEGREE")
    colind = torch.empty(nnz, dtype=torch.int32, device=dev)
    vals = torch.empty(nnz, dtype=torch.float32, device=dev)
    _lib.call("sn_laplacian_csr_from_mesh", _p(V), _p(F), nV, nF, 1, _p(rowptr), _p(colind), _p(vals), None, _p(ws), ws_bytes, _stream())
    return rowptr, colind, vals


_gemm_variant = None


def _split_gemm() -> bool:
    """True unless the process runs the fp32-MFMA A/B baseline of the Linear kernels (SN_GEMM_VARIANT=0) — asked of the library
    (sn_gemm_variant), which reads the switch once: the fused epilogues exist in the 16-bit matrix-pipe kernels only."""
    global _gemm_variant
    if _gemm_variant is None:
        _gemm_variant = int(_lib.load().sn_gemm_variant())
    return _gemm_variant != 0


def linear_fwd_supported(K: int, J: int) -> bool:
    """J = 128, or — split kernels only — any multiple of 4 up to 128 (the models' last layer has 120 outputs)."""
    return K in (128, 256) and (J == 128 or (_split_gemm() and 0 < J < 128 and J % 4 == 0))


def elu_stats_supported() -> bool:
    """Column statistics of the ELU output from the GEMM epilogue exist only in the 16-bit matrix-pipe kernels."""
    return _split_gemm()


def new_elu_stats_part(rows: int, device):
    """Workspace for the per-workgroup column sums / sums of squares of a forward GEMM's ELU output (128 columns)."""
    nblk = int(_lib.load().sn_linear_fwd_stats_blocks(rows))
    return torch.empty((max(nblk, 1), 2, 128), dtype=torch.float64, device=device)


def tile_sums_supported() -> bool:
    """The forward kernels can leave per-tile column sums of their ELU output (sn_linear_fwd_tiles_f32): split kernels only."""
    return elu_stats_supported()


def new_tile_sums(rows: int, device):
    """Buffer for the per-tile column sums of a forward GEMM's ELU output: ((rows + 31) // 32, 128) fp32."""
    return torch.empty(((rows + 31) // 32, 128), dtype=torch.float32, device=device)


def linear_fwd(x, W, bias, residual=None, y_elu=None, want_y: bool = True, elu_stats=None, tile_sums=None):
    """y = x·Wᵀ + bias (+ residual); optionally also writes elu(y) into the 2-D view `y_elu` (sn_linear_fwd_f32).
    want_y=False (with y_elu): only the activated copy is written and None is returned.
    elu_stats (from new_elu_stats_part): receives the column statistics of elu(y), see colstats_halves.
    tile_sums (from new_tile_sums, with elu_stats): receives the column sums of elu(y) per 32-row tile (avg_stats_from_tiles)."""
    _dev(x, W, bias, residual, y_elu, elu_stats, tile_sums)
    rows, K = x.shape
    J = W.shape[0]
    # (the fp32-MFMA A/B kernels, SN_GEMM_VARIANT=0, always write y)
    keep = want_y or y_elu is None
    y = torch.empty((rows, J), dtype=torch.float32, device=x.device) if (keep or not elu_stats_supported()) else None
    _lib.call("sn_linear_fwd_tiles_f32", _p(x), _ld(x), _p(W), _ld(W), _p(bias), _p(residual),
              _ld(residual) if residual is not None else 0, _p(y), J, _p(y_elu), _ld(y_elu) if y_elu is not None else 0,
              rows, K, J, _p(elu_stats), _p(tile_sums), _stream())
    return y if keep else None


def avg_stats_from_tiles(tile_sums, part, e, mask, inv_count, rows_per_seg: int, nseg: int):
    """(m, stats) of avg_stats WITHOUT a pass over e: from the per-tile column sums and the statistics partials the GEMM that
    wrote e left (sn_avg_stats_from_tiles_f32)."""
    _dev(tile_sums, part, e, mask, inv_count)
    rows, C = e.shape
    if tile_sums.shape != ((rows + 31) // 32, 128) or part.dim() != 3 or part.shape[1:] != (2, 128) or part.dtype != torch.float64:
        raise ValueError("avg_stats_from_tiles: tile sums / statistics partials of another operand")
    nblk = int(_lib.load().sn_linear_fwd_stats_blocks(rows))
    if part.shape[0] < nblk:
        raise ValueError("avg_stats_from_tiles: partial buffer smaller than the producing launch's grid")
    m = torch.empty((nseg, C), dtype=torch.float32, device=e.device)
    stats = torch.empty((2, 2 * C), dtype=torch.float64, device=e.device)
    ws = torch.empty((nseg, C), dtype=torch.float32, device=e.device)
    _lib.call("sn_avg_stats_from_tiles_f32", _p(tile_sums), _p(part), nblk, _p(e), _ld(e), _p(mask), _p(inv_count.contiguous()),
              rows_per_seg, nseg, C, _p(m), _p(stats), _p(ws), _stream())
    return m, stats


def avg_stats_from_tiles_ragged(tile_sums, part, e, seg):
    """(m, stats) of a global-average stage on a packed batch WITHOUT a pass over e (sn_avg_stats_from_tiles_ragged_f32): from the
    per-tile column sums and the statistics partials the GEMM that wrote e left; seg = operators.PackedSegments."""
    _dev(tile_sums, part, e)
    rows, C = e.shape
    if tile_sums.shape != ((rows + 31) // 32, 128) or part.dim() != 3 or part.shape[1:] != (2, 128) or part.dtype != torch.float64:
        raise ValueError("avg_stats_from_tiles_ragged: tile sums / statistics partials of another operand")
    nblk = int(_lib.load().sn_linear_fwd_stats_blocks(rows))
    if part.shape[0] < nblk or seg.rows != rows:
        raise ValueError("avg_stats_from_tiles_ragged: partial buffer smaller than the producing launch's grid / another batch")
    m = torch.empty((seg.nseg, C), dtype=torch.float32, device=e.device)
    stats = torch.empty((2, 2 * C), dtype=torch.float64, device=e.device)
    ws = torch.empty((seg.nseg, C), dtype=torch.float32, device=e.device)
    _lib.call("sn_avg_stats_from_tiles_ragged_f32", _p(tile_sums), _p(part), nblk, _p(e), _ld(e), _p(seg.off_dev), _p(seg.inv_count),
              seg.nseg, C, _p(m), _p(stats), _p(ws), _stream())
    return m, stats


def colstats_from_part(part, rows: int):
    """(2, 128) float64 statistics of a forward GEMM's ELU output from the partials it left (sn_colstats_merge_f64)."""
    _dev(part)
    if part.dim() != 3 or part.shape[1:] != (2, 128) or part.dtype != torch.float64:
        raise ValueError("colstats_from_part: expected (nblk, 2, 128) float64 partials")
    nblk = int(_lib.load().sn_linear_fwd_stats_blocks(rows))
    if part.shape[0] < nblk:
        raise ValueError("colstats_from_part: partial buffer smaller than the producing launch's grid")
    out = torch.empty((2, 128), dtype=torch.float64, device=part.device)
    _lib.call("sn_colstats_merge_f64", _p(part), nblk, 128, _p(out), 128, 0, _stream())
    return out


def colstats_merge_into(part, out, offset: int) -> None:
    """Combine (nblk, 2, C) float64 partials into columns offset .. offset+C of the (2, width) statistics tensor `out`
    (sn_colstats_merge_f64)."""
    _dev(part, out)
    C = part.shape[2]
    if part.dim() != 3 or part.shape[1] != 2 or part.dtype != torch.float64 or out.dtype != torch.float64 or \
            not out.is_contiguous() or offset + C > out.shape[1]:
        raise ValueError("colstats_merge_into: (nblk, 2, C) float64 partials into a contiguous (2, width >= offset + C) tensor")
    _lib.call("sn_colstats_merge_f64", _p(part), int(part.shape[0]), C, _p(out), out.shape[1], offset, _stream())


def colstats_into(x, out, offset: int):
    """Column sums / sums of squares of the 2-D view x written into columns offset .. offset+C of the (2, width) float64
    statistics tensor `out` (sn_colstats_into_f32)."""
    _dev(x, out)
    rows, C = x.shape
    if out.dim() != 2 or out.shape[0] != 2 or out.dtype != torch.float64 or not out.is_contiguous() or offset + C > out.shape[1]:
        raise ValueError("colstats_into: out must be a contiguous (2, width) float64 tensor with width >= offset + C")
    ws_bytes = int(_lib.load().sn_colstats_workspace_bytes(rows, C))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    _lib.call("sn_colstats_into_f32", _p(x), _ld(x), rows, C, _p(out), out.shape[1], offset, _p(ws), ws_bytes, _stream())
    return out


def colstats_halves(x, part, part_hi=None):
    """(2, 2C) float64 statistics of a stage's concat buffer x = [e | P·e] (rows, 2C), C = 128 or 64: the first half from the
    partials `part` the GEMM that wrote e left behind ((nblk, 2, 128) in the kernel's 128-column layout, of which the first C
    count); the second half from the partials `part_hi` ((nb, 2, C)) of the SpMM that wrote P·e (spmm_q3_stats) or, without
    them, from one pass over x[:, C:] only (sn_colstats_into_f32).  part=None: the first half by a pass over x[:, :C]."""
    _dev(x, part, part_hi)
    rows, C2 = x.shape
    C = C2 // 2
    if part is not None and (C not in (64, 128) or part.dim() != 3 or part.shape[1:] != (2, 128) or part.dtype != torch.float64):
        raise ValueError("colstats_halves: expected a (rows, 256 | 128) buffer and (nblk, 2, 128) float64 GEMM partials")
    if part_hi is not None and (C not in (64, 128) or part_hi.dim() != 3 or part_hi.shape[1:] != (2, C) or part_hi.dtype != torch.float64):
        raise ValueError("colstats_halves: expected (nb, 2, C) float64 partials for the propagated half")
    lib = _lib.load()
    out = torch.empty((2, C2), dtype=torch.float64, device=x.device)
    nlo = int(lib.sn_linear_fwd_stats_blocks(rows)) if part is not None else 0
    if part is not None and part.shape[0] < nlo:
        raise ValueError("colstats_halves: partial buffer smaller than the producing launch's grid")
    if part is not None and part_hi is not None:            # both producers left partials: one launch for the two halves
        _lib.call("sn_colstats_merge2_f64", _p(part), nlo, C, 128, _p(part_hi), int(part_hi.shape[0]), C, C, _p(out), _stream())
        return out
    for half, p_ in ((0, part), (1, part_hi)):
        if p_ is not None and (half == 1 or C == 128):
            nb = nlo if half == 0 else int(p_.shape[0])
            _lib.call("sn_colstats_merge_f64", _p(p_), nb, C, _p(out), C2, half * C, _stream())
        elif p_ is not None:                                 # (GEMM partials of a 64-channel half are read by the two-source launch only)
            raise ValueError("colstats_halves: 64-channel GEMM partials need the SpMM's partials too")
        else:
            xs = x[:, half * C:(half + 1) * C]
            ws_bytes = int(lib.sn_colstats_workspace_bytes(rows, C))
            ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
            _lib.call("sn_colstats_into_f32", _p(xs), _ld(xs), rows, C, _p(out), C2, half * C, _p(ws), ws_bytes, _stream())
    return out


def linear_dgrad_supported(J: int, C: int) -> bool:
    return C in (128, 256) and (J == 128 or (_split_gemm() and 0 < J < 128 and J % 4 == 0))


def linear_dgrad_elu_supported(J: int, C: int) -> bool:
    """The fused form exists only in the 16-bit matrix-pipe kernels (SN_GEMM_VARIANT != 0)."""
    return 0 < J <= 128 and J % 4 == 0 and C in (128, 256) and _split_gemm()


def linear_dgrad_elu(dy, W, x, center, B, Cc, gadd=None):
    """Input gradient of the folded BatchNorm+Linear whose operand x is a stage's concat buffer [e | P·e]
    (sn_linear_dgrad_elu_f32): returns (dx_hi, gact) with dx_hi = dx[:, C/2:] and gact = dx[:, :C/2] * elu'(e) + gadd."""
    _dev(dy, W, x, center, B, Cc, gadd)
    rows, J = dy.shape
    C = W.shape[1]
    h = C // 2
    dx_hi = torch.empty((rows, h), dtype=torch.float32, device=dy.device)
    gact = torch.empty((rows, h), dtype=torch.float32, device=dy.device)
    am = _new_dgrad_absmax(dy.device)
    _lib.call("sn_linear_dgrad_elu_absmax_f32", _p(dy), _ld(dy), _p(W), _ld(W), _p(x), _ld(x), _p(center), _p(B), _p(Cc),
              _p(dx_hi), h, _p(gact), h, _p(gadd), _ld(gadd) if gadd is not None else 0, rows, J, C, _p(am), _stream())
    note_absmax(gact, am)                # gact is the dy operand of the layer below
    return dx_hi, gact


def _new_dgrad_absmax(device):
    """Buffer for the per-workgroup maxima an input-gradient launch leaves (None when the two-piece weight gradient is off)."""
    if not absmax_wanted():
        return None
    return torch.empty(int(_lib.load().sn_linear_dgrad_absmax_blocks()), dtype=torch.float32, device=device)


def linear_dgrad(dy, W, x=None, center=None, B=None, Cc=None):
    """dx = dy·W (+ (x - center)*B + Cc): input gradient of the folded BatchNorm+Linear (sn_linear_dgrad_f32)."""
    _dev(dy, W, x, center, B, Cc)
    rows, J = dy.shape
    C = W.shape[1]
    dx = torch.empty((rows, C), dtype=torch.float32, device=dy.device)
    _lib.call("sn_linear_dgrad_f32", _p(dy), _ld(dy), _p(W), _ld(W), _p(x), _ld(x) if x is not None else 0, _p(center),
              _p(B), _p(Cc), _p(dx), C, rows, J, C, _stream())
    return dx


# ---- half-width global-average stage (see include/sn_spmm.h) ----------------------------------------------------------
def avg_stage_supported(C: int, J: int, rows_per_seg: int) -> bool:
    return C == 128 and J == 128 and rows_per_seg >= 32 and _split_gemm()


def avg_fwd_prep(segsum, inv_count, rows_per_seg: int, stats1):
    """(m (nseg, C) fp32, stats (2, 2C) fp64): per-mesh mean and the BatchNorm statistics of [e | mean broadcast]."""
    _dev(segsum, inv_count, stats1)
    nseg, C = segsum.shape
    m = torch.empty((nseg, C), dtype=torch.float32, device=segsum.device)
    stats = torch.empty((2, 2 * C), dtype=torch.float64, device=segsum.device)
    _lib.call("sn_avg_fwd_prep_f32", _p(segsum), _p(inv_count.contiguous()), nseg, C, rows_per_seg, _p(stats1), _p(m), _p(stats),
              _stream())
    return m, stats


def avg_stats(e, mask, inv_count, rows_per_seg: int, nseg: int):
    """(m, stats) of avg_fwd_prep from ONE pass over e (sn_avg_stats_f32)."""
    _dev(e, mask, inv_count)
    C = e.shape[1]
    m = torch.empty((nseg, C), dtype=torch.float32, device=e.device)
    stats = torch.empty((2, 2 * C), dtype=torch.float64, device=e.device)
    ws_bytes = int(_lib.load().sn_avg_stats_workspace_bytes(rows_per_seg, nseg, C))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=e.device)
    _lib.call("sn_avg_stats_f32", _p(e), _ld(e), _p(mask), _p(inv_count.contiguous()), rows_per_seg, nseg, C, _p(m), _p(stats),
              _p(ws), ws_bytes, _stream())
    return m, stats


def seg_affine(A, W, bias):
    """out[g, j] = bias[j] + sum_c A[g, c] * W[j, c]; W may be a column slice (row stride kept)."""
    _dev(A, W, bias)
    nseg, K = A.shape
    J = W.shape[0]
    out = torch.empty((nseg, J), dtype=torch.float32, device=A.device)
    _lib.call("sn_seg_affine_f32", _p(A), nseg, K, _p(W), _ld(W), _p(bias), J, _p(out), _stream())
    return out


def avg_stats_ragged(m, seg, part, nblk: int):
    """(2, 2C) float64 BatchNorm statistics of [e | per-mesh mean broadcast] on a packed batch (sn_avg_prep_ragged_f32): m (nseg, C)
    the per-mesh means, seg = operators.PackedSegments, part the (>= nblk, 2, C) float64 statistics partials of e (ready (2, C)
    statistics: pass them as one block)."""
    _dev(m, part)
    nseg, C = m.shape
    if part.dtype != torch.float64 or part.shape[-1] != C or not part.is_contiguous() or not m.is_contiguous():
        raise ValueError("avg_stats_ragged: float64 partials (nblk, 2, C) and contiguous means (nseg, C)")
    stats = torch.empty((2, 2 * C), dtype=torch.float64, device=m.device)
    _lib.call("sn_avg_prep_ragged_f32", _p(m), _p(seg.off_dev), nseg, C, _p(part), nblk, _p(stats), _stream())
    return stats


AVG_BWD_MERGE_MAX = 32        # meshes up to which a global-average stage's backward trio runs as one launch (A/B below)


def avg_merged_supported(J: int, C: int, nseg: int, which: int = 1) -> bool:
    """Shapes for which a global-average stage folds its per-mesh bias into the fold launch (bn_fold_seg, which = 1) / runs gc +
    coefficients + per-mesh vector of its backward as one launch (avg_bn_bwd, which = 2).

    The merged launches run per-channel (backward) / per-output-row (forward) work that the separate kernels spread over many
    more workgroups: they win while the per-mesh algebra is small.  Forward: up to 64 meshes (round 4: config-3 step 18.94 ->
    18.89 ms).  Backward: round 4's kernel lost beyond 8 meshes (57 us at 64 against 19.7 us for the three launches); round 6
    found why — a branch per row serialised its LDS reads and its operand loads, one workgroup per channel block walked every
    mesh's dot product — and the kernel now costs 10 us at one mesh (15), 21 us at 64: same-box config-3 steps 6.84 -> 6.79 ms at
    16 meshes, 10.97 -> 10.94 at 32, FAUST pair replayed 3.39 -> 3.28 ms.  At 64 meshes the one launch (21.0 us) and the three
    (19.6 us) are level within the noise of a step (19.26 / 19.25 ms) and the trace puts the three 1.3 us ahead per stage: the
    limit stays at 32."""
    from . import kernels as _self      # (the limit is read through the module so that tests / tools can move it)

    lim = 64 if which == 1 else _self.AVG_BWD_MERGE_MAX
    return J <= 128 and C % 32 == 0 and 2 * C <= 256 and 0 < nseg <= lim


def bn_fold_seg(stats, rows: int, gamma, beta, W, b, eps: float, momentum: float, running_mean, running_var, m, num_batches_tracked=None):
    """bn_fold (training mode) of a global-average stage's 2C-wide layer and its per-mesh bias m·Wf[:, C:]ᵀ + bf in one launch
    (sn_bn_fold_seg_f32).  Returns (mean, invstd, s, t, Wf, bf, segbias (nseg, J))."""
    _dev(stats, gamma, beta, W, b, running_mean, running_var, num_batches_tracked, m)
    if num_batches_tracked is not None and num_batches_tracked.dtype != torch.int64:
        raise TypeError("num_batches_tracked must be int64")
    J, C = W.shape
    nseg = m.shape[0]
    if m.shape[1] * 2 != C or not m.is_contiguous():
        raise ValueError("bn_fold_seg: per-mesh means of half the layer's width, contiguous")
    dev = W.device
    vec = torch.empty((4, C), dtype=torch.float32, device=dev)
    Wf = torch.empty((J, C), dtype=torch.float32, device=dev)
    bf = torch.empty(J, dtype=torch.float32, device=dev)
    segb = torch.empty((nseg, J), dtype=torch.float32, device=dev)
    _lib.call("sn_bn_fold_seg_f32", _p(stats), rows, _p(gamma), _p(beta), _p(W.contiguous()), _p(b), J, C, float(eps), float(momentum),
              _p(running_mean), _p(running_var), _p(vec[0]), _p(vec[1]), _p(vec[2]), _p(vec[3]), _p(Wf), _p(bf),
              _p(num_batches_tracked), _p(m), nseg, _p(segb), _stream())
    return vec[0], vec[1], vec[2], vec[3], Wf, bf, segb


def avg_bn_bwd(G1, dystats, seg_dy, m, mu2, W, s, invstd, beta, rows: int, has_bias: bool, Wf2, inv_count, rows_per_seg: int = 0,
               segoff=None):
    """(dW, db, dgamma, dbeta, Bc, Cc, segvec) of a global-average stage's backward from the first half's weight gradient G1, the
    column sums of dy and their per-mesh parts: avg_bwd_gc + bn_bwd_coeffs + avg_bwd_segvec[_ragged] in one launch
    (sn_avg_bn_bwd_f32; segoff: the row offsets of ragged meshes)."""
    _dev(G1, dystats, seg_dy, m, mu2, W, s, invstd, beta, Wf2, inv_count, segoff)
    J, C = G1.shape
    nseg = m.shape[0]
    dev = W.device
    dW = torch.empty((J, 2 * C), dtype=torch.float32, device=dev)
    db = torch.empty(J, dtype=torch.float32, device=dev) if has_bias else None
    vec = torch.empty((4, 2 * C), dtype=torch.float32, device=dev)
    segvec = torch.empty((nseg, C), dtype=torch.float32, device=dev)
    _lib.call("sn_avg_bn_bwd_f32", _p(G1), _p(dystats), _p(seg_dy), _p(m), _p(mu2), _p(W.contiguous()), _p(s), _p(invstd),
              _p(beta.contiguous()), rows, J, C, nseg, _p(Wf2), _ld(Wf2), _p(inv_count.contiguous()), rows_per_seg, _p(segoff),
              _p(dW), _p(db), _p(vec[0]), _p(vec[1]), _p(vec[2]), _p(vec[3]), _p(segvec), _stream())
    return dW, db, vec[0], vec[1], vec[2], vec[3], segvec


def avg_bwd_gc(G1, seg_dy, m, mu2):
    _dev(G1, seg_dy, m, mu2)
    J, C = G1.shape
    Gc = torch.empty((J, 2 * C), dtype=torch.float32, device=G1.device)
    _lib.call("sn_avg_bwd_gc_f32", _p(G1), _p(seg_dy), _p(m), _p(mu2), m.shape[0], J, C, _p(Gc), _stream())
    return Gc


def avg_bwd_segvec(seg_dy, Wf2, m, mu2, B2, C2, inv_count, rows_per_seg: int):
    _dev(seg_dy, Wf2, m, mu2, B2, C2, inv_count)
    nseg, C = m.shape
    J = seg_dy.shape[1]
    out = torch.empty((nseg, C), dtype=torch.float32, device=m.device)
    _lib.call("sn_avg_bwd_segvec_f32", _p(seg_dy), _p(Wf2), _ld(Wf2), _p(m), _p(mu2), _p(B2), _p(C2), _p(inv_count.contiguous()),
              rows_per_seg, nseg, J, C, _p(out), _stream())
    return out


def wgrad_seg(dy, x, center, rows_per_seg: int, bounds=None):
    """(G, colsum(dy) fp64, per-mesh colsum(dy) (nseg, J) fp32) in one pass (sn_wgrad_seg_f32; bounds: as kernels.wgrad)."""
    _dev(dy, x, center)
    rows, J = dy.shape
    C = x.shape[1]
    nseg = rows // rows_per_seg
    dev = dy.device
    G = torch.empty((J, C), dtype=torch.float32, device=dev)
    dysum = torch.empty(J, dtype=torch.float64, device=dev)
    seg = torch.empty((nseg, J), dtype=torch.float32, device=dev)
    ws_bytes = int(_lib.load().sn_wgrad_seg_workspace_bytes(rows, rows_per_seg, J, C))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    if bounds is not None:
        _lib.call("sn_wgrad_seg_bounded_f32", _p(dy), _ld(dy), _p(x), _ld(x), _p(center), rows, rows_per_seg, J, C, _p(G), _p(dysum),
                  _p(seg), _p(ws), ws_bytes, *_bounds_args(bounds, C), _stream())
    else:
        _lib.call("sn_wgrad_seg_f32", _p(dy), _ld(dy), _p(x), _ld(x), _p(center), rows, rows_per_seg, J, C, _p(G), _p(dysum), _p(seg),
                  _p(ws), ws_bytes, _stream())
    return G, dysum, seg


def wgrad_slabs(dy, x, center, seg, bounds=None):
    """wgrad_seg for RAGGED meshes (`seg`: operators.PackedSegments): (G, colsum(dy) fp64, per-mesh colsum(dy) (nseg, J))
    from one pass, the row slabs taken from seg's table (sn_wgrad_slabs_f32)."""
    _dev(dy, x, center)
    rows, J = dy.shape
    C = x.shape[1]
    dev = dy.device
    G = torch.empty((J, C), dtype=torch.float32, device=dev)
    dysum = torch.empty(J, dtype=torch.float64, device=dev)
    segsum = torch.empty((seg.nseg, J), dtype=torch.float32, device=dev)
    ws_bytes = seg.nslab * 128 * (C + 1) * 4
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    if bounds is not None:
        _lib.call("sn_wgrad_slabs_bounded_f32", _p(dy), _ld(dy), _p(x), _ld(x), _p(center), rows, _p(seg.slab_off), seg.nslab,
                  _p(seg.seg_slab_ptr), seg.nseg, J, C, _p(G), _p(dysum), _p(segsum), _p(ws), ws_bytes, *_bounds_args(bounds, C),
                  _stream())
    else:
        _lib.call("sn_wgrad_slabs_f32", _p(dy), _ld(dy), _p(x), _ld(x), _p(center), rows, _p(seg.slab_off), seg.nslab,
                  _p(seg.seg_slab_ptr), seg.nseg, J, C, _p(G), _p(dysum), _p(segsum), _p(ws), ws_bytes, _stream())
    return G, dysum, segsum


def avg_bwd_segvec_ragged(seg_dy, Wf2, m, mu2, B2, C2, seg):
    _dev(seg_dy, Wf2, m, mu2, B2, C2)
    nseg, C = m.shape
    J = seg_dy.shape[1]
    out = torch.empty((nseg, C), dtype=torch.float32, device=m.device)
    _lib.call("sn_avg_bwd_segvec_ragged_f32", _p(seg_dy), _p(Wf2), _ld(Wf2), _p(m), _p(mu2), _p(B2), _p(C2), _p(seg.inv_count),
              _p(seg.off_dev), nseg, J, C, _p(out), _stream())
    return out


def linear_fwd_segbias_ragged(x, W, segbias, seg, residual=None, y_elu=None, want_y: bool = True, elu_stats=None, tile_sums=None):
    """linear_fwd_segbias with the bias row of each RAGGED mesh (`seg`: operators.PackedSegments); tile_sums: as linear_fwd."""
    _dev(x, W, segbias, residual, y_elu, tile_sums)
    rows, K = x.shape
    J = W.shape[0]
    y = torch.empty((rows, J), dtype=torch.float32, device=x.device) if want_y else None
    _lib.call("sn_linear_fwd_segbias_ragged_tiles_f32", _p(x), _ld(x), _p(W), _ld(W), _p(segbias), _p(seg.off_dev), seg.nseg,
              _p(residual), _ld(residual) if residual is not None else 0, _p(y), J, _p(y_elu),
              _ld(y_elu) if y_elu is not None else 0, rows, K, J, _p(elu_stats), _p(tile_sums), _stream())
    return y


def linear_dgrad_eluseg_ragged(dy, W, x, center, B, Cc, segvec, seg, gadd=None):
    """linear_dgrad_eluseg with the vector of each RAGGED mesh, no row mask (a packed batch has no padding rows)."""
    _dev(dy, W, x, center, B, Cc, segvec, gadd)
    rows, J = dy.shape
    C = x.shape[1]
    gact = torch.empty((rows, C), dtype=torch.float32, device=dy.device)
    am = _new_dgrad_absmax(dy.device)
    _lib.call("sn_linear_dgrad_eluseg_ragged_absmax_f32", _p(dy), _ld(dy), _p(W), _ld(W), _p(x), _ld(x), _p(center), _p(B), _p(Cc),
              _p(segvec), _p(seg.off_dev), seg.nseg, _p(gact), C, _p(gadd), _ld(gadd) if gadd is not None else 0, rows, J, C,
              _p(am), _stream())
    note_absmax(gact, am)
    return gact


def avg_stage_ragged_supported(C: int, J: int, seg) -> bool:
    return C == 128 and J == 128 and seg.min_len >= 32 and _split_gemm()


def wgrad_thin_supported(J: int, C: int) -> bool:
    return 1 <= C <= 8 and J % 4 == 0 and 256 % (J // 4) == 0


def linear_thin_fwd(x, W, bias, y_elu=None):
    """y = x·W^T + bias for 1..8 input channels; optionally elu(y) into the 2-D view y_elu (sn_linear_thin_fwd_f32)."""
    _dev(x, W, bias, y_elu)
    rows, C = x.shape
    J = W.shape[0]
    y = torch.empty((rows, J), dtype=torch.float32, device=x.device)
    _lib.call("sn_linear_thin_fwd_f32", _p(x), _ld(x), _p(W), _ld(W), _p(bias), rows, C, J, _p(y), J, _p(y_elu),
              _ld(y_elu) if y_elu is not None else 0, _stream())
    return y


def wgrad_thin(dy, x, want_bias: bool = True):
    """(dW (J, C), db (J) | None) of a Linear with 1..8 input channels: dy^T x and colsum(dy) in one pass (sn_wgrad_thin_f32)."""
    _dev(dy, x)
    rows, J = dy.shape
    C = x.shape[1]
    dev = dy.device
    G = torch.empty((J, C), dtype=torch.float32, device=dev)
    db = torch.empty(J, dtype=torch.float32, device=dev) if want_bias else None
    ws_bytes = int(_lib.load().sn_wgrad_thin_workspace_bytes(rows, J, C))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    _lib.call("sn_wgrad_thin_f32", _p(dy), _ld(dy), _p(x), _ld(x), rows, J, C, _p(G), _p(db), _p(ws), ws_bytes, _stream())
    return G, db


def gather_segments(src, base, rows_per_item: int, row_stride: int, length: int):
    """(nitems, rows_per_item, length) fp32: row r of item i = src.flatten()[base[i] + r*row_stride : ... + length]
    (sn_gather_segments_f32); base: int64 device tensor of element offsets into the contiguous src."""
    _dev(src, base)
    if not src.is_contiguous() or src.dtype != torch.float32 or base.dtype != torch.int64:
        raise TypeError("gather_segments: contiguous float32 source and int64 offsets expected")
    n = base.numel()
    out = torch.empty((n, rows_per_item, length), dtype=torch.float32, device=src.device)
    _lib.call("sn_gather_segments_f32", _p(src), _p(base.contiguous()), n, rows_per_item, row_stride, length, _p(out), _stream())
    return out


def gather_segments_ragged(src, base, seg, row_stride: int, length: int):
    """(seg.rows, length) fp32: the packed counterpart of gather_segments — item i supplies its first len_i rows, stored at the
    rows of mesh i of the packed batch `seg` (operators.PackedSegments); sn_gather_segments_ragged_f32."""
    _dev(src, base)
    if not src.is_contiguous() or src.dtype != torch.float32 or base.dtype != torch.int64 or base.numel() != seg.nseg:
        raise TypeError("gather_segments_ragged: contiguous float32 source and one int64 offset per mesh expected")
    out = torch.empty((seg.rows, length), dtype=torch.float32, device=src.device)
    _lib.call("sn_gather_segments_ragged_f32", _p(src), _p(base.contiguous()), _p(seg.off_dev), seg.nseg, seg.rows, row_stride, length,
              _p(out), _stream())
    return out


def pair_argmin(GA, pa, GB, pb):
    """argmin_j (GA[r, pa[j]] + GB[pb[r], j]) for every r (sn_pair_argmin_f32): the target of the dense-correspondence
    loss, main.py:236-237, without the two gathered (NA, NB) matrices.  GA, GB: row-major fp32 matrices; pa (NB,), pb (NA,)
    int64.  Returns (NA,) int64."""
    _dev(GA, pa, GB, pb)
    if GA.dtype != torch.float32 or GB.dtype != torch.float32 or pa.dtype != torch.int64 or pb.dtype != torch.int64:
        raise TypeError("pair_argmin: float32 matrices and int64 index vectors expected")
    if GA.dim() != 2 or GB.dim() != 2 or GA.stride(1) != 1 or GB.stride(1) != 1:
        raise ValueError("pair_argmin: row-major 2-D matrices expected")
    NA, NB = pb.numel(), pa.numel()
    if NA > GA.shape[0] or NB > GB.shape[1]:
        raise ValueError("pair_argmin: more rows / columns asked for than the matrices hold")
    out = torch.empty(NA, dtype=torch.int64, device=GA.device)
    _lib.call("sn_pair_argmin_f32", _p(GA), GA.stride(0), GA.shape[1], _p(pa.contiguous()), _p(GB), GB.stride(0), _p(pb.contiguous()), NA, NB,
              _p(out), _stream())
    return out


def pair_ce_fwd(S, target, NA: int, NB: int):
    """(lse, rowloss), NA floats each, of the cross entropy over S[:NA, :NB] (sn_pair_ce_fwd_f32; main.py:238-239)."""
    _dev(S, target)
    if S.dtype != torch.float32 or target.dtype != torch.int64 or S.dim() != 2 or S.stride(1) != 1:
        raise TypeError("pair_ce_fwd: row-major float32 scores and int64 targets expected")
    if NA > S.shape[0] or NB > S.shape[1] or target.numel() != NA:
        raise ValueError("pair_ce_fwd: NA x NB does not fit the score matrix / target vector")
    lse = torch.empty(NA, dtype=torch.float32, device=S.device)
    rowloss = torch.empty(NA, dtype=torch.float32, device=S.device)
    _lib.call("sn_pair_ce_fwd_f32", _p(S), S.stride(0), _p(target.contiguous()), NA, NB, _p(lse), _p(rowloss), _stream())
    return lse, rowloss


def pair_ce_bwd(S, target, lse, gloss, NA: int, NB: int):
    """Gradient of mean_r rowloss[r] times the device scalar gloss, for the WHOLE score matrix (zeros outside NA x NB)."""
    _dev(S, target, lse, gloss)
    dS = torch.empty((S.shape[0], S.shape[1]), dtype=torch.float32, device=S.device)
    _lib.call("sn_pair_ce_bwd_f32", _p(S), S.stride(0), _p(target.contiguous()), _p(lse), _p(gloss), NA, NB, S.shape[0], S.shape[1],
              _p(dS), dS.stride(0), _stream())
    return dS


def pair_fused_fwd(FA, FB, target, NA: int, NB: int):
    """(lse, rowloss, workspace) of the correspondence cross entropy computed from the tower features FA (rowsA x K), FB
    (rowsB x K): the scores FA·FBᵀ (models.py:203) are formed tile by tile on the matrix pipe and never written
    (sn_pair_fused_fwd_f32).  `workspace` holds the split features for pair_fused_bwd."""
    _dev(FA, FB, target)
    for t in (FA, FB):
        if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise TypeError("pair_fused_fwd: row-major float32 feature matrices expected")
    if target.dtype != torch.int64 or FA.shape[1] != FB.shape[1]:
        raise TypeError("pair_fused_fwd: int64 targets and features of one width expected")
    if not (0 < NA <= FA.shape[0] and 0 < NB <= FB.shape[0]) or target.numel() != NA:
        raise ValueError("pair_fused_fwd: NA / NB do not fit the feature matrices / target vector")
    nbytes = _lib.load().sn_pair_fused_workspace_bytes(FA.shape[0], FB.shape[0])
    ws = torch.empty(nbytes, dtype=torch.uint8, device=FA.device)
    lse = torch.empty(NA, dtype=torch.float32, device=FA.device)
    rowloss = torch.empty(NA, dtype=torch.float32, device=FA.device)
    _lib.call("sn_pair_fused_fwd_f32", _p(FA), FA.stride(0), _p(FB), FB.stride(0), _p(target.contiguous()), NA, NB, FA.shape[0], FB.shape[0],
              FA.shape[1], _p(lse), _p(rowloss), _p(ws), nbytes, _stream())
    return lse, rowloss, ws


def pair_fused_bwd(target, lse, gloss, ws, NA: int, NB: int, rowsA: int, rowsB: int, K: int):
    """(dFA, dFB) of mean_r rowloss[r] times the device scalar gloss (sn_pair_fused_bwd_f32); rows past NA / NB are zero."""
    _dev(target, lse, gloss, ws)
    dFA = torch.empty((rowsA, K), dtype=torch.float32, device=lse.device)
    dFB = torch.empty((rowsB, K), dtype=torch.float32, device=lse.device)
    _lib.call("sn_pair_fused_bwd_f32", _p(target.contiguous()), _p(lse), _p(gloss), NA, NB, rowsA, rowsB, K, _p(dFA), K, _p(dFB), K,
              _p(ws), ws.numel(), _stream())
    return dFA, dFB


def masked_smooth_l1_fwd(out2d, target2d, rowmask, scale: float):
    """scale * sum smooth_l1(out*rowmask - target) as a 0-dim fp32 tensor (sn_masked_smooth_l1_fwd_f32)."""
    _dev(out2d, target2d, rowmask)
    rows, C = out2d.shape
    loss = torch.empty((), dtype=torch.float32, device=out2d.device)
    ws_bytes = int(_lib.load().sn_masked_smooth_l1_workspace_bytes(rows, C))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=out2d.device)
    _lib.call("sn_masked_smooth_l1_fwd_f32", _p(out2d), _ld(out2d), _p(target2d), _ld(target2d), _p(rowmask), rows, C,
              float(scale), _p(loss), _p(ws), ws_bytes, _stream())
    return loss


def masked_smooth_l1_bwd(out2d, target2d, rowmask, scale: float, gloss):
    """Gradient of masked_smooth_l1_fwd w.r.t. out2d, times the device scalar gloss (sn_masked_smooth_l1_bwd_f32)."""
    _dev(out2d, target2d, rowmask, gloss)
    rows, C = out2d.shape
    g = torch.empty((rows, C), dtype=torch.float32, device=out2d.device)
    _lib.call("sn_masked_smooth_l1_bwd_f32", _p(out2d), _ld(out2d), _p(target2d), _ld(target2d), _p(rowmask), rows, C,
              float(scale), _p(gloss), _p(g), C, _stream())
    return g


def linear_fwd_segbias(x, W, segbias, rows_per_seg: int, residual=None, y_elu=None, want_y: bool = True, elu_stats=None,
                       tile_sums=None):
    """y = x·W^T + segbias[row // rows_per_seg] (+ residual), optionally elu(y) into y_elu; want_y=False: only y_elu is written.
    tile_sums: as linear_fwd."""
    _dev(x, W, segbias, residual, y_elu, tile_sums)
    rows, K = x.shape
    J = W.shape[0]
    y = torch.empty((rows, J), dtype=torch.float32, device=x.device) if want_y else None
    _lib.call("sn_linear_fwd_segbias_tiles_f32", _p(x), _ld(x), _p(W), _ld(W), _p(segbias), rows_per_seg, _p(residual),
              _ld(residual) if residual is not None else 0, _p(y), J, _p(y_elu), _ld(y_elu) if y_elu is not None else 0,
              rows, K, J, _p(elu_stats), _p(tile_sums), _stream())
    return y


def linear_dgrad_eluseg(dy, W, x, center, B, Cc, segvec, rows_per_seg: int, rowmask=None, gadd=None):
    """(dy·W + (x - center) B + Cc + rowmask * segvec[row // rows_per_seg]) * elu'(x) + gadd for all C columns
    (segvec=None: no per-mesh vector)."""
    _dev(dy, W, x, center, B, Cc, segvec, rowmask, gadd)
    rows, J = dy.shape
    C = x.shape[1]
    gact = torch.empty((rows, C), dtype=torch.float32, device=dy.device)
    am = _new_dgrad_absmax(dy.device)
    _lib.call("sn_linear_dgrad_eluseg_absmax_f32", _p(dy), _ld(dy), _p(W), _ld(W), _p(x), _ld(x), _p(center), _p(B), _p(Cc),
              _p(segvec), rows_per_seg, _p(rowmask), _p(gact), C, _p(gadd), _ld(gadd) if gadd is not None else 0, rows, J, C,
              _p(am), _stream())
    note_absmax(gact, am)
    return gact
