"""ctypes binding of oracle/sn_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsn_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, "sn_oracle.c"), os.path.join(_HERE, "sn_host_twin.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s_) for s_ in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "--no-print-directory"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_bsr4_count.restype = C.c_int64
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


i64 = C.c_int64
i32 = C.c_int


def sparse_bmm(values, col_ind, col_ptr, dense):
    """oracle_sparse_bmm: batched CSR (B,R+1 global offsets) x dense (B,K,N) -> (B,R,N)."""
    values, col_ind, col_ptr, dense = _c(values, np.float32), _c(col_ind, np.int64), _c(col_ptr, np.int64), _c(dense, np.float32)
    B, R1 = col_ptr.shape
    _, K, N = dense.shape
    out = np.empty((B, R1 - 1, N), np.float32)
    lib().oracle_sparse_bmm(_p(out), _p(values), _p(col_ind), _p(col_ptr), _p(dense), i64(B), i64(R1 - 1), i64(K), i64(N))
    return out


def batch_csr(indices, B, R):
    """oracle_batch_csr: literal batch_csr.cu (keeps its interior-empty-row defect)."""
    indices = _c(indices, np.int64)
    nnz = indices.shape[1]
    col_ind = np.empty(nnz, np.int64)
    col_ptr = np.empty((B, R + 1), np.int64)
    lib().oracle_batch_csr(_p(indices), i64(nnz), i64(B), i64(R), _p(col_ind), _p(col_ptr))
    return col_ind, col_ptr


def coo_to_csr(idx_batch, idx_row, idx_col, B, R, Kb):
    idx_row, idx_col = _c(idx_row, np.int64), _c(idx_col, np.int64)
    ib = None if idx_batch is None else _c(idx_batch, np.int64)
    nnz = idx_row.shape[0]
    rowptr = np.empty(B * R + 1, np.int32)
    colind = np.empty(nnz, np.int32)
    lib().oracle_coo_to_csr_i32(_p(ib) if ib is not None else None, _p(idx_row), _p(idx_col), i64(nnz), i64(B), i64(R), i64(Kb), _p(rowptr), _p(colind))
    return rowptr, colind


def spmm_csr(rowptr, colind, vals, X, N, ldx=None, xg=1, Y=None, ldy=None, yg=1, M=None):
    """oracle_spmm_csr_f32.  X / Y are flat float32 buffers addressed with (ld, group); returns Y."""
    rowptr, colind, vals = _c(rowptr, np.int32), _c(colind, np.int32), _c(vals, np.float32)
    M = rowptr.shape[0] - 1 if M is None else M
    X = _c(X, np.float32)
    ldx = N * xg if ldx is None else ldx
    ldy = N * yg if ldy is None else ldy
    if Y is None:
        Y = np.zeros(((M + yg - 1) // yg) * ldy, np.float32)
    assert Y.dtype == np.float32 and Y.flags.c_contiguous
    lib().oracle_spmm_csr_f32(_p(rowptr), _p(colind), _p(vals), i64(M), _p(X), i64(ldx), i32(xg), i32(N), _p(Y), i64(ldy), i32(yg))
    return Y


def spmm_csr_f64(rowptr, colind, vals, X, N, ldx=None, xg=1):
    rowptr, colind, vals, X = _c(rowptr, np.int32), _c(colind, np.int32), _c(vals, np.float32), _c(X, np.float32)
    M = rowptr.shape[0] - 1
    ldx = N * xg if ldx is None else ldx
    Y = np.empty((M, N), np.float64)
    lib().oracle_spmm_csr_f64(_p(rowptr), _p(colind), _p(vals), i64(M), _p(X), i64(ldx), i32(xg), i32(N), _p(Y))
    return Y


def csr_transpose(rowptr, colind, vals, K):
    rowptr, colind, vals = _c(rowptr, np.int32), _c(colind, np.int32), _c(vals, np.float32)
    M = rowptr.shape[0] - 1
    t_rowptr = np.empty(K + 1, np.int32)
    t_colind = np.empty(colind.shape[0], np.int32)
    t_vals = np.empty(colind.shape[0], np.float32)
    lib().oracle_csr_transpose_f32(_p(rowptr), _p(colind), _p(vals), i64(M), i64(K), _p(t_rowptr), _p(t_colind), _p(t_vals))
    return t_rowptr, t_colind, t_vals


def blockdiag_concat(pool_rowptr, pool_colind, pool_vals, desc, size0, size1, total, vpe=1):
    pool_rowptr, pool_colind, pool_vals = _c(pool_rowptr, np.int32), _c(pool_colind, np.int32), _c(pool_vals, np.float32)
    desc = _c(desc, np.int64)
    B = desc.shape[0]
    out_rowptr = np.empty(B * size0 + 1, np.int32)
    out_colind = np.empty(total, np.int32)
    out_vals = np.empty(total * vpe, np.float32)
    lib().oracle_blockdiag_concat_i32(_p(pool_rowptr), _p(pool_colind), _p(pool_vals), _p(desc), i64(B), i64(size0), i64(size1), i64(total), i32(vpe), _p(out_rowptr), _p(out_colind), _p(out_vals))
    return out_rowptr, out_colind, out_vals


def csr_to_bsr4(rowptr, colind, vals):
    rowptr, colind, vals = _c(rowptr, np.int32), _c(colind, np.int32), _c(vals, np.float32)
    M = rowptr.shape[0] - 1
    assert M % 4 == 0
    b_rowptr = np.empty(M // 4 + 1, np.int32)
    nb = lib().oracle_bsr4_count(_p(rowptr), _p(colind), i64(M), _p(b_rowptr))
    b_colind = np.empty(nb, np.int32)
    b_vals = np.empty(nb * 16, np.float32)
    lib().oracle_bsr4_fill(_p(rowptr), _p(colind), _p(vals), i64(M), _p(b_rowptr), _p(b_colind), _p(b_vals))
    return b_rowptr, b_colind, b_vals


def elu(src):
    src = _c(src, np.float32)
    rows, Cc = src.reshape(-1, src.shape[-1]).shape
    dst = np.empty_like(src)
    lib().oracle_elu(_p(src), i64(Cc), _p(dst), i64(Cc), i64(rows), i32(Cc))
    return dst


def elu_bwd(gdst, out, gsrc=None):
    gdst, out = _c(gdst, np.float32), _c(out, np.float32)
    rows, Cc = out.reshape(-1, out.shape[-1]).shape
    acc = gsrc is not None
    g = _c(gsrc, np.float32).copy() if acc else np.empty_like(out)
    lib().oracle_elu_bwd(_p(gdst), i64(Cc), None, i64(0), None, i64(0), _p(out), i64(Cc), _p(g), i64(Cc), i64(rows), i32(Cc),
                         i32(1 if acc else 0))
    return g


# ---- raw-pointer variants (used by tests/cpu_kernels.py to run the product's host logic on CPU tensors) ----------
def spmm_csr_raw(rowptr, colind, vals, M, x_ptr, ldx, xg, N, y_ptr, ldy, yg):
    lib().oracle_spmm_csr_f32(_p(rowptr), _p(colind), _p(vals), i64(M), C.c_void_p(x_ptr), i64(ldx), i32(xg), i32(N),
                              C.c_void_p(y_ptr), i64(ldy), i32(yg))


def elu_raw(src_ptr, lds, dst_ptr, ldd, rows, Cc):
    lib().oracle_elu(C.c_void_p(src_ptr), i64(lds), C.c_void_p(dst_ptr), i64(ldd), i64(rows), i32(Cc))


def elu_bwd_raw(g_ptr, ldg, o_ptr, ldo, s_ptr, ldgs, rows, Cc, accumulate, g2_ptr=None, ldg2=0, ga_ptr=None, ldga=0):
    lib().oracle_elu_bwd(C.c_void_p(g_ptr), i64(ldg), C.c_void_p(g2_ptr) if g2_ptr else None, i64(ldg2),
                         C.c_void_p(ga_ptr) if ga_ptr else None, i64(ldga), C.c_void_p(o_ptr), i64(ldo), C.c_void_p(s_ptr),
                         i64(ldgs), i64(rows), i32(Cc), i32(1 if accumulate else 0))


def colstats_raw(x_ptr, ld, rows, Cc):
    out = np.empty(2 * Cc, np.float64)
    lib().oracle_colstats(C.c_void_p(x_ptr), i64(ld), i64(rows), i32(Cc), _p(out))
    return out.reshape(2, Cc)


def wgrad_raw(dy_ptr, lddy, x_ptr, ldx, rows, J, Cc, center=None):
    G = np.empty((J, Cc), np.float64)
    cen = None if center is None else _c(center, np.float32)
    lib().oracle_wgrad(C.c_void_p(dy_ptr), i64(lddy), C.c_void_p(x_ptr), i64(ldx), _p(cen) if cen is not None else None,
                       i64(rows), i32(J), i32(Cc), _p(G))
    return G


def affine_cols_acc_raw(dx_ptr, lddx, x_ptr, ldx, B, Cvec, rows, Cc, center=None):
    B, Cvec = _c(B, np.float32), _c(Cvec, np.float32)
    cen = None if center is None else _c(center, np.float32)
    lib().oracle_affine_cols_acc(C.c_void_p(dx_ptr), i64(lddx), C.c_void_p(x_ptr), i64(ldx), _p(cen) if cen is not None else None,
                                 _p(B), _p(Cvec), i64(rows), i32(Cc))
