"""Device-resident sparse operators of the Surface-Network path.

`SparseOperator`  one (usually batched block-diagonal) operator kept in HBM as CSR with int32 indices, plus —
                  built once and cached — the CSR of its transpose (for the backward) and the packed 4x4-block
                  BSR form used for the quaternionic Dirac operators.
`OperatorPool`    every mesh of a dataset converted ONCE (A and A^T, CSR and/or BSR4) and pooled in HBM; a batch
                  is assembled per step by one offset-concatenation launch instead of the reference's host-side
                  index shift + concat + coalesce() sort + H2D copy.

What this replaces in the reference (paths relative to the reference checkout):
    sp_sparse_to_pt_sparse      src/utils/utils_pt.py:56-69     (scipy -> torch COO, per sample per step)
    sparse_diag_cat             src/utils/utils_pt.py:41-53     (block-diagonal batching on the host, every step)
    sparse_cat                  src/utils/utils_pt.py:21-39     (3-D batched COO for SparseBMMFunc)
    transpose().coalesce()      src/utils/cuda/sparse_bmm_func.py:66-67  (re-done on every backward)
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import kernels

__all__ = ["SparseOperator", "OperatorPool", "as_operator"]

# A BSR4 copy is kept when zero-fill costs at most this much extra storage over CSR entries.
_BSR4_MAX_FILL = 1.6


class SparseOperator:
    """M x K fp32 sparse operator resident on one GPU (CSR, int32 indices).

    Quacks enough like the torch sparse tensors the reference models pass around: `.size()`, `.shape`,
    `.dim()`, `.is_cuda`, `.cuda()` (src/as_rigid_as_possible/models.py:133-136 inspects `Di.size()`).
    No gradient ever flows to an operator (src/utils/cuda/sparse_bmm_func.py:72).
    """

    layout = "sn_csr"
    requires_grad = False

    def __init__(self, rowptr, colind, vals, shape, *, batch: int = 1, transpose: "Optional[SparseOperator]" = None,
                 bsr4=None):
        M, K = int(shape[0]), int(shape[1])
        if rowptr.dtype != torch.int32 or colind.dtype != torch.int32 or vals.dtype != torch.float32:
            raise TypeError("SparseOperator wants int32 rowptr/colind and float32 vals")
        if rowptr.numel() != M + 1 or colind.numel() != vals.numel():
            raise ValueError("inconsistent CSR arrays")
        self.rowptr, self.colind, self.vals = rowptr, colind, vals
        self._shape = (M, K)
        self.batch = int(batch)                  # number of diagonal blocks (B of the reference's (B,R,K) operators)
        self._t = transpose
        self._bsr4 = bsr4                        # (b_rowptr, b_colind, b_vals) | None | False (= not worthwhile)

    # ---- tensor-like surface -------------------------------------------------------------------
    @property
    def shape(self):
        return torch.Size(self._shape)

    def size(self, dim: Optional[int] = None):
        return self.shape if dim is None else self._shape[dim]

    def dim(self) -> int:
        return 2

    @property
    def nnz(self) -> int:
        return int(self.colind.numel())

    def _nnz(self) -> int:
        return self.nnz

    @property
    def device(self):
        return self.rowptr.device

    @property
    def is_cuda(self) -> bool:
        return self.rowptr.is_cuda

    def cuda(self, *_, **__):
        return self.to("cuda")

    def to(self, device):
        device = torch.device(device)
        if device.type == self.device.type and (device.index is None or device.index == self.device.index):
            return self
        mv = lambda t: t.to(device)
        out = SparseOperator(mv(self.rowptr), mv(self.colind), mv(self.vals), self._shape, batch=self.batch)
        if isinstance(self._bsr4, tuple):
            out._bsr4 = tuple(mv(t) for t in self._bsr4)
        if self._t is not None:
            out._t = self._t.to(device)
            out._t._t = out
        return out

    def __repr__(self):
        return f"SparseOperator(shape={self._shape}, nnz={self.nnz}, batch={self.batch}, device={self.device})"

    # ---- derived forms --------------------------------------------------------------------------
    def t(self) -> "SparseOperator":
        """CSR of the transpose, built on first use by sn_csr_transpose_f32 and cached both ways."""
        if self._t is None:
            M, K = self._shape
            tr, tc, tv = kernels.csr_transpose(self.rowptr, self.colind, self.vals, M, K)
            self._t = SparseOperator(tr, tc, tv, (K, M), batch=self.batch, transpose=self)
        return self._t

    T = property(t)

    def bsr4(self):
        """(b_rowptr, b_colind, b_vals) or None when the 4x4-block form is not applicable / not worthwhile."""
        if self._bsr4 is None:
            M, K = self._shape
            if M % 4 or K % 4 or self.nnz == 0:
                self._bsr4 = False
            else:
                b = kernels.csr_to_bsr4(self.rowptr, self.colind, self.vals, M, K)
                self._bsr4 = b if 16 * b[1].numel() <= _BSR4_MAX_FILL * self.nnz else False
        return self._bsr4 or None

    # ---- constructors ----------------------------------------------------------------------------
    @classmethod
    def from_scipy(cls, A, device="cuda", batch: int = 1) -> "SparseOperator":
        A = A.tocsr()
        A.sort_indices()
        return cls(torch.from_numpy(A.indptr.astype(np.int32)).to(device),
                   torch.from_numpy(A.indices.astype(np.int32)).to(device),
                   torch.from_numpy(A.data.astype(np.float32)).to(device), A.shape, batch=batch)

    @classmethod
    def from_torch_coo(cls, A: torch.Tensor) -> "SparseOperator":
        """From a coalesced torch sparse COO tensor on the GPU: 2-D (block-diagonal, the output of
        sparse_diag_cat) or 3-D (B,R,K) (the output of sparse_cat).  Conversion runs on the device."""
        if A.layout != torch.sparse_coo:
            raise TypeError("expected a torch sparse COO tensor")
        if not A.is_coalesced():
            A = A.coalesce()          # the reference coalesces before use (utils_pt.py:39,53)
        idx, vals = A._indices(), A._values()
        if vals.dtype != torch.float32:
            vals = vals.float()
        if A.dim() == 2:
            M, K = A.shape
            rowptr, colind = kernels.coo_to_csr(None, idx[0], idx[1], 1, M, K)
            return cls(rowptr, colind, vals.contiguous(), (M, K), batch=1)
        if A.dim() == 3:
            B, R, Kb = A.shape
            rowptr, colind = kernels.coo_to_csr(idx[0], idx[1], idx[2], B, R, Kb)
            return cls(rowptr, colind, vals.contiguous(), (B * R, B * Kb), batch=B)
        raise ValueError("operator must be 2-D or 3-D")

    # ---- export (tests / debugging) --------------------------------------------------------------
    def to_scipy(self):
        import scipy.sparse as sp

        return sp.csr_matrix((self.vals.cpu().numpy(), self.colind.cpu().numpy(), self.rowptr.cpu().numpy()),
                             shape=self._shape)


def as_operator(A) -> SparseOperator:
    """Normalise whatever the reference drivers pass as L / Di / DiA (SparseOperator, torch sparse COO 2-D or
    3-D) to a SparseOperator.  The converted form is cached on the tensor object, so one operator that feeds
    several residual blocks in a step (and their backward passes) is converted once."""
    if isinstance(A, SparseOperator):
        return A
    if isinstance(A, torch.Tensor) and A.layout == torch.sparse_coo:
        cached = getattr(A, "_sn_operator", None)
        if cached is None:
            cached = SparseOperator.from_torch_coo(A)
            A._sn_operator = cached
        return cached
    raise TypeError(f"cannot interpret {type(A)} as a sparse operator")


class OperatorPool:
    """All per-mesh operators of one kind (e.g. every Di of a dataset) resident in HBM, A and A^T.

    `assemble(sel, size0, size1)` returns the batched block-diagonal SparseOperator of the selected meshes —
    identical (after to_dense) to the reference's `sparse_diag_cat([op[i] for i in sel], size0, size1)` — with its
    transpose attached, so neither forward nor backward ever sorts, transposes or touches the host.
    """

    def __init__(self, mats: Sequence, device="cuda", want_bsr4: bool = False):
        self.device = torch.device(device)
        self.n = len(mats)
        self.want_bsr4 = bool(want_bsr4)
        fwd = [m.tocsr() for m in mats]
        for m in fwd:
            m.sort_indices()
        self.rows = np.array([m.shape[0] for m in fwd], dtype=np.int64)
        self.cols = np.array([m.shape[1] for m in fwd], dtype=np.int64)
        self._fwd = self._upload(fwd)
        self._bwd = self._upload([m.T.tocsr() for m in fwd])   # one-time host transpose at load, like the dataset prep
        self._fwd_b = self._bwd_b = None
        if self.want_bsr4:
            self._fwd_b = self._pool_bsr4(self._fwd, self.rows, self.cols)
            self._bwd_b = self._pool_bsr4(self._bwd, self.cols, self.rows)

    # pooled CSR: dict(rowptr, colind, vals device tensors; rp_off, e_off, cnt host int64 arrays)
    def _upload(self, mats):
        for m in mats:
            m.sort_indices()
        rp_off = np.zeros(len(mats) + 1, dtype=np.int64)
        e_off = np.zeros(len(mats) + 1, dtype=np.int64)
        for i, m in enumerate(mats):
            rp_off[i + 1] = rp_off[i] + m.shape[0] + 1
            e_off[i + 1] = e_off[i] + m.nnz
        cat = lambda parts, dt: torch.from_numpy(np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)).to(self.device)
        return {
            "rowptr": cat([m.indptr for m in mats], np.int32),
            "colind": cat([m.indices for m in mats], np.int32),
            "vals": cat([m.data for m in mats], np.float32),
            "rp_off": rp_off, "e_off": e_off, "cnt": np.diff(e_off),
        }

    def _pool_bsr4(self, pool, rows, cols):
        """Convert each pooled mesh to BSR4 on the device, then re-pool (runs once per dataset)."""
        parts = []
        for i in range(self.n):
            rp = pool["rowptr"][pool["rp_off"][i]: pool["rp_off"][i + 1]]
            sl = slice(int(pool["e_off"][i]), int(pool["e_off"][i + 1]))
            parts.append(kernels.csr_to_bsr4(rp, pool["colind"][sl], pool["vals"][sl], int(rows[i]), int(cols[i])))
        rp_off = np.zeros(self.n + 1, dtype=np.int64)
        e_off = np.zeros(self.n + 1, dtype=np.int64)
        for i, (brp, bci, _) in enumerate(parts):
            rp_off[i + 1] = rp_off[i] + brp.numel()
            e_off[i + 1] = e_off[i] + bci.numel()
        return {
            "rowptr": torch.cat([p[0] for p in parts]), "colind": torch.cat([p[1] for p in parts]),
            "vals": torch.cat([p[2] for p in parts]), "rp_off": rp_off, "e_off": e_off, "cnt": np.diff(e_off),
        }

    def _concat(self, pool, sel, nrows, size0, size1, vpe):
        cnt = pool["cnt"][sel]
        out_off = np.zeros(len(sel) + 1, dtype=np.int64)
        np.cumsum(cnt, out=out_off[1:])
        desc = np.stack([pool["rp_off"][sel], pool["e_off"][sel], nrows, out_off[:-1]], axis=1)
        desc_d = torch.from_numpy(np.ascontiguousarray(desc)).to(self.device, non_blocking=True)
        return kernels.blockdiag_concat(pool["rowptr"], pool["colind"], pool["vals"], desc_d, size0, size1,
                                        int(out_off[-1]), vpe)

    def assemble(self, sel, size0: int, size1: int) -> SparseOperator:
        sel = np.asarray(sel, dtype=np.int64)
        B = len(sel)
        rows, cols = self.rows[sel], self.cols[sel]
        if (rows > size0).any() or (cols > size1).any():
            raise ValueError("size0/size1 smaller than a selected mesh")
        f = self._concat(self._fwd, sel, rows, size0, size1, 1)
        b = self._concat(self._bwd, sel, cols, size1, size0, 1)
        op = SparseOperator(*f, (B * size0, B * size1), batch=B)
        opt = SparseOperator(*b, (B * size1, B * size0), batch=B, transpose=op)
        op._t = opt
        if self.want_bsr4:
            if size0 % 4 or size1 % 4:
                raise ValueError("BSR4 pools need size0 and size1 to be multiples of 4")
            op._bsr4 = self._concat(self._fwd_b, sel, rows // 4, size0 // 4, size1 // 4, 16)
            opt._bsr4 = self._concat(self._bwd_b, sel, cols // 4, size1 // 4, size0 // 4, 16)
        else:
            op._bsr4 = opt._bsr4 = False
        return op
