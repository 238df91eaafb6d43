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

import os
import weakref
from typing import Optional, Sequence

import numpy as np
import torch

from . import kernels

__all__ = ["SparseOperator", "OperatorPool", "PackedSegments", "as_operator", "dirac_operators_from_mesh",
           "laplacian_operator_from_mesh"]

# A BSR4 copy is kept when zero-fill costs at most this much extra storage over CSR entries.
_BSR4_MAX_FILL = 1.6
# Form in which pools of Dirac operators assemble a batch: "q3" (quaternion-packed, default) or "bsr4"
# (functional.set_dirac_format switches both this and the kernel choice).
_POOL_FORMAT = "q3"
# SN_DEBUG_VALIDATE=1: every operator built from CSR arrays is checked on the device (SparseOperator.validate)
_DEBUG_VALIDATE = os.environ.get("SN_DEBUG_VALIDATE", "0") == "1"


class SparseOperator:
    """M x K fp32 sparse operator resident on one GPU (CSR, int32 indices).

    Quacks enough like the torch sparse tensors the reference models pass around: `.size()`, `.shape`,
    `.dim()`, `.is_cuda`, `.cuda()` (src/as_rigid_as_possible/models.py:133-136 inspects `Di.size()`).
    No gradient ever flows to an operator (src/utils/cuda/sparse_bmm_func.py:72).
    """

    layout = "sn_csr"
    requires_grad = False
    _format = None

    def __init__(self, rowptr, colind, vals, shape, *, batch: int = 1, transpose: "Optional[SparseOperator]" = None,
                 bsr4=None, q3=None):
        M, K = int(shape[0]), int(shape[1])
        self._shape = (M, K)
        if rowptr is None:
            if not isinstance(bsr4, tuple) and not isinstance(q3, tuple):
                raise ValueError("an operator needs CSR arrays, a BSR4 triple or a Q3 pair")
            self._csr = None                     # block-form-only operator (Dirac); CSR is expanded on demand
        else:
            if rowptr.dtype != torch.int32 or colind.dtype != torch.int32 or vals.dtype != torch.float32:
                raise TypeError("SparseOperator wants int32 rowptr/colind and float32 vals")
            if rowptr.numel() != M + 1 or colind.numel() != vals.numel():
                raise ValueError("inconsistent CSR arrays")
            self._csr = (rowptr, colind, vals)
            if _DEBUG_VALIDATE and rowptr.is_cuda:
                self.validate()
        self._nnz_cache = None
        self.batch = int(batch)                  # number of diagonal blocks (B of the reference's (B,R,K) operators)
        self.row_offsets = self.col_offsets = None   # packed (unpadded) batches: first row / column of every diagonal block
        # operator -> transpose is a strong reference, transpose -> operator a weak one: a strong pair would be a
        # reference cycle, and the per-step batch operators (hundreds of MB of HBM each) would then live until Python's
        # cyclic collector happens to run instead of being returned to the caching allocator when the step drops them
        self._t_strong = None
        self._t_weak = weakref.ref(transpose) if transpose is not None else None
        self._bsr4 = bsr4                        # (b_rowptr, b_colind, b_vals) | None | False (= not worthwhile)
        self._q3 = q3                            # (b_rowptr, q_blk) quaternion-packed form | None (unknown) | False (not a Dirac-type operator)
        self._rb4 = None                         # (b_ptr, b_col, b_val) 4x1 row-blocked form | None (not built) | False (not worthwhile)
        self._band = None                        # (max |column - row|, longest row, rows outside the ring window) | None (not measured)

    @property
    def format(self):
        """Storage form this operator's products take ("q3" | "bsr4" | "ring" | "rb4" | "csr"); None: the process default
        (functional.set_dirac_format / set_laplacian_format).  Shared with the attached transpose."""
        return self._format

    @format.setter
    def format(self, fmt):
        if fmt not in (None, "q3", "bsr4", "ring", "rb4", "csr"):
            raise ValueError(fmt)
        self._format = fmt
        t = self._t
        if t is not None:
            t._format = fmt

    @property
    def _t(self) -> "Optional[SparseOperator]":
        if self._t_strong is not None:
            return self._t_strong
        return self._t_weak() if self._t_weak is not None else None

    @_t.setter
    def _t(self, value):
        self._t_strong, self._t_weak = value, None

    # ---- tensor-like surface -------------------------------------------------------------------
    @property
    def shape(self):
        return torch.Size(self._shape)

    def size(self, dim: Optional[int] = None):
        return self.shape if dim is None else self._shape[dim]

    def dim(self) -> int:
        return 2

    # CSR arrays; for a BSR4-only operator they are expanded once (explicit zeros of the blocks dropped)
    def _ensure_csr(self):
        if self._csr is None:
            self._csr = _bsr4_to_csr(*self.bsr4(), self._shape[0])
        return self._csr

    @property
    def rowptr(self):
        return self._ensure_csr()[0]

    @property
    def colind(self):
        return self._ensure_csr()[1]

    @property
    def vals(self):
        return self._ensure_csr()[2]

    @property
    def nnz(self) -> int:
        if self._nnz_cache is None:
            if self._csr is not None:
                self._nnz_cache = int(self._csr[1].numel())
            elif isinstance(self._bsr4, tuple):
                self._nnz_cache = int(torch.count_nonzero(self._bsr4[2]).item())
            else:                                # Q3-only operator: every non-zero parameter appears four times in its block
                self._nnz_cache = 4 * int(torch.count_nonzero(self._q3[1][:, :3]).item())
        return self._nnz_cache

    def _nnz(self) -> int:
        return self.nnz

    @property
    def device(self):
        if self._csr is not None:
            return self._csr[0].device
        return (self._bsr4[0] if isinstance(self._bsr4, tuple) else self._q3[0]).device

    @property
    def is_cuda(self) -> bool:
        return self.device.type == "cuda"

    def cuda(self, *_, **__):
        return self.to("cuda")

    def _moved(self, device, transpose=None):
        """A copy of this operator's own arrays on `device` (the transpose is handled by to())."""
        mv = lambda t: t.to(device)
        b = tuple(mv(t) for t in self._bsr4) if isinstance(self._bsr4, tuple) else self._bsr4
        q = tuple(mv(t) for t in self._q3) if isinstance(self._q3, tuple) else self._q3
        csr = (None, None, None) if self._csr is None else tuple(mv(t) for t in self._csr)
        out = SparseOperator(*csr, self._shape, batch=self.batch, transpose=transpose, bsr4=b, q3=q)
        out._nnz_cache = self._nnz_cache
        out._band = self._band
        out.row_offsets, out.col_offsets = self.row_offsets, self.col_offsets
        return out

    def _cloned(self, transpose=None):
        cp = lambda t: t.clone()
        b = tuple(cp(t) for t in self._bsr4) if isinstance(self._bsr4, tuple) else self._bsr4
        q = tuple(cp(t) for t in self._q3) if isinstance(self._q3, tuple) else self._q3
        csr = (None, None, None) if self._csr is None else tuple(cp(t) for t in self._csr)
        out = SparseOperator(*csr, self._shape, batch=self.batch, transpose=transpose, bsr4=b, q3=q)
        out._rb4 = tuple(cp(t) for t in self._rb4) if isinstance(self._rb4, tuple) else self._rb4
        out._band = self._band
        out._nnz_cache = self._nnz_cache
        out.row_offsets, out.col_offsets = self.row_offsets, self.col_offsets
        return out

    def clone(self):
        """A deep copy (every materialised form, the attached transpose included): the static batch of a captured step is
        overwritten in place by every load, so it must not share arrays with a dataset's cached operators."""
        out = self._cloned()
        if self._t is not None:
            out._t = self._t._cloned(transpose=out)
        return out

    def to(self, device):
        device = torch.device(device)
        if device.type == self.device.type and (device.index is None or device.index == self.device.index):
            return self
        out = self._moved(device)
        t = self._t
        if t is not None:
            # the transpose's arrays are moved directly: t.to() would come back here through its own transpose link
            out._t = t._moved(device, transpose=out)
        return out

    def __repr__(self):
        return f"SparseOperator(shape={self._shape}, nnz={self.nnz}, batch={self.batch}, device={self.device})"

    def validate(self) -> None:
        """Debug check of the CSR arrays on the device (sn_validate_csr_i32): row pointers monotone from 0 to nnz, column
        indices inside [0, K) and strictly ascending per row, finite values.  Raises ValueError naming the defects.  Run on
        every operator built from CSR arrays when SN_DEBUG_VALIDATE=1 (the product kernels themselves do not bounds-check,
        like the reference's)."""
        rp, ci, va = self._ensure_csr()
        flags = kernels.validate_csr(rp, ci, va, self._shape[0], self._shape[1])
        if flags:
            names = {1: "rowptr[0] != 0", 2: "rowptr decreasing", 4: "rowptr[M] != nnz", 8: "column index out of range",
                     16: "row not sorted / duplicate columns", 32: "non-finite value"}
            raise ValueError("invalid CSR operator: " + ", ".join(v for k, v in names.items() if flags & k))

    # ---- derived forms --------------------------------------------------------------------------
    def t(self) -> "SparseOperator":
        """CSR of the transpose, built on first use by sn_csr_transpose_f32 and cached both ways."""
        t = self._t
        if t is None:
            M, K = self._shape
            tr, tc, tv = kernels.csr_transpose(self.rowptr, self.colind, self.vals, M, K)
            t = self._t = SparseOperator(tr, tc, tv, (K, M), batch=self.batch, transpose=self)
            t._format = self._format
        return t

    T = property(t)

    @classmethod
    def from_bsr4(cls, fwd, bwd, shape, batch: int = 1) -> "SparseOperator":
        """Operator and its transpose given directly as BSR4 triples (kernels.dirac_from_mesh)."""
        M, K = int(shape[0]), int(shape[1])
        op = cls(None, None, None, (M, K), batch=batch, bsr4=tuple(fwd))
        opt = cls(None, None, None, (K, M), batch=batch, bsr4=tuple(bwd), transpose=op)
        op._t = opt
        return op

    def q3(self):
        """(b_rowptr, q_blk) — the quaternion-packed form (three floats + the block column per 4x4 block) — or None when the
        operator's blocks are not pure-quaternion matrices.  Built from the BSR4 form on first use; the check reads one
        flag back from the device (operators assembled by an OperatorPool arrive with the form already attached)."""
        if self._q3 is None:
            b = self.bsr4()
            if b is None:
                self._q3 = False
            else:
                q, flag = kernels.bsr4_to_q3(b[1], b[2])
                self._q3 = (b[0], q.view(-1, 4)) if int(flag.item()) == 0 else False
        return self._q3 or None

    @classmethod
    def from_q3(cls, fwd, bwd, shape, batch: int = 1) -> "SparseOperator":
        """Operator and its transpose given as quaternion-packed pairs (b_rowptr, q_blk) — what an OperatorPool of Dirac
        operators assembles per step."""
        M, K = int(shape[0]), int(shape[1])
        op = cls(None, None, None, (M, K), batch=batch, q3=tuple(fwd))
        opt = cls(None, None, None, (K, M), batch=batch, q3=tuple(bwd), transpose=op)
        op._t = opt
        return op

    def rb4(self):
        """(b_ptr, b_col, b_val): the 4x1 row-blocked form used by the Laplacian-type (group-1) products, built on the
        device on first use without a host synchronisation; None for operators with fewer than 2 entries per row on
        average (nothing to share between the rows of a group)."""
        if self._rb4 is None:
            M, K = self._shape
            if self._band is not None and self._band[1] == 0x7fffffff:
                self._rb4 = False                  # a row with non-ascending columns (band probe): the 4-way merge needs sorted rows
            elif self.nnz < 2 * M:
                self._rb4 = False
            else:
                self._rb4 = kernels.csr_to_rb4(self.rowptr, self.colind, self.vals, M, K)
        return self._rb4 or None

    def band(self):
        """(max |column - row|, longest row, rows with an entry outside the ring kernel's window) of the CSR arrays — what
        decides between the sliding-window kernel and the gather kernels for Laplacian-type products.  Measured on the device on
        first use (one host synchronisation per operator); operators assembled by an OperatorPool arrive with the values
        already attached (maxima / sums over their meshes)."""
        if self._band is None:
            M, K = self._shape
            self._band = kernels.csr_band(self.rowptr, self.colind, M, K)
        return self._band

    def ring_ok(self, N: int) -> bool:
        """True when the Laplacian-type product of this operator at N dense columns should take the sliding-window kernel:
        square, large enough to fill the chip with persistent strips, (nearly) every entry within the kernel's window of its
        row (the few rows that reach further — the wrap-around rows of closed meshes — gather from global memory in the kernel)."""
        M, K = self._shape
        if not kernels.spmm_ring_supported(N, 1, M, K):
            return False
        if self._band is None and self.is_cuda and torch.cuda.is_current_stream_capturing():
            return False                                   # measuring the band reads back to the host: never inside a capture
        band, longest, outside = self.band()
        return longest <= 32 and outside <= kernels.RING_MAX_OUTSIDE * M

    def bsr4(self):
        """(b_rowptr, b_colind, b_vals) or None when the 4x4-block form is not applicable / not worthwhile."""
        if self._bsr4 is None and self._csr is None and isinstance(self._q3, tuple):
            self._bsr4 = _q3_to_bsr4(*self._q3)          # export / fallback only
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


def _q3_to_bsr4(b_rowptr, q_blk):
    """Expand quaternion-packed records to BSR4 (torch arithmetic; export / generic-kernel fallback only)."""
    p1, p2, p3 = q_blk[:, 0], q_blk[:, 1], q_blk[:, 2]
    col = q_blk[:, 3].contiguous().view(torch.int32)
    z = torch.zeros_like(p1)
    rows = [torch.stack([z, p1, p2, p3], 1), torch.stack([-p1, z, p3, -p2], 1), torch.stack([-p2, -p3, z, p1], 1),
            torch.stack([-p3, p2, -p1, z], 1)]
    return b_rowptr, col.clone(), torch.stack(rows, 1).reshape(-1).contiguous()


def _bsr4_to_csr(b_rowptr, b_colind, b_vals, M: int):
    """Expand BSR4 to CSR with torch index arithmetic (not on the hot path: export / generic-kernel fallback only).
    Explicit zeros of the blocks are dropped so that the result equals the coalesced operator."""
    dev = b_rowptr.device
    Mb = M // 4
    nblk = int(b_colind.numel())
    cnt = (b_rowptr[1:] - b_rowptr[:-1]).long()                                   # blocks per block row
    brow = torch.repeat_interleave(torch.arange(Mb, device=dev), cnt)             # block row of each block
    q = torch.arange(4, device=dev)
    rows = (4 * brow[:, None, None] + q[None, :, None]).expand(nblk, 4, 4)
    cols = (4 * b_colind.long()[:, None, None] + q[None, None, :]).expand(nblk, 4, 4)
    v = b_vals.view(nblk, 4, 4)
    keep = v != 0
    rows, cols, v = rows[keep], cols[keep], v[keep]
    order = torch.argsort(rows * (4 * (int(b_colind.max().item()) + 1 if nblk else 1)) + cols)
    rows, cols, v = rows[order], cols[order], v[order]
    rowptr = torch.zeros(M + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=M), 0)
    return rowptr.int(), cols.int(), v.contiguous()


def dirac_operators_from_mesh(V: torch.Tensor, F: torch.Tensor):
    """(Di, DiA) SparseOperators (with transposes attached) built on the device from vertex positions V and faces F —
    the on-the-fly replacement of the per-frame operators the reference precomputes with mesh.dirac
    (src/as_rigid_as_possible/add_laplacian.py:50-65).

    V: (nV,3) and F: (nF,3) for one mesh, or V: (B,nV,3) and F: (nF,3) shared / (B,nF,3) per-mesh faces for a batch
    of equally sized meshes, which yields the block-diagonal batched operators directly."""
    if V.dim() == 3:
        B, nV = V.shape[0], V.shape[1]
        Fb = F if F.dim() == 3 else F.unsqueeze(0).expand(B, -1, -1)
        nF = Fb.shape[1]
        off = (torch.arange(B, device=V.device, dtype=torch.int32) * nV).view(B, 1, 1)
        Fg = (Fb.to(torch.int32) + off).reshape(B * nF, 3)
        Vg = V.reshape(B * nV, 3)
    else:
        B, nV, nF = 1, V.shape[0], F.shape[0]
        Vg, Fg = V, F.to(torch.int32)
    di, diat, dia, dit = kernels.dirac_from_mesh(Vg.float(), Fg)
    Di = SparseOperator.from_bsr4(di, dit, (4 * B * nF, 4 * B * nV), batch=B)
    DiA = SparseOperator.from_bsr4(dia, diat, (4 * B * nV, 4 * B * nF), batch=B)
    for o in (Di, Di._t, DiA, DiA._t):           # quaternion blocks by construction: pack without reading the check flag
        o._q3 = (o._bsr4[0], kernels.bsr4_to_q3(o._bsr4[1], o._bsr4[2])[0].view(-1, 4))
    return Di, DiA


def laplacian_operator_from_mesh(V: torch.Tensor, F: torch.Tensor) -> SparseOperator:
    """L = A^-1 (D - W) built on the device (sn_laplacian_csr_from_mesh) — replaces the host pipeline
    mesh.cotangent_weights + graph.laplacian (src/mesh_mnist/add_laplacian.py:43-48).  V: (nV,3) or (B,nV,3) for a batch
    of equally sized meshes (block-diagonal result); F: (nF,3) shared or (B,nF,3)."""
    if V.dim() == 3:
        B, nV = V.shape[0], V.shape[1]
        Fb = F if F.dim() == 3 else F.unsqueeze(0).expand(B, -1, -1)
        off = (torch.arange(B, device=V.device, dtype=torch.int32) * nV).view(B, 1, 1)
        Fg = (Fb.to(torch.int32) + off).reshape(-1, 3)
        Vg = V.reshape(B * nV, 3)
    else:
        B, nV = 1, V.shape[0]
        Vg, Fg = V, F.to(torch.int32)
    rowptr, colind, vals = kernels.laplacian_from_mesh(Vg.float(), Fg)
    return SparseOperator(rowptr, colind, vals, (B * nV, B * nV), batch=B)


class PackedSegments:
    """Per-mesh row ranges of a PACKED batch (the concatenation of the meshes' node rows, no padding) — what takes the place
    of the reference's (B, Vmax, 1) `mask` tensor when a model runs on a packed batch: pass it as the `mask` argument of
    AvgResNet2 / global_average / the task models.  Holds the host lengths and, on the device, the tile table of
    sn_segment_colsum_ragged_f32 (tiles of <= 256 rows, a mesh's tiles consecutive) and 1 / vertex count per mesh."""

    TILE = 256

    def __init__(self, lengths, device="cuda"):
        self.lengths = np.asarray(lengths, dtype=np.int64)
        if self.lengths.ndim != 1 or (self.lengths < 1).any():
            raise ValueError("PackedSegments: one positive row count per mesh")
        self.nseg = int(len(self.lengths))
        self.offsets = np.zeros(self.nseg + 1, dtype=np.int64)
        np.cumsum(self.lengths, out=self.offsets[1:])
        self.rows = int(self.offsets[-1])
        per = (self.lengths + self.TILE - 1) // self.TILE
        seg_tile_ptr = np.zeros(self.nseg + 1, dtype=np.int64)
        np.cumsum(per, out=seg_tile_ptr[1:])
        seg = np.repeat(np.arange(self.nseg), per)
        k = np.arange(int(seg_tile_ptr[-1])) - seg_tile_ptr[seg]                   # tile index inside its mesh
        first = self.offsets[seg] + k * self.TILE
        cnt = np.minimum(self.TILE, self.offsets[seg + 1] - first)
        self.device = torch.device(device)
        self.tiles = h2d_async(np.stack([seg, first, cnt], axis=1), self.device)
        self.seg_tile_ptr = h2d_async(seg_tile_ptr, self.device)
        self.inv_count = h2d_async((1.0 / self.lengths).astype(np.float32), self.device)
        self.off_dev = h2d_async(self.offsets, self.device)                        # (nseg + 1,) int64: first row of every mesh
        # row slabs of the weight-gradient pass (sn_wgrad_slabs_f32): about 256 in all, none crossing a mesh boundary
        target = max(256, -(-self.rows // 256))
        nsl = np.maximum(1, -(-self.lengths // target))
        slab_ptr = np.zeros(self.nseg + 1, dtype=np.int64)
        np.cumsum(nsl, out=slab_ptr[1:])
        sseg = np.repeat(np.arange(self.nseg), nsl)
        sk = np.arange(int(slab_ptr[-1])) - slab_ptr[sseg]
        per = -(-self.lengths // nsl)
        per = (per + 15) // 16 * 16
        s0 = np.minimum(self.offsets[sseg] + sk * per[sseg], self.offsets[sseg + 1])
        self.nslab = int(slab_ptr[-1])
        self.slab_off = h2d_async(np.concatenate([s0, self.offsets[-1:]]), self.device)
        self.seg_slab_ptr = h2d_async(slab_ptr, self.device)
        self.len_f64 = h2d_async(self.lengths.astype(np.float64), self.device)
        self.min_len = int(self.lengths.min())

    _recent = {}

    @classmethod
    def cached(cls, lengths, device="cuda") -> "PackedSegments":
        """The segments of `lengths`, re-used when the same sizes come back (a sampler that draws the same meshes every step
        would otherwise upload the eight small tables again per step); holds the last 16 distinct size lists."""
        key = (tuple(int(v) for v in lengths), str(torch.device(device)))
        hit = cls._recent.get(key)
        if hit is None:
            if len(cls._recent) >= 16:
                cls._recent.pop(next(iter(cls._recent)))
            hit = cls._recent[key] = cls(lengths, device)
        return hit

    @classmethod
    def of_operator(cls, op: "SparseOperator", group: int = 1, side: str = "cols") -> "PackedSegments":
        """Row ranges of the dense operand of a packed operator: its column blocks (`side="cols"`, the input side) or its
        row blocks, divided by `group` (4 for the quaternion view of the Dirac operators)."""
        off = op.col_offsets if side == "cols" else op.row_offsets
        if off is None:
            raise ValueError("not a packed operator (OperatorPool.assemble(sel) without sizes)")
        return cls(np.diff(off) // group, op.device)

    def mean(self, x2d: torch.Tensor) -> torch.Tensor:
        """(nseg, C): per-mesh column means of the (rows, C) operand."""
        if x2d.shape[0] != self.rows:
            raise ValueError(f"PackedSegments: {self.rows} rows expected, got {x2d.shape[0]}")
        return kernels.segment_colsum_ragged(x2d, self.tiles, self.seg_tile_ptr, self.nseg, self.inv_count)


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


def h2d_async(arr, device) -> torch.Tensor:
    """Small host array -> device tensor WITHOUT stalling the host on the stream: staged through pinned memory (torch's
    caching host allocator keeps the staging block alive until the copy has run), so the per-step index / descriptor
    uploads of the samplers do not wait for the previous step's kernels — a pageable-source copy does."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    dev = torch.device(device)
    if dev.type != "cuda":
        return t.to(dev, non_blocking=True)
    stage = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    stage.copy_(t)
    return stage.to(dev, non_blocking=True)


class OperatorPool:
    """All per-mesh operators of one kind (e.g. every Di of a dataset) resident in HBM, A and A^T.

    `assemble(sel, size0, size1)` returns the batched block-diagonal SparseOperator of the selected meshes —
    identical (after to_dense) to the reference's `sparse_diag_cat([op[i] for i in sel], size0, size1)` — with its
    transpose attached, so neither forward nor backward ever sorts, transposes or touches the host.
    """

    def __init__(self, mats: Sequence, device="cuda", want_bsr4: bool = False, lean: bool = False):
        """lean=True: keep only the arrays batches are assembled from — for pools of Dirac operators that turn out to be
        quaternion-packed, the pooled CSR and BSR4 device arrays are dropped (the CSR entry counts stay on the host)."""
        self.device = torch.device(device)
        self.n = len(mats)
        self.want_bsr4 = bool(want_bsr4)
        self.lean = bool(lean)
        fwd = [m.tocsr() for m in mats]
        for m in fwd:
            m.sort_indices()
        self.rows = np.array([m.shape[0] for m in fwd], dtype=np.int64)
        self.cols = np.array([m.shape[1] for m in fwd], dtype=np.int64)
        self._fwd = self._upload(fwd)
        self._bwd = self._upload([m.T.tocsr() for m in fwd])   # one-time host transpose at load, like the dataset prep
        # per mesh: (max |column - row|, longest row, rows outside the ring window) of the operator and of its transpose — a
        # batch whose blocks sit on the diagonal (equal row and column offsets) inherits them without a device pass
        self._band_fwd = np.array([self._mesh_band(m) for m in fwd], dtype=np.int64).reshape(-1, 3)
        self._band_bwd = np.array([self._mesh_band(m.T.tocsr()) for m in fwd], dtype=np.int64).reshape(-1, 3)
        self._fwd_b = self._bwd_b = None
        self._fwd_q = self._bwd_q = None
        if self.want_bsr4:
            self._fwd_b = self._pool_bsr4(self._fwd, self.rows, self.cols)
            self._bwd_b = self._pool_bsr4(self._bwd, self.cols, self.rows)
            # quaternion-packed pools when every block of every mesh is a pure-quaternion matrix (Dirac operators are)
            fq, ff = kernels.bsr4_to_q3(self._fwd_b["colind"], self._fwd_b["vals"])
            bq, bf = kernels.bsr4_to_q3(self._bwd_b["colind"], self._bwd_b["vals"])
            if int(ff.item()) == 0 and int(bf.item()) == 0:
                self._fwd_q = dict(self._fwd_b, vals=fq.reshape(-1), colind=None)
                self._bwd_q = dict(self._bwd_b, vals=bq.reshape(-1), colind=None)
            if self.lean and self._fwd_q is not None:
                self._fwd_b = self._bwd_b = None
                for pool in (self._fwd, self._bwd):
                    pool["rowptr"] = pool["colind"] = pool["vals"] = None

    # ---- growing pools (utils_pt's resident cache appends the operators a driver shows it for the first time) -----------
    @staticmethod
    def _arena_append(buf, used: int, new):
        """`new` copied behind the first `used` elements of `buf`; the buffer doubles when it is full (offsets into it stay
        valid: the assembly kernels address pooled arrays by element offsets, never by their length)."""
        need = used + int(new.numel())
        if buf is None or need > buf.numel():
            cap = max(need, 2 * (int(buf.numel()) if buf is not None else 0), 1 << 16)
            grown = torch.empty(cap, dtype=new.dtype, device=new.device)
            if used:
                grown[:used].copy_(buf[:used])
            buf = grown
        buf[used:need].copy_(new.reshape(-1))
        return buf

    @classmethod
    def _merge(cls, dst, src, vpe: int):
        if src is None or dst is None:
            return None
        used_rp, used_e = int(dst["rp_off"][-1]), int(dst["e_off"][-1])
        for key, used, n_src in (("rowptr", used_rp, int(src["rp_off"][-1])), ("colind", used_e, int(src["e_off"][-1])),
                                 ("vals", used_e * vpe, int(src["e_off"][-1]) * vpe)):
            if dst[key] is not None and src[key] is not None:
                dst[key] = cls._arena_append(dst[key], used, src[key][:n_src])
        dst["rp_off"] = np.concatenate([dst["rp_off"], used_rp + src["rp_off"][1:]])
        dst["e_off"] = np.concatenate([dst["e_off"], used_e + src["e_off"][1:]])
        dst["cnt"] = np.diff(dst["e_off"])
        return dst

    def append(self, mats: Sequence) -> np.ndarray:
        """Add operators to the pool (converted like the constructor's: one-time host transpose, device BSR4 / quaternion
        packing); returns their indices for assemble().  A pool that was quaternion-packed stays so only while every added
        operator is — the caller (utils_pt's resident cache) keeps such operators in a pool of their own."""
        if not len(mats):
            return np.zeros(0, dtype=np.int64)
        return self.absorb(OperatorPool(mats, self.device, self.want_bsr4, self.lean))

    def absorb(self, chunk: "OperatorPool") -> np.ndarray:
        """Merge another pool of the same kind into this one (its arrays are copied behind this pool's); returns the indices
        its operators now have here."""
        if chunk.want_bsr4 != self.want_bsr4 or chunk.device != self.device:
            raise ValueError("OperatorPool.absorb: pools of different kinds")
        if (chunk._fwd_q is None) != (self._fwd_q is None) or (chunk._fwd_b is None) != (self._fwd_b is None):
            raise ValueError("OperatorPool.append: the new operators do not pack the way this pool's do")
        first = self.n
        self._fwd = self._merge(self._fwd, chunk._fwd, 1)
        self._bwd = self._merge(self._bwd, chunk._bwd, 1)
        fwd_rowptr_shared = self._fwd_q is not None and self._fwd_b is not None
        self._fwd_b = self._merge(self._fwd_b, chunk._fwd_b, 16)
        self._bwd_b = self._merge(self._bwd_b, chunk._bwd_b, 16)
        if fwd_rowptr_shared:                                 # (the quaternion dicts share block-row pointers and offsets)
            for q, b, cq in ((self._fwd_q, self._fwd_b, chunk._fwd_q), (self._bwd_q, self._bwd_b, chunk._bwd_q)):
                used = int(q["e_off"][-1])
                q["vals"] = self._arena_append(q["vals"], used * 4, cq["vals"][: int(cq["e_off"][-1]) * 4])
                q["rowptr"], q["rp_off"], q["e_off"], q["cnt"] = b["rowptr"], b["rp_off"], b["e_off"], b["cnt"]
        else:
            self._fwd_q = self._merge(self._fwd_q, chunk._fwd_q, 4)
            self._bwd_q = self._merge(self._bwd_q, chunk._bwd_q, 4)
        self.rows = np.concatenate([self.rows, chunk.rows])
        self.cols = np.concatenate([self.cols, chunk.cols])
        self._band_fwd = np.concatenate([self._band_fwd, chunk._band_fwd])
        self._band_bwd = np.concatenate([self._band_bwd, chunk._band_bwd])
        self.n += chunk.n
        return np.arange(first, self.n, dtype=np.int64)

    def device_bytes(self) -> int:
        tot = 0
        seen = set()
        for pool in (self._fwd, self._bwd, self._fwd_b, self._bwd_b, self._fwd_q, self._bwd_q):
            for key in ("rowptr", "colind", "vals"):
                t = pool[key] if pool is not None else None
                if t is not None and t.data_ptr() not in seen:
                    seen.add(t.data_ptr())
                    tot += t.numel() * t.element_size()
        return tot

    @staticmethod
    def _mesh_band(m):
        m = m.tocsr()
        if m.nnz == 0:
            return (0, 0, 0)
        counts = np.diff(m.indptr)
        rows = np.repeat(np.arange(m.shape[0]), counts)
        far = np.abs(m.indices.astype(np.int64) - rows)
        outside = np.unique(rows[far > kernels.ring_half_window()]).size if far.max() > 0 else 0
        return (int(far.max()), int(counts.max()), int(outside))

    def _attach_band(self, op, sel, diagonal: bool):
        if diagonal and len(sel):
            f, b = self._band_fwd[sel], self._band_bwd[sel]
            op._band = (int(f[:, 0].max()), int(f[:, 1].max()), int(f[:, 2].sum()))
            if op._t is not None:
                op._t._band = (int(b[:, 0].max()), int(b[:, 1].max()), int(b[:, 2].sum()))

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
        desc_d = h2d_async(desc, self.device)
        return kernels.blockdiag_concat(pool["rowptr"], pool["colind"], pool["vals"], desc_d, size0, size1,
                                        int(out_off[-1]), vpe)

    def _concat_ragged(self, pool, sel, nrows, row_off, col_off, vpe):
        cnt = pool["cnt"][sel]
        out_off = np.zeros(len(sel) + 1, dtype=np.int64)
        np.cumsum(cnt, out=out_off[1:])
        desc = np.stack([pool["rp_off"][sel], pool["e_off"][sel], nrows, out_off[:-1], row_off[:-1], col_off[:-1]], axis=1)
        desc_d = h2d_async(desc, self.device)
        return kernels.blockdiag_concat_ragged(pool["rowptr"], pool["colind"], pool["vals"], desc_d, int(row_off[-1]),
                                               int(col_off[-1]), int(out_off[-1]), vpe)

    def assemble_packed(self, sel) -> SparseOperator:
        """The batch WITHOUT padding: mesh i owns rows [row_offsets[i], row_offsets[i+1]) and columns
        [col_offsets[i], col_offsets[i+1]) (prefix sums of the meshes' own sizes), so the dense operands are the plain
        concatenation of the meshes' node rows — (sum V_i, C) instead of the reference's (B, Vmax, C)
        (src/as_rigid_as_possible/main.py:172-185 pads to the batch maxima).  The product then neither scans padding rows
        nor writes zeros into them.  The offsets (host int64 arrays, length B+1) ride on the operator."""
        sel = np.asarray(sel, dtype=np.int64)
        B = len(sel)
        rows, cols = self.rows[sel], self.cols[sel]
        ro = np.zeros(B + 1, dtype=np.int64)
        co = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(rows, out=ro[1:])
        np.cumsum(cols, out=co[1:])
        shape = (int(ro[-1]), int(co[-1]))
        if self.want_bsr4:
            if (rows % 4).any() or (cols % 4).any():
                raise ValueError("BSR4 pools need every mesh's sizes to be multiples of 4")
            if self._fwd_q is not None and (_POOL_FORMAT == "q3" or self._fwd_b is None):
                fr, _, fq = self._concat_ragged(self._fwd_q, sel, rows // 4, ro // 4, co // 4, 4)
                br_, _, bq = self._concat_ragged(self._bwd_q, sel, cols // 4, co // 4, ro // 4, 4)
                op = SparseOperator.from_q3((fr, fq.view(-1, 4)), (br_, bq.view(-1, 4)), shape, batch=B)
            else:
                fb = self._concat_ragged(self._fwd_b, sel, rows // 4, ro // 4, co // 4, 16)
                bb = self._concat_ragged(self._bwd_b, sel, cols // 4, co // 4, ro // 4, 16)
                op = SparseOperator.from_bsr4(fb, bb, shape, batch=B)
            op._nnz_cache = op._t._nnz_cache = int(self._fwd["cnt"][sel].sum())
        else:
            f = self._concat_ragged(self._fwd, sel, rows, ro, co, 1)
            b = self._concat_ragged(self._bwd, sel, cols, co, ro, 1)
            op = SparseOperator(*f, shape, batch=B)
            opt = SparseOperator(*b, (shape[1], shape[0]), batch=B, transpose=op)
            op._t = opt
            op._bsr4 = opt._bsr4 = False
            self._attach_band(op, sel, bool((rows == cols).all()))
        op.row_offsets, op.col_offsets = ro, co
        op._t.row_offsets, op._t.col_offsets = co, ro
        return op

    def assemble(self, sel, size0: Optional[int] = None, size1: Optional[int] = None) -> SparseOperator:
        """size0/size1 = the padded block size of the reference's sparse_diag_cat; both None: packed (assemble_packed)."""
        if size0 is None and size1 is None:
            return self.assemble_packed(sel)
        if size0 is None or size1 is None:
            raise ValueError("assemble: give both block sizes (padded batch) or neither (packed batch)")
        sel = np.asarray(sel, dtype=np.int64)
        B = len(sel)
        rows, cols = self.rows[sel], self.cols[sel]
        if (rows > size0).any() or (cols > size1).any():
            raise ValueError("size0/size1 smaller than a selected mesh")
        if self.want_bsr4:
            # Dirac pools: only the packed blocks are assembled per step (the products use them exclusively); the CSR
            # view of the batch is expanded from the blocks on demand (export / generic-kernel fallback).
            if size0 % 4 or size1 % 4:
                raise ValueError("BSR4 pools need size0 and size1 to be multiples of 4")
            if self._fwd_q is not None and (_POOL_FORMAT == "q3" or self._fwd_b is None):
                fr, _, fq = self._concat(self._fwd_q, sel, rows // 4, size0 // 4, size1 // 4, 4)
                br_, _, bq = self._concat(self._bwd_q, sel, cols // 4, size1 // 4, size0 // 4, 4)
                op = SparseOperator.from_q3((fr, fq.view(-1, 4)), (br_, bq.view(-1, 4)), (B * size0, B * size1), batch=B)
            else:
                fb = self._concat(self._fwd_b, sel, rows // 4, size0 // 4, size1 // 4, 16)
                bb = self._concat(self._bwd_b, sel, cols // 4, size1 // 4, size0 // 4, 16)
                op = SparseOperator.from_bsr4(fb, bb, (B * size0, B * size1), batch=B)
            op._nnz_cache = op._t._nnz_cache = int(self._fwd["cnt"][sel].sum())      # entries of the pooled CSR: no sync
            return op
        f = self._concat(self._fwd, sel, rows, size0, size1, 1)
        b = self._concat(self._bwd, sel, cols, size1, size0, 1)
        op = SparseOperator(*f, (B * size0, B * size1), batch=B)
        opt = SparseOperator(*b, (B * size1, B * size0), batch=B, transpose=op)
        op._t = opt
        op._bsr4 = opt._bsr4 = False
        self._attach_band(op, sel, size0 == size1)
        return op
