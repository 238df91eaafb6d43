"""Drop-in for the reference operator layer `utils.utils_pt` (src/utils/utils_pt.py) on MI355X.

    import surfacenetworks_amd.utils_pt as utils        # instead of: import utils.utils_pt as utils

Same public names, call signatures and `state_dict` keys as the reference module:
GraphConv1x1, GraphBatchNorm, global_average, LapResNet2, DenseLapResNet2, DirResNet2, AvgResNet2,
MlpResNet2, sparse_cat, sparse_diag_cat, sp_sparse_to_pt_sparse, to_dense_batched — plus SparseBMMFunc, the
name utils_pt.py:199,211 uses without importing.

What differs is how a block executes.  Every sparse product goes through the hand-written HIP kernels
(functional.py -> kernels.py -> csrc/sn_kernels.hip); ELU, the product and torch.cat are fused into one
(rows, 2C) buffer; BatchNorm runs on the (rows, C) view instead of two transposed copies.  Operators may be
`SparseOperator`s (resident CSR, see operators.py) or the torch sparse COO tensors the reference drivers
build (2-D block-diagonal from sparse_diag_cat or 3-D batched from sparse_cat) — those are converted on the
device once per tensor.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import blocks as snB
from . import functional as snF
from .operators import PackedSegments, SparseOperator, as_operator
from .resident import LazySparse, batch_operator, reference_cat, reference_diag_cat

__all__ = [
    "sparse_cat", "sparse_diag_cat", "sp_sparse_to_pt_sparse", "to_dense_batched", "GraphConv1x1",
    "GraphBatchNorm", "global_average", "DenseLapResNet2", "LapResNet2", "DirResNet2", "AvgResNet2",
    "MlpResNet2", "SparseBMMFunc", "PackedSegments",
]


# --------------------------------------------------------------------------------------------------
# host-side operator batching with the reference's return types (torch sparse COO, coalesced)
# --------------------------------------------------------------------------------------------------
def sp_sparse_to_pt_sparse(L):
    """scipy sparse matrix -> torch sparse COO (uncoalesced, dtype kept), as utils_pt.py:56-69 — returned as a `LazySparse`
    handle of the scipy matrix (resident.py): the index arrays are built only if something reads them; sparse_diag_cat /
    sparse_cat assemble batches of such handles on the device from resident copies."""
    return LazySparse.of_scipy(L)


def sparse_diag_cat(tensors, size0, size1):
    """Block-diagonal (len*size0, len*size1) operator from per-mesh COO operators, as utils_pt.py:41-53.
    Members that are handles of scipy matrices (what sp_sparse_to_pt_sparse returns) are batched ON THE DEVICE from their
    resident copies when a GPU is present (one offset-concatenation launch, no host sort, no index upload): the result is a
    `LazySparse` that stands for the reference's coalesced CPU tensor, whose `.cuda()` carries the assembled operator to the
    residual blocks.  Any other member list takes the reference's host path."""
    if len(tensors) and all(isinstance(t, LazySparse) for t in tensors):
        op = batch_operator(tensors, size0, size1)
        if op is not None:
            return LazySparse.of_batch("diag", tensors, size0, size1, op)
    return reference_diag_cat([t.materialize() if isinstance(t, LazySparse) else t for t in tensors], size0, size1)


def sparse_cat(tensors, size0, size1):
    """3-D batched (len, size0, size1) COO operator, as utils_pt.py:21-39 (same resident path as sparse_diag_cat: the batch
    operator the blocks multiply with is the block-diagonal one either way)."""
    if len(tensors) and all(isinstance(t, LazySparse) for t in tensors):
        op = batch_operator(tensors, size0, size1)
        if op is not None:
            return LazySparse.of_batch("cat", tensors, size0, size1, op)
    return reference_cat([t.materialize() if isinstance(t, LazySparse) else t for t in tensors], size0, size1)


def to_dense_batched(x, batch_size):
    return x.to_dense().unsqueeze(0).repeat(batch_size, 1, 1)


def global_average(x, mask):
    """Masked mean over the node axis, kept as (B,1,C) (utils_pt.py:120-122).  mask = PackedSegments (a packed batch
    (1, sum V_i, C)): the per-mesh means, (meshes, 1, C)."""
    if isinstance(mask, PackedSegments):
        return mask.mean(x.reshape(-1, x.shape[-1])).unsqueeze(1)
    m = mask.expand_as(x)
    return (x * m).sum(1, keepdim=True) / m.sum(1, keepdim=True)


class SparseBMMFunc:
    """`SparseBMMFunc()(A, X)` of src/utils/cuda/sparse_bmm_func.py:23-72 for 3-D batched operands:
    A (B,R,K) sparse COO / SparseOperator, X (B,K,N) dense -> (B,R,N); grad only w.r.t. X."""

    def __call__(self, matrix1, matrix2):
        op = as_operator(matrix1)
        B, _, N = matrix2.shape
        y = snF.spmm(op, matrix2.reshape(-1, N), 1)
        return y.view(B, -1, N)


# --------------------------------------------------------------------------------------------------
# modules
# --------------------------------------------------------------------------------------------------
class GraphConv1x1(nn.Module):
    """Optional BatchNorm1d ("pre" | "post" | None) + Linear over the channel axis of (B, Nodes, C)
    (utils_pt.py:76-104).  BatchNorm statistics run over all B*Nodes rows, padded rows included, exactly
    as BatchNorm1d on the reference's transposed (B,C,Nodes) view does."""

    def __init__(self, num_inputs, num_outputs, batch_norm=None):
        super().__init__()
        self.num_inputs, self.num_outputs, self.batch_norm = num_inputs, num_outputs, batch_norm
        if batch_norm == "pre":
            self.bn = nn.BatchNorm1d(num_inputs)
        if batch_norm == "post":
            self.bn = nn.BatchNorm1d(num_outputs)
        self.fc = nn.Linear(num_inputs, num_outputs)

    def forward2d(self, x2d, residual=None):
        """(rows, Cin) -> (rows, Cout) on the flattened node axis; `residual` (rows, Cout) is added to the result
        (inside the GEMM epilogue on the fused path)."""
        if self.batch_norm == "pre":
            if x2d.dtype == torch.float32 and self.bn.affine and self.bn.momentum is not None:
                return snF.bn_linear(x2d, self.bn, self.fc, residual)   # one statistics pass + folded GEMM (functional.py)
            x2d = self.bn(x2d)
        if self.batch_norm is None and snF.thin_linear_supported(x2d, self.fc):
            # first layer (3 / 6 coordinates in): output-streaming forward, one-pass weight gradient; elu(y) for the block
            # that follows leaves the same kernel (the activated hand-off of blocks.py)
            want_elu = residual is None and self.num_outputs % 4 == 0 and 256 % (self.num_outputs // 4) == 0
            x2d, cat = snF.thin_linear(x2d, self.fc, want_elu)
            if cat is not None:
                snB.attach_activated(x2d, cat)
        else:
            x2d = self.fc(x2d)
        if self.batch_norm == "post":
            x2d = self.bn(x2d)
        return x2d if residual is None else x2d + residual

    def forward(self, x):
        batch_size, num_nodes, num_inputs = x.size()
        assert num_inputs == self.num_inputs
        y2d = self.forward2d(x.reshape(-1, num_inputs))
        cat = snB.take_activated(y2d, y2d.shape[0], self.num_outputs)
        y = y2d.view(batch_size, num_nodes, self.num_outputs)
        return y if cat is None else snB.attach_activated(y, cat)


def elu_conv1x1(conv: GraphConv1x1, x: torch.Tensor) -> torch.Tensor:
    """conv(F.elu(x)) — how every model of the reference enters its last GraphConv1x1 (as_rigid_as_possible/models.py:148-150,
    dense_correspondence/models.py:177-179) — as one fused node where the shapes allow, else literally that."""
    if USE_WHOLE_BLOCKS and snB.elu_conv_ok(conv, x):
        return snB.elu_conv(conv, x)
    return conv(F.elu(x))


class GraphBatchNorm(nn.Module):
    """BatchNorm over B*Nodes rows that always uses batch statistics (utils_pt.py:107-118)."""

    def __init__(self, num_inputs):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_inputs)

    def forward(self, x):
        self.bn.train()
        b, n, c = x.size()
        return self.bn(x.reshape(b * n, c)).view(b, n, c)


def _blocks_ok(mod, x) -> bool:
    """Whole-block path (blocks.py): fp32, default affine BatchNorm with a fixed momentum, channel count the vector kernels
    take.  Everything else goes through the per-stage functions.  (Sub-modules and flags are read from the instances' own
    dictionaries: nn.Module.__getattr__ is a Python-level search, a dozen of them cost more than the rest of this check.)"""
    c = x.shape[-1]
    if not USE_WHOLE_BLOCKS or x.dtype != torch.float32 or c % 4 or 256 % (c // 4):
        return False
    mods = mod.__dict__["_modules"]
    for name in ("bn_fc0", "bn_fc1"):
        bn = mods[name].__dict__["_modules"]["bn"].__dict__
        if not (bn["affine"] and bn["momentum"] is not None and bn["track_running_stats"]):
            return False
    return True


USE_WHOLE_BLOCKS = True     # set False to run the per-stage autograd functions (functional.py) instead — for A/B tests


class _TwoStage(nn.Module):
    """Shared constructor: two BN("pre")+Linear(2C -> C) stages named bn_fc0 / bn_fc1 (utils_pt.py:156-157,
    187-188,227-228) — the names are part of the checkpoint format."""

    def __init__(self, num_outputs):
        super().__init__()
        self.num_outputs = num_outputs
        self.bn_fc0 = GraphConv1x1(2 * num_outputs, num_outputs, batch_norm="pre")
        self.bn_fc1 = GraphConv1x1(2 * num_outputs, num_outputs, batch_norm="pre")


class DenseLapResNet2(_TwoStage):
    """Laplacian block with a dense (B,V,V) operator: plain batched GEMM (utils_pt.py:124-148)."""

    def forward(self, L, mask, inputs):
        x = F.elu(inputs)
        x = self.bn_fc0(torch.cat([x, torch.bmm(L, x)], 2))
        x = F.elu(x)
        x = self.bn_fc1(torch.cat([x, torch.bmm(L, x)], 2))
        return x + inputs


class LapResNet2(_TwoStage):
    """x + Lin(BN([e1, L e1])), e1 = elu(Lin(BN([e0, L e0]))), e0 = elu(x)   (utils_pt.py:151-180)."""

    def forward(self, L, mask, inputs, avg_next=None):
        """avg_next (not in the reference's signature; True / False / None = not said): the output feeds an AvgResNet2 next — the
        second GEMM then also leaves the per-tile column sums that block needs of its operand (no statistics pass over it).  The
        unmodified models.py say nothing: the block then LEARNS it (blocks._wants_tiles: the AvgResNet2 that receives its hand-off
        without tile sums marks it), so from the second step on the hand-off exists exactly where it is read."""
        if isinstance(L, torch.Tensor) and L.layout == torch.strided:
            return DenseLapResNet2.forward(self, L, mask, inputs)
        batch, node, feat = inputs.size()
        op = as_operator(L)
        if _blocks_ok(self, inputs):
            return snB.lap_block(self, op, inputs, avg_next)                    # the whole block as one autograd node
        x2d = inputs.reshape(batch * node, feat)
        h = self.bn_fc0.forward2d(snF.lap_propagate(op, x2d))
        h = self.bn_fc1.forward2d(snF.lap_propagate(op, h), residual=x2d)        # "+ inputs" rides in the GEMM epilogue
        return h.view(batch, node, feat)


class DirResNet2(_TwoStage):
    """Dirac block (utils_pt.py:182-220): vertices -> faces through Di, faces -> vertices through DiA, on the
    quaternion view (rows*4, C/4); returns (v + v_out, f_out)."""

    def __init__(self, num_outputs, res_f=False):
        super().__init__(num_outputs)
        self.res_f = res_f          # accepted and unused, as in the reference (utils_pt.py:189)

    def forward(self, Di, DiA, v, f, f_out_needed=True, num_faces=None, avg_next=None):
        """f_out_needed=False (not in the reference's signature): the caller promises to use the returned face features ONLY
        as the `f` argument of the next DirResNet2 — the block then skips writing them (the next block reads the activated
        copy handed over internally) and returns a NaN placeholder of the right shape in their place.
        f=None with num_faces=F (not in the reference's signature either): the face features are all zero — what every
        model of the reference feeds its first Dirac block (as_rigid_as_possible/models.py:138) — and are not materialised:
        the face stage runs over the propagated half only (same values).
        avg_next (not in the reference's signature; True / False / None = not said): the vertex output feeds an AvgResNet2 next:
        see LapResNet2.forward (None: learnt from the block that consumes the hand-off)."""
        batch_size, num_nodes, num_inputs = v.size()
        if f is None:
            if num_faces is None:
                raise ValueError("DirResNet2: f=None needs num_faces")
            if _blocks_ok(self, v) and snB.zero_faces_ok(self, num_inputs):
                return snB.dirac_block(self, Di, DiA, v, None, f_out_needed, num_faces, avg_next)
            f = torch.zeros(batch_size, int(num_faces), num_inputs, dtype=v.dtype, device=v.device)
        _, num_faces, _ = f.size()
        if _blocks_ok(self, v):
            return snB.dirac_block(self, Di, DiA, v, f, f_out_needed, None, avg_next)   # the whole block as one autograd node
        v2d = v.reshape(batch_size * num_nodes, num_inputs)
        cat0, e_v = snF.dirac_face_stage(as_operator(Di), v2d, f.reshape(batch_size * num_faces, num_inputs))
        f_out = self.bn_fc0.forward2d(cat0)
        cat1 = snF.dirac_vert_stage(as_operator(DiA), f_out, e_v)
        v_new = self.bn_fc1.forward2d(cat1, residual=v2d)                        # v + v_out in the GEMM epilogue
        return v_new.view(batch_size, num_nodes, num_inputs), f_out.view(batch_size, num_faces, num_inputs)


class AvgResNet2(_TwoStage):
    """Global-average block (utils_pt.py:222-243): no sparse operator, plain PyTorch."""

    def forward(self, L, mask, inputs):
        b, n, c = inputs.size()
        if isinstance(mask, PackedSegments):
            # packed batch (1, sum V_i, C): per-mesh means over ragged row ranges, no padding rows anywhere; BatchNorm then
            # sees the real rows only (the reference's padded batch includes the padding rows, utils_pt.py:97-99)
            if self.training and snB.avg_block_ragged_ok(self, mask, inputs):
                return snB.avg_block_ragged(self, mask, inputs)                 # half width, one autograd node
            x2d = inputs.reshape(b * n, c)
            h = self.bn_fc0.forward2d(snF.avg_propagate_ragged(x2d, mask))
            h = self.bn_fc1.forward2d(snF.avg_propagate_ragged(h, mask), residual=x2d)
            return h.view(b, n, c)
        if c % 4 or 256 % (c // 4) or inputs.dtype != torch.float32:        # shapes the fused kernels do not cover
            x = F.elu(inputs)
            x = self.bn_fc0(torch.cat([x, global_average(x, mask).expand_as(x)], 2))
            x = F.elu(x)
            x = self.bn_fc1(torch.cat([x, global_average(x, mask).expand_as(x)], 2))
            return x + inputs
        if _blocks_ok(self, inputs):
            return snB.avg_block(self, mask, inputs)                            # the whole block as one autograd node
        h = self.bn_fc0.forward2d(snF.avg_propagate(inputs, mask))
        h = self.bn_fc1.forward2d(snF.avg_propagate(h.view(b, n, c), mask), residual=inputs.reshape(b * n, c))
        return h.view(b, n, c)


class MlpResNet2(nn.Module):
    """Per-node MLP block (utils_pt.py:245-263); checkpoint keys bn0.bn.*, fc0.fc.*, bn1.bn.*, fc1.fc.*."""

    def __init__(self, num_outputs):
        super().__init__()
        self.num_outputs = num_outputs
        self.bn0 = GraphBatchNorm(num_outputs)
        self.fc0 = GraphConv1x1(num_outputs, num_outputs, batch_norm=None)
        self.bn1 = GraphBatchNorm(num_outputs)
        self.fc1 = GraphConv1x1(num_outputs, num_outputs, batch_norm=None)

    def forward(self, L, mask, inputs):
        x = self.fc0(F.elu(self.bn0(inputs)))
        x = self.fc1(F.elu(self.bn1(x)))
        return x + inputs
