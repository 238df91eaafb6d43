"""TEST INFRASTRUCTURE ONLY — PyTorch restatement of the reference operator layer and task models on the
reference's own CPU path: `torch.mm(sparse_coo, dense)` with autograd from ATen
(src/utils/utils_pt.py:167,176,202,214).  This is "the repo's own CPU torch.sparse path" that BASELINE.md §3
names as the CPU baseline, and the oracle the HIP blocks/models are compared with.

Pinned against the imported reference in the build container (tests/golden/make_golden.py asserts equality
on every fixture before writing it) and against the committed fixtures in tests/test_oracle_golden.py.

Restated faithfully, including the parts that look odd:
  * BatchNorm1d runs on the transposed (B, C, Nodes) view, so statistics include zero-padded nodes
    (utils_pt.py:97-99); GraphBatchNorm forces train mode (utils_pt.py:113).
  * DirResNet2 returns (v + v_out, f_out): residual on vertices only (utils_pt.py:220).
  * The Dirac product acts on the plain view (B*Nodes*4, C/4) (utils_pt.py:201,213).
  * ARAP models add the last input frame repeated 40x (src/as_rigid_as_possible/models.py:52,152).
Deviations (the reference is broken there as shipped): 3-D batched sparse operators are
handled by block-diagonalising them (the reference calls an un-imported SparseBMMFunc, utils_pt.py:199,211),
and the Mesh-MNIST / FAUST Dirac models use the ARAP `num_faces` rule (as_rigid_as_possible/models.py:133-136)
instead of `DiA.size(2)` which raises on 2-D operators.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- operator handling ------------------------------------------------------------------------------
def block_diagonalise(A: torch.Tensor) -> torch.Tensor:
    """(B,R,K) sparse COO -> (B*R, B*K) sparse COO; 2-D operators pass through.  Mathematically what
    SparseBMMFunc's per-batch product is (src/utils/cuda/sparse_bmm.cu:16-61)."""
    if A.dim() == 2:
        return A
    A = A.coalesce()
    b, r, c = A.indices()
    B, R, K = A.shape
    return torch.sparse_coo_tensor(torch.stack([b * R + r, b * K + c]), A.values(), (B * R, B * K)).coalesce()


def sparse_mm(A: torch.Tensor, x2d: torch.Tensor) -> torch.Tensor:
    return torch.mm(block_diagonalise(A), x2d)


def sp_to_coo(M) -> torch.Tensor:
    """scipy -> torch COO, the conversion of utils_pt.py:56-69."""
    import numpy as np

    M = M.tocoo()
    idx = torch.from_numpy(np.stack([M.row, M.col]).astype("int64"))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(M.data), M.shape)


def diag_cat(tensors, size0, size1) -> torch.Tensor:
    """sparse_diag_cat of utils_pt.py:41-53."""
    idx, val = [], []
    for i, t in enumerate(tensors):
        off = torch.tensor([[i * size0], [i * size1]], dtype=torch.int64)
        idx.append(t._indices() + off)
        val.append(t._values())
    n = len(tensors)
    return torch.sparse_coo_tensor(torch.cat(idx, 1), torch.cat(val), (n * size0, n * size1)).coalesce()


def batch_cat(tensors, size0, size1) -> torch.Tensor:
    """sparse_cat of utils_pt.py:21-39 (3-D batched COO)."""
    idx, val = [], []
    for i, t in enumerate(tensors):
        ti = t._indices()
        idx.append(torch.cat([torch.full((1, ti.shape[1]), i, dtype=torch.int64), ti], 0))
        val.append(t._values())
    return torch.sparse_coo_tensor(torch.cat(idx, 1), torch.cat(val), (len(tensors), size0, size1)).coalesce()


def masked_mean(x, mask):
    m = mask.expand_as(x)                                   # utils_pt.py:120-122
    return (x * m).sum(1, keepdim=True) / m.sum(1, keepdim=True)


# ---- layers -----------------------------------------------------------------------------------------
class GraphConv1x1(nn.Module):
    def __init__(self, num_inputs, num_outputs, batch_norm=None):
        super().__init__()
        self.num_inputs, self.num_outputs, self.batch_norm = num_inputs, num_outputs, batch_norm
        if batch_norm in ("pre", "post"):
            self.bn = nn.BatchNorm1d(num_inputs if batch_norm == "pre" else num_outputs)
        self.fc = nn.Linear(num_inputs, num_outputs)

    def _bn(self, x):
        return self.bn(x.transpose(1, 2)).transpose(1, 2)  # utils_pt.py:98,101

    def forward(self, x):
        assert x.size(2) == self.num_inputs
        if self.batch_norm == "pre":
            x = self._bn(x)
        x = self.fc(x)
        if self.batch_norm == "post":
            x = self._bn(x)
        return x


class GraphBatchNorm(nn.Module):
    def __init__(self, num_inputs):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_inputs)

    def forward(self, x):
        self.bn.train()                                      # utils_pt.py:113
        b, n, c = x.size()
        return self.bn(x.view(b * n, c)).view(b, n, c)


class _Res2(nn.Module):
    def __init__(self, num_outputs):
        super().__init__()
        self.num_outputs = num_outputs
        self.bn_fc0 = GraphConv1x1(2 * num_outputs, num_outputs, batch_norm="pre")
        self.bn_fc1 = GraphConv1x1(2 * num_outputs, num_outputs, batch_norm="pre")


class LapResNet2(_Res2):
    @staticmethod
    def _lap(L, x):
        b, n, c = x.size()
        if L.layout is torch.strided:                       # dense (B,V,V): utils_pt.py:164-165
            return torch.bmm(L, x)
        return sparse_mm(L, x.reshape(-1, c)).view(b, n, c)  # utils_pt.py:167

    def forward(self, L, mask, inputs):
        x = F.elu(inputs)
        x = self.bn_fc0(torch.cat([x, self._lap(L, x)], 2))
        x = F.elu(x)
        x = self.bn_fc1(torch.cat([x, self._lap(L, x)], 2))
        return x + inputs


DenseLapResNet2 = LapResNet2


class DirResNet2(_Res2):
    def __init__(self, num_outputs, res_f=False):
        super().__init__(num_outputs)
        self.res_f = res_f

    def forward(self, Di, DiA, v, f):
        b, nv, c = v.size()
        nf = f.size(1)
        x_in, f_in = F.elu(v), F.elu(f)
        y = sparse_mm(Di, x_in.reshape(b * nv * 4, c // 4)).view(b, nf, c)       # utils_pt.py:201-203
        f_out = self.bn_fc0(torch.cat([f_in, y], 2))
        z = sparse_mm(DiA, F.elu(f_out).reshape(b * nf * 4, c // 4)).view(b, nv, c)  # utils_pt.py:213-215
        v_out = self.bn_fc1(torch.cat([x_in, z], 2))
        return v + v_out, f_out


class AvgResNet2(_Res2):
    def forward(self, L, mask, inputs):
        x = F.elu(inputs)
        x = self.bn_fc0(torch.cat([x, masked_mean(x, mask).expand_as(x).contiguous()], 2))
        x = F.elu(x)
        x = self.bn_fc1(torch.cat([x, masked_mean(x, mask).expand_as(x).contiguous()], 2))
        return x + inputs


class MlpResNet2(nn.Module):
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


# ---- task models ------------------------------------------------------------------------------------
def _num_faces(Di, DiA, batch_size):
    """src/as_rigid_as_possible/models.py:133-136."""
    return DiA.size(2) // 4 if len(Di.size()) == 3 else DiA.size(1) // 4 // batch_size


class ArapLapModel(nn.Module):
    """src/as_rigid_as_possible/models.py:21-52 (`Model`) — also the FAUST tower with 3 input channels
    (src/dense_correspondence/models.py:21-49)."""

    def __init__(self, layer=15, in_channels=6):
        super().__init__()
        self.conv1 = GraphConv1x1(in_channels, 128, batch_norm=None)
        self.layer = layer
        for i in range(layer):
            self.add_module(f"rn{i}", LapResNet2(128) if i % 2 == 0 else AvgResNet2(128))
        self.conv2 = GraphConv1x1(128, 120, batch_norm="pre")

    def forward(self, L, mask, inputs):
        x = self.conv1(inputs)
        for i in range(self.layer):
            x = self._modules[f"rn{i}"](L, mask, x)
        x = self.conv2(F.elu(x))
        return x + inputs[:, :, -3:].repeat(1, 1, 40)


class ArapDirModel(nn.Module):
    """src/as_rigid_as_possible/models.py:108-152 (`DirModel`)."""

    def __init__(self, layer=15, in_channels=6):
        super().__init__()
        self.conv1 = GraphConv1x1(in_channels, 128, batch_norm=None)
        self.layer = layer
        for i in range(layer):
            self.add_module(f"rn{i}", DirResNet2(128) if i % 2 == 0 else AvgResNet2(128))
        self.do = nn.Dropout2d()                             # declared, never called (models.py:123)
        self.conv2 = GraphConv1x1(128, 120, batch_norm="pre")

    def forward(self, Di, DiA, mask, inputs):
        b = inputs.size(0)
        v = self.conv1(inputs)
        f = torch.zeros(b, _num_faces(Di, DiA, b), 128, dtype=v.dtype, device=v.device)
        for i in range(self.layer):
            blk = self._modules[f"rn{i}"]
            if i % 2 == 0:
                v, f = blk(Di, DiA, v, f)
            else:
                v = blk(None, mask, v)
        x = self.conv2(F.elu(v))
        return x + inputs[:, :, -3:].repeat(1, 1, 40)


class MnistLapModel(nn.Module):
    """src/mesh_mnist/models.py:22-53 (`Model`); dropout p given so tests can switch it off."""

    def __init__(self, p_drop=0.5):
        super().__init__()
        self.conv1 = GraphConv1x1(3, 64, batch_norm=None)
        for i in range(5):
            self.add_module(f"rn{i}", LapResNet2(64))
        self.bn_conv2 = GraphConv1x1(64, 64, batch_norm="pre")
        self.fc1 = nn.Linear(64, 10)
        self.p_drop = p_drop

    def _head(self, x, mask):
        x = F.elu(self.bn_conv2(F.elu(x)))
        x = masked_mean(x, mask).squeeze(1)                  # reference .squeeze() (models.py:49); B>1 assumed there
        x = F.dropout(x, p=self.p_drop, training=self.training)
        return F.log_softmax(self.fc1(x), dim=1)

    def forward(self, inputs, L, mask):
        x = self.conv1(inputs)
        for i in range(5):
            x = self._modules[f"rn{i}"](L, mask, x)
        return self._head(x, mask)


class MnistDirModel(MnistLapModel):
    """src/mesh_mnist/models.py:122-159 (`DirModel`)."""

    def __init__(self, p_drop=0.5):
        nn.Module.__init__(self)
        self.conv1 = GraphConv1x1(3, 64, batch_norm=None)
        for i in range(5):
            self.add_module(f"rn{i}", DirResNet2(64))
        self.bn_conv2 = GraphConv1x1(64, 64, batch_norm="pre")
        self.fc1 = nn.Linear(64, 10)
        self.p_drop = p_drop

    def forward(self, inputs, Di, DiA, mask):
        b = inputs.size(0)
        v = self.conv1(inputs)
        f = torch.zeros(b, _num_faces(Di, DiA, b), 64, dtype=v.dtype, device=v.device)
        for i in range(5):
            v, f = self._modules[f"rn{i}"](Di, DiA, v, f)
        return self._head(v, mask)


class SiameseModel(nn.Module):
    """src/dense_correspondence/models.py:184-203: one tower applied to both shapes, bmm(FA, FB^T)."""

    def __init__(self, model="lap", layer=15):
        super().__init__()
        self.model = ArapDirModel(layer, 3) if "dir" in model else ArapLapModel(layer, 3)

    def forward(self, OperationA, OperationB, inputA, inputB):
        FA = self.model(*OperationA, inputA)
        FB = self.model(*OperationB, inputB)
        return torch.bmm(FA, FB.transpose(1, 2))


# ---- losses (the unit the `metric` times) -----------------------------------------------------------------
def arap_loss(outputs, targets, mask, batch_size):
    """src/as_rigid_as_possible/main.py:225-226."""
    outputs = outputs * mask.expand_as(outputs)
    return F.smooth_l1_loss(outputs, targets, reduction="sum") / batch_size


def delta_cross_entropy(outputs, targetX, targetY):
    """loss_fun_delta_cross_entropy, src/dense_correspondence/main.py:229-240 (note: always reads outputs[0])."""
    loss = outputs.new_zeros(1)
    for i in range(outputs.size(0)):
        GA, lA, liA = targetX[i]
        GB, lB, liB = targetY[i]
        NA, NB = lA.size(0), lB.size(0)
        _, GAB = torch.min(GA[:, liA[lB]] + GB[liB[lA], :], dim=1)
        loss = loss + F.cross_entropy(outputs[0, :NA, :NB], GAB)
    return loss / outputs.size(0)
