"""Mesh-MNIST classification — counterpart of src/mesh_mnist/models.py and main.py:79-167.

`Model` (Laplacian) and `DirModel` (Dirac): conv1 3->64, five residual blocks at 64 channels, ELU, BN+Linear 64->64,
ELU, masked global average, dropout, fc1 64->10, log_softmax; NLL loss; Adam(1e-3, wd 1e-5) (main.py:139,159).
`state_dict` keys match the reference (conv1.fc.*, rn{i}.*, bn_conv2.*, fc1.*).  The reference's DirModel reads
`DiA.size(2)` and so only works with 3-D batched operators (models.py:142); here the ARAP rule
(as_rigid_as_possible/models.py:133-136) makes both operator layouts work.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels, mesh_ops
from . import utils_pt as utils
from .arap import make_adam
from .operators import OperatorPool


class _Head(nn.Module):
    def _classify(self, x, mask):
        x = F.elu(self.bn_conv2(F.elu(x)))
        x = utils.global_average(x, mask).squeeze(1)
        x = F.dropout(x, training=self.training)
        return F.log_softmax(self.fc1(x), dim=1)


class Model(_Head):
    def __init__(self):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(3, 64, batch_norm=None)
        for i in range(5):
            self.add_module("rn{}".format(i), utils.LapResNet2(64))
        self.bn_conv2 = utils.GraphConv1x1(64, 64, batch_norm="pre")
        self.fc1 = nn.Linear(64, 10)

    def forward(self, inputs, L, mask):
        x = self.conv1(inputs)
        for i in range(5):
            x = self._modules["rn{}".format(i)](L, mask, x)
        return self._classify(x, mask)


class DirModel(_Head):
    def __init__(self):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(3, 64, batch_norm=None)
        for i in range(5):
            self.add_module("rn{}".format(i), utils.DirResNet2(64))
        self.bn_conv2 = utils.GraphConv1x1(64, 64, batch_norm="pre")
        self.fc1 = nn.Linear(64, 10)

    def forward(self, inputs, Di, DiA, mask):
        batch_size = inputs.size(0)
        v = self.conv1(inputs)
        num_faces = DiA.size(2) // 4 if len(Di.size()) == 3 else DiA.size(1) // 4 // batch_size
        f = None                                # zeros(batch, faces, 64) (models.py:144), materialised only where needed
        for i in range(5):
            v, f = self._modules["rn{}".format(i)](Di, DiA, v, f, num_faces=num_faces)
        return self._classify(v, mask)


class _BlockStack(_Head):
    """Shared body of the block-type variants (models.py:55-119): conv1, five blocks of one type, the classifier head."""

    block = None

    def __init__(self):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(3, 64, batch_norm=None)
        for i in range(5):
            self.add_module("rn{}".format(i), self.block(64))
        self.bn_conv2 = utils.GraphConv1x1(64, 64, batch_norm="pre")
        self.fc1 = nn.Linear(64, 10)

    def forward(self, inputs, L, mask):
        x = self.conv1(inputs)
        for i in range(5):
            x = self._modules["rn{}".format(i)](L, mask, x)
        return self._classify(x, mask)


class AvgModel(_BlockStack):
    """Global-average blocks only (src/mesh_mnist/models.py:55-87)."""
    block = utils.AvgResNet2


class MlpModel(_BlockStack):
    """Per-node MLP blocks only (src/mesh_mnist/models.py:89-119)."""
    block = utils.MlpResNet2


def make_optimizer(model):
    return make_adam(model)       # main.py:139


def halve_lr(optimizer, epoch: int) -> bool:
    """main.py:174-176: halve the learning rate at the end of every tenth epoch past the twentieth."""
    from .arap import halve_lr as _halve

    return _halve(optimizer, epoch, after=20, every=10)


@dataclass
class Batch:
    inputs: torch.Tensor      # (B, Vmax, 3)
    targets: torch.Tensor     # (B,)
    mask: torch.Tensor        # (B, Vmax, 1)
    L: Optional[object]
    Di: Optional[object]
    DiA: Optional[object]


class MeshDigits:
    """Synthetic stand-in for train_plus.np (mesh_mnist/add_laplacian.py:63-71): seeded Delaunay meshes of ~150
    vertices with the reference scaling, a label in 0..9, operators resident in HBM pools."""

    def __init__(self, count, seed=2, device="cuda", vmin=140, vmax=230, fixed_vertices=None, model="dir", reorder="auto"):
        """reorder: the meshes are STORED in a locality numbering (mesh_ops.MeshOrder; Delaunay vertices come in random order).
        The model's output is per MESH (ten class scores), so nothing maps back; `orders[i].vorder` (stored position -> generated
        vertex index) is kept for callers that look at per-vertex activations."""
        rng = np.random.default_rng(seed)
        self.device = torch.device(device)
        self.kind = model
        Vs, mats = [], {"L": [], "Di": [], "DiA": []}
        self.nv, self.nf = [], []
        for _ in range(count):
            n = fixed_vertices or int(rng.integers(vmin, vmax + 1))
            V, F_ = mesh_ops.delaunay_disc(n, rng)
            order = mesh_ops.MeshOrder.of_mesh(F_, V.shape[0], reorder)
            self.orders = getattr(self, "orders", []) + [order]
            V, F_ = order.mesh(V, F_)
            Vs.append(V.astype(np.float32))
            self.nv.append(V.shape[0])
            self.nf.append(F_.shape[0])
            if model == "dir":
                Di, DiA = mesh_ops.dirac(V, F_)
                mats["Di"].append(Di.astype(np.float32))
                mats["DiA"].append(DiA.astype(np.float32))
            else:
                mats["L"].append(mesh_ops.laplacian(V, F_).astype(np.float32))
        self.nv, self.nf = np.array(self.nv), np.array(self.nf)
        self.n = count
        xyz = np.zeros((count, int(self.nv.max()), 3), np.float32)
        for i, V in enumerate(Vs):
            xyz[i, : V.shape[0]] = V
        self.xyz = torch.from_numpy(xyz).to(self.device)
        self.vcount = torch.from_numpy(self.nv).to(self.device)
        self.labels = torch.from_numpy(rng.integers(0, 10, size=count)).to(self.device)
        if model == "dir":
            self.pool_Di = OperatorPool(mats["Di"], self.device, want_bsr4=True)
            self.pool_DiA = OperatorPool(mats["DiA"], self.device, want_bsr4=True)
        else:
            self.pool_L = OperatorPool(mats["L"], self.device)
        # the reference keeps a running maximum of the padded sizes across steps (main.py:83-84,119-120)
        self.run_nv = self.run_nf = 0

    def sample_batch(self, batch_size, rng, ids=None) -> Batch:
        ids = rng.integers(0, self.n, size=batch_size) if ids is None else np.asarray(ids)
        self.run_nv = max(self.run_nv, int(self.nv[ids].max()))
        self.run_nf = max(self.run_nf, int(self.nf[ids].max()))
        nv, nf = self.run_nv, self.run_nf
        sid = torch.from_numpy(ids).to(self.device)
        inputs = self.xyz[sid][:, :nv].contiguous()
        if inputs.shape[1] < nv:
            inputs = F.pad(inputs, (0, 0, 0, nv - inputs.shape[1]))
        mask = (torch.arange(nv, device=self.device)[None, :] < self.vcount[sid][:, None]).float().unsqueeze(2)
        L = Di = DiA = None
        if self.kind == "dir":
            Di = self.pool_Di.assemble(ids, 4 * nf, 4 * nv)
            DiA = self.pool_DiA.assemble(ids, 4 * nv, 4 * nf)
        else:
            L = self.pool_L.assemble(ids, nv, nv)
        return Batch(inputs, self.labels[sid], mask, L, Di, DiA)


def forward_loss(model, batch: Batch):
    out = model(batch.inputs, batch.Di, batch.DiA, batch.mask) if batch.Di is not None else model(batch.inputs, batch.L, batch.mask)
    return F.nll_loss(out, batch.targets), out                                  # main.py:159


def graphed_train_step(model, optimizer, example: Batch, bucket=None):
    """Training step with forward + loss + backward replayed from one hipGraph (graphs.GraphedTrainStep) — these models
    are launch-bound when run eagerly (≈500 launches of a few microseconds per step).  Needs batches of one signature
    (e.g. MeshDigits(fixed_vertices=...)); check `.matches(batch)`."""
    from .graphs import GraphedTrainStep

    return GraphedTrainStep(model, optimizer, example, lambda m, b: forward_loss(m, b)[0], bucket)


def train_step(model, optimizer, batch: Batch, grad_sync=None):
    loss, _ = forward_loss(model, batch)
    optimizer.zero_grad(set_to_none=False)
    loss.backward()
    kernels.clear_absmax()
    if grad_sync is not None:
        grad_sync()
    optimizer.step()
    return loss
