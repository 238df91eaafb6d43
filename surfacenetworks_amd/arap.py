"""as_rigid_as_possible temporal prediction — the build's counterpart of the reference harness
(src/as_rigid_as_possible/models.py, main.py): 2 input frames -> 40 predicted frames, 15 residual blocks at
128 channels (Dirac/Laplacian blocks on even layers, global-average blocks on odd ones), masked smooth-L1 loss.

Model classes keep the reference's names, layer order, widths and `state_dict` keys (conv1.fc.*, rn{i}.bn_fc{0,1}.*,
conv2.*), so reference checkpoints load.  What differs is the data path: every sequence's
operators are converted once into device-resident CSR / CSR^T / BSR4 pools (operators.OperatorPool) and a batch is
assembled on the GPU per step, instead of per-sample scipy->COO conversion, sparse_diag_cat + coalesce on the host
and an H2D copy of the operators every step (main.py:98-185).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels, mesh_ops
from . import utils_pt as utils
from .operators import OperatorPool, PackedSegments, h2d_async

INPUT_FRAMES = 2       # main.py:105
OUTPUT_FRAMES = 40     # main.py:106


def _num_faces(Di, DiA, batch_size):
    """models.py:133-136 — works for 3-D batched and 2-D block-diagonal operators."""
    return DiA.size(2) // 4 if len(Di.size()) == 3 else DiA.size(1) // 4 // batch_size



def _add_last_frame(x, inputs, frames):
    """x + inputs[:, :, -3:].repeat(1, 1, frames) (models.py:105,152) as one broadcast add — same values, the repeated
    tensor is never written."""
    B, V, _ = x.shape
    return (x.reshape(B, V, frames, 3) + inputs[:, :, None, -3:]).reshape(B, V, 3 * frames)


class Model(nn.Module):
    """Laplacian variant (models.py:21-52)."""

    def __init__(self, layer=15, dense=False):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(6, 128, batch_norm=None)
        self.layer = layer
        for i in range(layer):
            if i % 2 == 0:
                blk = utils.DenseLapResNet2(128) if dense else utils.LapResNet2(128)
            else:
                blk = utils.AvgResNet2(128)
            self.add_module("rn{}".format(i), blk)
        self.conv2 = utils.GraphConv1x1(128, 120, batch_norm="pre")

    def forward(self, L, mask, inputs):
        x = self.conv1(inputs)
        for i in range(self.layer):
            blk = self._modules["rn{}".format(i)]
            if i % 2 == 0 and isinstance(blk, utils.LapResNet2):
                x = blk(L, mask, x, avg_next=i + 1 < self.layer)      # (a global-average block follows: hand it the tile sums)
            else:
                x = blk(L, mask, x)
        x = utils.elu_conv1x1(self.conv2, x)
        return _add_last_frame(x, inputs, OUTPUT_FRAMES)


class AvgModel(nn.Module):
    """Global-average blocks only (models.py:54-78)."""

    def __init__(self):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(6, 128, batch_norm=None)
        for i in range(15):
            self.add_module("rn{}".format(i), utils.AvgResNet2(128))
        self.conv2 = utils.GraphConv1x1(128, 120, batch_norm="pre")

    def forward(self, L, mask, inputs):
        x = self.conv1(inputs)
        for i in range(15):
            x = self._modules["rn{}".format(i)](L, mask, x)
        x = utils.elu_conv1x1(self.conv2, x)
        return _add_last_frame(x, inputs, OUTPUT_FRAMES)


class MlpModel(nn.Module):
    """Per-node MLP blocks only (models.py:81-105): conv1, 15 x MlpResNet2, GraphBatchNorm, ELU, conv2 without BatchNorm."""

    def __init__(self):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(6, 128, batch_norm=None)
        for i in range(15):
            self.add_module("rn{}".format(i), utils.MlpResNet2(128))
        self.bn = utils.GraphBatchNorm(128)
        self.conv2 = utils.GraphConv1x1(128, 120, batch_norm=None)

    def forward(self, L, mask, inputs):
        x = self.conv1(inputs)
        for i in range(15):
            x = self._modules["rn{}".format(i)](L, mask, x)
        x = self.conv2(F.elu(self.bn(x)))
        return _add_last_frame(x, inputs, OUTPUT_FRAMES)


class DirModel(nn.Module):
    """Dirac variant (models.py:108-152): the `metric`'s "Dirac temporal-predict" model, 1 018 872 parameters."""

    def __init__(self):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(6, 128, batch_norm=None)
        for i in range(15):
            self.add_module("rn{}".format(i), utils.DirResNet2(128) if i % 2 == 0 else utils.AvgResNet2(128))
        self.do = nn.Dropout2d()          # declared and never applied, as in the reference (models.py:123)
        self.conv2 = utils.GraphConv1x1(128, 120, batch_norm="pre")

    def forward(self, Di, DiA, mask, inputs):
        batch_size = inputs.size(0)
        v = self.conv1(inputs)
        f = None                                                   # f = zeros(batch, faces, 128) (models.py:138), not materialised
        nf = _num_faces(Di, DiA, batch_size)
        for i in range(15):
            blk = self._modules["rn{}".format(i)]
            if i % 2 == 0:
                v, f = blk(Di, DiA, v, f, f_out_needed=False, num_faces=nf,   # f only feeds the next Dirac block (models.py:139-147)
                           avg_next=i + 1 < 15)                                # (a global-average block follows: hand it the tile sums)
            else:
                v = blk(None, mask, v)
        x = utils.elu_conv1x1(self.conv2, v)
        return _add_last_frame(x, inputs, OUTPUT_FRAMES)


class _MaskedSmoothL1(torch.autograd.Function):
    """sum smooth_l1(outputs * mask, targets) / batch_size as one pass forward (fp64 sums) and one pass backward
    (kernels.masked_smooth_l1_*) instead of five elementwise passes over the (B, V, 120) tensors."""

    @staticmethod
    def forward(ctx, out2d, tgt2d, rowmask, scale):
        ctx.save_for_backward(out2d, tgt2d, rowmask)
        ctx.scale = scale
        return kernels.masked_smooth_l1_fwd(out2d, tgt2d, rowmask, scale)

    @staticmethod
    def backward(ctx, gloss):
        out2d, tgt2d, rowmask = ctx.saved_tensors
        return kernels.masked_smooth_l1_bwd(out2d, tgt2d, rowmask, ctx.scale, gloss.contiguous()), None, None, None


def loss_fn(outputs, targets, mask, batch_size):
    """Masked smooth-L1, summed, divided by the batch size (main.py:225-226).  Under data parallelism pass the
    GLOBAL batch size so that the all-reduced (summed) gradients equal the single-process ones."""
    if isinstance(mask, PackedSegments):               # packed batch: every row is a real vertex
        mask = torch.ones(outputs.shape[0] * outputs.shape[1], dtype=torch.float32, device=outputs.device)
    if outputs.dim() == 3 and outputs.dtype == torch.float32 and targets.dtype == torch.float32 and \
            not targets.requires_grad and mask.numel() == outputs.shape[0] * outputs.shape[1]:
        C = outputs.shape[2]
        out = _MaskedSmoothL1.apply(outputs.reshape(-1, C), targets.reshape(-1, C).contiguous(),
                                    mask.reshape(-1).to(torch.float32).contiguous(), 1.0 / batch_size)
        return out
    outputs = outputs * mask.expand_as(outputs)
    return F.smooth_l1_loss(outputs, targets, reduction="sum") / batch_size


def make_adam(model):
    """torch.optim.Adam(lr 1e-3, weight_decay 1e-5) as all three drivers construct it (main.py:207; mesh_mnist/main.py:139;
    dense_correspondence/main.py:285).  On the GPU torch's own fused implementation (same update, 2 launches per step
    instead of 17 multi-tensor ones)."""
    params = list(model.parameters())
    fused = len(params) > 0 and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    return torch.optim.Adam(params, 1e-3, weight_decay=1e-5, fused=fused)


def make_optimizer(model):
    """Adam(lr 1e-3, weight_decay 1e-5), main.py:207."""
    return make_adam(model)


def halve_lr(optimizer, epoch: int, after: int = 50, every: int = 10) -> bool:
    """The drivers' epoch-level schedule: the learning rate of every parameter group is halved at the end of each epoch with
    `epoch > after and epoch % every == 0` — after = 50 for ARAP (main.py:237-239), 20 for Mesh-MNIST
    (mesh_mnist/main.py:174-176).  Returns whether it did."""
    if not (epoch > after and epoch % every == 0):
        return False
    for group in optimizer.param_groups:
        group["lr"] *= 0.5
    return True


# --------------------------------------------------------------------------------------------------
# data: synthetic cloth sequences, resident on the GPU
# --------------------------------------------------------------------------------------------------
@dataclass
class Batch:
    inputs: torch.Tensor      # (B, Vmax, 6)
    targets: torch.Tensor     # (B, Vmax, 120)
    mask: torch.Tensor        # (B, Vmax, 1)
    L: Optional[object]
    Di: Optional[object]
    DiA: Optional[object]
    num_meshes: int


class ClothSequences:
    """Synthetic stand-in for the reference's data_plus/*.npy sequences (as_rigid_as_possible/add_laplacian.py:39-70):
    each sequence is one grid-cloth mesh whose vertices follow a smooth travelling deformation over `frames` frames;
    operators are precomputed for the first `op_frames` frames (the reference does it for frames < 10).

    Everything the training loop touches lives in HBM: frame coordinates as one (S, T, Vmax, 3) tensor and the
    operators in OperatorPools (entry s*op_frames + t is sequence s at frame t)."""

    def __init__(self, grids, frames=50, op_frames=2, seed=3, device="cuda", model="dir", permute=False,
                 operators="pool", reorder="auto"):
        """permute: False (row-major grid numbering), True / "vertices" (seeded random vertex numbering) or
        "both" (faces shuffled too: what a scanned mesh looks like).
        reorder: "auto" (default) / True / False — the dataset is STORED in a locality numbering of its vertices and faces
        (mesh_ops.MeshOrder: reverse Cuthill-McKee; "auto" keeps a numbering that is already local, e.g. a generated grid), so
        that every operator is banded and the products' gathers stay in cache whatever order the meshes arrive in.  The model
        is equivariant to the numbering and the loss invariant, so training is unchanged; `to_dataset_order` maps per-vertex
        outputs back for callers that index vertices by the dataset's own numbers.
        operators="pool": per-frame operators precomputed (host, fp64 coordinates) and pooled in HBM, as the
        reference's dataset does.  operators="device": nothing is precomputed — the Dirac operators of the sampled
        frames are built on the GPU every step from the stored fp32 coordinates (sn_dirac_bsr4_from_mesh); needs
        meshes of one size (no padding) and model="dir"."""
        rng = np.random.default_rng(seed)
        self.device = torch.device(device)
        self.frames, self.op_frames, self.kind = frames, op_frames, model
        self.operators = operators
        if operators not in ("pool", "device"):
            raise ValueError(operators)
        if operators == "device" and (model != "dir" or len(set(grids)) != 1):
            raise ValueError("on-device operator construction supports the Dirac model on equally sized meshes")
        assert frames >= INPUT_FRAMES + OUTPUT_FRAMES + 1 and 1 <= op_frames
        Vs, Fs, mats = [], [], {"L": [], "Di": [], "DiA": []}
        coords, faces_all, orders = [], [], []
        for (n, m) in grids:
            V0, F_ = mesh_ops.grid_cloth(n, m, rng, permute=permute)
            order = mesh_ops.MeshOrder.of_mesh(F_, V0.shape[0], reorder)
            V0, F_ = order.mesh(V0, F_)
            orders.append(order)
            amp = 0.03 * (0.5 + rng.random())
            k = 2 * np.pi * (1 + rng.integers(0, 3))
            ph = rng.random() * 2 * np.pi
            t = np.arange(frames)[:, None]
            disp = amp * np.sin(k * V0[None, :, 0] + 0.25 * t + ph)            # (T, V)
            Vt = np.repeat(V0[None], frames, 0)
            Vt[:, :, 2] += disp
            Vt[:, :, 1] += 0.5 * amp * np.cos(k * V0[None, :, 1] + 0.2 * t)
            coords.append(Vt.astype(np.float32))
            Vs.append(V0.shape[0])
            Fs.append(F_.shape[0])
            faces_all.append(F_.astype(np.int32))
            for tf in range(op_frames if operators == "pool" else 0):
                if model == "dir":
                    Di, DiA = mesh_ops.dirac(Vt[tf].astype(np.float64), F_)
                    mats["Di"].append(Di.astype(np.float32))
                    mats["DiA"].append(DiA.astype(np.float32))
                else:
                    mats["L"].append(mesh_ops.laplacian(Vt[tf].astype(np.float64), F_).astype(np.float32))
        self.num_vertices = np.array(Vs)
        self.num_faces = np.array(Fs)
        self.n = len(grids)
        self.orders = orders
        vmax = int(self.num_vertices.max())
        xyz = np.zeros((self.n, frames, vmax, 3), np.float32)
        for s, c in enumerate(coords):
            xyz[s, :, : c.shape[1]] = c
        self.xyz = torch.from_numpy(xyz).to(self.device)
        self.vcount = torch.from_numpy(self.num_vertices).to(self.device)
        self._mask_cache = {}
        if operators == "device":
            self.faces = torch.from_numpy(np.stack(faces_all)).to(self.device)          # (S, F, 3) int32
        elif model == "dir":
            self.pool_Di = OperatorPool(mats["Di"], self.device, want_bsr4=True)
            self.pool_DiA = OperatorPool(mats["DiA"], self.device, want_bsr4=True)
        else:
            self.pool_L = OperatorPool(mats["L"], self.device, want_bsr4=False)

    def _mask_of(self, seq_ids, nv):
        """(B, nv, 1) float mask of the real vertex rows of the padded batch (main.py:126-130 pads to the batch maximum).  It
        depends on the SELECTION only, and loops that visit every mesh once per step pass the same selection each time: the
        last few masks are kept (read-only downstream) — four elementwise launches per step otherwise."""
        cache = self.__dict__.setdefault("_mask_cache", {})      # (datasets.arap_from_files builds the object field by field)
        key = (seq_ids.tobytes(), int(nv))
        hit = cache.get(key)
        if hit is None:
            sid = h2d_async(seq_ids, self.device)
            hit = (torch.arange(nv, device=self.device)[None, :] < self.vcount[sid][:, None]).float().unsqueeze(2)
            if len(cache) >= 8:
                cache.pop(next(iter(cache)))
            cache[key] = hit
        return hit

    def to_dataset_order(self, x: torch.Tensor, seq_ids) -> torch.Tensor:
        """Per-vertex rows (B, nv, C) of the samples `seq_ids` from the STORED numbering (what sample_batch hands out and the
        model returns) back into the dataset's own vertex numbering; padding rows stay where they are."""
        return reorder_rows(x, self.orders, seq_ids, "vrank")

    def from_dataset_order(self, x: torch.Tensor, seq_ids) -> torch.Tensor:
        """The inverse: per-vertex rows given in the dataset's numbering, as the stored numbering wants them."""
        return reorder_rows(x, self.orders, seq_ids, "vorder")

    def _vertex_major(self):
        """Vertex-major (n, vmax, frames*3) copy of self.xyz (n, frames, vmax, 3): the 42 frames a sample needs of one
        vertex are then ONE contiguous run, gathered by sn_gather_segments_f32; rebuilt when self.xyz is replaced."""
        cached = getattr(self, "_xyz_vm", None)
        if cached is None or cached[0] is not self.xyz:
            n, fr, vmax, _ = self.xyz.shape
            cached = (self.xyz, self.xyz.permute(0, 2, 1, 3).reshape(n, vmax, fr * 3).contiguous())
            self._xyz_vm = cached
        return cached[1]

    def sample_batch(self, batch_size, rng: np.random.Generator, seq_ids=None, offsets=None, packed: bool = False) -> Batch:
        """Counterpart of sample_batch (main.py:98-185): random sequence + random start frame per sample; operator of
        the last input frame (main.py:156); everything zero-padded to the batch maximum (main.py:126-130).
        packed=True (no counterpart in the reference): the batch WITHOUT padding — inputs (1, sum V_i, 6), targets
        (1, sum V_i, 120), `mask` = PackedSegments, packed block-diagonal operators; `num_meshes` stays the sample count."""
        if seq_ids is None:
            seq_ids = rng.integers(0, self.n, size=batch_size)
        if offsets is None:
            hi = min(self.op_frames - INPUT_FRAMES + 1, self.frames - INPUT_FRAMES - OUTPUT_FRAMES)
            offsets = rng.integers(0, max(hi, 1), size=batch_size)
        seq_ids = np.asarray(seq_ids)
        offsets = np.asarray(offsets)
        B = len(seq_ids)
        nv = int(self.num_vertices[seq_ids].max())
        nf = int(self.num_faces[seq_ids].max())
        # a sample's 42 frames of a vertex are one contiguous run of the vertex-major copy: inputs and targets are gathered
        # straight into their final (B, nv, frames*3) layout (no permute / slice copies of the 160 MB window)
        vm = self._vertex_major()                                  # (n, vmax, frames*3)
        vmax, f3 = vm.shape[1], vm.shape[2]
        # element offsets of (sample, vertex 0, start frame) for the input and the target window: host arithmetic on the B
        # sample indices, ONE small upload (was: two uploads and four elementwise launches per step)
        hb = (seq_ids.astype(np.int64) * vmax) * f3 + 3 * offsets.astype(np.int64)
        bases = h2d_async(np.stack([hb, hb + 3 * INPUT_FRAMES]), self.device)
        base, base_t = bases[0], bases[1]
        op_ids = seq_ids * self.op_frames + (offsets + INPUT_FRAMES - 1)
        L = Di = DiA = None
        if packed:
            if self.operators == "device":
                raise ValueError("packed batches need pooled operators")
            # the real rows of every sample, mesh-major, gathered straight into the packed layout (no padded intermediate)
            seg = PackedSegments.cached(self.num_vertices[seq_ids], self.device)
            inputs = kernels.gather_segments_ragged(vm, base, seg, f3, 3 * INPUT_FRAMES).unsqueeze(0)
            targets = kernels.gather_segments_ragged(vm, base_t, seg, f3, 3 * OUTPUT_FRAMES).unsqueeze(0)
            if self.kind == "dir":
                Di, DiA = self.pool_Di.assemble(op_ids), self.pool_DiA.assemble(op_ids)
            else:
                L = self.pool_L.assemble(op_ids)
            return Batch(inputs, targets, seg, L, Di, DiA, B)
        inputs = kernels.gather_segments(vm, base, nv, f3, 3 * INPUT_FRAMES)
        targets = kernels.gather_segments(vm, base_t, nv, f3, 3 * OUTPUT_FRAMES)
        mask = self._mask_of(seq_ids, nv)
        if self.operators == "device":
            from .operators import dirac_operators_from_mesh

            sid = h2d_async(seq_ids, self.device)
            off = h2d_async(offsets, self.device)
            Vop = self.xyz[sid, off + INPUT_FRAMES - 1]                                   # (B, nv, 3): last input frame
            Di, DiA = dirac_operators_from_mesh(Vop, self.faces[sid])
        elif self.kind == "dir":
            Di = self.pool_Di.assemble(op_ids, 4 * nf, 4 * nv)
            DiA = self.pool_DiA.assemble(op_ids, 4 * nv, 4 * nf)
        else:
            L = self.pool_L.assemble(op_ids, nv, nv)
        return Batch(inputs, targets, mask, L, Di, DiA, B)


def reorder_rows(x: torch.Tensor, orders, seq_ids, which: str) -> torch.Tensor:
    """out[b, k] = x[b, table_b[k]] for k < V_b (table = MeshOrder.vrank: stored -> dataset order; .vorder: the inverse), rows
    past a mesh's size unchanged.  One gather over the tensor; evaluation / export only — training never needs it."""
    seq_ids = np.asarray(seq_ids)
    if all(orders[int(s_)].identity for s_ in seq_ids):
        return x
    B, nv = x.shape[0], x.shape[1]
    idx = np.tile(np.arange(nv, dtype=np.int64), (B, 1))
    for b, s_ in enumerate(seq_ids):
        t = getattr(orders[int(s_)], which)
        idx[b, : t.size] = t
    idx_d = h2d_async(idx, x.device)
    return torch.gather(x, 1, idx_d[:, :, None].expand(-1, -1, x.shape[2]))


def forward_loss(model, batch: Batch, global_batch: Optional[int] = None):
    if batch.Di is not None:
        out = model(batch.Di, batch.DiA, batch.mask, batch.inputs)
    else:
        out = model(batch.L, batch.mask, batch.inputs)
    return loss_fn(out, batch.targets, batch.mask, global_batch or batch.num_meshes), out


class GraphedTrainStep:
    """train_step with forward + loss + backward replayed from one hipGraph (graphs.GraphedTrainStep); the gradient
    all-reduce and the optimizer stay eager.  `example` fixes the batch signature (and becomes the static batch)."""

    def __init__(self, model, optimizer, example: Batch, global_batch: Optional[int] = None, bucket=None):
        import dataclasses

        from .graphs import GraphedTrainStep as _G

        # the example's tensors become the graph's static inputs and every load() overwrites them: the sampler's CACHED mask
        # (ClothSequences._mask_of hands the same tensor out again when a selection recurs) must not be one of them
        if isinstance(example.mask, torch.Tensor):
            example = dataclasses.replace(example, mask=example.mask.clone())
        self._g = _G(model, optimizer, example, lambda m, b: forward_loss(m, b, global_batch)[0], bucket)
        self.step, self.optimizer = self._g.step, optimizer

    def matches(self, batch: Batch) -> bool:
        return self._g.matches(batch)

    def __call__(self, batch: Batch, grad_sync=None):
        return self._g(batch, grad_sync)


def train_step(model, optimizer, batch: Batch, global_batch: Optional[int] = None, grad_sync=None, zero_grads=None):
    """One update (main.py:217-232): forward, masked smooth-L1, backward, [gradient all-reduce], Adam.
    zero_grads: e.g. FlatGradBucket.zero_ — one memset of the flat buffer instead of one launch per parameter."""
    loss, _ = forward_loss(model, batch, global_batch)
    if zero_grads is not None:
        zero_grads()
    else:
        optimizer.zero_grad(set_to_none=False)
    loss.backward()
    kernels.clear_absmax()
    if grad_sync is not None:
        grad_sync()
    optimizer.step()
    return loss
