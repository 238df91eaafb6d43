"""FAUST dense correspondence — counterpart of src/dense_correspondence/models.py and main.py:106-240,310-327.

One tower (`Model`: Laplacian, `DirModel`: Dirac; conv1 3->128, 15 residual blocks at 128 channels, conv2 128->120 plus
the input coordinates repeated 40x) is applied to both shapes; `SiameseModel` returns bmm(FA, FB^T), a (B, NA, NB)
score matrix, trained with the argmin-target cross entropy of loss_fun_delta_cross_entropy (main.py:229-240).
`state_dict` keys match the reference (model.conv1.*, model.rn{i}.*, model.conv2.*).
"""
from __future__ import annotations

import os
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels, mesh_ops
from . import utils_pt as utils
from .arap import make_adam
from .operators import OperatorPool, SparseOperator



def _add_last_frame(x, inputs, frames):
    """x + inputs[:, :, -3:].repeat(1, 1, frames) (models.py:105,152) as one broadcast add — same values, the repeated
    tensor is never written."""
    B, V, _ = x.shape
    return (x.reshape(B, V, frames, 3) + inputs[:, :, None, -3:]).reshape(B, V, 3 * frames)


class Model(nn.Module):
    def __init__(self, layer):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(3, 128, batch_norm=None)
        self.layer = layer
        for i in range(layer):
            self.add_module("rn{}".format(i), utils.LapResNet2(128) if i % 2 == 0 else utils.AvgResNet2(128))
        self.conv2 = utils.GraphConv1x1(128, 120, batch_norm="pre")

    def forward(self, L, mask, inputs):
        x = self.conv1(inputs)
        for i in range(self.layer):
            blk = self._modules["rn{}".format(i)]
            if i % 2 == 0:
                x = blk(L, mask, x, avg_next=i + 1 < self.layer)      # (a global-average block follows: hand it the tile sums)
            else:
                x = blk(L, mask, x)
        x = utils.elu_conv1x1(self.conv2, x)
        return _add_last_frame(x, inputs, 40)


class DirModel(nn.Module):
    def __init__(self, layer):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(3, 128, batch_norm=None)
        self.layer = layer
        for i in range(layer):
            self.add_module("rn{}".format(i), utils.DirResNet2(128) if i % 2 == 0 else utils.AvgResNet2(128))
        self.do = nn.Dropout2d()
        self.conv2 = utils.GraphConv1x1(128, 120, batch_norm="pre")

    def forward(self, Di, DiA, mask, inputs):
        batch_size = inputs.size(0)
        v = self.conv1(inputs)
        num_faces = DiA.size(2) // 4 if len(Di.size()) == 3 else DiA.size(1) // 4 // batch_size
        f = None                                # zeros(batch, faces, 128) (models.py:168), not materialised
        for i in range(self.layer):
            blk = self._modules["rn{}".format(i)]
            if i % 2 == 0:
                v, f = blk(Di, DiA, v, f, num_faces=num_faces, avg_next=i + 1 < self.layer)
            else:
                v = blk(None, mask, v)
        x = utils.elu_conv1x1(self.conv2, v)
        return _add_last_frame(x, inputs, 40)


class AmplifyModel(nn.Module):
    """models.py:51-82: the Laplacian tower fed a SEQUENCE of operators — block pair i uses L_sequence[i // 2], the last
    one from there on (main.py:72-83 builds the sequence from normalised powers of L)."""

    def __init__(self, layer):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(3, 128, batch_norm=None)
        self.layer = layer
        for i in range(layer):
            self.add_module("rn{}".format(i), utils.LapResNet2(128) if i % 2 == 0 else utils.AvgResNet2(128))
        self.conv2 = utils.GraphConv1x1(128, 120, batch_norm="pre")

    def forward(self, L_sequence, mask, inputs):
        x = self.conv1(inputs)
        for i in range(self.layer):
            L = L_sequence[-1] if i // 2 >= len(L_sequence) else L_sequence[i // 2]
            x = self._modules["rn{}".format(i)](L, mask, x)
        x = utils.elu_conv1x1(self.conv2, x)
        return _add_last_frame(x, inputs, 40)


class AvgModel(nn.Module):
    """models.py:84-109: global-average blocks only."""

    def __init__(self, layer):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(3, 128, batch_norm=None)
        self.layer = layer
        for i in range(layer):
            self.add_module("rn{}".format(i), utils.AvgResNet2(128))
        self.conv2 = utils.GraphConv1x1(128, 120, batch_norm="pre")

    def forward(self, L, mask, inputs):
        x = self.conv1(inputs)
        for i in range(self.layer):
            x = self._modules["rn{}".format(i)](L, mask, x)
        x = utils.elu_conv1x1(self.conv2, x)
        return _add_last_frame(x, inputs, 40)


class MlpModel(nn.Module):
    """models.py:111-138: per-node MLP blocks, GraphBatchNorm before the last layer, conv2 without BatchNorm."""

    def __init__(self, layer):
        super().__init__()
        self.conv1 = utils.GraphConv1x1(3, 128, batch_norm=None)
        self.layer = layer
        for i in range(layer):
            self.add_module("rn{}".format(i), utils.MlpResNet2(128))
        self.bn = utils.GraphBatchNorm(128)
        self.conv2 = utils.GraphConv1x1(128, 120, batch_norm=None)

    def forward(self, L, mask, inputs):
        x = self.conv1(inputs)
        for i in range(self.layer):
            x = self._modules["rn{}".format(i)](L, mask, x)
        x = self.conv2(F.elu(self.bn(x)))
        return _add_last_frame(x, inputs, 40)


class _TwoReaders(torch.autograd.Function):
    """(p_1 .. p_k) -> (p_1 .. p_k, p_1 .. p_k): two aliases of every parameter, one per application of a shared tower.
    The backward receives both gradients of every parameter together and adds them with one multi-tensor launch; autograd
    then stores the sums (`.grad is None`) or accumulates them as usual."""

    @staticmethod
    def forward(ctx, *params):
        ctx.set_materialize_grads(False)
        return tuple(q.view_as(q) for q in params) + tuple(q.view_as(q) for q in params)

    @staticmethod
    def backward(ctx, *grads):
        k = len(grads) // 2
        out = [a if b is None else b for a, b in zip(grads[:k], grads[k:])]
        both = [i for i in range(k) if grads[i] is not None and grads[k + i] is not None]
        if both:
            for i, t in zip(both, torch._foreach_add([grads[i] for i in both], [grads[k + i] for i in both])):
                out[i] = t
        return tuple(out)


_TOWER_SLOTS = weakref.WeakKeyDictionary()
_TOWER_STREAMS = weakref.WeakKeyDictionary()      # model -> (side stream, per-BatchNorm capture buffers)
# The two applications of the shared tower are independent until the score matrix; a 7000-row tower fills 219 of the chip's
# 512 workgroup slots and its step is ~500 launches of a few microseconds, so running them on two streams (inside the one
# captured hipGraph: two concurrent kernel chains) hides most of that latency.  (_TWO_STREAMS = False: one stream.)
_TWO_STREAMS = True          # (tests may set False: both towers on the caller's stream)


class SiameseModel(nn.Module):
    """models.py:184-203."""

    def __init__(self, model="dirac", layer=15):
        super().__init__()
        if "dir" in model:                       # same dispatch order as models.py:188-199
            self.model = DirModel(layer)
        elif "amp" in model:
            self.model = AmplifyModel(layer)
        elif "lap" in model:
            self.model = Model(layer)
        elif "avg" in model:
            self.model = AvgModel(layer)
        elif "mlp" in model:
            self.model = MlpModel(layer)
        else:
            raise ValueError("towers: 'dir', 'amp', 'lap', 'avg', 'mlp'")

    def _tower_slots(self):
        """[(owner module's parameter dict, key, index into the unique-parameter list)], the list itself; rebuilt when a
        slot no longer holds the parameter it was built from."""
        cached = _TOWER_SLOTS.get(self)
        if cached is not None:
            slots, plist = cached
            if all(d.get(k) is plist[i] and plist[i].requires_grad for d, k, i in slots):
                return cached
        slots, plist, index = [], [], {}
        for mod in self.model.modules():
            for k, q in mod._parameters.items():
                if q is not None and q.requires_grad:
                    if id(q) not in index:
                        index[id(q)] = len(plist)
                        plist.append(q)
                    slots.append((mod._parameters, k, index[id(q)]))
        _TOWER_SLOTS[self] = (slots, plist)      # kept outside the module: nothing of it is pickled or deep-copied with it
        return slots, plist

    def towers(self, OperationA, OperationB, inputA, inputB):
        """The two applications of the shared tower (models.py:201-202).  When gradients are wanted, each application reads
        the parameters through its own set of aliases (`_TwoReaders`): the two gradients of a parameter then meet in ONE
        multi-tensor addition at the end of the backward instead of in one accumulation launch per parameter (the tower
        has ~90 parameters; at a FAUST pair every launch is a few per cent of the step).  Same values: a + b either way.
        The aliases are put into the modules' parameter slots for the duration of each application (a few dozen dict
        assignments; torch.func.functional_call does the same with ~3 ms of bookkeeping per step)."""
        if not torch.is_grad_enabled():
            return self.model(*OperationA, inputA), self.model(*OperationB, inputB)
        slots, plist = self._tower_slots()
        if not plist:
            return self.model(*OperationA, inputA), self.model(*OperationB, inputB)
        alias = _TwoReaders.apply(*plist)
        k = len(plist)
        from . import functional as snF

        # (not with synchronised BatchNorm: its collectives would be issued from two streams)
        two = _TWO_STREAMS and inputA.is_cuda and snF._BN_SYNC is None
        if two:
            # Two streams need two INDEPENDENT applications.  (i) A pair of the same frame (`PairBatch(ds, i, i)`: the reference
            # samples the two frames independently, main.py:106-191) hands both towers the same operator / mask objects, whose
            # derived forms (row-blocked arrays, transposes, the mask's 1/count) are built lazily by the first application on
            # ITS stream and would be read by the other without an ordering edge.  (ii) A BatchNorm with momentum=None keeps
            # a cumulative average that the capture-buffer trick below (momentum 1) cannot express.
            flat = lambda ops: [o for o in (ops if isinstance(ops, (tuple, list)) else (ops,)) if o is not None]
            ida = {id(o) for o in flat(OperationA)} | {id(inputA)}
            shared = any(id(o) in ida for o in flat(OperationB)) or inputB is inputA
            cumulative = any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.momentum is None for m in self.model.modules())
            two = not shared and not cumulative
        try:
            for d, key, i in slots:
                d[key] = alias[i]
            if two:
                main = torch.cuda.current_stream()
                side, bns = self._tower_streams()
                side.wait_stream(main)                       # (fork: the second tower starts from everything issued so far)
            FA = self.model(*OperationA, inputA)
            for d, key, i in slots:
                d[key] = alias[k + i]
            if not two:
                FB = self.model(*OperationB, inputB)
            else:
                # The towers share their BatchNorm modules: the reference applies A's running-statistics update, then B's.
                # On two streams the second application writes ITS batch statistics into capture buffers instead (momentum 1
                # makes the fold kernel store them as they are) and the update is applied after the join, in that order.
                saved = []
                for bn, cap_mean, cap_var in bns:
                    if not (bn.training and bn.track_running_stats):
                        continue                             # (eval mode normalises WITH the running statistics: leave them)
                    saved.append((bn, bn._buffers["running_mean"], bn._buffers["running_var"], bn._buffers["num_batches_tracked"], bn.momentum))
                    bn._buffers["running_mean"], bn._buffers["running_var"], bn._buffers["num_batches_tracked"] = cap_mean, cap_var, None
                    bn.momentum = 1.0
                try:
                    with torch.cuda.stream(side):
                        FB = self.model(*OperationB, inputB)
                finally:
                    for bn, rm, rv, nbt, mom in saved:
                        bn._buffers["running_mean"], bn._buffers["running_var"], bn._buffers["num_batches_tracked"] = rm, rv, nbt
                        bn.momentum = mom
                main.wait_stream(side)                       # (join)
                FB.record_stream(main)
                live = [(bn, cm, cv) for bn, cm, cv in bns if bn.training and bn.track_running_stats]
                if live:
                    with torch.no_grad():
                        means, vars_ = [bn.running_mean for bn, _, _ in live], [bn.running_var for bn, _, _ in live]
                        moms = {bn.momentum for bn, _, _ in live}
                        if len(moms) == 1:                   # one multi-tensor launch per operation
                            m_ = moms.pop()
                            torch._foreach_mul_(means + vars_, 1.0 - m_)
                            torch._foreach_add_(means + vars_, [cm for _, cm, _ in live] + [cv for _, _, cv in live], alpha=m_)
                        else:
                            for bn, cm, cv in live:
                                bn.running_mean.mul_(1.0 - bn.momentum).add_(cm, alpha=bn.momentum)
                                bn.running_var.mul_(1.0 - bn.momentum).add_(cv, alpha=bn.momentum)
                        torch._foreach_add_([bn.num_batches_tracked for bn, _, _ in live], 1)
        finally:
            for d, key, i in slots:
                d[key] = plist[i]
        return FA, FB

    def _tower_streams(self):
        """(side stream, [(BatchNorm module, capture buffer for its batch mean, ... for its unbiased batch variance)]) — kept
        outside the module (nothing of it is pickled or deep-copied with it), rebuilt when the modules or their device change."""
        cached = _TOWER_STREAMS.get(self)
        mods = [m for m in self.model.modules() if isinstance(m, nn.BatchNorm1d) and m.running_mean is not None]
        if cached is not None and len(cached[1]) == len(mods) and all(
                a is b and cm.device == b.running_mean.device for (a, cm, _), b in zip(cached[1], mods)):
            return cached
        dev = next(self.model.parameters()).device
        cached = (torch.cuda.Stream(device=dev),
                  [(m, torch.zeros_like(m.running_mean), torch.zeros_like(m.running_var)) for m in mods])
        _TOWER_STREAMS[self] = cached
        return cached

    def forward(self, OperationA, OperationB, inputA, inputB):
        FA, FB = self.towers(OperationA, OperationB, inputA, inputB)
        return torch.bmm(FA, FB.transpose(1, 2))


def correspondence_target(GA, lA, liA, GB, lB, liB):
    """main.py:236-237: `_, GAB = torch.min(GA[:, liA[lB]] + GB[liB[lA], :], dim=1)` in one kernel that reads each matrix
    once and keeps neither gathered copy nor their sum (sn_pair_argmin_f32)."""
    if GA.dtype != torch.float32 or GB.dtype != torch.float32 or GA.stride(-1) != 1 or GB.stride(-1) != 1:
        return torch.min(GA[:, liA[lB]] + GB[liB[lA], :], dim=1)[1]      # (the kernel takes row-major fp32 matrices)
    return kernels.pair_argmin(GA, liA[lB], GB, liB[lA])


class _PairCrossEntropy(torch.autograd.Function):
    """F.cross_entropy(S[:NA, :NB], target) (main.py:238-239) in two passes over the score matrix instead of torch's five
    (log_softmax, nll_loss, their backward passes, the zero padding of the slice's gradient): sn_pair_ce_fwd/bwd_f32."""

    @staticmethod
    def forward(ctx, scores, target, NA, NB):
        S = scores[0] if scores.dim() == 3 else scores       # (B, N, N): sample 0 is the one scored (main.py:238)
        lse, rowloss = kernels.pair_ce_fwd(S, target, NA, NB)
        ctx.save_for_backward(S, target, lse)
        ctx.dims = (NA, NB)
        ctx.batch = scores.shape[0] if scores.dim() == 3 else None
        return rowloss.sum() / NA

    @staticmethod
    def backward(ctx, g):
        S, target, lse = ctx.saved_tensors
        dS = kernels.pair_ce_bwd(S, target, lse, g.reshape(1).contiguous(), *ctx.dims)
        if ctx.batch is None:
            return dS, None, None, None
        if ctx.batch == 1:                   # the gradient of `outputs[0]` as a view: no zero fill + copy of the (1, N, N) matrix
            return dS.unsqueeze(0), None, None, None
        full = dS.new_zeros((ctx.batch,) + tuple(dS.shape))
        full[0] = dS
        return full, None, None, None


def pair_cross_entropy(scores, target, NA: int, NB: int):
    """Cross entropy of the NA x NB corner of the (padded) score matrix against `target`, mean over rows.  `scores`: the
    (N, N) matrix, or the (B, N, N) output of SiameseModel, of which sample 0 is scored."""
    if scores.dtype != torch.float32 or scores.stride(-1) != 1:
        S = scores[0] if scores.dim() == 3 else scores
        return F.cross_entropy(S[:NA, :NB], target)
    return _PairCrossEntropy.apply(scores, target, NA, NB)


def loss_fun_delta_cross_entropy(outputs, targetX, targetY):
    """main.py:229-240, including its quirk of always scoring outputs[0] (the reference runs batch 1, main.py:40)."""
    loss = outputs.new_zeros(1)
    for i in range(outputs.size(0)):
        GA, lA, liA = targetX[i]
        GB, lB, liB = targetY[i]
        NA, NB = lA.size(0), lB.size(0)
        GAB = correspondence_target(GA, lA, liA, GB, lB, liB)
        loss = loss + pair_cross_entropy(outputs, GAB, NA, NB)
    return loss / outputs.size(0)


class _StreamedCorrespondenceCE(torch.autograd.Function):
    """mean_i CE(FA[i]·FBᵀ, target[i]) without materialising the (NA, NB) score matrix (196 MB at 7000 x 7000, plus its
    softmax and gradient in the reference: models.py:203, main.py:238-239).  Row blocks of FA are multiplied against FB,
    reduced to a log-sum-exp and the target logit, and discarded; the backward recomputes each block."""

    @staticmethod
    def forward(ctx, FA, FB, target, block):
        NA = FA.shape[0]
        loss = FA.new_zeros((), dtype=torch.float64)
        for r0 in range(0, NA, block):
            logits = FA[r0:r0 + block] @ FB.t()
            lse = torch.logsumexp(logits, dim=1)
            tgt = logits.gather(1, target[r0:r0 + block, None]).squeeze(1)
            loss += (lse - tgt).double().sum()
        ctx.save_for_backward(FA, FB, target)
        ctx.block = block
        return (loss / NA).to(FA.dtype)

    @staticmethod
    def backward(ctx, g):
        FA, FB, target = ctx.saved_tensors
        NA = FA.shape[0]
        gFA = torch.empty_like(FA)
        gFB = torch.zeros_like(FB)
        scale = g / NA
        for r0 in range(0, NA, ctx.block):
            a = FA[r0:r0 + ctx.block]
            p = torch.softmax(a @ FB.t(), dim=1)
            p.scatter_add_(1, target[r0:r0 + ctx.block, None], -torch.ones_like(p[:, :1]))
            p *= scale
            gFA[r0:r0 + ctx.block] = p @ FB
            gFB += p.t() @ a
        return gFA, gFB, None, None


class _FusedCorrespondenceCE(torch.autograd.Function):
    """mean_r CE(FA[r]·FBᵀ, target[r]) over the NA x NB corner, from the (rows, K) tower features: scores, soft-max and both
    gradient products in hand-written matrix-pipe kernels, the score matrix never in memory (sn_pair_fused_fwd/bwd_f32;
    replaces models.py:203 + main.py:238-239 and their backward passes — two 7000 x 7000 x 120 library GEMMs among them)."""

    @staticmethod
    def forward(ctx, FA, FB, target, NA, NB):
        lse, rowloss, ws = kernels.pair_fused_fwd(FA, FB, target, NA, NB)
        ctx.save_for_backward(target, lse, ws)
        ctx.dims = (NA, NB, FA.shape[0], FB.shape[0], FA.shape[1])
        return rowloss.sum() / NA

    @staticmethod
    def backward(ctx, g):
        target, lse, ws = ctx.saved_tensors
        dFA, dFB = kernels.pair_fused_bwd(target, lse, g.reshape(1).contiguous(), ws, *ctx.dims)
        return dFA, dFB, None, None, None


_PREFETCH_TARGET = True      # the argmin target of the NEXT pair on a side stream (False: on the caller's stream)
_TARGET_STREAMS = {}


def _target_stream(device):
    key = torch.device(device).index
    st = _TARGET_STREAMS.get(key)
    if st is None:
        st = _TARGET_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


_FUSED_SCORES = os.environ.get("SN_PAIR_FUSED", "1") != "0"      # A/B switch: "0" = bmm + sn_pair_ce_* on the score matrix


def fused_pair_supported(FA, FB) -> bool:
    return (_FUSED_SCORES and FA.is_cuda and FA.dtype == torch.float32 and FB.dtype == torch.float32 and FA.dim() == 3
            and FA.shape[-1] == FB.shape[-1] <= 128 and FA[0].is_contiguous() and FB[0].is_contiguous())


def fused_pair_cross_entropy(FA, FB, target, NA: int, NB: int):
    """The loss of one pair from the (B, N, K) tower outputs — sample 0 is the one scored (main.py:238)."""
    return _FusedCorrespondenceCE.apply(FA[0], FB[0], target, NA, NB)


def streamed_delta_cross_entropy(FA, FB, targetX, targetY, block=1024):
    """loss_fun_delta_cross_entropy (main.py:229-240) computed from the tower features instead of bmm(FA, FBᵀ):
    FA, FB are the (B, N, 120) tower outputs.  Same value/gradients as SiameseModel + loss_fun_delta_cross_entropy,
    including the reference's use of sample 0's scores for every i."""
    B = FA.size(0)
    loss = FA.new_zeros(())
    for i in range(B):
        GA, lA, liA = targetX[i]
        GB, lB, liB = targetY[i]
        NA, NB = lA.size(0), lB.size(0)
        tgt = correspondence_target(GA, lA, liA, GB, lB, liB)
        loss = loss + _StreamedCorrespondenceCE.apply(FA[0, :NA], FB[0, :NB], tgt, block)
    return loss / B


def make_optimizer(model):
    return make_adam(model)       # main.py:285


class TorusBodies:
    """Synthetic stand-in for the FAUST .npz frames (main.py:65-104): torus-grid meshes (65 x 106 -> 6890 vertices,
    13 780 faces), padded to 7000 vertices (main.py:193), a random label permutation pair and a synthetic
    'geodesic' matrix per shape."""

    def __init__(self, count, n=65, m=106, pad_to=7000, seed=4, device="cuda"):
        rng = np.random.default_rng(seed)
        self.device = torch.device(device)
        self.pad_to = pad_to
        self.frames = []
        mats = []
        for _ in range(count):
            V, F_ = mesh_ops.torus_grid(n, m, rng)
            nv = V.shape[0]
            mats.append(mesh_ops.laplacian(V, F_).astype(np.float32))
            label = rng.permutation(nv)
            G = torch.from_numpy(V.astype(np.float32)).to(self.device)
            self.frames.append({
                "V": torch.from_numpy(V.astype(np.float32)).to(self.device),
                "label": torch.from_numpy(label).to(self.device),
                "label_inv": torch.from_numpy(np.argsort(label)).to(self.device),
                "G": torch.cdist(G, G),                    # synthetic stand-in for dist_mat
            })
        self.pool_L = OperatorPool(mats, self.device)
        self.n = count
        self._samples = {}
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)        # the frames are read from other streams later (PairBatch's target)

    def sample(self, idx):
        """(inputs, target triple, mask, operator) of frame idx, padded to `pad_to`.  A frame never changes, so its padded
        tensors and its assembled operator (with the forms the products derive from it) are built once and handed out again:
        they are the dataset's own — read, do not write (PairBatch.owned() copies what a captured step overwrites)."""
        hit = self._samples.get(idx)
        if hit is None:
            fr = self.frames[idx]
            nv = fr["V"].shape[0]
            inputs = torch.zeros(1, self.pad_to, 3, device=self.device)
            inputs[0, :nv] = fr["V"]
            mask = torch.zeros(1, self.pad_to, 1, device=self.device)
            mask[0, :nv] = 1
            L = self.pool_L.assemble([idx], self.pad_to, self.pad_to)
            hit = self._samples[idx] = (inputs, [(fr["G"], fr["label"], fr["label_inv"])], mask, L)
        return hit


class PairBatch:
    """One training pair (two independent samples, main.py:310-316) as a flat set of device tensors — what the hipGraph
    replay needs as static inputs (graphs.GraphedTrainStep).  The loss target (main.py:236-237) is computed here, from the
    dataset's resident geodesic matrices: the replay then takes a 6890-entry index vector as input instead of two
    190 MB matrices copied into static buffers every step."""

    def __init__(self, ds: TorusBodies, ia: int, ib: int):
        self.inX, self.tX, self.mX, self.LX = ds.sample(ia)
        self.inY, self.tY, self.mY, self.LY = ds.sample(ib)
        (GA, lA, liA), (GB, lB, liB) = self.tX[0], self.tY[0]
        self.NA, self.NB = int(lA.size(0)), int(lB.size(0))
        # The target reads two resident 190 MB matrices (~0.14 ms) and depends on nothing of the model: on a device it is
        # computed on a stream of its own, so that the target of the NEXT pair overlaps the step of the current one; the
        # consumer waits for `_target_ready` (target_tensor()).
        self._target_ready = None
        if GA.is_cuda and _PREFETCH_TARGET:
            side = _target_stream(GA.device)             # (the datasets synchronise the device once, after building their tensors)
            with torch.cuda.stream(side):
                self.target = correspondence_target(GA, lA, liA, GB, lB, liB)
                self._target_ready = side.record_event()
        else:
            self.target = correspondence_target(GA, lA, liA, GB, lB, liB)

    def target_tensor(self):
        """The loss target, safe to use on the CURRENT stream."""
        if self._target_ready is not None:
            cur = torch.cuda.current_stream(self.target.device)
            cur.wait_event(self._target_ready)
            self.target.record_stream(cur)
            self._target_ready = None
        return self.target

    def owned(self) -> "PairBatch":
        """A copy for the static batch of a captured step (overwritten in place by every `load()`): nothing in it is the
        dataset's own tensor."""
        import copy

        b = copy.copy(self)
        b.inX, b.inY, b.mX, b.mY, b.target = (t.clone() for t in (self.inX, self.inY, self.mX, self.mY, self.target_tensor()))
        b._target_ready = None
        own = lambda ops: type(ops)(o.clone() for o in ops) if isinstance(ops, (tuple, list)) else ops.clone()
        b.LX, b.LY = own(self.LX), own(self.LY)
        b.tX = b.tY = None
        return b

    def graph_constants(self):
        """Host values baked into the captured kernels' arguments (the corner of the score matrix the cross entropy reads):
        part of the batch signature, so a pair with other vertex counts cannot be replayed through this capture."""
        return (self.NA, self.NB)

    def graph_tensors(self):
        from .graphs import operator_tensors

        out = [self.inX, self.inY, self.mX, self.mY, self.target_tensor()]
        for ops in (self.LX, self.LY):
            for o in (ops if isinstance(ops, (tuple, list)) else (ops,)):
                out += operator_tensors(o)
        return out


def forward_loss(model, b: PairBatch):
    """loss_fun_delta_cross_entropy for the one pair of a PairBatch (main.py:229-240 at batch size 1), target precomputed."""
    if isinstance(model, SiameseModel):
        # same value as model(...) + pair_cross_entropy, without the (1, N, N) score matrix in between
        FA, FB = model.towers(_operation(b.LX, b.mX), _operation(b.LY, b.mY), b.inX, b.inY)
        if fused_pair_supported(FA, FB):
            return fused_pair_cross_entropy(FA, FB, b.target_tensor(), b.NA, b.NB).reshape(1)
        out = torch.bmm(FA, FB.transpose(1, 2))
    else:
        out = model(_operation(b.LX, b.mX), _operation(b.LY, b.mY), b.inX, b.inY)
    return pair_cross_entropy(out, b.target_tensor(), b.NA, b.NB).reshape(1)


def graphed_train_step(model, optimizer, example: PairBatch, bucket=None, global_pairs: int = 1):
    """Training step with forward + loss + backward replayed from one hipGraph: a pair of 7000-row shapes is ~1000 launches
    of a few microseconds, i.e. launch-bound when issued from Python.
    Data parallel (BASELINE config 4, one pair per GPU and step, main.py:40,310-327): every rank captures its own step with
    global_pairs = the number of ranks (the loss is the mean over the job's pairs) and a `bucket` (dp.FlatGradBucket): the
    replay is followed by pack + SUM all-reduce + Adam."""
    from .graphs import GraphedTrainStep

    if global_pairs == 1:
        return GraphedTrainStep(model, optimizer, example.owned(), forward_loss, bucket)
    scale = 1.0 / float(global_pairs)
    return GraphedTrainStep(model, optimizer, example.owned(), lambda m, b: forward_loss(m, b) * scale, bucket)


class LSequence(list):
    """The operator list of the 'amp' tower (main.py:72-83, 187-188): ONE positional argument of AmplifyModel.forward."""


def _operation(ops, mask):
    """[L, mask], [L_sequence, mask] or [Di, DiA, mask]: the `Operation` lists SiameseModel.forward unpacks (main.py:317-320)."""
    if isinstance(ops, LSequence):
        return [ops, mask]
    return [*ops, mask] if isinstance(ops, (tuple, list)) else [ops, mask]


def amplify_sequence(L):
    """main.py:72-83: the stored Laplacian scaled by D^-1/2 on both sides (D = entries per row - 1), then twice
    `L <- (D^-1/2 L D^-1/2)^2`; three float32 CSR matrices."""
    import scipy.sparse as sp

    L = L.astype("f").tocsr()
    idp = L.indptr
    Dsq = sp.diags(1 / np.sqrt(idp[1:] - idp[:-1] - 1)).astype("f")
    L = Dsq.dot(L).dot(Dsq).astype("f")
    out = [L.tocsr()]
    for _ in range(2):
        L = Dsq.dot(L).dot(Dsq).astype("f")
        L = L.dot(L).tocsr()
        out.append(L)
    return out


class FaustFrames:
    """FAUST frames read from the reference's .npz files (datasets.load_faust_frame; src/dense_correspondence/main.py:66-102)
    as a resident dataset with the TorusBodies interface: coordinates, label permutations and geodesic matrices as device
    tensors, the operators of the requested tower in OperatorPools; `sample(idx)` pads to `pad_to` vertices (main.py:193)."""

    def __init__(self, frames, model="lap", pad_to=None, device="cuda", reorder="auto"):
        """reorder ("auto" / True / False): scans arrive in whatever vertex order the scanner wrote (main.py:66-102 loads them as
        stored); the frames are STORED in a locality numbering (mesh_ops.MeshOrder) — coordinates, operators, label permutations
        and the geodesic matrix renumbered once.  The loss (rows of the score matrix against targets from G, label, label_inv)
        is invariant to the numbering; `orders[i].vorder` maps stored positions back to the file's vertex ids."""
        self.device = torch.device(device)
        self.kind = "dir" if "dir" in model else ("amp" if "amp" in model else "lap")      # dispatch order of main.py:72-95
        self.orders = [mesh_ops.MeshOrder.of_mesh(fr["F"].cpu().numpy(), int(fr["V"].shape[0]), reorder) for fr in frames]
        frames = [fr if o.identity else _renumbered_frame(fr, o) for fr, o in zip(frames, self.orders)]
        self.frames = frames
        self.n = len(frames)
        nv = max(int(fr["V"].shape[0]) for fr in frames)
        nf = max(int(fr["F"].shape[0]) for fr in frames)
        self.pad_to = int(pad_to) if pad_to is not None else nv
        self.pad_faces = nf
        if self.kind == "dir":
            self.pool_Di = OperatorPool([fr["Di"] for fr in frames], self.device, want_bsr4=True)
            self.pool_DiA = OperatorPool([fr["DiA"] for fr in frames], self.device, want_bsr4=True)
        elif self.kind == "amp":
            seqs = [amplify_sequence(fr["L"]) for fr in frames]
            self.pool_Lseq = [OperatorPool([s[j] for s in seqs], self.device) for j in range(3)]
        else:
            self.pool_L = OperatorPool([fr["L"] for fr in frames], self.device)
        self._samples = {}
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)        # (as TorusBodies: the frames are read from other streams later)

    def to_dataset_order(self, x: torch.Tensor, idx: int) -> torch.Tensor:
        """Per-vertex rows (1, pad_to, C) of frame `idx` — tower features, per-vertex predictions — from the STORED numbering
        (what sample() hands out and the towers return) back into the file's own vertex numbering; padding rows stay where they
        are.  Evaluation / export only: the loss is invariant to the numbering."""
        from .arap import reorder_rows

        return reorder_rows(x, self.orders, [idx], "vrank")

    def from_dataset_order(self, x: torch.Tensor, idx: int) -> torch.Tensor:
        """The inverse: per-vertex rows given in the file's numbering, as the stored numbering wants them."""
        from .arap import reorder_rows

        return reorder_rows(x, self.orders, [idx], "vorder")

    def sample(self, idx):
        """As TorusBodies.sample: built once per frame, the dataset's own tensors from then on."""
        hit = self._samples.get(idx)
        if hit is None:
            fr = self.frames[idx]
            nv = fr["V"].shape[0]
            inputs = torch.zeros(1, self.pad_to, 3, device=self.device)
            inputs[0, :nv] = fr["V"]
            mask = torch.zeros(1, self.pad_to, 1, device=self.device)
            mask[0, :nv] = 1
            if self.kind == "dir":
                ops = (self.pool_Di.assemble([idx], 4 * self.pad_faces, 4 * self.pad_to),
                       self.pool_DiA.assemble([idx], 4 * self.pad_to, 4 * self.pad_faces))
            elif self.kind == "amp":
                ops = LSequence(pl.assemble([idx], self.pad_to, self.pad_to) for pl in self.pool_Lseq)
            else:
                ops = self.pool_L.assemble([idx], self.pad_to, self.pad_to)
            hit = self._samples[idx] = (inputs, [(fr["G"], fr["label"], fr["label_inv"])], mask, ops)
        return hit


def _renumbered_frame(fr, order):
    """A frame dict (datasets.load_faust_frame) in the stored numbering of `order`: vertex k is the file's vertex vorder[k]."""
    dev = fr["V"].device
    vo = torch.from_numpy(order.vorder).to(dev)
    vr = torch.from_numpy(order.vrank).to(dev)
    out = dict(fr)
    out["V"] = fr["V"][vo]
    out["F"] = vr[fr["F"].long()[torch.from_numpy(order.forder).to(dev)]]
    out["label"] = fr["label"][vo]                          # label of the vertex now stored at position k
    out["label_inv"] = vr[fr["label_inv"].long()]           # stored position of the vertex with label c
    out["G"] = fr["G"][vo][:, vo].contiguous()
    for key, rows, cols, group in (("L", "vorder", "vorder", 1), ("Di", "forder", "vorder", 4), ("DiA", "vorder", "forder", 4)):
        if fr.get(key) is not None:
            out[key] = mesh_ops.permute_operator(fr[key], getattr(order, rows), getattr(order, cols), group).astype(np.float32)
    return out


def forward_pair_loss(model, ds, ia: int, ib: int, streamed: bool = False, block: int = 1024):
    """Siamese forward + delta cross entropy of one pair (main.py:310-324).  streamed=True: the towers' features go straight
    into streamed_delta_cross_entropy — the (N, N) score matrix, its softmax and its gradient are never materialised."""
    inX, tX, mX, LX = ds.sample(ia)
    inY, tY, mY, LY = ds.sample(ib)
    if streamed:
        FA, FB = model.towers(_operation(LX, mX), _operation(LY, mY), inX, inY)
        return streamed_delta_cross_entropy(FA, FB, tX, tY, block)
    out = model(_operation(LX, mX), _operation(LY, mY), inX, inY)
    return loss_fun_delta_cross_entropy(out, tX, tY)


def train_step(model, optimizer, ds, ia: int, ib: int, grad_sync=None, streamed: bool = False, global_pairs: int = 1,
               zero_grads=None):
    """main.py:310-327: two independent samples, siamese forward, delta-CE loss, Adam.
    Data parallel (BASELINE config 4: one pair per GPU and step): every rank passes its own pair and `global_pairs` = the
    number of ranks, so the loss is the mean over the job's pairs and the SUM all-reduce of `grad_sync`
    (dp.FlatGradBucket.sync / all_reduce) needs no averaging pass.  zero_grads: e.g. FlatGradBucket.detach_grads."""
    loss = forward_pair_loss(model, ds, ia, ib, streamed)
    if global_pairs != 1:
        loss = loss / global_pairs
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
