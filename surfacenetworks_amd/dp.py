"""Data parallelism for the Surface-Network training step: one process per GPU, the mesh batch sharded across ranks,
ONE RCCL all-reduce per step over a single flat fp32 gradient bucket.

The reference is single-process/single-GPU; this layer is what `north_star` adds.  Design for MI355X:
  * The batched operator is block-diagonal, so meshes never interact inside SpMM / Linear / ELU: a mesh never spans
    GPUs and the data path needs NO collective.  Only gradients cross xGMI.
  * ~1.02 M parameters = 4.08 MB of fp32 gradients per step (ARAP / FAUST towers).  On the fully connected xGMI
    topology (7 links x ~153 GB/s per GPU) that is latency-bound, so the right shape is one contiguous bucket and
    one collective — not per-parameter hooks or many small buckets.  Parameter .grad tensors are views into the
    bucket, so there is no flatten/unflatten copy either.
  * BatchNorm statistics stay per replica (standard DDP semantics); `sync_batchnorm()` is an opt-in used to show
    parity with the single-process step at the same global batch.
  * Loss normalisation uses the GLOBAL batch (arap.loss_fn(..., global_batch)), so a SUM all-reduce of the shard
    gradients equals the single-process gradient — no averaging pass.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["init_distributed", "shard_round_robin", "shard_balanced", "FlatGradBucket", "broadcast_parameters",
           "sync_batchnorm", "world_info"]


def world_info():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend: Optional[str] = None, single_rank_group: bool = False):
    """Join the job described by RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).
    backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests.  Returns (rank, local_rank, world, device).
    single_rank_group: create the process group even when WORLD_SIZE is 1 — a one-rank RCCL communicator (ncclCommInitRank +
    a real ncclAllReduce per step through FlatGradBucket(always_reduce=True)): what a 1-GPU box can prove of the N > 1 path."""
    rank, local_rank, world = world_info()
    use_gpu = torch.cuda.is_available() and os.environ.get("SN_DP_FORCE_CPU", "0") != "1"
    if use_gpu:
        # one process per GPU; the modulo only matters for functional tests that put several gloo ranks on one GPU
        device = torch.device("cuda", local_rank % torch.cuda.device_count())
        torch.cuda.set_device(device)
    else:
        device = torch.device("cpu")
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if use_gpu else "gloo")
        kw = {"device_id": device} if (use_gpu and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world, device


def shard_round_robin(n_items: int, rank: int, world: int) -> np.ndarray:
    """Rank r owns items {i : i mod world == r}."""
    return np.arange(rank, n_items, world)


def shard_balanced(weights: Sequence[float], rank: int, world: int) -> np.ndarray:
    """Size-balanced bins for ragged batches (e.g. weight = nnz of a mesh): longest-processing-time greedy.
    Deterministic, identical on every rank."""
    w = np.asarray(weights, dtype=np.float64)
    order = np.argsort(-w, kind="stable")
    load = np.zeros(world)
    owner = np.empty(len(w), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += w[i]
    return np.flatnonzero(owner == rank)


class FlatGradBucket:
    """All gradients of a module in one contiguous fp32 buffer; `.grad` of every parameter is a view into it."""

    def __init__(self, params: Iterable[torch.nn.Parameter], always_reduce: bool = False):
        """always_reduce: run the pack + all-reduce even in a one-rank process group (the sum over one rank is the identity;
        used to put the collective itself on the measured path of a 1-GPU run)."""
        self.always_reduce = bool(always_reduce)
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off: off + n].view_as(p)
            off += n
        self._work = None

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def zero_(self):
        self.flat.zero_()

    def check_views(self) -> bool:
        """True while every .grad still aliases the bucket (optimizer.zero_grad(set_to_none=True) would break it)."""
        base = self.flat.untyped_storage().data_ptr()
        return all(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in self.params)

    def all_reduce(self, async_op: bool = False):
        """SUM over ranks (gradients were computed with the loss normalised by the global batch)."""
        if not self._collective():
            return None
        self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        return self._work

    def detach_grads(self):
        """Leave `.grad = None` on every parameter: the next backward then STORES each gradient instead of adding it into
        its bucket slice (one small kernel launch per parameter saved), and no zeroing pass is needed.  `sync()` brings
        the gradients back into the bucket when there is anything to reduce."""
        for p in self.params:
            p.grad = None

    def sync(self):
        """After a backward that followed `detach_grads()`: with one rank nothing is copied (the optimizer reads the
        fresh gradients); with several ranks the gradients are packed into the flat buffer by one multi-tensor copy,
        all-reduced (SUM) and the parameters' `.grad` re-pointed at the bucket slices."""
        if not self._collective():
            return None
        grads = [p.grad for p in self.params]
        have = [(v, g) for v, g in zip(self._views(), grads) if g is not None]
        missing = [v for v, g in zip(self._views(), grads) if g is None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if missing:
            torch._foreach_zero_(missing)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        for p, v in zip(self.params, self._views()):
            p.grad = v
        return None

    def _collective(self) -> bool:
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or self.always_reduce

    def _views(self):
        if getattr(self, "_view_list", None) is None:
            out, off = [], 0
            for p in self.params:
                n = p.numel()
                out.append(self.flat[off: off + n].view_as(p))
                off += n
            self._view_list = out
        return self._view_list

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every replica start from rank `src`'s parameters and buffers (one flat broadcast per dtype)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    tensors = [t for t in list(module.parameters()) + list(module.buffers())]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt, ts in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src)
        off = 0
        with torch.no_grad():
            for t in ts:
                t.copy_(flat[off: off + t.numel()].view_as(t))
                off += t.numel()


def sync_batchnorm(enabled: bool = True, group=None) -> None:
    """Opt-in: BatchNorm statistics of the GLOBAL batch in every fused BatchNorm+Linear (functional.set_bn_sync): the
    per-channel sums and the row count are all-reduced in the forward, the two backward reductions in the backward, so N
    replicas reproduce the single-process result at the same global batch (tests/test_dp_gloo.py)."""
    from . import functional as snF

    snF.set_bn_sync(enabled, group)
