"""hipGraph capture of one training step (forward + loss + backward) of the Surface-Network models.

A step of the ARAP Dirac model is ~800 kernel launches of 5-400 us; launched one by one from Python the host needs
30-50 us per launch, which is the same order as the GPU time, so any host slowdown (a busy node, eight ranks sharing
the cores) turns the step launch-bound.  The reference has the same structure (one Python-dispatched torch op after
the other, src/as_rigid_as_possible/main.py:217-232).  Here the whole forward/backward is recorded once into a
hipGraph on static buffers and replayed with a single launch per step:

    batch  = dataset.sample_batch(...)        # eager: a few launches, new tensors
    step.load(batch)                          # device-to-device copies into the static batch
    step.replay()                             # ONE hipGraphLaunch: zero grads, forward, loss, backward
    bucket.all_reduce(); optimizer.step()     # eager (RCCL and the optimizer stay outside the graph)

Only the kernel launches are recorded — same kernels, same order, same arithmetic — so a replay is bit-identical to
the eager step (tests/test_graph_gpu.py).  The graph is tied to the batch *signature* (tensor shapes and operator
entry counts); `GraphedStep.matches(batch)` tells the caller whether a batch can be replayed or needs a new capture.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from . import kernels
from .operators import SparseOperator

__all__ = ["operator_tensors", "batch_tensors", "batch_signature", "GraphedStep", "GraphedTrainStep"]


# The ring-vs-row-blocked choice of a Laplacian-type operator (SparseOperator.ring_ok) depends on the band of the batch's
# VALUES, not on its shapes, and it decides whether the row-blocked arrays are graph inputs.  A capture therefore FREEZES the
# choice of every operator of its example (in listing order) and every later batch is listed — and, were it multiplied
# eagerly, dispatched — with those choices: `_ring_choices` is ("record", list) while the example is listed, ("apply", iterator)
# while a later batch is, None otherwise.
_ring_choices = None


class _RingChoices:
    def __init__(self, mode, store):
        self.state = (mode, store if mode == "record" else iter(store))

    def __enter__(self):
        global _ring_choices
        self.prev, _ring_choices = _ring_choices, self.state

    def __exit__(self, *exc):
        global _ring_choices
        _ring_choices = self.prev


class _SignatureMismatch(Exception):
    """A batch whose operators cannot take the kernels the capture froze (listing aborted: the batch does not match)."""


def _ring_choice(o: SparseOperator) -> bool:
    """The ring-vs-row-blocked choice used to LIST an operator's arrays.  The frozen choices live on the GraphedStep (the
    `_RingChoices` context), never on the operator: an operator that was merely compared with, or loaded into, a capture keeps
    dispatching by its own band when it is multiplied eagerly later."""
    if _ring_choices is not None and _ring_choices[0] == "apply":
        take = next(_ring_choices[1])
        # the sliding-window kernel bounds a row by its first and last entry: an operator with non-ascending columns must
        # never be loaded into a capture that multiplies with it (band()[1] == 0x7fffffff marks such rows)
        if take and o.band()[1] == 0x7fffffff:
            raise _SignatureMismatch()
        return take
    take = bool(o.ring_ok(128) or o.ring_ok(64))
    if _ring_choices is not None:
        _ring_choices[1].append(take)
    return take


def operator_tensors(op: Optional[SparseOperator]) -> List[torch.Tensor]:
    """Every materialised device array of an operator and its attached transpose, in a fixed order."""
    if op is None:
        return []
    out: List[torch.Tensor] = []
    # A CSR operator that arrives without its transpose (laplacian_operator_from_mesh, as_operator) would build it lazily in
    # the warm-up backward — after the static list was taken — and every replay would then multiply by the EXAMPLE's
    # transpose.  Built here, it is listed (and reloaded) with the operator's own arrays, as the pools' operators are.
    if op._t is None and op._csr is not None and op.is_cuda:
        op.t()
    for o in (op, op._t):
        if o is None:
            continue
        if o._csr is not None:
            out.extend(o._csr)
        if isinstance(o._bsr4, tuple):
            out.extend(o._bsr4)
        if isinstance(o._q3, tuple):
            out.extend(o._q3)
        # A plain CSR operator (Laplacian-type) is multiplied through its 4x1 row-blocked form, derived from the CSR arrays
        # on first use and cached on the operator.  Under a graph that derived form must be an INPUT like the arrays it
        # comes from — a form cached during warm-up would otherwise be replayed against every later batch — so it is built
        # here (for the example and for every batch that is loaded) and listed with them.
        if o._csr is not None and not isinstance(o._bsr4, tuple) and not isinstance(o._q3, tuple):
            from . import functional as snF

            lfmt = o.format if o.format in ("ring", "rb4", "csr") else snF._LAPLACIAN_FORMAT
            if lfmt in ("ring", "rb4") and o.is_cuda:
                # (an operator that takes the sliding-window kernel is multiplied straight from its CSR arrays: nothing
                #  derived to list; the decision — the band of the operator — is measured here, before the capture)
                if lfmt == "ring" and _ring_choice(o):
                    continue
                r = o.rb4()
                if r is not None:
                    out.extend(r)
    return out


def batch_tensors(batch) -> List[torch.Tensor]:
    """Device tensors of a Batch-like object (inputs, targets, mask, then L / Di / DiA arrays); a batch type with another
    layout supplies them itself through a `graph_tensors()` method."""
    if hasattr(batch, "graph_tensors"):
        return list(batch.graph_tensors())
    out = [t for t in (batch.inputs, batch.targets, batch.mask) if t is not None]
    for name in ("L", "Di", "DiA"):
        out.extend(operator_tensors(getattr(batch, name, None)))
    return out


class BatchAhead:
    """Input pipeline for small batches: `make_batch()` (sampling + block-diagonal assembly: a dozen small launches) runs on a
    stream of its own, ONE batch ahead of the consumer — batch t+1 is assembled while step t computes.  `get()` hands the
    finished batch to the current stream (event wait + record_stream on its tensors) and starts the next one.  Worth it when
    the step leaves the GPU room (Mesh-MNIST batch, FAUST pair); at 64 large meshes the assembly only competes with the step."""

    def __init__(self, make_batch, device=None):
        self.make = make_batch
        self.side = torch.cuda.Stream(device=device)
        torch.cuda.synchronize(device)                     # whatever the sampler reads (resident dataset tensors) is complete
        self.nxt = self._produce()

    def _produce(self):
        with torch.cuda.stream(self.side):
            b = self.make()
            return b, self.side.record_event()

    def get(self):
        b, ready = self.nxt
        cur = torch.cuda.current_stream(self.side.device)
        cur.wait_event(ready)
        for t in batch_tensors(b):
            t.record_stream(cur)
        self.nxt = self._produce()
        return b


def batch_signature(batch):
    """Shapes and dtypes of the batch tensors, followed by the batch's `graph_constants()` — host values a capture bakes into
    kernel arguments (e.g. PairBatch.NA / NB)."""
    consts = tuple(batch.graph_constants()) if hasattr(batch, "graph_constants") else ()
    return tuple((tuple(t.shape), t.dtype) for t in batch_tensors(batch)) + (("const",) + consts,)


def batch_signature_constants(batch):
    return ("const",) + (tuple(batch.graph_constants()) if hasattr(batch, "graph_constants") else ())


_capture_serial = 0
_active_capture = None


def active_capture():
    """Serial number of the GraphedStep capture in progress on this thread of control, else None.  Values derived from a
    batch tensor while THAT capture records (e.g. the per-mesh 1/count of a mask) are part of its graph — recomputed by every
    replay — and may be shared by the blocks of the same capture; they must never outlive it (blocks.avg_block)."""
    return _active_capture


class GraphedStep:
    """`body(batch) -> loss` (forward, loss, backward into pre-existing .grad buffers) captured in a hipGraph.

    `example` supplies shapes and becomes the static batch (its tensors are the graph's inputs; `load` overwrites them).
    `zero_grads()` is recorded at the head of the graph, so a replay leaves exactly this step's gradients in `.grad`.
    """

    def __init__(self, body: Callable, example, zero_grads: Callable[[], None], warmup: int = 2,
                 preserve: Optional[List[torch.Tensor]] = None):
        """`preserve`: tensors the body updates in place (BatchNorm running statistics) — restored after the eager
        warm-up runs so that capturing leaves the model state untouched."""
        if not batch_tensors(example)[0].is_cuda:
            raise RuntimeError("GraphedStep needs a GPU batch (hipGraph capture)")
        self.static = example
        self._ring = []                                    # frozen ring-vs-row-blocked choices of the example's operators
        with _RingChoices("record", self._ring):
            self._static_tensors = batch_tensors(example)
        with _RingChoices("apply", self._ring):
            self.signature = batch_signature(example)
        self._body, self._zero = body, zero_grads
        # eager warm-up on a side stream (lazy conversions, allocator, autotuned library kernels), then capture
        saved = [t.clone() for t in (preserve or [])]
        # AccumulateGrad nodes created by earlier eager steps live on the default stream; the warm-up below runs on a
        # side stream on purpose, so the (harmless) stream-mismatch warning is silenced for its duration
        warn_ctl = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if warn_ctl is not None:
            warn_ctl(False)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for t, s0 in zip(preserve or [], saved):
                t.copy_(s0)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        global _capture_serial, _active_capture
        _capture_serial += 1
        _active_capture = _capture_serial
        try:
            with torch.cuda.graph(self.graph):
                self.loss = self._run()
        finally:
            _active_capture = None
        torch.cuda.synchronize()
        if warn_ctl is not None:
            warn_ctl(True)

    def _run(self):
        self._zero()
        return self._body(self.static)

    def matches(self, batch) -> bool:
        try:
            with _RingChoices("apply", self._ring):
                return batch_signature(batch) == self.signature
        except (StopIteration, _SignatureMismatch):        # more Laplacian-type operators than the example had / unsorted rows
            return False

    def load(self, batch) -> None:
        """Copy a freshly sampled batch into the static buffers (device-to-device, on the current stream)."""
        try:
            with _RingChoices("apply", self._ring):
                src = batch_tensors(batch)
        except (StopIteration, _SignatureMismatch):
            raise ValueError("batch does not match the captured signature; capture a new GraphedStep") from None
        if len(src) != len(self._static_tensors) or any(
                s.shape != d.shape or s.dtype != d.dtype for s, d in zip(src, self._static_tensors)):
            raise ValueError("batch does not match the captured signature; capture a new GraphedStep")
        if self.signature[-1] != batch_signature_constants(batch):
            raise ValueError(f"batch constants {batch_signature_constants(batch)[1:]} differ from the captured ones "
                             f"{self.signature[-1][1:]}; capture a new GraphedStep")
        with torch.no_grad():
            # one multi-tensor launch per dtype: a mixed list (fp32 values, int32 indices, int64 tables) would take the
            # op's slow path, one device-to-device copy per tensor (37 copies of ~10 us for a FAUST pair)
            groups = {}
            for s_, d_ in zip(src, self._static_tensors):
                if d_.numel():
                    g = groups.setdefault(d_.dtype, ([], []))
                    g[0].append(d_)
                    g[1].append(s_)
            for dst, srcs in groups.values():
                torch._foreach_copy_(dst, srcs)

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        return self.loss

    def __call__(self, batch) -> torch.Tensor:
        self.load(batch)
        return self.replay()



class GraphedTrainStep:
    """A training step whose forward + loss + backward is one hipGraph replay; the gradient all-reduce and the optimizer
    stay eager.  `loss_of(model, batch) -> loss`; `example` fixes the batch signature and becomes the static batch.

    Gradients are STORED, not accumulated: every parameter's `.grad` is None while the step is captured, so autograd
    assigns the freshly computed gradient tensors (static buffers of the graph's memory pool) instead of adding them into
    pre-existing ones — no zeroing pass and no add per parameter in the replay (124 small launches for the ARAP models).
    A replay rewrites those buffers in place; `.grad` is re-pointed at them after every replay because a gradient
    reduction (`FlatGradBucket.sync`) may have re-pointed it at the bucket slices."""

    def __init__(self, model, optimizer, example, loss_of: Callable, bucket=None):
        self.optimizer = optimizer
        self.params = [p for p in model.parameters() if p.requires_grad]

        def drop():
            for p in self.params:
                p.grad = None

        def body(b):
            loss = loss_of(model, b)
            loss.backward()
            kernels.clear_absmax()
            return loss

        # (the `zero_grads` slot of GraphedStep runs before the body, at warm-up and at capture time: host code, not recorded)
        self.step = GraphedStep(body, example, drop, preserve=list(model.buffers()))
        self.grads = [p.grad for p in self.params]      # static; None for a parameter the loss does not reach
        # A replay leaves its gradients in the graph's own buffers, NOT in a FlatGradBucket's slices: the only reduction that
        # sees them is one that packs `.grad` first (`bucket.sync`).  With `bucket` given the step runs that itself.
        self.bucket = bucket

    def matches(self, batch) -> bool:
        return self.step.matches(batch)

    def __call__(self, batch, grad_sync=None):
        if grad_sync is not None and getattr(grad_sync, "__func__", None) is getattr(type(self.bucket), "all_reduce", False):
            # (the pre-round-2 pattern `graphed(batch, grad_sync=bucket.all_reduce)` would reduce a buffer the replay never
            # wrote and every rank would then step on its local gradients only)
            raise ValueError("GraphedTrainStep stores its gradients outside the bucket: pass grad_sync=bucket.sync "
                             "(or nothing: a step built with bucket= reduces by itself)")
        loss = self.step(batch)
        for p, g in zip(self.params, self.grads):
            p.grad = g
        if grad_sync is None and self.bucket is not None:
            grad_sync = self.bucket.sync
        if grad_sync is not None:
            grad_sync()
        self.optimizer.step()
        return loss
