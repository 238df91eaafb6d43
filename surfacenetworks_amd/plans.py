"""Launch plans: a whole residual block enqueued by ONE host call (csrc/sn_plan.hip, include/sn_spmm.h "launch plans").

The reference's training loops (src/as_rigid_as_possible/main.py:217-232, src/mesh_mnist/main.py:151-167,
src/dense_correspondence/main.py:310-327) launch every op from the interpreter and cannot be graph-captured by an unmodified
driver; with this package's kernels a block is 10-25 launches of 5-400 us and the host cost of a launch issued from Python
(~22 us: ctypes marshalling, one or more torch.empty per output, the Function bodies) bounds every small-batch configuration.

What happens here, per block call site and per distinct shape signature:
  * the FIRST call runs the block's ordinary host code (blocks.py / functional.py / kernels.py) as a DRY RUN: `_lib.call` records
    (entry point, arguments) instead of launching, a TorchDispatchMode notes every allocation the code makes and refuses any
    torch op that computes (such a block is not plannable and keeps the eager path);
  * every recorded pointer is expressed as (slot, offset): slots 0 / 1 are the block's two workspace ARENAS (all allocations of
    the dry run laid out back to back; large ones in the first, parameter-sized ones in the second, so that a parameter gradient
    kept by the optimizer does not pin hundreds of MB), the rest are the tensors the block was given (features, weights, operator
    arrays, gradients);
  * the list goes into a C plan (sn_plan_add_*); this and every later call allocate the two arenas (two torch.empty instead of
    ~40), hand the slots' base addresses to sn_plan_run — which calls the same launchers in the same order — and rebuild only
    the block's OUTPUTS as views of the arenas.  What the backward needs stays a description (offsets into the forward arenas,
    which autograd keeps alive) until the backward's own plan runs.
  * a plan run whose slot addresses have come back several times (a training loop's caching allocator repeats them from step to
    step) is captured once at those addresses and from then on enqueued as ONE graph launch of the same kernels
    (sn_plan_instantiate / sn_plan_exec_launch; small plans only: GRAPH_MAX_ARENA_BYTES).
Same kernels, same order, same arguments: results are bit-identical to the eager path (tests/test_plans_gpu.py).

SN_PLANS=0 disables the mechanism (every block launches its kernels from Python, as before); SN_PLAN_GRAPHS=0 only the graph
launches (every plan run walks its list)."""
from __future__ import annotations

import bisect
import collections
import ctypes as C
import os
import threading
from typing import List, Optional, Sequence

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import _lib, kernels

__all__ = ["enabled", "set_enabled", "set_graphs", "graph_stats", "Site", "stats", "reset", "PlanError"]

_ENABLED = os.environ.get("SN_PLANS", "1") != "0"
# A plan run whose slot addresses were seen before is enqueued as ONE graph launch (sn_plan_instantiate: csrc/sn_plan.hip says why the
# addresses repeat and what it saves).  SN_PLAN_GRAPHS=0: every run walks its launch list.
_GRAPHS = os.environ.get("SN_PLAN_GRAPHS", "1") != "0"
MAX_EXECS_PER_PLAN = 256      # address sets kept per plan: a plan shared by the middle blocks of a model runs on one set per block, tower and
                              # batch of a cycling loader (the FAUST loop behind the reference's names: 56-80 per plan)
# A graph launch costs the DEVICE ~7 us more than the same launches issued one by one (measured: +0.23 ms on the 32 plan runs of the
# 64-mesh step, which the device bounds) and saves the HOST ~30 us.  It pays where the host is the bound, i.e. where a block direction
# is shorter on the device than its ~100 us of host time: plans whose workspace is below this size (a Dirac block direction runs ~1 us
# per MiB of workspace; ARAP at 4 meshes, 110 MiB, is the crossover: 5.3 -> 4.7 ms per step on a slow host, 3.64 -> 3.82 on a fast one).
GRAPH_MAX_ARENA_BYTES = 64 << 20
_graveyard = collections.deque()  # (event, handle, lib) of graphs dropped while a launch of theirs may still be in flight


def _bury(lib, handle) -> None:
    try:
        ev = torch.cuda.Event()
        ev.record()
    except Exception:  # noqa: BLE001 — no device any more (interpreter shutdown)
        ev = None
    _graveyard.append((ev, handle, lib))
    while len(_graveyard) > 32:
        ev0, h0, lib0 = _graveyard.popleft()
        if ev0 is not None:
            ev0.synchronize()
        lib0.sn_plan_exec_destroy(h0)
_NEVER = object()
# Instantiating a graph costs 270-540 us (6-9 kernel nodes), a graph launch saves ~30: an address set pays for its graph after 9-18
# more runs.  The classic rent-or-buy answer: walk the list until the set has come back about that often, then buy (a loop that cycles
# few sets gets its graphs within a dozen steps; address sets that do not keep coming back never cost an instantiation).
GRAPH_AFTER_SIGHTINGS = 12
_graph_counts = {"instantiated": 0, "launched": 0, "refused": 0, "evicted": 0}


def graph_stats():
    return dict(_graph_counts)


def set_graphs(on: bool) -> None:
    global _GRAPHS
    _GRAPHS = bool(on)
_BIG_BYTES = 1 << 20          # allocations from this size on live in the first arena
_ALIGN = 256
_ALLOW_CPU = False            # tests: record (never run) plans on CPU tensors

K_INT, K_DOUBLE, K_PTR, K_NULL, K_STREAM = 0, 1, 2, 3, 4


class PlanError(RuntimeError):
    """The dry run met something a plan cannot express (the block then keeps its eager path)."""


def enabled() -> bool:
    return _ENABLED


def set_enabled(on: bool) -> None:
    global _ENABLED
    _ENABLED = bool(on)


# ---- descriptions of tensors relative to slots --------------------------------------------------------------------------------
class _Desc:
    __slots__ = ("slot", "off", "eoff", "shape", "stride", "dtype", "attrs", "exact")

    def __init__(self, slot, off, shape, stride, dtype, exact=False):
        self.slot, self.off, self.shape, self.stride, self.dtype = slot, off, shape, stride, dtype
        self.eoff = off // torch.empty((), dtype=dtype).element_size()      # offset in elements of its own type
        self.attrs = None         # [(name, _Desc)]: `_sn_*` tensor attributes the block left on this output (activated hand-offs)
        self.exact = exact        # an ext slot returned as it came


def _span_bytes(t: torch.Tensor) -> int:
    if t.numel() == 0:
        return 0
    return (sum((s - 1) * st for s, st in zip(t.shape, t.stride())) + 1) * t.element_size()


_ALLOC_OPS = {"aten::empty", "aten::empty_strided", "aten::empty_like", "aten::new_empty", "aten::new_empty_strided"}
_ZERO_ALLOC_OPS = {"aten::zeros", "aten::zeros_like", "aten::new_zeros"}
_HARMLESS_OPS = {"aten::detach", "aten::alias", "aten::_unsafe_view", "aten::lift_fresh", "aten::detach_"}


class _Recorder(TorchDispatchMode):
    """Dry run of a block's host code: launches recorded, allocations noted, computing torch ops refused."""

    def __init__(self, ext: Sequence[Optional[torch.Tensor]]):
        super().__init__()
        self.ext = list(ext)
        self.nodes: List[tuple] = []          # ("call", name, args) | ("memset", tensor, byte) | ("copy", dst, src)
        self.allocs: List[torch.Tensor] = []  # kept alive until the layout is done: no address is handed out twice
        self.stray: List[str] = []
        self.notes: List[tuple] = []          # (tensor, maxima) recorded by kernels.note_absmax and not taken again
        self.allow_cpu = _ALLOW_CPU
        self.tags: List[tuple] = []           # (tag, entry count | None) of every sparse product, in launch order (SpmmTimer)

    # -- launches --------------------------------------------------------------------------------------------------------------
    def record_call(self, name, args):
        self.nodes.append(("call", name, args))

    # -- torch ops -------------------------------------------------------------------------------------------------------------
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        if name in _ALLOC_OPS:
            self.allocs.append(out)
        elif name in _ZERO_ALLOC_OPS:
            self.allocs.append(out)
            self.nodes.append(("memset", out, 0))
        elif getattr(func, "is_view", False) or name in _HARMLESS_OPS:
            pass
        elif name == "aten::zero_":
            self.nodes.append(("memset", args[0], 0))
        elif name == "aten::fill_" and not isinstance(args[1], torch.Tensor) and float(args[1]) == 0.0:
            self.nodes.append(("memset", args[0], 0))
        elif name == "aten::copy_" and isinstance(args[1], torch.Tensor) and args[0].dtype == args[1].dtype and \
                args[0].shape == args[1].shape and args[0].device == args[1].device:
            self.nodes.append(("copy", args[0], args[1]))
        else:
            self.stray.append(name)
        return out


class _Layout:
    """Arena offsets of the dry run's allocations and the (slot, offset) of any pointer into them or into an ext tensor."""

    def __init__(self, rec: _Recorder, alias_ok=()):
        regions = {}
        for t in rec.allocs:
            st = t.untyped_storage()
            n = int(st.nbytes())
            if n and st.data_ptr() not in regions:
                regions[st.data_ptr()] = n
        self.starts = sorted(regions)
        self.sizes = [regions[p] for p in self.starts]
        self.where = []
        tot = [0, 0]
        for n in self.sizes:
            a = 0 if n >= _BIG_BYTES else 1
            self.where.append((a, tot[a]))
            tot[a] += (n + _ALIGN - 1) // _ALIGN * _ALIGN        # (2 MiB alignment of the large ones measured: no effect on the step)
        self.arena_bytes = tuple(tot)
        self.ext = []
        for j, t in enumerate(rec.ext):
            if t is not None and t.numel():
                self.ext.append((t.data_ptr(), _span_bytes(t), j))
        # A pointer is attributed to the FIRST operand whose range holds it: two operands that overlap in memory during the dry
        # run and do not on a later call (or the other way round in a way the key does not see) would make the plan read the
        # wrong one.  Same first byte among the positions the caller keys the sharing pattern of (`alias_ok`): fine; any other
        # overlap: no plan for this signature.
        by_addr = sorted(self.ext)
        ok = set(alias_ok)
        for (p0, n0, j0), (p1, n1, j1) in zip(by_addr, by_addr[1:]):
            if p1 < p0 + n0 and not (p1 == p0 and j0 in ok and j1 in ok):
                raise PlanError("two operands of the block overlap in memory")
        self.used_ext = set()

    def resolve(self, p: int):
        i = bisect.bisect_right(self.starts, p) - 1
        if i >= 0 and p < self.starts[i] + self.sizes[i]:
            a, off = self.where[i]
            return a, off + (p - self.starts[i])
        for base, span, j in self.ext:
            if base <= p < base + span:
                self.used_ext.add(j)
                return 2 + j, p - base
        return None

    def describe(self, t: torch.Tensor, rec: _Recorder) -> _Desc:
        if t.numel() == 0:
            raise PlanError("an empty tensor among the block's results")
        for j, e in enumerate(rec.ext):
            if e is t:
                self.used_ext.add(j)
                return _Desc(2 + j, 0, tuple(t.shape), tuple(t.stride()), t.dtype, exact=True)
        hit = self.resolve(t.data_ptr())
        if hit is None:
            raise PlanError("a result of the block lives in memory the plan does not know (neither its arenas nor its operands)")
        if hit[1] % t.element_size():
            raise PlanError("misaligned result")
        return _Desc(hit[0], hit[1], tuple(t.shape), tuple(t.stride()), t.dtype)


def _attr_tensors(t: torch.Tensor):
    """(name, tensor) of the `_sn_*` tensor attributes riding on a tensor object (statistics partials, tile sums, batch counter)."""
    d = getattr(t, "__dict__", None)
    if not d:
        return []
    return [(k, v) for k, v in d.items() if k.startswith("_sn_") and isinstance(v, torch.Tensor)]


def _map_structure(x, fn):
    if isinstance(x, torch.Tensor):
        return fn(x)
    if isinstance(x, (tuple, list)):
        return tuple(_map_structure(v, fn) for v in x)
    if x is None or isinstance(x, (bool, int, float, str)):
        return x
    raise PlanError(f"a block result of type {type(x).__name__} cannot be described")


class Plan:
    """The C launch list of one block direction and the descriptions of what it leaves."""

    _n_alive = 0

    def __init__(self, rec: _Recorder, result, device, alias_ok=()):
        if rec.stray:
            raise PlanError("torch ops that compute inside the block: " + ", ".join(sorted(set(rec.stray))))
        lay = _Layout(rec, alias_ok)
        lib = _lib.load()
        handle = C.c_void_p()
        _lib.check(lib.sn_plan_create(C.byref(handle)), "sn_plan_create")
        self.handle = handle
        self._lib = lib
        Plan._n_alive += 1
        self.device = device
        self.launches = 0
        for node in rec.nodes:
            if node[0] == "call":
                self._add_call(node[1], node[2], lay)
            elif node[0] == "memset":
                self._add_memset(node[1], node[2], lay)
            else:
                self._add_copy(node[1], node[2], lay)
        self.result = _map_structure(result, lambda t: self._describe(t, lay, rec))
        self.effects = [(lay.describe(t, rec), lay.describe(m, rec)) for t, m in rec.notes]
        # what a per-launch timer (functional.SpmmTimer) wants to know of this plan's sparse products; None: one of them has no
        # entry count on the host (the plan then steps aside while a timer runs)
        self.tags = list(rec.tags) if all(k is not None for _, k in rec.tags) else None
        self.arena_bytes = lay.arena_bytes
        self.n_ext = len(rec.ext)
        self.used_ext = sorted(lay.used_ext)
        self._BasesT = C.c_uint64 * (2 + self.n_ext)
        self.bwd = {}                       # backward plans recorded against this forward's layout, by their own key
        self.execs = collections.OrderedDict()   # slot addresses (the bytes of the slot table) -> sightings (int) | graph handle | _NEVER
        self.graphable = 0 < sum(lay.arena_bytes) <= GRAPH_MAX_ARENA_BYTES and self.launches >= 2

    def _describe(self, t, lay, rec):
        d = lay.describe(t, rec)
        attrs = _attr_tensors(t)
        if attrs and not d.exact:
            d.attrs = [(k, lay.describe(v, rec)) for k, v in attrs]
        return d

    @staticmethod
    def _2d(t: torch.Tensor):
        """(pitch, width, rows) in bytes of a tensor a fill / copy node can address, or None."""
        es = t.element_size()
        if t.is_contiguous():
            return t.numel() * es, t.numel() * es, 1
        if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
            return t.stride(0) * es, t.shape[1] * es, t.shape[0]
        return None

    def _slot_of(self, t, lay):
        hit = lay.resolve(t.data_ptr())
        if hit is None:
            raise PlanError("a fill / copy touches memory the plan does not know")
        return hit

    def _add_memset(self, t, byte, lay):
        if t.numel() == 0:
            return
        g = self._2d(t)
        if g is None:
            raise PlanError("fill of a tensor that is neither contiguous nor a 2-D row-strided view")
        slot, off = self._slot_of(t, lay)
        _lib.check(self._lib.sn_plan_add_memset(self.handle, slot, off, byte, g[0], g[1], g[2]), "sn_plan_add_memset")

    def _add_copy(self, dst, src, lay):
        if dst.numel() == 0:
            return
        gd, gs = self._2d(dst), self._2d(src)
        if gd is None or gs is None or (gd[2] != gs[2] and not (gd[2] == 1 and gs[2] == 1)) or gd[1] != gs[1]:
            # (a contiguous side against a row-strided one of the same shape: address both by rows)
            if gd is not None and gs is not None and dst.dim() == 2 and src.dim() == 2:
                es = dst.element_size()
                gd = (dst.stride(0) * es, dst.shape[1] * es, dst.shape[0])
                gs = (src.stride(0) * es, src.shape[1] * es, src.shape[0])
            else:
                raise PlanError("copy between tensors a 2-D copy cannot address")
        ds, do = self._slot_of(dst, lay)
        ss, so = self._slot_of(src, lay)
        _lib.check(self._lib.sn_plan_add_copy(self.handle, ds, do, gd[0], ss, so, gs[0], gd[1], gd[2]), "sn_plan_add_copy")

    def _add_call(self, name, args, lay):
        fn = int(self._lib.sn_plan_lookup(name.encode()))
        if fn < 0:
            raise PlanError(f"{name} is not an entry point a plan can call")
        argtypes = _lib.SIGNATURES[name][1]
        n = len(argtypes)
        if len(args) != n:
            raise PlanError(f"{name}: {len(args)} arguments recorded, {n} declared")
        kind = (C.c_int32 * n)()
        slot = (C.c_int32 * n)()
        ival = (C.c_int64 * n)()
        dval = (C.c_double * n)()
        for i, (a, ty) in enumerate(zip(args, argtypes)):
            if ty is C.c_void_p:
                if i == n - 1:
                    kind[i] = K_STREAM
                elif a is None or a == 0:
                    kind[i] = K_NULL
                else:
                    hit = lay.resolve(int(a))
                    if hit is None:
                        raise PlanError(f"{name}: argument {i} points into memory the plan does not know")
                    kind[i], slot[i], ival[i] = K_PTR, hit[0], hit[1]
            elif ty is C.c_double or ty is C.c_float:
                kind[i], dval[i] = K_DOUBLE, float(a)
            else:
                kind[i], ival[i] = K_INT, int(a)
        _lib.check(self._lib.sn_plan_add_call(self.handle, fn, n, kind, slot, ival, dval), f"sn_plan_add_call({name})")
        self.launches += 1

    def __del__(self):
        try:
            for x in self.execs.values():
                if x.__class__ is not int and x is not _NEVER:
                    _bury(self._lib, x)
            self.execs.clear()
            if self.handle:
                self._lib.sn_plan_destroy(self.handle)
                self.handle = None
                Plan._n_alive -= 1
        except Exception:  # noqa: BLE001 — interpreter shutdown
            pass

    # -- per call ---------------------------------------------------------------------------------------------------------------
    def new_arenas(self, dev=None):
        dev = self.device if dev is None else dev
        big = torch.empty(self.arena_bytes[0] // 4, dtype=torch.float32, device=dev) if self.arena_bytes[0] else None
        small = torch.empty(self.arena_bytes[1] // 4, dtype=torch.float32, device=dev) if self.arena_bytes[1] else None
        return big, small

    def run(self, big, small, ext):
        """Enqueue the plan on torch's current stream.  (The slot table is made per call: a plan may be run from several threads —
        the autograd engine's device thread runs backward plans while the caller's thread runs forward ones.)"""
        b = self._BasesT()
        b[0] = big.data_ptr() if big is not None else 0
        b[1] = small.data_ptr() if small is not None else 0
        for j in self.used_ext:
            b[2 + j] = ext[j].data_ptr()
        self._launch(b)

    def _launch(self, b):
        failed = C.c_int32(-1)
        if _GRAPHS and self.graphable:
            k = bytes(b)
            execs = self.execs
            x = execs.get(k)
            if x is None:
                execs[k] = 1                                     # first sighting: remember the addresses, walk the list
                if len(execs) > MAX_EXECS_PER_PLAN:
                    old = execs.popitem(last=False)[1]
                    if old.__class__ is not int and old is not _NEVER:
                        _bury(self._lib, old)
                        _graph_counts["evicted"] += 1
            elif x is not _NEVER:
                execs.move_to_end(k)
                if x.__class__ is int:
                    if x + 1 < GRAPH_AFTER_SIGHTINGS:
                        execs[k] = x + 1
                    else:
                        x = execs[k] = self._instantiate(b, x)   # these addresses keep coming back
                if x.__class__ is not int and x is not _NEVER:
                    st = self._lib.sn_plan_exec_launch(x, self.handle, b, 2 + self.n_ext, kernels._stream(), C.byref(failed))
                    if st == 0:
                        _graph_counts["launched"] += 1
                        return
                    if failed.value >= 0:                        # (it walked the list — a capture of the caller's — and an entry failed)
                        _lib.check(st, f"sn_plan_exec_launch (entry {failed.value})")
                    execs[k] = _NEVER                            # the runtime would not launch the graph: these addresses walk the list
                    _bury(self._lib, x)
                    _graph_counts["refused"] += 1
        st = self._lib.sn_plan_run(self.handle, b, 2 + self.n_ext, kernels._stream(), C.byref(failed))
        if st != 0:
            _lib.check(st, f"sn_plan_run (entry {failed.value})")

    def _instantiate(self, b, sightings):
        """The graph of this plan at the addresses `b`; `sightings` back when a capture of the caller's is in progress on this thread
        (not now); _NEVER when the per-launch timer is on or the runtime refuses (these addresses then always walk the list)."""
        if torch.cuda.is_current_stream_capturing():
            return sightings
        x = C.c_void_p()
        failed = C.c_int32(-1)
        st = self._lib.sn_plan_instantiate(self.handle, b, 2 + self.n_ext, C.byref(x), C.byref(failed))
        if st != 0 or not x:
            _graph_counts["refused"] += 1
            return _NEVER
        _graph_counts["instantiated"] += 1
        return x


class _Builder:
    """Tensors from descriptions: views of the arenas / of the operands of one run (one torch.as_strided per tensor)."""

    __slots__ = ("big", "small", "ext", "typed")

    def __init__(self, big, small, ext):
        self.big, self.small, self.ext = big, small, ext
        self.typed = None

    def __call__(self, d):
        c = d.__class__
        if c is _Desc:
            return self.tensor(d)
        if c is tuple:
            return tuple([self(v) for v in d])
        return d

    def tensor(self, d):
        if d.slot >= 2:
            e = self.ext[d.slot - 2]
            if d.exact:
                return e
            if e.dtype != d.dtype:
                raise PlanError("a result aliases an operand of another dtype")
            t = torch.as_strided(e, d.shape, d.stride, e.storage_offset() + d.off // e.element_size())
        else:
            base = self.small if d.slot else self.big
            if d.dtype is not torch.float32:
                if self.typed is None:
                    self.typed = {}
                key = (d.slot, d.dtype)
                tb = self.typed.get(key)
                if tb is None:
                    tb = self.typed[key] = base.view(d.dtype)
                base = tb
            t = torch.as_strided(base, d.shape, d.stride, d.eoff)
        if d.attrs:
            for k, dv in d.attrs:
                setattr(t, k, self.tensor(dv))
        return t


# ---- call sites ------------------------------------------------------------------------------------------------------------------
_SITES = []


MAX_PLANS_PER_SITE = 64       # distinct shape signatures kept per call site (least recently used beyond that)
THRASH_WINDOW = 16            # a site whose last THRASH_WINDOW lookups were (nearly) all new signatures ...
THRASH_MISSES = 12
COOLDOWN_CALLS = 256          # ... stops recording for this many calls (known signatures still replay; the rest run eagerly)


class Site:
    """One block call site (e.g. the forward of the Dirac block): its plans by shape signature.

    Recording a plan costs a dry run of the block's host code (~1 ms) and every plan keeps a launch list in host memory: a
    caller whose shapes change every step — random ragged batches — must not pay that per step nor grow the table without
    bound.  Hence: at most MAX_PLANS_PER_SITE signatures per site (least recently used dropped), and a site that keeps
    missing (THRASH_MISSES of the last THRASH_WINDOW lookups) leaves new signatures to the eager path for COOLDOWN_CALLS calls."""

    def __init__(self, name: str):
        self.name = name
        self.plans = collections.OrderedDict()
        self.recorded = self.replayed = self.refused = self.skipped = 0
        self.reasons = {}
        self.recent = collections.deque(maxlen=THRASH_WINDOW)
        self.cooldown = 0
        _SITES.append(self)

    def lookup(self, key):
        """(plan | None, may_record): the plan of `key` if there is one (None also for a signature that was refused before);
        may_record says whether a missing one should be recorded now."""
        plan = self.plans.get(key, self)
        if plan is not self:
            self.recent.append(0)
            if plan is not None:
                self.plans.move_to_end(key)
            return plan, False
        self.recent.append(1)
        if self.cooldown > 0:
            self.cooldown -= 1
            self.skipped += 1
            return None, False
        if len(self.recent) == THRASH_WINDOW and sum(self.recent) >= THRASH_MISSES:
            self.cooldown = COOLDOWN_CALLS
            self.recent.clear()
            self.skipped += 1
            return None, False
        return None, True

    def store(self, key, plan) -> None:
        self.plans[key] = plan
        while len(self.plans) > MAX_PLANS_PER_SITE:
            self.plans.popitem(last=False)

    def refuse(self, key, why: str):
        self.store(key, None)
        self.refused += 1
        self.reasons[why] = self.reasons.get(why, 0) + 1
        if os.environ.get("SN_STRICT", "0") == "1":
            raise PlanError(f"{self.name}: {why}")


def stats():
    """{site: {"plans", "recorded", "replayed", "refused", "reasons"}} — bench.py / tests read it."""
    return {s.name: {"plans": sum(1 for p in s.plans.values() if p is not None), "recorded": s.recorded, "replayed": s.replayed,
                     "refused": s.refused, "skipped": s.skipped, "reasons": dict(s.reasons)} for s in _SITES}


def reset() -> None:
    for k in _graph_counts:
        _graph_counts[k] = 0
    for s in _SITES:
        s.plans.clear()
        s.recorded = s.replayed = s.refused = s.skipped = 0
        s.reasons.clear()
        s.recent.clear()
        s.cooldown = 0


def usable(*tensors) -> bool:
    """Plans apply: switched on, no synchronised BatchNorm (its collectives sit between the launches), operands on the GPU."""
    if not _ENABLED:
        return False
    from .functional import _BN_SYNC

    if _BN_SYNC is not None:
        return False
    for t in tensors:
        if t is not None:
            return t.is_cuda or _ALLOW_CPU
    return False


def expand_ext(tensors: Sequence[Optional[torch.Tensor]], scan: Sequence[int]):
    """(ext list, key part): the tensors themselves followed by the `_sn_*` tensor attributes riding on those at the positions
    `scan` (the block's features — statistics partials / tile sums of an activated hand-off — and the running-mean buffers —
    the batch counter of this call)."""
    ext = list(tensors)
    names = []
    for i in scan:
        t = tensors[i]
        if t is not None:
            d = t.__dict__
            if d:
                for k, v in d.items():
                    if k.startswith("_sn_") and isinstance(v, torch.Tensor):
                        ext.append(v)
                        names.append((i, k, v.shape))
    return ext, tuple(names)


def _dry_run(impl, args, ext):
    rec = _Recorder(ext)
    keep = [(t, dict(t.__dict__)) for t in ext if t is not None and hasattr(t, "__dict__")]
    _lib._record_lock.acquire()                    # one dry run at a time; calls of OTHER threads keep launching (_lib.recorder)
    prev, prev_tid = _lib._recorder, _lib._recorder_tid
    _lib._recorder_tid = threading.get_ident()
    _lib._recorder = rec
    try:
        with torch.no_grad(), rec:
            result = impl(*args)
    finally:
        _lib._recorder, _lib._recorder_tid = prev, prev_tid
        _lib._record_lock.release()
        for t, m in rec.notes:                      # bounds noted on the dry run's scratch tensors: the replay notes the real ones
            ent = kernels._absmax_table.get(t.data_ptr())
            if ent is not None and ent[2] is m:
                del kernels._absmax_table[t.data_ptr()]
        for t, before in keep:                      # attributes the dry run hung on (or took from) the operands: as they were
            t.__dict__.clear()
            t.__dict__.update(before)
    return rec, result


def record(site: Site, key, impl, args, ext, device, alias_ok=()) -> Optional[Plan]:
    """Dry-run `impl(*args)` and turn what it would have launched into a plan (None, and the reason noted, when it cannot be).
    alias_ok: positions of `ext` that may start at the same address because the caller's key holds their sharing pattern."""
    try:
        rec, result = _dry_run(impl, args, ext)
        plan = Plan(rec, result, device, alias_ok)
    except PlanError as exc:
        site.refuse(key, str(exc)[:200])
        return None
    site.store(key, plan)
    site.recorded += 1
    return plan


def renote(grads, maxima) -> None:
    for g, m in zip(grads, maxima):
        if g is not None and m is not None:
            kernels.note_absmax(g, m)


def apply_effects(plan: Plan, build: _Builder) -> None:
    for dt, dm in plan.effects:
        kernels.note_absmax(build(dt), build(dm))
