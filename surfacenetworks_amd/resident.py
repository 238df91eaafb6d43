"""The fast path behind the reference's own batching names (src/utils/utils_pt.py:21-69).

An unmodified reference driver builds its batch operators on the host every step:

    Di.append(utils.sp_sparse_to_pt_sparse(sequences[ind][t]['Di']))          src/as_rigid_as_possible/main.py:161-162
    Di = utils.sparse_diag_cat(Di, 4 * num_faces, 4 * num_vertices)           src/as_rigid_as_possible/main.py:180-181
    ... (Di).cuda()                                                           src/as_rigid_as_possible/main.py:184
    Di = utils.sparse_cat(Di, ...); Variable(Di).cuda()                       src/mesh_mnist/main.py:109-115
    utils.sparse_diag_cat(laplacian, 7000, 7000).coalesce() ... .cuda()       src/dense_correspondence/main.py:180-190

i.e. scipy -> COO index arithmetic -> concatenation -> coalesce() sort -> a 24-byte-per-entry host-to-device copy (3.2 s per
step at BASELINE configs[2]).  Here the same calls return `LazySparse` tensors: torch.Tensor subclasses with the reference's
shape / dtype / layout / device that carry WHAT they stand for instead of index arrays —

    sp_sparse_to_pt_sparse(A)   -> the scipy matrix itself;
    sparse_diag_cat / sparse_cat -> the member list and block sizes, and (when a GPU is present) the batch operator already
                                    assembled ON THE DEVICE from a resident copy of every member: `ResidentCache` keeps each
                                    source matrix it has seen (keyed on the scipy object; the drivers pass the same dataset
                                    objects every step) packed in HBM — CSR for Laplacians, quaternion records for Dirac
                                    operators, with transposes — inside growing `OperatorPool`s, so a batch is the same single
                                    offset-concatenation launch as the product's own samplers use;
    .coalesce() / .cuda()       -> the same handle (already sorted; already resident).

The residual blocks pick the assembled `SparseOperator` up through `operators.as_operator` (`_sn_operator`).  ANY other use
— `.to_dense()`, `._indices()`, `torch.mm(A, x)`, arithmetic — materialises the tensor exactly as the reference's code would
have built it (host COO arithmetic, coalesce) and proceeds on the real tensor: nothing the reference's API promises is lost,
only deferred.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional, Sequence

import numpy as np
import torch
import torch.utils._pytree as pytree

__all__ = ["LazySparse", "ResidentCache", "resident_cache", "reference_sp_to_coo", "reference_diag_cat", "reference_cat"]


# ---- the reference's own constructions (what a LazySparse materialises to) ------------------------------------------------
def reference_sp_to_coo(L):
    """scipy sparse matrix -> torch sparse COO (uncoalesced, dtype kept), as utils_pt.py:56-69."""
    coo = L.tocoo()
    index = torch.from_numpy(np.stack([coo.row, coo.col]).astype(np.int64))
    return torch.sparse_coo_tensor(index, torch.from_numpy(coo.data), torch.Size(coo.shape))


def reference_diag_cat(tensors, size0, size1):
    """Block-diagonal (len*size0, len*size1) operator from per-mesh COO operators, as utils_pt.py:41-53."""
    shift = torch.tensor([[size0], [size1]], dtype=torch.int64)
    index = torch.cat([t._indices() + i * shift for i, t in enumerate(tensors)], dim=1)
    values = torch.cat([t._values() for t in tensors], dim=0)
    n = len(tensors)
    return torch.sparse_coo_tensor(index, values, torch.Size((n * size0, n * size1))).coalesce()


def reference_cat(tensors, size0, size1):
    """3-D batched (len, size0, size1) COO operator, as utils_pt.py:21-39."""
    index = torch.cat([torch.cat([torch.full((1, t._nnz()), i, dtype=torch.int64), t._indices()], dim=0)
                       for i, t in enumerate(tensors)], dim=1)
    values = torch.cat([t._values() for t in tensors], dim=0)
    return torch.sparse_coo_tensor(index, values, torch.Size((len(tensors), size0, size1))).coalesce()


_TORCH_DTYPE = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.float16): torch.float16,
                np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64}


class _Payload:
    """What a LazySparse stands for; shared by the handles `.coalesce()`, `.cuda()`, `.detach()` hand out."""

    __slots__ = ("kind", "source", "members", "size0", "size1", "operator", "real", "coalesced")

    def __init__(self, kind, source=None, members=None, size0=0, size1=0, operator=None):
        self.kind = kind                    # "source" | "diag" | "cat"
        self.source = source                # the scipy matrix (kind "source")
        self.members = members              # list of LazySparse / torch COO tensors (kinds "diag", "cat")
        self.size0, self.size1 = size0, size1
        self.operator = operator            # the batch assembled on the device (SparseOperator) or None
        self.real = {}                      # materialised torch tensors by (device, coalesced)
        self.coalesced = kind != "source"

    def build(self, coalesced: bool):
        """The tensor the reference's code would have returned (CPU)."""
        if self.kind == "source":
            t = reference_sp_to_coo(self.source)
            return t.coalesce() if coalesced else t
        members = [m.materialize() if isinstance(m, LazySparse) else m for m in self.members]
        return (reference_diag_cat if self.kind == "diag" else reference_cat)(members, self.size0, self.size1)


class LazySparse(torch.Tensor):
    """A sparse-COO tensor of the reference's batching API that is materialised only when something reads it (module
    docstring).  `_sn_operator`: the device-resident operator the residual blocks multiply with (None: not resident)."""

    @staticmethod
    def __new__(cls, payload: _Payload, shape, dtype, device, coalesced: bool):
        t = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=dtype, layout=torch.sparse_coo,
                                                device=torch.device(device), requires_grad=False)
        t._sn_payload = payload
        t._sn_coalesced = bool(coalesced)
        # the assembled batch rides along only on the device it lives on: a handle moved to ANOTHER GPU (.cuda(1)) materialises
        # there like any tensor the cache does not know (operators.as_operator -> from_torch_coo on that device)
        dev, op = torch.device(device), payload.operator
        if dev.type != "cuda" or op is None:
            op = None
        else:
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            if op.device.type != "cuda" or op.device.index != idx:
                op = None
        t._sn_operator = op
        return t

    def __init__(self, *args, **kwargs):
        super().__init__()

    # ---- what stays lazy ------------------------------------------------------------------------------------------------
    def _like(self, device=None, coalesced=None) -> "LazySparse":
        return LazySparse(self._sn_payload, self.shape, self.dtype, self.device if device is None else device,
                          self._sn_coalesced if coalesced is None else coalesced)

    def coalesce(self):
        return self if self._sn_coalesced else self._like(coalesced=True)

    def is_coalesced(self):
        return self._sn_coalesced

    def detach(self):
        return self._like()

    def cuda(self, device=None, non_blocking=False, **_):
        if isinstance(device, int):
            device = torch.device("cuda", device)
        return self.to(torch.device("cuda") if device is None else device)

    def cpu(self, *_, **__):
        return self.to("cpu")

    def to(self, *args, **kwargs):
        device = kwargs.get("device")
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, (str, torch.device)):
                device = a
            elif isinstance(a, torch.dtype):
                dtype = a
            elif isinstance(a, torch.Tensor):
                device, dtype = a.device, a.dtype
        if dtype is not None and dtype != self.dtype:
            return self.materialize().to(*args, **kwargs)
        if device is None:
            return self
        device = torch.device(device)
        if device.type == "cuda":
            if not torch.cuda.is_available():
                raise RuntimeError("LazySparse.cuda(): no GPU is available (torch.cuda.is_available() is False)")
            if device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
        if device == self.device:
            return self
        return self._like(device=device)

    def __repr__(self, *_, **__):
        p = self._sn_payload
        what = "scipy matrix" if p.kind == "source" else f"{p.kind} of {len(p.members)} operators"
        return (f"LazySparse({what}, size={tuple(self.shape)}, dtype={self.dtype}, device={self.device}, "
                f"resident={self._sn_operator is not None})")

    def __reduce_ex__(self, proto):
        return self.materialize().__reduce_ex__(proto)

    # ---- everything else reads the real tensor ------------------------------------------------------------------------------
    def materialize(self) -> torch.Tensor:
        """The real torch sparse COO tensor this handle stands for, built the way the reference builds it (and cached)."""
        p = self._sn_payload
        key = (str(self.device), self._sn_coalesced or p.coalesced)          # (the full device: cuda:0 and cuda:1 are two tensors)
        t = p.real.get(key)
        if t is None:
            base = p.real.get(("cpu", key[1]))
            if base is None:
                base = p.real[("cpu", key[1])] = p.build(key[1])
            t = base if self.device.type == "cpu" else base.to(self.device)
            p.real[key] = t
        return t

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name == "__get__" or name in ("size", "dim", "ndimension", "is_sparse", "type", "element_size", "is_floating_point",
                                         "is_complex", "get_device", "sparse_dim", "dense_dim", "__len__"):
            with torch._C.DisableTorchFunctionSubclass():          # metadata: answered by the wrapper itself
                return func(*args, **kwargs)
        args, kwargs = pytree.tree_map(lambda a: a.materialize() if isinstance(a, LazySparse) else a, (args, kwargs))
        return func(*args, **kwargs)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        if func in (torch.ops.aten.detach.default, torch.ops.aten.alias.default) and isinstance(args[0], LazySparse):
            return args[0]._like()                                  # (Variable(A) of the Mesh-MNIST driver, src/mesh_mnist/main.py:115)
        args, kwargs = pytree.tree_map(lambda a: a.materialize() if isinstance(a, LazySparse) else a, (args, kwargs or {}))
        return func(*args, **kwargs)

    # ---- constructors ---------------------------------------------------------------------------------------------------------
    @classmethod
    def of_scipy(cls, A) -> "LazySparse":
        dt = _TORCH_DTYPE.get(np.dtype(A.dtype))
        if dt is None:
            raise TypeError(f"unsupported scipy dtype {A.dtype}")
        return cls(_Payload("source", source=A), A.shape, dt, "cpu", False)

    @classmethod
    def of_batch(cls, kind, members, size0, size1, operator) -> "LazySparse":
        n = len(members)
        shape = (n * size0, n * size1) if kind == "diag" else (n, size0, size1)
        return cls(_Payload(kind, members=list(members), size0=int(size0), size1=int(size1), operator=operator),
                   shape, members[0].dtype, "cpu", True)


# ---- resident copies of the source matrices ----------------------------------------------------------------------------------
def _digest(A) -> int:
    """Checksum of EVERY byte of a scipy matrix's arrays (xxh3 when the module is there, else zlib.adler32): ~0.3 ms for a
    360 k-entry Dirac operator."""
    arrays = [getattr(A, n, None) for n in ("data", "indices", "indptr", "row", "col")]
    try:
        import xxhash

        h = xxhash.xxh3_64()
        for a in arrays:
            if isinstance(a, np.ndarray):
                h.update(np.ascontiguousarray(a).view(np.uint8).data)
        return h.intdigest()
    except ImportError:
        import zlib

        v = 1
        for a in arrays:
            if isinstance(a, np.ndarray):
                v = zlib.adler32(np.ascontiguousarray(a).view(np.uint8).data, v)
        return v


class ResidentCache:
    """Every source matrix the batching functions have been shown, packed in HBM once (module docstring).

    Entries are keyed on the scipy OBJECT (`id`, guarded by a weak reference and the addresses / sizes of its arrays): the
    reference drivers keep their datasets as lists of scipy matrices and pass the same objects every step.  A matrix whose
    arrays are replaced is converted again.  One whose arrays are overwritten IN PLACE is caught in two ways: every call
    compares eight probed values of each of `data` / `indices` / `indptr` (O(1)), and every call re-verifies the FULL checksum
    of one resident member of the batch, round-robin — so any in-place edit is noticed within as many calls as the batch has
    members (then the matrix is converted again), deterministically, at O(nnz of one mesh) per call.  `freeze = True`
    (resident_cache().freeze = True, before the first batch) instead makes the admitted arrays read-only, so that an in-place
    edit RAISES in the driver at the line that does it.
    A source that dies takes its entry with it (weak-reference callback); the HBM of dead entries is reclaimed by starting the
    pools over once it exceeds half of what is held (or the budget is reached): SN_RESIDENT_MAX_GB, at most half of the
    device memory that was free when the cache was made.  Three growing pools per device: quaternion-packed Dirac operators
    ("q3"), other operators with a 4x4 block structure ("bsr4"), plain CSR ("csr")."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.pools = {}
        self.index = {}
        self.max_bytes = int(float(os.environ.get("SN_RESIDENT_MAX_GB", "96")) * 2**30)
        try:
            free, _total = torch.cuda.mem_get_info(self.device)
            self.max_bytes = min(self.max_bytes, int(free) // 2)      # (a pool that grows holds old + new buffer for a moment)
        except Exception:  # noqa: BLE001 — no device / no such query: the configured budget stands
            pass
        self.hits = self.misses = self.resets = self.stale = 0
        self.dead_bytes = 0
        self.freeze = False
        self._rr = 0

    def clear(self):
        self.pools.clear()
        self.index.clear()
        self.dead_bytes = 0

    @staticmethod
    def _probe(a):
        return a.reshape(-1)[:: max(1, a.size // 8)][:8].tobytes() if isinstance(a, np.ndarray) and a.size else b""

    @staticmethod
    def _signature(A):
        """What is compared on EVERY call (a few hundred matrices per batch, twice per step: O(1) each and cheap): class, shape, the
        identity of the arrays and eight evenly spaced values of the value array — the usual in-place edits (A.data *= 2) are
        caught at once; whatever this misses, the round-robin checksum of assemble() finds."""
        g = A.__dict__.get
        d = g("data")
        if d.__class__ is np.ndarray and d.size:
            probe = d.reshape(-1)[:: max(1, d.size // 8)][:8].tobytes()
        else:
            probe = b""
        return (A.__class__, g("_shape"), id(d), id(g("indices")), id(g("indptr")), id(g("row")), id(g("col")), probe)

    @staticmethod
    def _canonical(A):
        """CSR with sorted, duplicate-free rows (what coalesce() makes of the reference's COO); the driver's matrix is left
        untouched (a copy is made when it is not canonical CSR already)."""
        C = A if A.format == "csr" else A.tocsr()
        if not C.has_canonical_format:
            C = C.copy() if C is A else C
            C.sum_duplicates()
        return C

    @staticmethod
    def _blocky(A) -> bool:
        M, K = A.shape
        if M % 4 or K % 4 or A.nnz == 0:
            return False
        rows = np.repeat(np.arange(M, dtype=np.int64) // 4, np.diff(A.indptr))
        blocks = np.unique(rows * (K // 4) + A.indices.astype(np.int64) // 4).size
        return 16 * blocks <= 1.6 * A.nnz

    def _admit(self, mats) -> None:
        """Convert the matrices in `mats` (not seen before) and add them to the pools."""
        from .operators import OperatorPool

        held = sum(p.device_bytes() for p in self.pools.values())
        if held > self.max_bytes or (self.dead_bytes > (64 << 20) and 2 * self.dead_bytes > held):
            self.clear()                                    # generational: start over rather than compact the pools
            self.resets += 1
        canon = [self._canonical(A) for A in mats]
        blocky = [i for i, C in enumerate(canon) if self._blocky(C)]
        plain = [i for i in range(len(mats)) if i not in set(blocky)]
        groups = []
        if blocky:
            chunk = OperatorPool([canon[i] for i in blocky], self.device, want_bsr4=True, lean=True)
            groups.append(("q3" if chunk._fwd_q is not None else "bsr4", blocky, chunk))
        if plain:
            groups.append(("csr", plain, OperatorPool([canon[i] for i in plain], self.device, want_bsr4=False)))
        for kind, members, chunk in groups:
            pool = self.pools.get(kind)
            if pool is None:
                self.pools[kind] = chunk
                slots = np.arange(chunk.n)
            else:
                slots = pool.absorb(chunk)
            for i, slot in zip(members, slots):
                A = mats[i]
                if self.freeze:
                    for n in ("data", "indices", "indptr", "row", "col"):
                        a = getattr(A, n, None)
                        if isinstance(a, np.ndarray):
                            a.flags.writeable = False
                key, nbytes = id(A), int(24 * A.nnz)            # (A and A^T, packed: the order of magnitude is what matters)
                self.index[key] = (weakref.ref(A, lambda _r, key=key, nbytes=nbytes: self._died(key, nbytes)), self._signature(A),
                                   kind, int(slot), _digest(A))

    def _died(self, key, nbytes) -> None:
        ent = self.index.get(key)
        if ent is not None and ent[0]() is None:
            del self.index[key]
            self.dead_bytes += nbytes

    def _entry(self, A):
        ent = self.index.get(id(A))
        return ent if ent is not None and ent[0]() is A and ent[1] == self._signature(A) else None

    def assemble(self, sources: Sequence, size0: int, size1: int):
        """The block-diagonal batch operator of the scipy matrices `sources` (padded blocks size0 x size1), assembled on the
        device from their resident copies; None when the batch cannot take the resident path (mixed kinds, unsupported dtype)."""
        for A in sources:
            if A.dtype != np.float32 or A.ndim != 2 or A.shape[0] > size0 or A.shape[1] > size1:
                return None
        # one member's full checksum per call, round-robin over the batch: an in-place edit the probes miss is found within
        # len(sources) calls
        if len(sources):
            A = sources[self._rr % len(sources)]
            self._rr += 1
            ent = self._entry(A)
            if ent is not None and ent[4] != _digest(A):
                del self.index[id(A)]
                self.dead_bytes += int(24 * A.nnz)
                self.stale += 1
        ents = [self._entry(A) for A in sources]
        new = []
        for A, e in zip(sources, ents):
            if e is None and not any(A is b for b in new):
                new.append(A)
        if new:
            self.misses += len(new)
            self._admit(new)
            ents = [self._entry(A) for A in sources]
            if any(e is None for e in ents):                 # (a reset dropped members admitted before this call)
                self._admit([A for A, e in zip(sources, ents) if e is None])
                ents = [self._entry(A) for A in sources]
        self.hits += len(sources) - len(new)
        kinds = {e[2] for e in ents}
        if len(kinds) != 1:
            return None
        kind = kinds.pop()
        if kind != "csr" and (size0 % 4 or size1 % 4):
            return None
        sel = np.array([e[3] for e in ents], dtype=np.int64)
        return self.pools[kind].assemble(sel, int(size0), int(size1))


_CACHES = {}


def resident_cache(device=None) -> Optional[ResidentCache]:
    """The cache of the current (or given) GPU; None without one."""
    if not torch.cuda.is_available():
        return None
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    c = _CACHES.get(dev.index)
    if c is None:
        c = _CACHES[dev.index] = ResidentCache(dev)
    return c


def batch_operator(tensors: Sequence, size0: int, size1: int):
    """The resident batch operator of a list of per-mesh tensors, or None (no GPU, SN_RESIDENT=0, members that are not
    LazySparse handles of scipy matrices)."""
    if os.environ.get("SN_RESIDENT", "1") == "0" or not len(tensors):
        return None
    if not all(isinstance(t, LazySparse) and t._sn_payload.kind == "source" for t in tensors):
        return None
    cache = resident_cache()
    if cache is None:
        return None
    return cache.assemble([t._sn_payload.source for t in tensors], size0, size1)
