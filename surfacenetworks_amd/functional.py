"""Autograd entry points of the hot path.  Each Function is a thin shell around launches in kernels.py.

    spmm(A, x)                  y = A·x ; backward grad_x = A^T·grad_y through the cached CSR of A^T, grad_A = None
                                (contract of SparseBMMFunc, src/utils/cuda/sparse_bmm_func.py:27-72, and of
                                torch.mm(sparse, dense) at src/utils/utils_pt.py:167,176,202,214)
    lap_propagate(L, x)         [elu(x), L·elu(x)] written straight into one (rows, 2C) buffer
                                (F.elu + torch.mm + torch.cat of src/utils/utils_pt.py:161-168 and :171-177)
    dirac_face_stage(Di, v, f)  ([elu(f), Di·elu(v)], elu(v))         (src/utils/utils_pt.py:195-204)
    dirac_vert_stage(DiA, f_out, e_v)  [e_v, DiA·elu(f_out)]          (src/utils/utils_pt.py:208-216)

Dense operands are 2-D (rows, C) fp32 tensors; `group` says how many operator rows one tensor row holds
(1 for the Laplacian, 4 for the quaternion view (B*V*4, C/4) of the Dirac path, utils_pt.py:201,213).
"""
from __future__ import annotations

import os

import torch

from . import _lib as _lib_mod
from . import kernels
from .operators import SparseOperator, as_operator

__all__ = ["spmm", "lap_propagate", "dirac_face_stage", "dirac_vert_stage", "avg_propagate", "avg_propagate_ragged", "bn_linear", "bnlin_forward",
           "bnlin_backward", "bn_prepare", "avg_stage_forward_ragged", "avg_stage_backward_ragged", "set_dirac_format", "set_laplacian_format", "SpmmTimer", "thin_linear", "thin_linear_supported"]

_DIRAC_FORMAT = "q3"
_LAPLACIAN_FORMAT = "ring"       # set_laplacian_format


def set_laplacian_format(fmt: str) -> None:
    """PROCESS DEFAULT of the kernel / storage form of the group-1 (Laplacian-type) products at 64 / 128 dense columns (one
    operator can choose for itself: `op.format = "rb4"`):
    'ring' (default) the sliding-window kernel straight from the CSR arrays (X rows within +-160 of the current rows in an
                    LDS ring, everything requested once by LDS-DMA) for square operators of >= 131 072 rows whose entries all
                    lie in that window (SparseOperator.ring_ok: batches of meshes in a locality-preserving vertex order);
                    every other operator as 'rb4';
    'rb4'           4x1 row blocks — four consecutive rows share one gather per distinct column; built from the CSR
                    arrays on the device the first time an operator is used (bit-identical results for finite inputs);
    'csr'           always the generic CSR kernel."""
    global _LAPLACIAN_FORMAT
    if fmt not in ("ring", "rb4", "csr"):
        raise ValueError(fmt)
    _LAPLACIAN_FORMAT = fmt


def set_dirac_format(fmt: str) -> None:
    """PROCESS DEFAULT of the kernel / storage form of the group-4 (quaternionic Dirac) products (one operator can choose for
    itself: `op.format = "bsr4"`; batches a pool assembles in the packed form only multiply in that form):
    'q3'   (default) quaternion-packed blocks, 16 bytes each, when the operator's blocks are pure-quaternion matrices;
                     operators that are not fall back to 'bsr4';
    'bsr4' packed 4x4 blocks (68 bytes each) when the operator has them;
    'csr'  always the generic CSR kernel."""
    global _DIRAC_FORMAT
    if fmt not in ("q3", "bsr4", "csr"):
        raise ValueError(fmt)
    _DIRAC_FORMAT = fmt
    from . import operators

    operators._POOL_FORMAT = "q3" if fmt == "q3" else "bsr4"


class SpmmTimer:
    """Optional per-launch timing of every SpMM (bench.py measures the dominant kernel live with it).  Uses the
    library's timing facility (sn_timing_*): hipExtLaunchKernelGGL stamps each kernel's own start/stop into HIP events
    on the launch stream, so the durations agree with rocprofv3's kernel trace.  Off by default: zero overhead."""

    active = None

    def __init__(self):
        # one (tag, nnz | operator) per launch, in launch order.  Operators whose entry count is already known on the
        # host (pool-assembled batches) are recorded by that number; only the others are kept alive until results()
        # reads their nnz (no sync inside the timed region either way)
        self.tags = []

    def __enter__(self):
        from . import _lib

        _lib.call("sn_timing_drain", None, None, 0, _ctypes_i64_ref())       # drop stale records
        _lib.call("sn_timing_enable", 1)
        SpmmTimer.active = self
        return self

    def __exit__(self, *exc):
        from . import _lib

        SpmmTimer.active = None
        _lib.call("sn_timing_enable", 0)

    @staticmethod
    def time_linear(on: bool) -> None:
        """While a timer is active: also time the Linear-layer launches (default) or the sparse products only."""
        from . import _lib

        if SpmmTimer.active is not None:
            _lib.call("sn_timing_enable", 1 if on else 2)

    def results(self):
        """[(tag, M, K, nnz, N, milliseconds)] for every launch recorded while the timer was active; the tag ends in
        /csr, /bsr4 or /q3, then +e (fused ELU-backward epilogue: E read), +g (G read too), +s (column statistics left)."""
        import ctypes
        import numpy as np

        from . import _lib

        n = int(_lib.load().sn_timing_count())
        ms = np.zeros(max(n, 1), np.float64)
        meta = np.zeros((max(n, 1), 5), np.int64)
        written = ctypes.c_int64(0)
        _lib.call("sn_timing_drain", ms.ctypes.data, meta.ctypes.data, n, ctypes.addressof(written))
        # Linear-layer launches (kind >= 0x100: forward / input gradient / weight gradient, recorded by their launchers) are
        # interleaved with the SpMM records; they go to self.linear as (kernel, rows, width, out_width, bytes, ms)
        lin_names = {0x100: "linear_fwd", 0x200: "linear_dgrad", 0x400: "wgrad"}
        self.linear = []
        keep = []
        for i in range(written.value):
            kind = int(meta[i][0])
            if kind >= 0x100:
                base = kind & 0x700
                self.linear.append((f"{lin_names.get(base, 'linear')}[{kind & 0xff}]", int(meta[i][1]), int(meta[i][2]), int(meta[i][4]),
                                    int(meta[i][3]), float(ms[i])))
            else:
                keep.append(i)
        if len(keep) != len(self.tags):
            raise RuntimeError(f"timing records ({len(keep)}) do not match launches ({len(self.tags)})")
        meta, ms = meta[keep], ms[keep]
        out = []
        for i, (tag, op) in enumerate(self.tags):
            kind, M, K, _, N = meta[i]
            nnz = op if isinstance(op, int) else op.nnz
            fmt = ("/ring" if kind & 64 else "/rb4" if kind & 32 else "/q3" if kind & 8 else "/bsr4" if kind & 1 else "/csr") + ("+e" if kind & 2 else "") + \
                ("+g" if kind & 4 else "") + ("+s" if kind & 16 else "")
            out.append((tag + fmt, int(M), int(K), int(nnz), int(N), float(ms[i])))
        return out




def _ctypes_i64_ref():
    import ctypes

    _ctypes_i64_ref.slot = ctypes.c_int64(0)          # kept alive on the function object
    return ctypes.addressof(_ctypes_i64_ref.slot)


def product_form(op: SparseOperator, group: int, ncols: int):
    """(kind, arrays, (M, K, count)): the storage form in which `_launch` multiplies `op` with a dense operand of `ncols` columns in
    groups of `group` — "q3" | "bsr4" | "ring" | "rb4" | "csr" — chosen by the operator's own `format`, else the process defaults
    (set_dirac_format / set_laplacian_format).  A derived form is built on first use and cached on the operator; the choice
    itself is cached per (group, width, defaults), so a launch plan (plans.py) can name the operator's arrays without launching."""
    M, K = op.shape
    own = getattr(op, "format", None)
    dfmt = own if own in ("q3", "bsr4", "csr") else _DIRAC_FORMAT
    lfmt = own if own in ("ring", "rb4", "csr") else _LAPLACIAN_FORMAT
    ck = (group, ncols, dfmt, lfmt, kernels.RING_MIN_ROWS)
    cache = op.__dict__.get("_form_cache")
    if cache is not None and cache[0] == ck:
        return cache[1]
    # the block-form kernels and every fused epilogue exist for N in {16, 32, 64, 128} dense columns (the widths the
    # reference models use: 64 / 128 channels); any other width takes the generic CSR kernel and an unfused epilogue
    vec = (ncols // group) in (16, 32, 64, 128)
    form = None
    if group == 4 and dfmt == "q3" and vec:
        q = op.q3()
        if q is not None:
            form = ("q3", q, (M, K, int(q[1].shape[0])))
    if form is None:
        b = op.bsr4() if (dfmt != "csr" and group == 4 and vec) else None
        if b is not None:
            form = ("bsr4", b, (M, K, int(b[1].numel())))
        elif lfmt == "ring" and group == 1 and op.ring_ok(ncols):
            # banded square operator on a batch that fills the chip: sliding window over X in LDS, straight from the CSR arrays
            form = ("ring", (op.rowptr, op.colind, op.vals), (M, K, int(op.colind.numel())))
        elif lfmt in ("ring", "rb4") and kernels.spmm_rb4_supported(ncols // group, group) and op.rb4() is not None:
            r = op.rb4()                                   # Laplacian-type operator: one gather per listed column of a 4-row group
            form = ("rb4", r, (M, K, int(r[1].numel())))
        else:
            form = ("csr", (op.rowptr, op.colind, op.vals), (M, K, int(op.colind.numel())))
    if not (op.is_cuda and torch.cuda.is_current_stream_capturing()):     # (ring_ok answers False for an unmeasured band in a capture)
        op.__dict__["_form_cache"] = (ck, form)
    return form


def _launch(op: SparseOperator, x: torch.Tensor, y: torch.Tensor, group: int, tag: str = "", elubwd=None, stats: bool = False):
    """y <- op·x with the best resident format of `op` (product_form).  elubwd = (e, g): y <- (op·x) * elu'(e) + g fused into the
    store (the backward of an ELU-activated propagation stage; g may be None).  stats=True: where the kernel offers it (packed
    Dirac operators and Laplacian-type CSR operators, 128 channels) the launch also leaves the column statistics of y and
    their partials are returned (kernels.spmm_q3_stats / spmm_csr_stats), else None."""
    M, K = op.shape
    timer = SpmmTimer.active
    rec = _lib_mod.recorder() if _lib_mod._recorder is not None else None
    if timer is not None or rec is not None:
        known = op._nnz_cache if op._nnz_cache is not None else (int(op._csr[1].numel()) if op._csr is not None else None)
        if rec is not None:
            rec.tags.append((tag, known))                # (a launch plan replays its products' tags into an active timer)
        else:
            timer.tags.append((tag, op if known is None else known))
    e, g = elubwd if elubwd is not None else (None, None)
    vec = (y.shape[1] // group) in (16, 32, 64, 128)
    kind, arr, _ = product_form(op, group, y.shape[1])
    if kind == "q3":
        if stats and e is None and kernels.spmm_q3_stats_supported(y.shape[1] // group, group):
            return kernels.spmm_q3_stats(arr[0], arr[1], M // 4, K // 4, x, y, group)
        am = kernels.spmm_q3(arr[0], arr[1], M // 4, K // 4, x, y, group, e, g, want_absmax=e is not None and kernels.absmax_wanted())
        if am is not None:
            kernels.note_absmax(y, am)           # a fused ELU-backward product writes a gradient: the dy of the layer below
    elif kind == "bsr4":
        if elubwd is None:
            kernels.spmm_bsr4(arr[0], arr[1], arr[2], M // 4, K // 4, x, y, group)
        else:
            kernels.spmm_bsr4_elubwd(arr[0], arr[1], arr[2], M // 4, K // 4, x, e, g, y, group)
    elif kind == "ring":
        if stats and e is None and y.shape[1] == 128:
            return kernels.spmm_ring_stats(arr[0], arr[1], arr[2], M, K, x, y)
        am = kernels.spmm_ring(arr[0], arr[1], arr[2], M, K, x, y, e, g, want_absmax=e is not None and kernels.absmax_wanted())
        if am is not None:
            kernels.note_absmax(y, am)               # (as the packed Dirac product above: y is the dy of the layer below)
    elif kind == "rb4":
        if stats and e is None and y.shape[1] == 128:
            return kernels.spmm_rb4_stats(arr[0], arr[1], arr[2], M, K, x, y)
        am = kernels.spmm_rb4(arr[0], arr[1], arr[2], M, K, x, y, e, g, want_absmax=e is not None and kernels.absmax_wanted())
        if am is not None:
            kernels.note_absmax(y, am)
    elif elubwd is None:
        if stats and kernels.spmm_csr_stats_supported(y.shape[1] // group, group):
            return kernels.spmm_csr_stats(arr[0], arr[1], arr[2], M, K, x, y)
        kernels.spmm_csr(arr[0], arr[1], arr[2], M, K, x, y, group)
    elif vec:
        kernels.spmm_csr_elubwd(arr[0], arr[1], arr[2], M, K, x, e, g, y, group)
    else:
        tmp = torch.empty(y.shape, dtype=torch.float32, device=y.device)
        kernels.spmm_csr(arr[0], arr[1], arr[2], M, K, x, tmp, group)
        kernels.elu_bwd(tmp, e, y, False, None, g)                      # y = tmp * elu'(e) + g
    return None


def _rows2d(x: torch.Tensor) -> torch.Tensor:
    """A 2-D fp32 view with contiguous rows (copies only if the rows themselves are strided)."""
    if x.dim() != 2:
        raise ValueError("expected a 2-D dense operand")
    if x.dtype != torch.float32:
        raise TypeError("the Surface-Network path is fp32")
    if x.shape[1] > 1 and x.stride(1) != 1:
        x = x.contiguous()
    return x


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op, group):
        x = _rows2d(x)
        M, K = op.shape
        N = x.shape[1] // group
        y = torch.empty((M // group, group * N), dtype=torch.float32, device=x.device)
        _launch(op, x, y, group, "fwd")
        ctx.op, ctx.group = op, group
        return y

    @staticmethod
    def backward(ctx, gy):
        if not ctx.needs_input_grad[0]:
            return None, None, None
        op, group = ctx.op, ctx.group
        gy = _rows2d(gy)
        M, K = op.shape
        gx = torch.empty((K // group, gy.shape[1]), dtype=torch.float32, device=gy.device)
        _launch(op.t(), gy, gx, group, "bwd")
        return gx, None, None            # no gradient w.r.t. the operator (sparse_bmm_func.py:72)


def spmm(A, x: torch.Tensor, group: int = 1) -> torch.Tensor:
    """A·x for a 2-D dense x of shape (K/group, group*N); returns (M/group, group*N)."""
    op = as_operator(A)
    if x.shape[0] * group != op.shape[1]:
        raise ValueError(f"spmm: operator is {tuple(op.shape)} but x has {x.shape[0]}x{group} rows")
    return _SpMM.apply(x, op, group)


class _LapPropagate(torch.autograd.Function):
    """cat = [e, L·e], e = elu(x).  Backward: g_e = g_cat[:, :C] + L^T·g_cat[:, C:]; g_x = g_e * elu'(e)."""

    @staticmethod
    def forward(ctx, x, op):
        x = _rows2d(x)
        rows, C = x.shape
        cat = torch.empty((rows, 2 * C), dtype=torch.float32, device=x.device)
        e, le = cat[:, :C], cat[:, C:]
        kernels.elu_into(x, e)
        _launch(op, e, le, 1, "fwd")
        ctx.op = op
        ctx.save_for_backward(cat)
        return cat

    @staticmethod
    def backward(ctx, g_cat):
        (cat,) = ctx.saved_tensors
        g_cat = _rows2d(g_cat)
        C = cat.shape[1] // 2
        g_e = torch.empty((cat.shape[0], C), dtype=torch.float32, device=cat.device)
        _launch(ctx.op.t(), g_cat[:, C:], g_e, 1, "bwd")
        g_x = torch.empty_like(g_e)
        kernels.elu_bwd(g_e, cat[:, :C], g_x, False, g_cat[:, :C])       # (g_e + g_cat[:, :C]) * elu'(e) in one pass
        return g_x, None


def lap_propagate(L, x2d: torch.Tensor) -> torch.Tensor:
    op = as_operator(L)
    if op.shape[0] != op.shape[1] or op.shape[1] != x2d.shape[0]:
        raise ValueError(f"lap_propagate: operator {tuple(op.shape)} vs {x2d.shape[0]} rows")
    return _LapPropagate.apply(x2d, op)


class _DiracFaceStage(torch.autograd.Function):
    """cat0 = [elu(f), Di·elu(v)] (face rows) and e_v = elu(v) (vertex rows, reused by the vertex stage)."""

    @staticmethod
    def forward(ctx, v, f, op):
        v, f = _rows2d(v), _rows2d(f)
        C = v.shape[1]
        e_v = torch.empty_like(v, memory_format=torch.contiguous_format)
        kernels.elu_into(v, e_v)
        cat0 = torch.empty((f.shape[0], 2 * C), dtype=torch.float32, device=v.device)
        kernels.elu_into(f, cat0[:, :C])
        _launch(op, e_v, cat0[:, C:], 4, "fwd")
        ctx.op = op
        ctx.save_for_backward(cat0, e_v)
        ctx.set_materialize_grads(False)
        return cat0, e_v

    @staticmethod
    def backward(ctx, g_cat0, g_ev):
        cat0, e_v = ctx.saved_tensors
        C = e_v.shape[1]
        g_v = g_f = None
        if g_cat0 is not None:
            g_cat0 = _rows2d(g_cat0)
        if ctx.needs_input_grad[1] and g_cat0 is not None:
            g_f = torch.empty((cat0.shape[0], C), dtype=torch.float32, device=cat0.device)
            kernels.elu_bwd(g_cat0[:, :C], cat0[:, :C], g_f, False)
        if ctx.needs_input_grad[0]:
            g_e = torch.empty_like(e_v)
            if g_cat0 is not None:
                _launch(ctx.op.t(), g_cat0[:, C:], g_e, 4, "bwd")
            else:
                g_e.zero_()
            g_v = torch.empty_like(e_v)
            kernels.elu_bwd(g_e, e_v, g_v, False, _rows2d(g_ev) if g_ev is not None else None)
        return g_v, g_f, None


def dirac_face_stage(Di, v2d: torch.Tensor, f2d: torch.Tensor):
    op = as_operator(Di)
    if op.shape[1] != 4 * v2d.shape[0] or op.shape[0] != 4 * f2d.shape[0]:
        raise ValueError(f"dirac_face_stage: Di is {tuple(op.shape)}, v rows {v2d.shape[0]}, f rows {f2d.shape[0]}")
    return _DiracFaceStage.apply(v2d, f2d, op)


class _DiracVertStage(torch.autograd.Function):
    """cat1 = [e_v, DiA·elu(f_out)] (vertex rows)."""

    @staticmethod
    def forward(ctx, f_out, e_v, op):
        f_out, e_v = _rows2d(f_out), _rows2d(e_v)
        C = e_v.shape[1]
        e_f = torch.empty_like(f_out, memory_format=torch.contiguous_format)
        kernels.elu_into(f_out, e_f)
        cat1 = torch.empty((e_v.shape[0], 2 * C), dtype=torch.float32, device=e_v.device)
        cat1[:, :C].copy_(e_v)
        _launch(op, e_f, cat1[:, C:], 4, "fwd")
        ctx.op = op
        ctx.save_for_backward(e_f)
        return cat1

    @staticmethod
    def backward(ctx, g_cat1):
        (e_f,) = ctx.saved_tensors
        g_cat1 = _rows2d(g_cat1)
        C = e_f.shape[1]
        g_fo = None
        if ctx.needs_input_grad[0]:
            g_e = torch.empty_like(e_f)
            _launch(ctx.op.t(), g_cat1[:, C:], g_e, 4, "bwd")
            g_fo = torch.empty_like(e_f)
            kernels.elu_bwd(g_e, e_f, g_fo, False)
        g_ev = g_cat1[:, :C] if ctx.needs_input_grad[1] else None
        return g_fo, g_ev, None


def dirac_vert_stage(DiA, f_out2d: torch.Tensor, e_v2d: torch.Tensor) -> torch.Tensor:
    op = as_operator(DiA)
    if op.shape[0] != 4 * e_v2d.shape[0] or op.shape[1] != 4 * f_out2d.shape[0]:
        raise ValueError(f"dirac_vert_stage: DiA is {tuple(op.shape)}, v rows {e_v2d.shape[0]}, f rows {f_out2d.shape[0]}")
    return _DiracVertStage.apply(f_out2d, e_v2d, op)


# ---- optional synchronised BatchNorm statistics over data-parallel replicas --------------------------------
_BN_SYNC = None          # None: per-replica statistics (DDP semantics, default) | True / a process group: global statistics


def set_bn_sync(enabled: bool = True, group=None) -> None:
    """Make every fused BatchNorm+Linear use statistics of the GLOBAL batch: the per-channel sums (fp64) and the row count
    are all-reduced in the forward, G = dy^T(x - mean) and colsum(dy) in the backward, so that N replicas reproduce the
    single-process result at the same global batch (parameter gradients are returned divided by N: the gradient all-reduce
    (SUM) then restores them).  Costs one small collective per layer and direction and a host read of the global row
    count (shards may have different padded sizes)."""
    global _BN_SYNC
    _BN_SYNC = (group if group is not None else True) if enabled else None


def _sync_world():
    import torch.distributed as dist

    if _BN_SYNC is None or not (dist.is_available() and dist.is_initialized()):
        return None, 1
    grp = None if _BN_SYNC is True else _BN_SYNC
    n = dist.get_world_size(grp)
    return (grp, n) if n > 1 else (None, 1)


def _sync_stats(stats, rows: int):
    grp, n = _sync_world()
    if n == 1:
        return stats, rows
    import torch.distributed as dist

    buf = torch.cat([stats.reshape(-1), stats.new_tensor([float(rows)])])
    dist.all_reduce(buf, group=grp)
    k = stats.numel()
    return buf[:k].view_as(stats).contiguous(), int(round(float(buf[k].item())))


def _sync_grad_stats(Gc, sdy):
    """All-reduced (Gc, colsum(dy)) and the factor the parameter gradients derived from them must be multiplied by."""
    grp, n = _sync_world()
    if n == 1:
        return Gc, sdy, 1.0
    import torch.distributed as dist

    Gc = Gc.contiguous()
    dist.all_reduce(Gc, group=grp)
    sdy = sdy.contiguous()
    dist.all_reduce(sdy, group=grp)
    return Gc, sdy, 1.0 / n


def _scale_param_grads(scale, *grads):
    return grads if scale == 1.0 else tuple(None if g is None else g * scale for g in grads)


class _Const:
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v


def stash(ctx, *groups) -> None:
    """Keep tuples of (tensors | plain values) for backward THROUGH ctx.save_for_backward, so that autograd releases the
    buffers as soon as the node has run (a plain ctx attribute would pin hundreds of MB of HBM per block until the
    whole graph object dies — e.g. until the caller drops the previous step's loss)."""
    tensors, spec = [], []
    for g in groups:
        gs = []
        for item in g:
            if isinstance(item, torch.Tensor):
                gs.append(len(tensors))
                tensors.append(item)
            else:
                gs.append(_Const(item))
        spec.append(gs)
    ctx.save_for_backward(*tensors)
    ctx._sn_spec = spec


def unstash(ctx):
    t = ctx.saved_tensors
    return [tuple(t[i] if isinstance(i, int) else i.v for i in gs) for gs in ctx._sn_spec]


def _take_counter(running_mean):
    """The num_batches_tracked tensor bn_prepare left on the running-mean buffer for THIS call (one-shot), or None."""
    d = getattr(running_mean, "__dict__", None)
    return d.pop("_sn_nbt", None) if d is not None else None


def bnlin_forward(x, gamma, beta, W, b, running_mean, running_var, training, momentum, eps, residual=None, elu_out=None,
                  want_y=True, elu_stats=None, pre_stats=None, tile_sums=None):
    """Forward of the folded BatchNorm1d("pre") + Linear on a (rows, C) operand (no autograd): statistics in one pass
    (fp64 accumulation), BN folded into the weights  y = x·(W·diag(s))ᵀ + (b + W·t),  s = gamma*invstd, t = beta - mean*s,
    optional residual add and ELU copy in the GEMM epilogue.  Returns (y, state) with `state` for bnlin_backward."""
    # x may carry `_sn_part`: the column statistics of its first half, left by the GEMM that wrote it (blocks.py) — the
    # statistics pass then reads only the propagated half
    part, part_hi = getattr(x, "_sn_part", None), getattr(x, "_sn_part_hi", None)   # (hi: left by the SpMM that wrote P·e)
    x = _rows2d(x)
    rows = x.shape[0]
    nbt = _take_counter(running_mean)
    if not training:
        stats = None
    elif pre_stats is not None:                 # (2, C) float64 statistics of x supplied by its producer
        stats = pre_stats
    elif ((part is not None or part_hi is not None) and x.shape[1] == 256) or \
            (part is not None and part_hi is not None and x.shape[1] == 128):      # (64-channel stages: both producers or none)
        stats = kernels.colstats_halves(x, part, part_hi)
    else:
        stats = kernels.colstats(x)
    rows_g = rows
    if training:
        stats, rows_g = _sync_stats(stats, rows)
    mean, invstd, s, t, Wf, bf = kernels.bn_fold(stats, rows_g, gamma, beta, W, b, eps, momentum, training, running_mean,
                                                 running_var, nbt)
    if residual is not None:
        residual = _rows2d(residual)
    if kernels.linear_fwd_supported(x.shape[1], W.shape[0]):
        y = kernels.linear_fwd(x, Wf, bf, residual, elu_out, want_y, elu_stats, tile_sums)   # row-streaming GEMM, weights in registers
    else:
        if elu_stats is not None:
            raise ValueError("bnlin_forward: elu_stats needs a shape the fused GEMM covers")
        _library_gemm("the forward", x.shape[1], W.shape[0])
        y = torch.addmm(bf, x, Wf.t())
        if residual is not None:
            y += residual
        if elu_out is not None:
            kernels.elu_into(y, elu_out)
    return y, (x, W, Wf, s, mean, invstd, beta, training, b is not None, rows_g)


_STRICT = os.environ.get("SN_STRICT", "0") == "1"


def _library_gemm(what: str, K: int, J: int) -> None:
    """A Linear of this shape is not covered by the hand-written kernels (forward / input gradient: K in {128, 256} inputs and
    J <= 128 outputs, J a multiple of 4; weight gradient: J <= 128, C in {128, 256}, or the 64 x 64 head of Mesh-MNIST) and takes
    torch's GEMM (hipBLASLt).  None of the reference's models has such a layer (their widths are 3/6 -> C, 2C -> C, C -> 10/120
    with C in {64, 128}; the 3/6-input and 10-output ones have kernels of their own).  SN_STRICT=1 turns the silent library
    call into an error, so that a model landing on it is visible."""
    if _STRICT:
        raise RuntimeError(f"SN_STRICT: {what} of a {K} -> {J} Linear would run on the library GEMM (no hand-written kernel for this shape)")


def _dy_bounds(dy, invstd, rows_g, training):
    """Bounds for the two-piece weight gradient (kernels.wgrad(bounds=...)): the maxima the producer of dy left for it, BatchNorm's
    inverse standard deviations of the operand's columns and the row count behind them — or None (eval-mode BatchNorm has no
    batch statistics; dy from a kernel that leaves no maxima): then the three-piece bf16 form runs."""
    if not training or not kernels.absmax_wanted():
        return None
    am = kernels.take_absmax(dy)
    if am is None:
        return None
    return am, invstd, rows_g


def _centered_wgrad(dy, x, mean, bounds=None):
    """G = dyᵀ·(x - mean) and colsum(dy) (fp64 statistics layout of kernels.wgrad).  Widths the split-K kernel takes go to it
    directly.  A 64-wide x (the classifier head of the Mesh-MNIST models, mesh_mnist/models.py: bn_conv2) is read as rows/2
    rows of 128 — two consecutive rows side by side, dy likewise — so the same kernel applies: the product of the paired
    matrices holds G of the even rows in its upper-left block and G of the odd rows in its lower-right one (the cross blocks
    are discarded).  A library GEMM has no tile for a 64 x 64 output with K = 8e4 (217 us at 77 k rows against ~25 us)."""
    rows, C = x.shape
    J = dy.shape[1]
    if kernels.wgrad_supported(J, C):
        return kernels.wgrad(dy, x, mean, want_colsum=True, bounds=bounds)       # colsum(dy) rides on the same pass over dy
    if C == 64 and rows % 2 == 0 and rows > 0 and x.is_contiguous() and dy.is_contiguous() and kernels.wgrad_supported(2 * J, 2 * C):
        if bounds is not None:
            bounds = (bounds[0], torch.cat([bounds[1], bounds[1]]), bounds[2])
        G2, s2 = kernels.wgrad(dy.view(rows // 2, 2 * J), x.view(rows // 2, 2 * C), torch.cat([mean, mean]), want_colsum=True,
                               bounds=bounds)
        return G2[:J, :C] + G2[J:, C:], s2[:J] + s2[J:]
    _library_gemm("the weight gradient", C, J)
    return dy.t().mm(x - mean), kernels.colstats(dy)


def _wgrad_and_coeffs(dy, x, W, s, mean, invstd, beta, training, has_bias, rows_g):
    """(dW, db, dgamma, dbeta, Bc, Cc) of a folded BatchNorm+Linear: the centred weight gradient G = dyᵀ·(x - mean), colsum(dy) and
    every BatchNorm reduction they give algebraically: the product, [the all-reduce of synchronised statistics,] the coefficients."""
    bounds = _dy_bounds(dy, invstd, rows_g, training)
    Gc, sdy = _centered_wgrad(dy, x, mean, bounds)
    scale = 1.0
    if training:
        Gc, sdy, scale = _sync_grad_stats(Gc, sdy)
    dW, db, dgamma, dbeta, Bc, Cc = kernels.bn_bwd_coeffs(Gc, sdy, W, s, invstd, beta, rows_g, has_bias)
    dW, db, dgamma, dbeta = _scale_param_grads(scale, dW, db, dgamma, dbeta)
    return dW, db, dgamma, dbeta, Bc, Cc


def bnlin_backward(state, dy, need_dx=True, through_elu=None):
    """Backward of bnlin_forward: G = dyᵀ·(x - mean) (split-K MFMA kernel) and colsum(dy) give every BatchNorm
    reduction algebraically (sum_r dz = colsum(dy)·W, sum_r dz∘(x-mean) = sum_j W∘G); dx = dy·(W·diag(s)) + (x-mean)∘B + C
    in ONE GEMM with the tail in its epilogue.  Returns (dx, dgamma, dbeta, dW, db).

    through_elu=(gadd,): x is a stage's concat buffer [e | P·e]; instead of dx the first element returned is the pair
    (dx[:, C/2:],  dx[:, :C/2] * elu'(e) + gadd) — the operand of the transposed propagation and the gradient that has
    already passed the activation (gadd may be None), produced by the GEMM epilogue when the fused kernel applies."""
    x, W, Wf, s, mean, invstd, beta, training, has_bias, rows_g = state
    dy = dy.contiguous()
    rows, C = x.shape
    J = dy.shape[1]
    # centring inside the kernel leaves no fp32 cancellation against mean·colsum(dy)
    dW, db, dgamma, dbeta, Bc, Cc = _wgrad_and_coeffs(dy, x, W, s, mean, invstd, beta, training, has_bias, rows_g)
    dx = None
    if through_elu is not None and training and kernels.linear_dgrad_elu_supported(J, C):
        dx = kernels.linear_dgrad_elu(dy, Wf, x, mean, Bc, Cc, through_elu[0])
    elif need_dx or through_elu is not None:
        if kernels.linear_dgrad_supported(J, C):
            dx = kernels.linear_dgrad(dy, Wf, x, mean, Bc, Cc) if training else kernels.linear_dgrad(dy, Wf)
        else:
            _library_gemm("the input gradient", C, J)
            dx = dy.mm(Wf)
            if training:
                kernels.affine_cols_acc(dx, x, Bc, Cc, mean)
        if through_elu is not None:                                  # unfused composition of the same result
            h = C // 2
            gact = torch.empty((rows, h), dtype=torch.float32, device=dx.device)
            kernels.elu_bwd(dx[:, :h], x[:, :h], gact, False, None, through_elu[0])
            dx = (dx[:, h:], gact)
    return dx, dgamma, dbeta, dW, db


def zero_first_supported(C: int, J: int) -> bool:
    """Shapes for which the stage [0 | p] -> Lin(BN(.)) runs at half width (bnlin_forward_zero_first)."""
    return C == 128 and J == 128 and kernels.linear_fwd_supported(C, J) and kernels.wgrad_supported(J, C) and \
        kernels.linear_dgrad_supported(J, C)


def bnlin_forward_zero_first(p, gamma, beta, W, b, running_mean, running_var, training, momentum, eps, elu_out=None,
                             want_y=True, elu_stats=None):
    """bnlin_forward of the concat buffer [0 | p] WITHOUT the zero half (the first Dirac block of a model: its face
    features are all zero, src/as_rigid_as_possible/models.py:138).  The statistics of the zero columns are (0, 0); folded
    into the Linear they only contribute the constant W[:, :C]·beta[:C] to the bias, so the product runs over the C real
    columns: same y, same running statistics, half the GEMM, no zero buffer, no ELU / statistics pass over it."""
    part_hi = getattr(p, "_sn_part_hi", None)       # statistics of p left by the SpMM that wrote it
    p = _rows2d(p)
    rows, C = p.shape
    stats = None
    rows_g = rows
    nbt = _take_counter(running_mean)
    if training:
        stats = torch.zeros((2, 2 * C), dtype=torch.float64, device=p.device)
        if part_hi is not None and C == 128:
            kernels.colstats_merge_into(part_hi, stats, C)
        else:
            kernels.colstats_into(p, stats, C)
        stats, rows_g = _sync_stats(stats, rows)
    mean, invstd, s, t, Wf, bf = kernels.bn_fold(stats, rows_g, gamma, beta, W, b, eps, momentum, training, running_mean,
                                                 running_var, nbt)
    y = kernels.linear_fwd(p, Wf[:, C:], bf, None, elu_out, want_y, elu_stats)
    return y, (p, W, Wf, s, mean, invstd, beta, training, b is not None, rows_g)


def bnlin_backward_zero_first(state, dy):
    """Backward of bnlin_forward_zero_first: (dp, dgamma, dbeta, dW, db) — the gradient w.r.t. the zero half is not formed
    (nothing upstream of it)."""
    p, W, Wf, s, mean, invstd, beta, training, has_bias, rows_g = state
    dy = dy.contiguous()
    rows, C = p.shape
    J = dy.shape[1]
    b2 = _dy_bounds(dy, invstd[C:], rows_g, training)
    G2, sdy = kernels.wgrad(dy, p, mean[C:], want_colsum=True, bounds=b2)
    # zero half: x - mean = -mean there, i.e. G[:, :C] = -colsum(dy) (x) mean[:C] — zero with batch statistics (mean = 0),
    # the running mean in eval mode
    Gc = torch.empty((J, 2 * C), dtype=torch.float32, device=G2.device)
    if training:
        Gc[:, :C].zero_()
    else:
        Gc[:, :C].copy_(-(sdy.to(torch.float32)[:, None] * mean[None, :C]))
    Gc[:, C:].copy_(G2)
    scale = 1.0
    if training:
        Gc, sdy, scale = _sync_grad_stats(Gc, sdy)
    dW, db, dgamma, dbeta, Bc, Cc = kernels.bn_bwd_coeffs(Gc, sdy, W, s, invstd, beta, rows_g, has_bias)
    dW, db, dgamma, dbeta = _scale_param_grads(scale, dW, db, dgamma, dbeta)
    Wf2 = Wf[:, C:]
    dp = kernels.linear_dgrad(dy, Wf2, p, mean[C:], Bc[C:], Cc[C:]) if training else kernels.linear_dgrad(dy, Wf2)
    return dp, dgamma, dbeta, dW, db


def bnlin_backward_elu_input(state, dy):
    """Backward of bnlin_forward whose operand x is the OUTPUT of an ELU (`conv(F.elu(v))`), continued through that
    activation: returns (dL/dv, dgamma, dbeta, dW, db).  The BatchNorm tail and the activation derivative are one in-place
    pass over the input gradient (sn_affine_cols_elu_bwd_f32) instead of a tail pass plus torch's ELUBackward."""
    x, W, Wf, s, mean, invstd, beta, training, has_bias, rows_g = state
    dy = dy.contiguous()
    J, C = dy.shape[1], x.shape[1]
    dW, db, dgamma, dbeta, Bc, Cc = _wgrad_and_coeffs(dy, x, W, s, mean, invstd, beta, training, has_bias, rows_g)
    if training and kernels.linear_dgrad_elu_supported(128, C) and kernels.linear_dgrad_supported(J, C):
        # product, BatchNorm tail and activation derivative in one kernel (epilogue of the input-gradient GEMM)
        dx = kernels.linear_dgrad_eluseg(dy, Wf, x, mean, Bc, Cc, None, 0)
        return dx, dgamma, dbeta, dW, db
    dx = kernels.linear_dgrad(dy, Wf) if kernels.linear_dgrad_supported(J, C) else dy.mm(Wf)
    if training:
        kernels.affine_cols_elu_bwd(dx, x, Bc, Cc, mean)
    else:
        kernels.affine_cols_elu_bwd(dx, x)
    return dx, dgamma, dbeta, dW, db


def avg_stage_forward(e, mask_rows, inv_count, nseg, per, gamma, beta, W, b, running_mean, running_var, training, momentum,
                      eps, residual=None, elu_out=None, want_y=True, elu_stats=None, tile_sums=None, e_tiles=None):
    """One stage of AvgResNet2 (utils_pt.py:230-243), Lin(BN([e | global_average(e) broadcast])), at HALF width: the second
    half of the concat buffer is a per-mesh constant m, so it is never materialised — its BatchNorm statistics follow from
    m (nseg x C numbers), its share of the Linear product is a per-mesh bias m·Wf[:, C:]^T + bf, and the GEMM runs over the
    C real columns only.  Training-mode BatchNorm only (the caller checks kernels.avg_stage_supported)."""
    e = _rows2d(e)
    rows, C = e.shape
    if e_tiles is not None:
        # the GEMM that wrote e left its per-tile column sums and its statistics partials: no pass over e
        m, stats = kernels.avg_stats_from_tiles(e_tiles[0], e_tiles[1], e, mask_rows, inv_count, per, nseg)
    else:
        m, stats = kernels.avg_stats(e, mask_rows, inv_count, per, nseg)       # per-mesh mean + BatchNorm statistics, one pass over e
    stats, rows_g = _sync_stats(stats, rows)
    if kernels.avg_merged_supported(W.shape[0], C, m.shape[0]):
        mean, invstd, s, t, Wf, bf, segb = kernels.bn_fold_seg(stats, rows_g, gamma, beta, W, b, eps, momentum, running_mean,
                                                               running_var, m, _take_counter(running_mean))
    else:
        mean, invstd, s, t, Wf, bf = kernels.bn_fold(stats, rows_g, gamma, beta, W, b, eps, momentum, True, running_mean, running_var,
                                                     _take_counter(running_mean))
        segb = kernels.seg_affine(m, Wf[:, C:], bf)
    if residual is not None:
        residual = _rows2d(residual)
    y = kernels.linear_fwd_segbias(e, Wf[:, :C], segb, per, residual, elu_out, want_y, elu_stats, tile_sums)
    return y, (e, m, W, Wf, s, mean, invstd, beta, b is not None, rows_g)


def avg_stage_backward(state, mask_rows, inv_count, nseg, per, dy, gadd):
    """Backward of avg_stage_forward THROUGH the ELU that produced e: returns (dL/d(pre-activation of e) + gadd, dgamma,
    dbeta, dW, db).  Second-half terms: G[:, C:] = sum_mesh S^T (m - mu2) with S the per-mesh column sums of dy; the gradient
    of the mean path, inv_count * (S·Wf2 + per ((m - mu2) B2 + C2)), is added per row inside the dgrad GEMM's epilogue."""
    e, m, W, Wf, s, mean, invstd, beta, has_bias, rows_g = state
    dy = dy.contiguous()
    rows, C = e.shape
    bounds = _dy_bounds(dy, invstd[:C], rows_g, True)
    G1, sdy, Sg = kernels.wgrad_seg(dy, e, mean[:C], per, bounds=bounds)   # per-mesh column sums of dy from the same pass
    if _BN_SYNC is None and kernels.avg_merged_supported(dy.shape[1], C, m.shape[0], 2):
        dW, db, dgamma, dbeta, Bc, Cc, segvec = kernels.avg_bn_bwd(G1, sdy, Sg, m, mean[C:], W, s, invstd, beta, rows_g, has_bias,
                                                                   Wf[:, C:], inv_count, rows_per_seg=per)
        g = kernels.linear_dgrad_eluseg(dy, Wf[:, :C], e, mean[:C], Bc[:C], Cc[:C], segvec, per, mask_rows, gadd)
        return g, dgamma, dbeta, dW, db
    Gc = kernels.avg_bwd_gc(G1, Sg, m, mean[C:])
    Gc, sdy, scale = _sync_grad_stats(Gc, sdy)
    dW, db, dgamma, dbeta, Bc, Cc = kernels.bn_bwd_coeffs(Gc, sdy, W, s, invstd, beta, rows_g, has_bias)
    dW, db, dgamma, dbeta = _scale_param_grads(scale, dW, db, dgamma, dbeta)
    segvec = kernels.avg_bwd_segvec(Sg, Wf[:, C:], m, mean[C:], Bc[C:], Cc[C:], inv_count, per)
    g = kernels.linear_dgrad_eluseg(dy, Wf[:, :C], e, mean[:C], Bc[:C], Cc[:C], segvec, per, mask_rows, gadd)
    return g, dgamma, dbeta, dW, db


def avg_stage_forward_ragged(e, seg, gamma, beta, W, b, running_mean, running_var, momentum, eps, residual=None, elu_out=None,
                             want_y=True, elu_stats=None, part=None, tile_sums=None, e_tiles=None):
    """avg_stage_forward on a PACKED batch (`seg`: operators.PackedSegments — ragged meshes, no padding rows, every row
    real): per-mesh means by sn_segment_colsum_ragged_f32; BatchNorm statistics of the first half from the partials `part`
    the GEMM that wrote e left (else one more pass over e), of the broadcast half from the means (mesh i contributes
    len_i·m_i and len_i·m_i²); the per-mesh bias enters the GEMM by mesh offsets (sn_linear_fwd_segbias_ragged_f32)."""
    e = _rows2d(e)
    rows, C = e.shape
    if e_tiles is not None:
        # the GEMM that wrote e left its per-tile column sums and its statistics partials: no pass over e at all
        m, stats = kernels.avg_stats_from_tiles_ragged(e_tiles[0], e_tiles[1], e, seg)
    else:
        m = seg.mean(e).contiguous()
    # statistics of [e | mean broadcast] in one launch: the partials of the kernel that wrote e (or one more pass over e) and
    # len_i m_i, len_i m_i^2 per mesh
    if e_tiles is not None:
        pass
    elif part is not None and C == 128:
        stats = kernels.avg_stats_ragged(m, seg, part, kernels.linear_fwd_stats_blocks(rows))
    else:
        stats = kernels.avg_stats_ragged(m, seg, kernels.colstats(e).reshape(1, 2, C), 1)
    stats, rows_g = _sync_stats(stats, rows)
    if kernels.avg_merged_supported(W.shape[0], C, m.shape[0]):
        mean, invstd, s, t, Wf, bf, segb = kernels.bn_fold_seg(stats, rows_g, gamma, beta, W, b, eps, momentum, running_mean,
                                                               running_var, m.contiguous(), _take_counter(running_mean))
    else:
        mean, invstd, s, t, Wf, bf = kernels.bn_fold(stats, rows_g, gamma, beta, W, b, eps, momentum, True, running_mean, running_var,
                                                     _take_counter(running_mean))
        segb = kernels.seg_affine(m, Wf[:, C:], bf)
    if residual is not None:
        residual = _rows2d(residual)
    y = kernels.linear_fwd_segbias_ragged(e, Wf[:, :C], segb, seg, residual, elu_out, want_y, elu_stats, tile_sums)
    return y, (e, m, W, Wf, s, mean, invstd, beta, b is not None, rows_g)


def avg_stage_backward_ragged(state, seg, dy, gadd):
    """Backward of avg_stage_forward_ragged through the ELU that produced e (as avg_stage_backward; the mean-path gradient of
    mesh i is inv_count_i (S_i·Wf2 + len_i ((m_i - mu2) B2 + C2)))."""
    e, m, W, Wf, s, mean, invstd, beta, has_bias, rows_g = state
    dy = dy.contiguous()
    rows, C = e.shape
    bounds = _dy_bounds(dy, invstd[:C], rows_g, True)
    G1, sdy, Sg = kernels.wgrad_slabs(dy, e, mean[:C], seg, bounds=bounds)
    if _BN_SYNC is None and kernels.avg_merged_supported(dy.shape[1], C, m.shape[0], 2):
        dW, db, dgamma, dbeta, Bc, Cc, segvec = kernels.avg_bn_bwd(G1, sdy, Sg, m.contiguous(), mean[C:], W, s, invstd, beta, rows_g,
                                                                   has_bias, Wf[:, C:], seg.inv_count, segoff=seg.off_dev)
        g = kernels.linear_dgrad_eluseg_ragged(dy, Wf[:, :C], e, mean[:C], Bc[:C], Cc[:C], segvec, seg, gadd)
        return g, dgamma, dbeta, dW, db
    Gc = kernels.avg_bwd_gc(G1, Sg, m, mean[C:])
    Gc, sdy, scale = _sync_grad_stats(Gc, sdy)
    dW, db, dgamma, dbeta, Bc, Cc = kernels.bn_bwd_coeffs(Gc, sdy, W, s, invstd, beta, rows_g, has_bias)
    dW, db, dgamma, dbeta = _scale_param_grads(scale, dW, db, dgamma, dbeta)
    segvec = kernels.avg_bwd_segvec_ragged(Sg, Wf[:, C:], m, mean[C:], Bc[C:], Cc[C:], seg)
    g = kernels.linear_dgrad_eluseg_ragged(dy, Wf[:, :C], e, mean[:C], Bc[:C], Cc[:C], segvec, seg, gadd)
    return g, dgamma, dbeta, dW, db


class _BNLinear(torch.autograd.Function):
    """autograd wrapper of bnlin_forward / bnlin_backward (BatchNorm1d("pre") + Linear of GraphConv1x1,
    src/utils/utils_pt.py:83-99, without materialising the normalised tensor)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, W, b, running_mean, running_var, training, momentum, eps, residual):
        y, state = bnlin_forward(x, gamma, beta, W, b, running_mean, running_var, training, momentum, eps, residual)
        stash(ctx, state)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        (state,) = unstash(ctx)
        dx, dgamma, dbeta, dW, db = bnlin_backward(state, dy, ctx.needs_input_grad[0])
        return dx, dgamma, dbeta, dW, db, None, None, None, None, None, (dy if ctx.has_res else None)


def bn_prepare(bn: torch.nn.BatchNorm1d):
    """Per-call bookkeeping nn.BatchNorm1d does in Python: returns (training, momentum, eps).  The num_batches_tracked
    counter is bumped by the fold kernel of this call (sn_bn_fold_f32) instead of a launch of its own: it rides on the
    running-mean buffer as a one-shot attribute that bnlin_forward / avg_stage_forward pick up."""
    d = bn.__dict__
    track = d["track_running_stats"]
    training = d["training"] or not track
    if d["momentum"] is None or not d["affine"]:
        raise NotImplementedError("the fused BatchNorm+Linear supports the default affine BatchNorm1d with a fixed momentum")
    if d["training"] and track:
        buf = d["_buffers"]
        nbt, rm = buf.get("num_batches_tracked"), buf.get("running_mean")
        if nbt is not None and rm is not None:
            rm._sn_nbt = nbt
    return training, d["momentum"], d["eps"]


def bn_linear(x2d: torch.Tensor, bn: torch.nn.BatchNorm1d, fc: torch.nn.Linear, residual=None) -> torch.Tensor:
    """Fused BatchNorm1d + Linear on a (rows, C) fp32 operand (+ `residual`, added in the GEMM epilogue); updates bn's
    running statistics like nn.BatchNorm1d."""
    training, momentum, eps = bn_prepare(bn)
    return _BNLinear.apply(x2d, bn.weight, bn.bias, fc.weight, fc.bias, bn.running_mean, bn.running_var, training,
                           momentum, eps, residual)


class _ThinLinear(torch.autograd.Function):
    """nn.Linear with a handful of input channels on (rows, Cin <= 8) — the models' first layer,
    GraphConv1x1(6 | 3, C, batch_norm=None) (src/utils/utils_pt.py:99).  Forward: one output-streaming kernel that can also
    leave elu(y) in the next block's concat buffer (sn_linear_thin_fwd_f32); the backward's weight and bias gradients are
    ONE pass over dy (sn_wgrad_thin_f32) instead of a 128 x 6, K = rows GEMM the library has no tile for (490 us -> ~40 us
    at the ARAP batch) plus a column reduction."""

    @staticmethod
    def forward(ctx, x, W, b, want_elu):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        cat = None
        if want_elu:                             # elu(y) straight into the next block's concat buffer (blocks.py hand-off)
            cat = torch.empty((x.shape[0], 2 * W.shape[0]), dtype=torch.float32, device=x.device)
        y = kernels.linear_thin_fwd(x, W, b, cat[:, :W.shape[0]] if cat is not None else None)
        if cat is None:
            return y, None
        ctx.mark_non_differentiable(cat)
        ctx.set_materialize_grads(False)
        return y, cat

    @staticmethod
    def backward(ctx, dy, _gcat=None):
        x, W = ctx.saved_tensors
        if dy is None:
            return None, None, None, None
        dy = dy.contiguous()
        dx = dy.mm(W) if ctx.needs_input_grad[0] else None
        dW, db = kernels.wgrad_thin(dy, x, ctx.has_bias)
        return dx, dW, db, None


def thin_linear_supported(x2d: torch.Tensor, fc: torch.nn.Linear) -> bool:
    return x2d.dtype == torch.float32 and fc.weight.dtype == torch.float32 and \
        kernels.wgrad_thin_supported(fc.out_features, fc.in_features)


def thin_linear(x2d: torch.Tensor, fc: torch.nn.Linear, want_elu: bool = False):
    """(y, cat): cat is None or a (rows, 2J) buffer whose first half holds elu(y) (the blocks' activated hand-off)."""
    return _ThinLinear.apply(x2d.contiguous(), fc.weight, fc.bias, bool(want_elu))


class _AvgPropagate(torch.autograd.Function):
    """cat = [e, mean_mesh(e)], e = elu(x): the propagate step of AvgResNet2 (src/utils/utils_pt.py:230-241) with
    global_average (utils_pt.py:120-122) fused: ELU straight into the concat buffer, per-mesh masked column sums in one
    pass, the mean broadcast written into the second half; backward folds the mean-path gradient into the ELU backward."""

    @staticmethod
    def forward(ctx, x, mask_rows, inv_count, nseg):
        x = _rows2d(x)
        rows, C = x.shape
        per = rows // nseg
        cat = torch.empty((rows, 2 * C), dtype=torch.float32, device=x.device)
        e = cat[:, :C]
        kernels.elu_into(x, e)
        mean = kernels.segment_colsum(e, mask_rows, per, nseg) * inv_count          # (nseg, C) * (nseg, 1)
        kernels.bcast_rows(mean, cat[:, C:], per)
        ctx.save_for_backward(cat, mask_rows, inv_count)
        ctx.per, ctx.nseg = per, nseg
        return cat

    @staticmethod
    def backward(ctx, g_cat):
        cat, mask_rows, inv_count = ctx.saved_tensors
        g_cat = _rows2d(g_cat)
        C = cat.shape[1] // 2
        gm = kernels.segment_colsum(g_cat[:, C:], None, ctx.per, ctx.nseg) * inv_count   # d/d(mean) / count
        g_x = torch.empty((cat.shape[0], C), dtype=torch.float32, device=cat.device)
        kernels.elu_bwd_bcast(g_cat[:, :C], cat[:, :C], gm.contiguous(), mask_rows, g_x, ctx.per)
        return g_x, None, None, None


class _AvgPropagateRagged(torch.autograd.Function):
    """_AvgPropagate on a PACKED batch: cat = [e, mean_mesh(e)] with meshes of different sizes and no padding rows
    (operators.PackedSegments; sn_segment_colsum_ragged_f32 / sn_bcast_rows_ragged_f32)."""

    @staticmethod
    def forward(ctx, x, seg):
        x = _rows2d(x)
        rows, C = x.shape
        cat = torch.empty((rows, 2 * C), dtype=torch.float32, device=x.device)
        kernels.elu_into(x, cat[:, :C])
        kernels.bcast_rows_ragged(seg.mean(cat[:, :C]), seg.tiles, cat[:, C:])
        ctx.seg = seg
        ctx.save_for_backward(cat)
        return cat

    @staticmethod
    def backward(ctx, g_cat):
        (cat,) = ctx.saved_tensors
        seg = ctx.seg
        g_cat = _rows2d(g_cat)
        C = cat.shape[1] // 2
        gm = seg.mean(g_cat[:, C:])                               # (sum over the mesh's rows of d/d(mean)) / count
        gb = torch.empty((cat.shape[0], C), dtype=torch.float32, device=cat.device)
        kernels.bcast_rows_ragged(gm, seg.tiles, gb)
        g_x = torch.empty_like(gb)
        kernels.elu_bwd(g_cat[:, :C], cat[:, :C], g_x, False, gb)      # (g_e + broadcast mean-path gradient) * elu'(e)
        return g_x, None


def avg_propagate_ragged(x2d: torch.Tensor, seg) -> torch.Tensor:
    """x2d: (sum V_i, C) packed rows, seg: operators.PackedSegments -> (sum V_i, 2C) [elu(x), per-mesh mean of elu(x)]."""
    return _AvgPropagateRagged.apply(x2d, seg)


def avg_propagate(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """x: (B, V, C), mask: (B, V, 1) -> (B*V, 2C) concat buffer [elu(x), per-mesh masked mean of elu(x)]."""
    B, V, C = x.shape
    mask_rows = mask.reshape(B * V).contiguous()
    inv_count = 1.0 / mask.reshape(B, V).sum(1, keepdim=True)
    return _AvgPropagate.apply(x.reshape(B * V, C), mask_rows, inv_count, B)
