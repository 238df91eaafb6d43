"""Whole residual blocks as single autograd nodes.

Each of the three blocks of the reference operator layer (src/utils/utils_pt.py: LapResNet2 :151-180, DirResNet2 :182-220,
AvgResNet2 :222-243) runs here as ONE torch.autograd.Function whose forward and backward chain the kernels by hand.  Owning
the whole block removes what per-op autograd cannot:
  * the GEMM epilogue of one stage writes elu(y) straight into the first half of the NEXT stage's (rows, 2C) concat buffer
    (sn_linear_fwd_f32 `y_elu`), inside a block and — through the `_sn_cat` hand-off — across blocks, so an ELU pass and the
    re-read of its input disappear per stage;
  * the residual `x + block(x)` rides in the second GEMM's epilogue and its gradient is added inside the last ELU-backward
    kernel (`gadd`), and a tensor that feeds two consumers (f_out: this block's vertex stage and the next block's face
    stage) gets its two gradients summed in that same kernel — no separate accumulation passes.

Activated hand-off: a block returns its outputs with the attribute `_sn_cat` = a fresh (rows, 2C) buffer whose first half
already holds elu(output).  The next block takes (and removes) it instead of running its own ELU pass.  The attribute lives
on one tensor object only: an out-of-place op yields an object without it, and an in-place op (which keeps the object)
bumps the tensor's version counter, which take_activated compares with the one recorded at hand-off — so a stale buffer
cannot be picked up either way.
"""
from __future__ import annotations

import weakref

import torch

from . import kernels, plans
from .functional import (SpmmTimer, _launch, _rows2d, product_form, avg_stage_backward, avg_stage_backward_ragged, avg_stage_forward,
                         avg_stage_forward_ragged, bn_prepare, bnlin_backward,
                         bnlin_backward_elu_input, bnlin_backward_zero_first, bnlin_forward, bnlin_forward_zero_first, stash, unstash,
                         zero_first_supported)
from .graphs import active_capture
from .operators import as_operator

__all__ = ["lap_block", "dirac_block", "avg_block", "avg_block_ragged", "avg_block_ragged_ok", "take_activated", "attach_activated", "zero_faces_ok", "elu_conv", "elu_conv_ok"]


def _version_of(t: torch.Tensor):
    """The tensor's version counter; None for tensors made under torch.inference_mode() — they carry none (reading it
    raises) and cannot be edited in place outside inference mode either, so there is nothing to compare."""
    return None if t.is_inference() else t._version


def attach_activated(t: torch.Tensor, cat: torch.Tensor) -> torch.Tensor:
    t._sn_cat = (cat, _version_of(t))
    return t


def take_activated(t: torch.Tensor, rows: int, C: int):
    """The (rows, 2C) buffer whose first half holds elu(t), if the producer of `t` left one AND `t` still has the values
    it was computed from: an in-place edit of `t` (`v += x`, `v.mul_(mask)`, `v[:, idx] = 0` keep the object and its
    attributes) bumps the tensor's version counter, and the stale activation is then dropped, not consumed."""
    d = getattr(t, "__dict__", None)
    entry = d.pop("_sn_cat", None) if d is not None else None
    if entry is None:
        return None
    cat, version = entry
    if version != _version_of(t) or tuple(cat.shape) != (rows, 2 * C) or cat.device != t.device or cat.dtype != torch.float32:
        return None
    return cat


def _new_cat(rows, C, device):
    return torch.empty((rows, 2 * C), dtype=torch.float32, device=device)


def _new_part(rows, C, device, narrow=False):
    """Workspace in which the GEMM that writes elu(y) into a concat buffer's first half also leaves that half's column
    statistics (kernels.linear_fwd `elu_stats`), or None where the fused GEMM does not offer it.  narrow: 64-channel stages
    too (the Dirac blocks, whose other half gets its statistics from the quaternion SpMM: only both together are used)."""
    if not (C == 128 or (narrow and C == 64 and kernels.linear_fwd_supported(2 * C, C))) or not kernels.elu_stats_supported():
        return None
    return kernels.new_elu_stats_part(rows, device)


def _attach_part(cat, part, tiles=None):
    """The statistics travel with the buffer object: bnlin_forward(cat, ...) then reads only the propagated half.
    tiles: the per-tile column sums the same GEMM left (kernels.new_tile_sums) — a global-average block that consumes the buffer
    then needs no statistics pass over it (functional.avg_stage_forward)."""
    if part is not None:
        cat._sn_part = part
        if tiles is not None:
            cat._sn_tiles = tiles


_TILE_SUMS_MIN_ROWS = 32768


def _new_tiles(rows, C, device, part, wanted):
    """Tile-sum buffer for a forward GEMM whose activated output feeds a global-average block (or None).  `wanted`: _wants_tiles
    (the block's `avg_next` argument, or what the block has learnt when the caller cannot say).  Only for operands
    large enough that the statistics pass they save costs more than the per-tile path: a 7000-row FAUST tower runs FASTER with
    the pass (its per-mesh sums from tiles are four workgroups walking 218 tiles: replayed pair step 3.25 ms against 3.45, same
    box), the 322 624-row ARAP batch 0.28 ms per step slower."""
    if not wanted or part is None or C != 128 or rows < _TILE_SUMS_MIN_ROWS or not kernels.tile_sums_supported():
        return None
    return kernels.new_tile_sums(rows, device)


def _wants_tiles(mod, avg_next) -> bool:
    """Whether a Dirac / Laplacian block leaves the per-tile column sums a following global-average block reads instead of a
    statistics pass (5 MB of stores per launch at the ARAP batch).  avg_next True / False: the caller says (the product's own
    models do).  None — the reference's calling sequence, which cannot say: what the block has LEARNT: a global-average block
    that receives this block's activated hand-off without tile sums (and is large enough to want them) marks the producer
    (_learn_avg_next), so from the second step on exactly the blocks that feed one leave them and a model's last block never does."""
    if avg_next is None:
        return bool(mod.__dict__.get("_sn_avg_next", False))
    return bool(avg_next)


def _mark_producer(cat, mod, avg_next) -> None:
    if avg_next is None and not mod.__dict__.get("_sn_avg_next", False):
        cat._sn_producer = weakref.ref(mod)


def _learn_avg_next(pre, rows) -> None:
    if pre is not None and rows >= _TILE_SUMS_MIN_ROWS and _tiles_of(pre) is None:
        ref = pre.__dict__.get("_sn_producer")
        mod = ref() if ref is not None else None
        if mod is not None:
            mod._sn_avg_next = True


def _tiles_of(cat):
    """(tile sums, statistics partials) left on a concat buffer by the GEMM that wrote its first half, or None."""
    t, p = getattr(cat, "_sn_tiles", None), getattr(cat, "_sn_part", None)
    return (t, p) if (t is not None and p is not None) else None


def _attach_hi(cat, part):
    """Statistics of the PROPAGATED half, left by the SpMM that wrote it (functional._launch(..., stats=True))."""
    if part is not None:
        cat._sn_part_hi = part


def _activated(x2d, pre):
    """(rows, 2C) buffer whose first half is elu(x2d): the handed-off one, or a new one filled here."""
    if pre is not None:
        return pre
    rows, C = x2d.shape
    cat = _new_cat(rows, C, x2d.device)
    kernels.elu_into(x2d, cat[:, :C])
    return cat


def _bn_args(conv):
    """Flatten a GraphConv1x1("pre") into the tensors/flags bnlin_forward needs (and do BatchNorm's Python bookkeeping).
    Parameters and buffers are read from the modules' own dictionaries: nn.Module.__getattr__ (a Python-level search through
    three dictionaries per access) costs more than everything else here, twelve times per block."""
    mods = conv.__dict__["_modules"]
    bn, fc = mods["bn"], mods["fc"]
    training, momentum, eps = bn_prepare(bn)
    bp, bb, fp = bn.__dict__["_parameters"], bn.__dict__["_buffers"], fc.__dict__["_parameters"]
    return (bp["weight"], bp["bias"], fp["weight"], fp["bias"], bb["running_mean"], bb["running_var"], training, momentum, eps)


# ---- launch plans (plans.py): one host call per block direction ---------------------------------------------------------------
# Every block below is written as two plain functions — `*_fwd(tensors..., operators..., constants...) -> (outputs, saved)` and
# `*_bwd(saved, operators..., gradients..., constants...) -> gradients` — that launch through kernels.py.  The autograd nodes call
# them directly (eager) or hand them to `_plan_forward` / `_plan_backward`, which record their launch list once per shape
# signature and from then on enqueue it with one sn_plan_run.
_MISSING = object()


def _op_operands(ops, ncols):
    """(arrays, key) of the forms in which the block multiplies its operators and their transposes (functional.product_form:
    building a derived form launches and may synchronise — here, before anything is recorded)."""
    arrays, key = [], []
    for op, group in ops:
        for o in (op, op.t()):
            kind, arr, scal = product_form(o, group, ncols)
            arrays.extend(arr)
            key.append((kind, scal))
    return arrays, tuple(key)


def _plan_forward(ctx, site, impl, tensors, ops, consts, ncols, n_dyn, scan):
    """Run `impl(*tensors, *operators, *consts)` through its launch plan.  Returns the block's outputs (views of the plan's
    arenas) or None when the block is not plannable (the caller then runs `impl` eagerly).
    tensors[:n_dyn] are the block's features (any shape / stride: part of the plan's key in full), the rest its parameters and
    buffers (keyed by shape; a non-contiguous one makes the dry run refuse the plan); scan: positions whose `_sn_*` tensor
    attributes are operands too (plans.expand_ext)."""
    op_arrays, op_key = _op_operands(ops, ncols) if ops else ((), ())
    ext, attr_key = plans.expand_ext(tensors, scan)
    ext.extend(op_arrays)
    dyn = tensors[:n_dyn]
    ptrs = [t.data_ptr() if t is not None else 0 for t in dyn]
    key = (tuple([None if t is None else (t.shape, t.stride(), t.dtype) for t in dyn]),
           tuple([None if t is None else t.shape for t in tensors[n_dyn:]]), attr_key, op_key, consts,
           tuple([ptrs.index(p_) for p_ in ptrs]),               # features that share memory must do so on every run of the plan
           _TILE_SUMS_MIN_ROWS)
    plan, may_record = site.lookup(key)
    if plan is None:
        if not may_record:
            return None
        plan = plans.record(site, key, impl, (*tensors, *[o for o, _ in ops], *consts), ext, tensors[0].device, range(n_dyn))
        if plan is None:
            return None
        need = set()

        def slots(d):
            if isinstance(d, plans._Desc):
                if d.slot >= 2:
                    need.add(d.slot - 2)
            elif isinstance(d, tuple):
                for v in d:
                    slots(v)
        slots(plan.result[1])
        plan.saved_ext = sorted(need)                              # the operands the backward reaches through what was saved
    timer = SpmmTimer.active
    if timer is not None:
        if plan.tags is None:
            return None                                            # (a product without a host-side entry count: eager under a timer)
        timer.tags.extend(plan.tags)
    big, small = plan.new_arenas(tensors[0].device)
    plan.run(big, small, ext)
    site.replayed += 1
    build = plans._Builder(big, small, ext)
    outs = build(plan.result[0])
    if plan.effects:
        plans.apply_effects(plan, build)
    # What the backward needs: the arenas and the operands the saved descriptions point into.  Plain attributes, dropped by
    # the backward itself: autograd's saved-tensor slots would tie the arena's version counter — shared by every view of
    # it, i.e. by every output of the block — to the backward (an in-place edit of any output would then be an error).
    ctx._sn_plan = (plan, big, small, [ext[j] for j in plan.saved_ext], ops, ncols)
    return outs


def _plan_backward(ctx, site, impl, grads, consts):
    """The backward of a block whose forward ran through `_plan_forward`: `impl(saved, *operators, *grads, *consts)` through its
    own plan, recorded against the forward plan's arena layout.  Returns the gradients (nested as `impl` returns them)."""
    state = ctx._sn_plan
    if state is None:
        raise RuntimeError("a second backward through a block that ran from a launch plan (retain_graph): its workspace was "
                           "released by the first one; run with SN_PLANS=0 for that")
    fplan, big, small, kept, ops, ncols = state
    ctx._sn_plan = None
    op_arrays, op_key = _op_operands(ops, ncols) if ops else ((), ())
    maxima = [kernels.take_absmax(g) if g is not None else None for g in grads]
    ext = [big, small, *kept, *grads, *maxima, *op_arrays]
    gptrs = [g.data_ptr() if g is not None else 0 for g in grads]
    key = (tuple([None if g is None else (g.shape, g.stride(), g.dtype) for g in grads]), tuple([m is not None for m in maxima]),
           op_key, consts, tuple([gptrs.index(p_) for p_ in gptrs]))      # (one gradient tensor handed in for two outputs)
    plan = fplan.bwd.get(key, _MISSING)
    if plan is _MISSING or plan is None:
        fext = [None] * fplan.n_ext
        for j, t in zip(fplan.saved_ext, kept):
            fext[j] = t
        saved = plans._Builder(big, small, fext)(fplan.result[1])
        plans.renote(grads, maxima)
        if plan is None:                                           # not plannable: the eager backward on the saved tensors
            return impl(saved, *[o for o, _ in ops], *grads, *consts)
        g0 = 2 + len(kept)
        plan = plans.record(site, (id(fplan), key), impl, (saved, *[o for o, _ in ops], *grads, *consts), ext,
                            grads_device(grads, big, small), range(g0, g0 + len(grads)))
        fplan.bwd[key] = plan
        if plan is None:
            plans.renote(grads, maxima)                            # (the dry run took them)
            return impl(saved, *[o for o, _ in ops], *grads, *consts)
        for g in grads:                                            # (bounds the dry run did not take)
            if g is not None:
                kernels.take_absmax(g)
    timer = SpmmTimer.active
    if timer is not None:
        if plan.tags is None:
            fext = [None] * fplan.n_ext
            for j, t in zip(fplan.saved_ext, kept):
                fext[j] = t
            plans.renote(grads, maxima)
            return impl(plans._Builder(big, small, fext)(fplan.result[1]), *[o for o, _ in ops], *grads, *consts)
        timer.tags.extend(plan.tags)
    b2, s2 = plan.new_arenas(grads_device(grads, big, small))
    plan.run(b2, s2, ext)
    site.replayed += 1
    build = plans._Builder(b2, s2, ext)
    out = build(plan.result)
    if plan.effects:
        plans.apply_effects(plan, build)
    return out


def grads_device(grads, *others):
    for t in (*grads, *others):
        if t is not None:
            return t.device
    raise ValueError("no tensor to take the device from")


_SITES = {name: plans.Site(name) for name in ("dirac_fwd", "dirac_bwd", "propagate_fwd", "propagate_bwd", "avg_fwd", "avg_bwd",
                                              "avg_ragged_fwd", "avg_ragged_bwd", "elu_conv_fwd", "elu_conv_bwd")}


# ------------------------------------------------------------------------------------------------------------
_NAN = {}


def _nan_placeholder(device):
    """One NaN element per device, made once: the stand-in for an output that is not written costs no launch per block (a
    4-byte fill is a 4 us launch on the step's critical path, eight times per ARAP step).  Never cached from inside a graph
    capture — the element would live in that graph's memory pool."""
    t = _NAN.get(device)
    if t is None:
        t = torch.full((1, 1), float("nan"), dtype=torch.float32, device=device)
        if device.type != "cuda" or not torch.cuda.is_current_stream_capturing():
            _NAN[device] = t
    return t


def _dirac_fwd(v, f, pre_v, pre_f, g0, b0, W0, c0, rm0, rv0, g1, b1, W1, c1, rm1, rv1, opDi, opDiA, need_f, avg_next, tr0, mo0,
               ep0, tr1, mo1, ep1):
    """DirResNet2 (utils_pt.py:191-220):
         cat0 = [elu(f), Di·elu(v)]  -> f_out = Lin(BN(cat0));   cat1 = [elu(v), DiA·elu(f_out)] -> v + Lin(BN(cat1)).
    Returns ((v_new, f_out | None, nxt_v, nxt_f), saved)."""
    rv, C = v.shape
    rf = opDi.shape[0] // 4
    cat1 = _activated(v, pre_v)
    nxt_f = _new_cat(rf, C, v.device)                        # the next Dirac block's cat0; first half = elu(f_out)
    pf = _new_part(rf, C, v.device, narrow=True)
    if f is None:
        # all-zero face features (the first Dirac block of a model): cat0 = [0 | Di·elu(v)] runs at half width
        cat0 = torch.empty((rf, C), dtype=torch.float32, device=v.device)      # only the propagated half exists
        _attach_hi(cat0, _launch(opDi, cat1[:, :C], cat0, 4, "fwd", stats=tr0))
        f_out, st0 = bnlin_forward_zero_first(cat0, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, nxt_f[:, :C], need_f, pf)
    else:
        cat0 = pre_f if pre_f is not None else _activated(f, None)             # (f's values are not touched when handed off)
        _attach_hi(cat0, _launch(opDi, cat1[:, :C], cat0[:, C:], 4, "fwd", stats=tr0))
        f_out, st0 = bnlin_forward(cat0, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, None, nxt_f[:, :C], want_y=need_f,
                                   elu_stats=pf)
    _attach_part(nxt_f, pf)
    _attach_hi(cat1, _launch(opDiA, nxt_f[:, :C], cat1[:, C:], 4, "fwd", stats=tr1))
    nxt_v = _new_cat(rv, C, v.device)
    pv = _new_part(rv, C, v.device, narrow=True)
    tv = _new_tiles(rv, C, v.device, pv, avg_next)
    v_new, st1 = bnlin_forward(cat1, g1, b1, W1, c1, rm1, rv1, tr1, mo1, ep1, v, nxt_v[:, :C], elu_stats=pv, tile_sums=tv)
    _attach_part(nxt_v, pv, tv)
    return (v_new, f_out, nxt_v, nxt_f), ((cat0, cat1, nxt_f), st0, st1)


def _dirac_bwd(saved, opDi, opDiA, g_vnew, g_fo, f_zero, need_gv, need_gf):
    """Backward of _dirac_fwd: (g_v, g_f, dgamma0, dbeta0, dW0, db0, dgamma1, dbeta1, dW1, db1)."""
    (cat0, cat1, nxt_f), st0, st1 = saved
    C = cat1.shape[1] // 2
    dev = cat1.device
    # Every ELU backward of the block is fused: the dgrad GEMM's epilogue sends the first half of a stage's input
    # gradient through the activation (h = dx[:, :C]·elu'(e) + the gradient of the other branch), and the transposed
    # product's store does the same for the propagated half:  (opᵀ·dx[:, C:])·elu'(e) + h.
    # ---- second stage (vertex rows) ----
    gp1 = (None,) * 4
    h1 = None                                                                   # dx1[:, :C]·elu'(e_v) + g_vnew
    if g_vnew is not None:
        (dx1_hi, h1), dg1, db1, dW1, dc1 = bnlin_backward(st1, g_vnew, through_elu=(g_vnew,))
        gp1 = (dg1, db1, dW1, dc1)
        g_sum = torch.empty((nxt_f.shape[0], C), dtype=torch.float32, device=dev)
        # (DiA^T·dx1_hi)·elu'(e_f)  +  the gradient f_out receives from the next block
        _launch(opDiA.t(), dx1_hi, g_sum, 4, "bwd", elubwd=(nxt_f[:, :C], g_fo))
        g_fo = g_sum
    # ---- first stage (face rows) ----
    gp0 = (None,) * 4
    g_v = g_f = None
    dx0_hi = None
    if g_fo is not None and f_zero:
        dx0_hi, dg0, db0, dW0, dc0 = bnlin_backward_zero_first(st0, g_fo)          # no gradient for the zero half
        gp0 = (dg0, db0, dW0, dc0)
    elif g_fo is not None:
        (dx0_hi, g_f), dg0, db0, dW0, dc0 = bnlin_backward(st0, g_fo, through_elu=(None,))   # g_f = dx0[:, :C]·elu'(e_f)
        gp0 = (dg0, db0, dW0, dc0)
    if need_gv:
        if dx0_hi is not None:
            g_v = torch.empty((cat1.shape[0], C), dtype=torch.float32, device=dev)
            _launch(opDi.t(), dx0_hi, g_v, 4, "bwd", elubwd=(cat1[:, :C], h1))   # (Di^T·dx0_hi)·elu'(e_v) + h1
        else:
            g_v = h1
    if f_zero or not need_gf:
        g_f = None
    return (g_v, g_f) + gp0 + gp1


class _DiracBlock(torch.autograd.Function):
    """DirResNet2 as one autograd node (_dirac_fwd / _dirac_bwd), through a launch plan when the block is plannable."""

    @staticmethod
    def forward(ctx, v, f, opDi, opDiA, pre_v, pre_f, need_f, avg_next, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, g1, b1, W1, c1,
                rm1, rv1, tr1, mo1, ep1):
        v = _rows2d(v)
        if f is not None and pre_f is None:
            f = _rows2d(f)                     # (with a hand-off f is only a carrier — possibly the zero-stride NaN placeholder of
        ctx.f_zero = f is None                 #  need_f=False: making THAT contiguous wrote 321 MB per block at the ARAP batch)
        ctx.ops = (opDi, opDiA)
        tensors = (v, f, pre_v, pre_f, g0, b0, W0, c0, rm0, rv0, g1, b1, W1, c1, rm1, rv1)
        consts = (need_f, avg_next, tr0, mo0, ep0, tr1, mo1, ep1)
        outs = None
        ctx._sn_plan = None
        if plans.usable(v):
            outs = _plan_forward(ctx, _SITES["dirac_fwd"], _dirac_fwd, tensors, ((opDi, 4), (opDiA, 4)), consts, v.shape[1], 4, (2, 3, 8, 14))
            if outs is not None:
                _drop_counters(rm0, rm1)
        ctx._sn_planned = outs is not None
        if outs is None:
            outs, saved = _dirac_fwd(*tensors, opDi, opDiA, *consts)
            stash(ctx, *saved)
        v_new, f_out, nxt_v, nxt_f = outs
        if f_out is None:
            # The caller only chains f into the next Dirac block, which consumes the ACTIVATED hand-off: the pre-activation
            # face features are not written (321 MB per block at the ARAP batch).  What is returned in their place is a
            # zero-stride NaN view, so that any other use of it is loud instead of silently wrong.
            f_out = _nan_placeholder(v.device).expand(nxt_f.shape[0], v.shape[1])
        ctx.mark_non_differentiable(nxt_v, nxt_f)
        ctx.set_materialize_grads(False)
        return v_new, f_out, nxt_v, nxt_f

    @staticmethod
    def backward(ctx, g_vnew, g_fout, _gv, _gf):
        if g_vnew is None and g_fout is None:
            return (None,) * 26
        opDi, opDiA = ctx.ops
        g_vnew = g_vnew.contiguous() if g_vnew is not None else None
        g_fout = g_fout.contiguous() if g_fout is not None else None
        consts = (ctx.f_zero, bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1]))
        if ctx._sn_planned:
            r = _plan_backward(ctx, _SITES["dirac_bwd"], _dirac_bwd, (g_vnew, g_fout), consts)
        else:
            r = _dirac_bwd(unstash(ctx), opDi, opDiA, g_vnew, g_fout, *consts)
        none5 = (None,) * 5
        return (r[0], r[1], None, None, None, None, None, None) + r[2:6] + none5 + r[6:10] + none5


def _drop_counters(*running_means) -> None:
    """bn_prepare hangs the batch counter on the running-mean buffer for the fold launch of THIS call; a planned call reaches it
    as a plan operand — the one-shot attribute must not survive the call either way."""
    for rm in running_means:
        d = getattr(rm, "__dict__", None)
        if d is not None:
            d.pop("_sn_nbt", None)


def dirac_block(mod, Di, DiA, v, f, need_f=True, num_faces=None, avg_next=None):
    """DirResNet2.forward on (B, V, C) / (B, F, C) tensors; `mod` supplies bn_fc0 / bn_fc1.  need_f=False: the returned
    face features are only a carrier of the activated hand-off for the next Dirac block (see _DiracBlock.forward).
    f=None (with num_faces): all-zero face features, never materialised (zero_faces_ok says when).
    avg_next: the vertex output feeds a global-average block next: the GEMM that writes its activated copy also leaves the
    per-tile column sums that block needs (no statistics pass over its operand)."""
    B, V, C = v.shape
    F_ = f.shape[1] if f is not None else int(num_faces)
    rv, rf = B * V, B * F_
    opDi, opDiA = as_operator(Di), as_operator(DiA)
    if opDi.shape != (4 * rf, 4 * rv) or opDiA.shape != (4 * rv, 4 * rf):
        raise ValueError(f"DirResNet2: Di {tuple(opDi.shape)} / DiA {tuple(opDiA.shape)} do not match v rows {rv}, f rows {rf}")
    v_new, f_out, nxt_v, nxt_f = _DiracBlock.apply(v.reshape(rv, C), f.reshape(rf, C) if f is not None else None, opDi, opDiA,
                                                  take_activated(v, rv, C),
                                                  take_activated(f, rf, C) if f is not None else None, bool(need_f),
                                                  _wants_tiles(mod, avg_next), *_bn_args(mod._modules["bn_fc0"]), *_bn_args(mod._modules["bn_fc1"]))
    _mark_producer(nxt_v, mod, avg_next)
    return attach_activated(v_new.view(B, V, C), nxt_v), attach_activated(f_out.view(B, F_, C), nxt_f)


def zero_faces_ok(mod, C: int) -> bool:
    """Can dirac_block take f=None (all-zero face features at half width) for this module?"""
    return zero_first_supported(C, mod.bn_fc0.fc.weight.shape[0]) and mod.bn_fc0.fc.weight.shape[1] == 2 * C


# ------------------------------------------------------------------------------------------------------------
def _propagate_fwd(x, pre, mask_rows, inv_count, g0, b0, W0, c0, rm0, rv0, g1, b1, W1, c1, rm1, rv1, op, nseg, avg_next, tr0, mo0,
                   ep0, tr1, mo1, ep1):
    """LapResNet2 (utils_pt.py:159-180) and AvgResNet2 (utils_pt.py:230-243): two stages  [e, P(e)] -> Lin(BN(.))  with the
    same propagation P — the sparse product with L, or the per-mesh masked mean broadcast back (global_average)."""
    rows, C = x.shape
    per = rows // nseg if nseg else 0

    def propagate(cat, training):
        if op is not None:
            # (the Laplacian product leaves the BatchNorm statistics of the half it writes, like the Dirac products)
            _attach_hi(cat, _launch(op, cat[:, :C], cat[:, C:], 1, "fwd", stats=training))
        else:
            mean = kernels.segment_colsum(cat[:, :C], mask_rows, per, nseg) * inv_count
            kernels.bcast_rows(mean, cat[:, C:], per)

    cat_a = _activated(x, pre)
    propagate(cat_a, tr0)
    cat_b = _new_cat(rows, C, x.device)
    pb = _new_part(rows, C, x.device)
    _, st0 = bnlin_forward(cat_a, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, None, cat_b[:, :C],
                           want_y=False, elu_stats=pb)       # only elu(h) is consumed
    _attach_part(cat_b, pb)
    propagate(cat_b, tr1)
    nxt = _new_cat(rows, C, x.device)
    pn = _new_part(rows, C, x.device)
    tn = _new_tiles(rows, C, x.device, pn, avg_next)
    out, st1 = bnlin_forward(cat_b, g1, b1, W1, c1, rm1, rv1, tr1, mo1, ep1, x, nxt[:, :C], elu_stats=pn, tile_sums=tn)
    _attach_part(nxt, pn, tn)
    return (out, nxt), ((cat_a, cat_b), st0, st1, (mask_rows, inv_count))


def _propagate_bwd(saved, op, g_out, nseg, need_gx):
    (cat_a, cat_b), st0, st1, (mask_rows, inv_count) = saved
    C = cat_a.shape[1] // 2
    per = cat_a.shape[0] // nseg if nseg else 0

    def stage_backward(st, g_in, cat, gadd):
        """gradient w.r.t. the stage input: ((dcat[:, :C] + P^T dcat[:, C:]) * elu'(e)) + gadd, and the parameter
        gradients of the stage's BatchNorm+Linear."""
        if op is not None:                       # sparse propagation: both ELU backward passes fused (see _dirac_bwd)
            (d_hi, h), dg, db, dW, dc = bnlin_backward(st, g_in, through_elu=(gadd,))
            g = torch.empty((cat.shape[0], C), dtype=torch.float32, device=cat.device)
            _launch(op.t(), d_hi, g, 1, "bwd", elubwd=(cat[:, :C], h))
        else:                                    # global average: per-mesh column sums of dcat[:, C:], broadcast back
            dcat, dg, db, dW, dc = bnlin_backward(st, g_in)
            g = torch.empty((cat.shape[0], C), dtype=torch.float32, device=cat.device)
            gm = (kernels.segment_colsum(dcat[:, C:], None, per, nseg) * inv_count).contiguous()
            kernels.elu_bwd_bcast(dcat[:, :C], cat[:, :C], gm, mask_rows, g, per, gadd)
        return g, dg, db, dW, dc

    g_h, dg1, db1, dW1, dc1 = stage_backward(st1, g_out, cat_b, None)
    g_x, dg0, db0, dW0, dc0 = stage_backward(st0, g_h, cat_a, g_out)              # + residual-path gradient
    return (g_x if need_gx else None, dg0, db0, dW0, dc0, dg1, db1, dW1, dc1)


class _PropagateBlock(torch.autograd.Function):
    """LapResNet2 / the full-width AvgResNet2 as one autograd node (_propagate_fwd / _propagate_bwd)."""

    @staticmethod
    def forward(ctx, x, op, mask_rows, inv_count, nseg, pre, avg_next, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, g1, b1, W1, c1,
                rm1, rv1, tr1, mo1, ep1):
        x = _rows2d(x)
        tensors = (x, pre, mask_rows, inv_count, g0, b0, W0, c0, rm0, rv0, g1, b1, W1, c1, rm1, rv1)
        consts = (nseg, avg_next, tr0, mo0, ep0, tr1, mo1, ep1)
        ctx.op, ctx.nseg = op, nseg
        ctx._sn_plan = None
        outs = None
        if op is not None and plans.usable(x):          # (the full-width average stage multiplies by 1/count with a torch op)
            outs = _plan_forward(ctx, _SITES["propagate_fwd"], _propagate_fwd, tensors, ((op, 1),), consts, x.shape[1], 4, (1, 8, 14))
            if outs is not None:
                _drop_counters(rm0, rm1)
        ctx._sn_planned = outs is not None
        if outs is None:
            outs, saved = _propagate_fwd(*tensors, op, *consts)
            stash(ctx, *saved)
        out, nxt = outs
        ctx.mark_non_differentiable(nxt)
        ctx.set_materialize_grads(False)          # else the engine fills a (rows, 2C) zero gradient for `nxt` every backward
        return out, nxt

    @staticmethod
    def backward(ctx, g_out, _gn):
        if g_out is None:
            return (None,) * 25
        g_out = g_out.contiguous()
        consts = (ctx.nseg, bool(ctx.needs_input_grad[0]))
        if ctx._sn_planned:
            r = _plan_backward(ctx, _SITES["propagate_bwd"], _propagate_bwd, (g_out,), consts)
        else:
            r = _propagate_bwd(unstash(ctx), ctx.op, g_out, *consts)
        none5 = (None,) * 5
        return (r[0], None, None, None, None, None, None) + r[1:5] + none5 + r[5:9] + none5


def _avg_fwd(x, pre, mask_rows, inv_count, g0, b0, W0, c0, rm0, rv0, g1, b1, W1, c1, rm1, rv1, nseg, tr0, mo0, ep0, tr1, mo1, ep1):
    """AvgResNet2 (utils_pt.py:230-243) at half width (functional.avg_stage_forward): the broadcast mean is never written,
    both Linear layers run over C instead of 2C columns, and the whole backward of a stage — BatchNorm tail, mean-path
    gradient, ELU derivative, residual-path gradient — leaves the dgrad GEMM's epilogue."""
    rows, C = x.shape
    per = rows // nseg
    cat = _activated(x, pre)
    e_a = cat[:, :C]
    e_b = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    # per-mesh means and BatchNorm sums of a stage's operand: from what the GEMM that wrote it left (per-tile column sums +
    # statistics partials) when there is such a producer — else one statistics pass over the operand
    pb = _new_part(rows, C, x.device)
    tb = _new_tiles(rows, C, x.device, pb, True)
    _, st0 = avg_stage_forward(e_a, mask_rows, inv_count, nseg, per, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, None, e_b,
                               want_y=False, elu_stats=pb if tb is not None else None, tile_sums=tb,
                               e_tiles=_tiles_of(cat))             # only elu(h) is needed downstream
    nxt = _new_cat(rows, C, x.device)
    pn = _new_part(rows, C, x.device)
    out, st1 = avg_stage_forward(e_b, mask_rows, inv_count, nseg, per, g1, b1, W1, c1, rm1, rv1, tr1, mo1, ep1, x,
                                 nxt[:, :C], elu_stats=pn, e_tiles=(tb, pb) if tb is not None else None)
    _attach_part(nxt, pn)
    return (out, nxt), (st0, st1, (mask_rows, inv_count))


def _avg_bwd(saved, g_out, nseg, need_gx):
    st0, st1, (mask_rows, inv_count) = saved
    per = st0[0].shape[0] // nseg
    g_h, dg1, db1, dW1, dc1 = avg_stage_backward(st1, mask_rows, inv_count, nseg, per, g_out, None)
    g_x, dg0, db0, dW0, dc0 = avg_stage_backward(st0, mask_rows, inv_count, nseg, per, g_h, g_out)   # + residual path
    return (g_x if need_gx else None, dg0, db0, dW0, dc0, dg1, db1, dW1, dc1)


class _AvgBlock(torch.autograd.Function):
    """AvgResNet2 at half width as one autograd node (_avg_fwd / _avg_bwd)."""

    @staticmethod
    def forward(ctx, x, mask_rows, inv_count, nseg, pre, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, g1, b1, W1, c1, rm1, rv1,
                tr1, mo1, ep1):
        x = _rows2d(x)
        tensors = (x, pre, mask_rows, inv_count, g0, b0, W0, c0, rm0, rv0, g1, b1, W1, c1, rm1, rv1)
        consts = (nseg, tr0, mo0, ep0, tr1, mo1, ep1)
        ctx.nseg = nseg
        ctx._sn_plan = None
        outs = None
        if plans.usable(x):
            outs = _plan_forward(ctx, _SITES["avg_fwd"], _avg_fwd, tensors, (), consts, x.shape[1], 4, (1, 8, 14))
            if outs is not None:
                _drop_counters(rm0, rm1)
        ctx._sn_planned = outs is not None
        if outs is None:
            outs, saved = _avg_fwd(*tensors, *consts)
            stash(ctx, *saved)
        out, nxt = outs
        ctx.mark_non_differentiable(nxt)
        ctx.set_materialize_grads(False)          # else the engine fills a (rows, 2C) zero gradient for `nxt` every backward
        return out, nxt

    @staticmethod
    def backward(ctx, g_out, _gn):
        if g_out is None:
            return (None,) * 23
        g_out = g_out.contiguous()
        consts = (ctx.nseg, bool(ctx.needs_input_grad[0]))
        if ctx._sn_planned:
            r = _plan_backward(ctx, _SITES["avg_bwd"], _avg_bwd, (g_out,), consts)
        else:
            r = _avg_bwd(unstash(ctx), g_out, *consts)
        none5 = (None,) * 5
        return (r[0], None, None, None, None) + r[1:5] + none5 + r[5:9] + none5


def _seg_tensors(seg):
    """The device tables of a PackedSegments (operands of a plan) and what identifies its layout."""
    return (seg.tiles, seg.seg_tile_ptr, seg.inv_count, seg.off_dev, seg.slab_off, seg.seg_slab_ptr, seg.len_f64), \
        (seg.nseg, seg.rows, seg.nslab, seg.min_len, int(seg.tiles.shape[0]))


def _avg_ragged_fwd(x, pre, g0, b0, W0, c0, rm0, rv0, g1, b1, W1, c1, rm1, rv1, t0, t1, t2, t3, t4, t5, t6, seg, seg_key, mo0, ep0,
                    mo1, ep1):
    """_avg_fwd on a PACKED batch: `seg` (operators.PackedSegments) gives the meshes' row ranges; no mask, every row is real
    (BatchNorm over real rows only — the reference's padded batch includes the padding rows, utils_pt.py:97-99).
    (t0..t6: seg's device tables, listed so that a launch plan knows them as operands; seg_key: their layout.)"""
    rows, C = x.shape
    cat = _activated(x, pre)
    e_a = cat[:, :C]
    e_b = torch.empty((rows, C), dtype=torch.float32, device=x.device)
    pb = _new_part(rows, C, x.device)
    tb = _new_tiles(rows, C, x.device, pb, True)      # (as _avg_fwd: means and statistics from what the producing GEMM left)
    _, st0 = avg_stage_forward_ragged(e_a, seg, g0, b0, W0, c0, rm0, rv0, mo0, ep0, None, e_b, want_y=False, elu_stats=pb,
                                      part=getattr(cat, "_sn_part", None), tile_sums=tb, e_tiles=_tiles_of(cat))
    nxt = _new_cat(rows, C, x.device)
    pn = _new_part(rows, C, x.device)
    out, st1 = avg_stage_forward_ragged(e_b, seg, g1, b1, W1, c1, rm1, rv1, mo1, ep1, x, nxt[:, :C], elu_stats=pn, part=pb,
                                        e_tiles=(tb, pb) if tb is not None else None)
    _attach_part(nxt, pn)
    return (out, nxt), (st0, st1)


def _avg_ragged_bwd(saved, g_out, t0, t1, t2, t3, t4, t5, t6, seg, seg_key, need_gx):
    st0, st1 = saved
    g_h, dg1, db1, dW1, dc1 = avg_stage_backward_ragged(st1, seg, g_out, None)
    g_x, dg0, db0, dW0, dc0 = avg_stage_backward_ragged(st0, seg, g_h, g_out)      # + residual path
    return (g_x if need_gx else None, dg0, db0, dW0, dc0, dg1, db1, dW1, dc1)


class _AvgBlockRagged(torch.autograd.Function):
    """_AvgBlock on a PACKED batch (_avg_ragged_fwd / _avg_ragged_bwd)."""

    @staticmethod
    def forward(ctx, x, seg, pre, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, g1, b1, W1, c1, rm1, rv1, tr1, mo1, ep1):
        x = _rows2d(x)
        seg_t, seg_key = _seg_tensors(seg)
        tensors = (x, pre, g0, b0, W0, c0, rm0, rv0, g1, b1, W1, c1, rm1, rv1, *seg_t)
        consts = (_Opaque(seg), seg_key, mo0, ep0, mo1, ep1)
        ctx.seg = seg
        ctx._sn_plan = None
        outs = None
        if plans.usable(x):
            outs = _plan_forward(ctx, _SITES["avg_ragged_fwd"], _avg_ragged_fwd_u, tensors, (), consts, x.shape[1], 2, (1, 6, 12))
            if outs is not None:
                _drop_counters(rm0, rm1)
        ctx._sn_planned = outs is not None
        if outs is None:
            outs, saved = _avg_ragged_fwd(*tensors, seg, seg_key, mo0, ep0, mo1, ep1)
            stash(ctx, *saved)
        out, nxt = outs
        ctx.mark_non_differentiable(nxt)
        ctx.set_materialize_grads(False)
        return out, nxt

    @staticmethod
    def backward(ctx, g_out, _gn):
        if g_out is None:
            return (None,) * 21
        g_out = g_out.contiguous()
        seg = ctx.seg
        seg_t, seg_key = _seg_tensors(seg)
        need = bool(ctx.needs_input_grad[0])
        if ctx._sn_planned:
            r = _plan_backward(ctx, _SITES["avg_ragged_bwd"], _avg_ragged_bwd_u, (g_out, *seg_t), (_Opaque(seg), seg_key, need))
        else:
            r = _avg_ragged_bwd(unstash(ctx), g_out, *seg_t, seg, seg_key, need)
        none5 = (None,) * 5
        return (r[0], None, None) + r[1:5] + none5 + r[5:9] + none5


class _Opaque:
    """A host object a block needs (a PackedSegments) among a plan's constants: it takes no part in the plan's key — what
    identifies it is listed next to it (`seg_key`) — and is handed to the block's function as it is."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v

    def __hash__(self):
        return 0

    def __eq__(self, other):
        return isinstance(other, _Opaque)


def _avg_ragged_fwd_u(*a):
    return _avg_ragged_fwd(*a[:21], a[21].v, *a[22:])


def _avg_ragged_bwd_u(saved, g_out, *a):
    return _avg_ragged_bwd(saved, g_out, *a[:7], a[7].v, *a[8:])


def avg_block_ragged_ok(mod, seg, inputs) -> bool:
    C = inputs.shape[-1]
    a0, a1 = _bn_args(mod._modules["bn_fc0"]), _bn_args(mod._modules["bn_fc1"])
    return bool(a0[6] and a1[6]) and inputs.dtype == torch.float32 and mod.bn_fc0.fc.weight.shape == (C, 2 * C) and \
        mod.bn_fc1.fc.weight.shape == (C, 2 * C) and kernels.avg_stage_ragged_supported(C, C, seg) and \
        seg.rows == inputs.shape[0] * inputs.shape[1]


def avg_block_ragged(mod, seg, inputs):
    """AvgResNet2 on a packed (1, sum V_i, C) batch as one autograd node at half width."""
    B, V, C = inputs.shape
    rows = B * V
    pre = take_activated(inputs, rows, C)
    _learn_avg_next(pre, rows)
    out, nxt = _AvgBlockRagged.apply(inputs.reshape(rows, C), seg, pre, *_bn_args(mod.bn_fc0),
                                     *_bn_args(mod.bn_fc1))
    return attach_activated(out.view(B, V, C), nxt)


def _elu_conv_fwd(v, pre, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0):
    C = v.shape[1]
    cat = _activated(v, pre)
    part = getattr(cat, "_sn_part", None)        # statistics of elu(v) left by the GEMM that wrote it
    pre_stats = kernels.colstats_from_part(part, v.shape[0]) if (part is not None and tr0 and C == 128) else None
    y, st = bnlin_forward(cat[:, :C], g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, pre_stats=pre_stats)
    return (y,), (st,)


def _elu_conv_bwd(saved, dy, need_gv):
    (st,) = saved
    g_v, dg, db, dW, dc = bnlin_backward_elu_input(st, dy)
    return (g_v if need_gv else None, dg, db, dW, dc)


class _EluConv(torch.autograd.Function):
    """GraphConv1x1("pre")(F.elu(v)) — the models' last layer (src/as_rigid_as_possible/models.py:148-150) — as one node:
    elu(v) is the activated hand-off of the preceding block when there is one (no ELU pass), and the backward runs the
    BatchNorm tail and the activation derivative as one pass."""

    @staticmethod
    def forward(ctx, v, pre, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0):
        v = _rows2d(v)
        tensors = (v, pre, g0, b0, W0, c0, rm0, rv0)
        consts = (tr0, mo0, ep0)
        ctx._sn_plan = None
        outs = None
        if plans.usable(v):
            outs = _plan_forward(ctx, _SITES["elu_conv_fwd"], _elu_conv_fwd, tensors, (), consts, v.shape[1], 2, (1, 6))
            if outs is not None:
                _drop_counters(rm0)
        ctx._sn_planned = outs is not None
        if outs is None:
            outs, saved = _elu_conv_fwd(*tensors, *consts)
            stash(ctx, *saved)
        return outs[0]

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        need = bool(ctx.needs_input_grad[0])
        if ctx._sn_planned:
            r = _plan_backward(ctx, _SITES["elu_conv_bwd"], _elu_conv_bwd, (dy,), (need,))
        else:
            r = _elu_conv_bwd(unstash(ctx), dy, need)
        return (r[0], None) + r[1:5] + (None,) * 5


def elu_conv(conv, v):
    """conv(F.elu(v)) for a GraphConv1x1 with batch_norm="pre" on a (B, N, C) tensor, as one autograd node."""
    B, N, C = v.shape
    rows = B * N
    y = _EluConv.apply(v.reshape(rows, C), take_activated(v, rows, C), *_bn_args(conv))
    return y.view(B, N, conv.num_outputs)


def elu_conv_ok(conv, v) -> bool:
    c = v.shape[-1]
    return getattr(conv, "batch_norm", None) == "pre" and v.dtype == torch.float32 and c % 4 == 0 and 256 % (c // 4) == 0 and \
        conv.bn.affine and conv.bn.momentum is not None and conv.bn.track_running_stats


def lap_block(mod, L, inputs, avg_next=None):
    B, V, C = inputs.shape
    rows = B * V
    op = as_operator(L)
    if op.shape != (rows, rows):
        raise ValueError(f"LapResNet2: operator {tuple(op.shape)} vs {rows} rows")
    out, nxt = _PropagateBlock.apply(inputs.reshape(rows, C), op, None, None, 0, take_activated(inputs, rows, C),
                                     _wants_tiles(mod, avg_next), *_bn_args(mod._modules["bn_fc0"]), *_bn_args(mod._modules["bn_fc1"]))
    _mark_producer(nxt, mod, avg_next)
    return attach_activated(out.view(B, V, C), nxt)


def avg_block(mod, mask, inputs):
    B, V, C = inputs.shape
    rows = B * V
    # The 7 global-average blocks of a model share one mask: its flattened copy and the per-mesh 1/count are cached on the
    # mask object, keyed by its version (in-place edits invalidate).  NOT while a hipGraph is being captured: a value
    # computed eagerly during warm-up would be baked into the graph as a constant, and every replay on another batch
    # loaded into the static mask would divide by the example batch's vertex counts.  Under capture the reduction is
    # recorded (once per block: a (B, 1) reduction) and nothing is cached.
    # Inside a GraphedStep capture the blocks of THAT capture share the recorded value (it is recomputed by every replay
    # before its first use; the key carries the capture's serial number, so it is never taken for an eager value or for
    # another capture's).
    capturing = mask.is_cuda and torch.cuda.is_current_stream_capturing()
    gen = active_capture() if capturing else None
    slot = "_sn_avg" if not capturing else ("_sn_avg_cap" if gen is not None else None)
    cached = getattr(mask, slot, None) if slot else None
    key = (B, V, _version_of(mask), gen)
    if cached is None or cached[0] != key:
        cached = (key, mask.reshape(rows).contiguous(), 1.0 / mask.reshape(B, V).sum(1, keepdim=True))
        if slot:
            try:
                setattr(mask, slot, cached)
            except AttributeError:
                pass
    _, mask_rows, inv_count = cached
    a0, a1 = _bn_args(mod._modules["bn_fc0"]), _bn_args(mod._modules["bn_fc1"])
    if a0[6] and a1[6] and kernels.avg_stage_supported(C, a0[2].shape[0], V) and a0[2].shape[1] == 2 * C and a1[2].shape[0] == C:
        pre = take_activated(inputs, rows, C)
        _learn_avg_next(pre, rows)
        out, nxt = _AvgBlock.apply(inputs.reshape(rows, C), mask_rows, inv_count, B, pre, *a0, *a1)
    else:
        out, nxt = _PropagateBlock.apply(inputs.reshape(rows, C), None, mask_rows, inv_count, B, take_activated(inputs, rows, C),
                                         False, *a0, *a1)
    return attach_activated(out.view(B, V, C), nxt)
