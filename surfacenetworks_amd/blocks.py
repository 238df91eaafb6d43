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

import torch

from . import kernels
from .functional import (_launch, _rows2d, avg_stage_backward, avg_stage_backward_ragged, avg_stage_forward,
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
    """Tile-sum buffer for a forward GEMM whose activated output feeds a global-average block (or None).  `wanted` comes from the
    block's `avg_next` argument: True / None (not said: the unmodified reference models — every Dirac / Laplacian block but the
    last is followed by an AvgResNet2, as_rigid_as_possible/models.py:115-121 — get the hand-off too; a buffer nobody picks up
    costs 5 MB of stores per launch) or False.  Only for operands
    large enough that the statistics pass they save costs more than the per-tile path: a 7000-row FAUST tower runs FASTER with
    the pass (its per-mesh sums from tiles are four workgroups walking 218 tiles: replayed pair step 3.25 ms against 3.45, same
    box), the 322 624-row ARAP batch 0.28 ms per step slower."""
    if not wanted or part is None or C != 128 or rows < _TILE_SUMS_MIN_ROWS or not kernels.tile_sums_supported():
        return None
    return kernels.new_tile_sums(rows, device)


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
    """Flatten a GraphConv1x1("pre") into the tensors/flags bnlin_forward needs (and do BatchNorm's Python bookkeeping)."""
    training, momentum, eps = bn_prepare(conv.bn)
    return (conv.bn.weight, conv.bn.bias, conv.fc.weight, conv.fc.bias, conv.bn.running_mean, conv.bn.running_var,
            training, momentum, eps)


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


class _DiracBlock(torch.autograd.Function):
    """DirResNet2 (utils_pt.py:191-220):
         cat0 = [elu(f), Di·elu(v)]  -> f_out = Lin(BN(cat0));   cat1 = [elu(v), DiA·elu(f_out)] -> v + Lin(BN(cat1))."""

    @staticmethod
    def forward(ctx, v, f, opDi, opDiA, pre_v, pre_f, need_f, avg_next, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, g1, b1, W1, c1,
                rm1, rv1, tr1, mo1, ep1):
        v = _rows2d(v)
        rv, C = v.shape
        rf = opDi.shape[0] // 4
        cat1 = _activated(v, pre_v)
        nxt_f = _new_cat(rf, C, v.device)                        # the next Dirac block's cat0; first half = elu(f_out)
        pf = _new_part(rf, C, v.device, narrow=True)
        ctx.f_zero = f is None
        if f is None:
            # all-zero face features (the first Dirac block of a model): cat0 = [0 | Di·elu(v)] runs at half width
            cat0 = torch.empty((rf, C), dtype=torch.float32, device=v.device)      # only the propagated half exists
            _attach_hi(cat0, _launch(opDi, cat1[:, :C], cat0, 4, "fwd", stats=tr0))
            f_out, st0 = bnlin_forward_zero_first(cat0, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, nxt_f[:, :C], need_f, pf)
        else:
            cat0 = pre_f if pre_f is not None else _activated(_rows2d(f), None)    # (f's values are not touched when handed off)
            _attach_hi(cat0, _launch(opDi, cat1[:, :C], cat0[:, C:], 4, "fwd", stats=tr0))
            f_out, st0 = bnlin_forward(cat0, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, None, nxt_f[:, :C], want_y=need_f,
                                       elu_stats=pf)
        _attach_part(nxt_f, pf)
        if f_out is None:
            # The caller only chains f into the next Dirac block, which consumes the ACTIVATED hand-off: the pre-activation
            # face features are not written (321 MB per block at the ARAP batch).  What is returned in their place is a
            # zero-stride NaN view, so that any other use of it is loud instead of silently wrong.
            f_out = _nan_placeholder(v.device).expand(rf, C)
        _attach_hi(cat1, _launch(opDiA, nxt_f[:, :C], cat1[:, C:], 4, "fwd", stats=tr1))
        nxt_v = _new_cat(rv, C, v.device)
        pv = _new_part(rv, C, v.device, narrow=True)
        tv = _new_tiles(rv, C, v.device, pv, avg_next)
        v_new, st1 = bnlin_forward(cat1, g1, b1, W1, c1, rm1, rv1, tr1, mo1, ep1, v, nxt_v[:, :C], elu_stats=pv, tile_sums=tv)
        _attach_part(nxt_v, pv, tv)
        ctx.ops = (opDi, opDiA)
        stash(ctx, (cat0, cat1, nxt_f), st0, st1)
        ctx.mark_non_differentiable(nxt_v, nxt_f)
        ctx.set_materialize_grads(False)
        return v_new, f_out, nxt_v, nxt_f

    @staticmethod
    def backward(ctx, g_vnew, g_fout, _gv, _gf):
        opDi, opDiA = ctx.ops
        (cat0, cat1, nxt_f), st0, st1 = unstash(ctx)
        C = cat1.shape[1] // 2
        dev = cat1.device
        none9 = (None,) * 9
        if g_vnew is None and g_fout is None:
            return (None,) * 26
        # Every ELU backward of the block is fused: the dgrad GEMM's epilogue sends the first half of a stage's input
        # gradient through the activation (h = dx[:, :C]·elu'(e) + the gradient of the other branch), and the transposed
        # product's store does the same for the propagated half:  (opᵀ·dx[:, C:])·elu'(e) + h.
        # ---- second stage (vertex rows) ----
        g_fo = g_fout.contiguous() if g_fout is not None else None
        gp1 = none9
        h1 = None                                                                   # dx1[:, :C]·elu'(e_v) + g_vnew
        if g_vnew is not None:
            g_vnew = g_vnew.contiguous()
            (dx1_hi, h1), dg1, db1, dW1, dc1 = bnlin_backward(st1, g_vnew, through_elu=(g_vnew,))
            gp1 = (dg1, db1, dW1, dc1, None, None, None, None, None)
            g_sum = torch.empty((nxt_f.shape[0], C), dtype=torch.float32, device=dev)
            # (DiA^T·dx1_hi)·elu'(e_f)  +  the gradient f_out receives from the next block
            _launch(opDiA.t(), dx1_hi, g_sum, 4, "bwd", elubwd=(nxt_f[:, :C], g_fo))
            g_fo = g_sum
        # ---- first stage (face rows) ----
        gp0 = none9
        g_v = g_f = None
        dx0_hi = None
        if g_fo is not None and ctx.f_zero:
            dx0_hi, dg0, db0, dW0, dc0 = bnlin_backward_zero_first(st0, g_fo)          # no gradient for the zero half
            gp0 = (dg0, db0, dW0, dc0, None, None, None, None, None)
        elif g_fo is not None:
            (dx0_hi, g_f), dg0, db0, dW0, dc0 = bnlin_backward(st0, g_fo, through_elu=(None,))   # g_f = dx0[:, :C]·elu'(e_f)
            gp0 = (dg0, db0, dW0, dc0, None, None, None, None, None)
        if ctx.needs_input_grad[0]:
            if dx0_hi is not None:
                g_v = torch.empty((cat1.shape[0], C), dtype=torch.float32, device=dev)
                _launch(opDi.t(), dx0_hi, g_v, 4, "bwd", elubwd=(cat1[:, :C], h1))   # (Di^T·dx0_hi)·elu'(e_v) + h1
            else:
                g_v = h1
        if ctx.f_zero or not ctx.needs_input_grad[1]:
            g_f = None
        return (g_v, g_f, None, None, None, None, None, None) + gp0 + gp1


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
                                                  avg_next is not False, *_bn_args(mod.bn_fc0), *_bn_args(mod.bn_fc1))
    return attach_activated(v_new.view(B, V, C), nxt_v), attach_activated(f_out.view(B, F_, C), nxt_f)


def zero_faces_ok(mod, C: int) -> bool:
    """Can dirac_block take f=None (all-zero face features at half width) for this module?"""
    return zero_first_supported(C, mod.bn_fc0.fc.weight.shape[0]) and mod.bn_fc0.fc.weight.shape[1] == 2 * C


# ------------------------------------------------------------------------------------------------------------
class _PropagateBlock(torch.autograd.Function):
    """LapResNet2 (utils_pt.py:159-180) and AvgResNet2 (utils_pt.py:230-243): two stages  [e, P(e)] -> Lin(BN(.))  with the
    same propagation P — the sparse product with L, or the per-mesh masked mean broadcast back (global_average)."""

    @staticmethod
    def forward(ctx, x, op, mask_rows, inv_count, nseg, pre, avg_next, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, g1, b1, W1, c1,
                rm1, rv1, tr1, mo1, ep1):
        x = _rows2d(x)
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
        ctx.op, ctx.seg = op, (mask_rows, inv_count, nseg, per)
        stash(ctx, (cat_a, cat_b), st0, st1)
        ctx.mark_non_differentiable(nxt)
        ctx.set_materialize_grads(False)          # else the engine fills a (rows, 2C) zero gradient for `nxt` every backward
        return out, nxt

    @staticmethod
    def backward(ctx, g_out, _gn):
        op = ctx.op
        mask_rows, inv_count, nseg, per = ctx.seg
        (cat_a, cat_b), st0, st1 = unstash(ctx)
        C = cat_a.shape[1] // 2
        if g_out is None:
            return (None,) * 25
        g_out = g_out.contiguous()

        def stage_backward(st, g_in, cat, gadd):
            """gradient w.r.t. the stage input: ((dcat[:, :C] + P^T dcat[:, C:]) * elu'(e)) + gadd, and the parameter
            gradients of the stage's BatchNorm+Linear."""
            if op is not None:                       # sparse propagation: both ELU backward passes fused (see _DiracBlock)
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
        if not ctx.needs_input_grad[0]:
            g_x = None
        return (g_x, None, None, None, None, None, None, dg0, db0, dW0, dc0, None, None, None, None, None, dg1, db1, dW1, dc1, None,
                None, None, None, None)


class _AvgBlock(torch.autograd.Function):
    """AvgResNet2 (utils_pt.py:230-243) at half width (functional.avg_stage_forward): the broadcast mean is never written,
    both Linear layers run over C instead of 2C columns, and the whole backward of a stage — BatchNorm tail, mean-path
    gradient, ELU derivative, residual-path gradient — leaves the dgrad GEMM's epilogue."""

    @staticmethod
    def forward(ctx, x, mask_rows, inv_count, nseg, pre, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, g1, b1, W1, c1, rm1, rv1,
                tr1, mo1, ep1):
        x = _rows2d(x)
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
        stash(ctx, st0, st1, (mask_rows, inv_count))
        ctx.seg = (nseg, per)
        ctx.mark_non_differentiable(nxt)
        ctx.set_materialize_grads(False)          # else the engine fills a (rows, 2C) zero gradient for `nxt` every backward
        return out, nxt

    @staticmethod
    def backward(ctx, g_out, _gn):
        st0, st1, (mask_rows, inv_count) = unstash(ctx)
        nseg, per = ctx.seg
        if g_out is None:
            return (None,) * 23
        g_out = g_out.contiguous()
        g_h, dg1, db1, dW1, dc1 = avg_stage_backward(st1, mask_rows, inv_count, nseg, per, g_out, None)
        g_x, dg0, db0, dW0, dc0 = avg_stage_backward(st0, mask_rows, inv_count, nseg, per, g_h, g_out)   # + residual path
        if not ctx.needs_input_grad[0]:
            g_x = None
        return (g_x, None, None, None, None, dg0, db0, dW0, dc0, None, None, None, None, None, dg1, db1, dW1, dc1, None,
                None, None, None, None)


class _AvgBlockRagged(torch.autograd.Function):
    """_AvgBlock on a PACKED batch: `seg` (operators.PackedSegments) gives the meshes' row ranges; no mask, every row is real
    (BatchNorm over real rows only — the reference's padded batch includes the padding rows, utils_pt.py:97-99)."""

    @staticmethod
    def forward(ctx, x, seg, pre, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, g1, b1, W1, c1, rm1, rv1, tr1, mo1, ep1):
        x = _rows2d(x)
        rows, C = x.shape
        cat = _activated(x, pre)
        e_a = cat[:, :C]
        e_b = torch.empty((rows, C), dtype=torch.float32, device=x.device)
        pb = _new_part(rows, C, x.device)
        tb = _new_tiles(rows, C, x.device, pb, True)      # (as _AvgBlock: means and statistics from what the producing GEMM left)
        _, st0 = avg_stage_forward_ragged(e_a, seg, g0, b0, W0, c0, rm0, rv0, mo0, ep0, None, e_b, want_y=False, elu_stats=pb,
                                          part=getattr(cat, "_sn_part", None), tile_sums=tb, e_tiles=_tiles_of(cat))
        nxt = _new_cat(rows, C, x.device)
        pn = _new_part(rows, C, x.device)
        out, st1 = avg_stage_forward_ragged(e_b, seg, g1, b1, W1, c1, rm1, rv1, mo1, ep1, x, nxt[:, :C], elu_stats=pn, part=pb,
                                            e_tiles=(tb, pb) if tb is not None else None)
        _attach_part(nxt, pn)
        stash(ctx, st0, st1)
        ctx.seg = seg
        ctx.mark_non_differentiable(nxt)
        ctx.set_materialize_grads(False)
        return out, nxt

    @staticmethod
    def backward(ctx, g_out, _gn):
        st0, st1 = unstash(ctx)
        if g_out is None:
            return (None,) * 21
        g_out = g_out.contiguous()
        g_h, dg1, db1, dW1, dc1 = avg_stage_backward_ragged(st1, ctx.seg, g_out, None)
        g_x, dg0, db0, dW0, dc0 = avg_stage_backward_ragged(st0, ctx.seg, g_h, g_out)      # + residual path
        if not ctx.needs_input_grad[0]:
            g_x = None
        return (g_x, None, None, dg0, db0, dW0, dc0, None, None, None, None, None, dg1, db1, dW1, dc1, None, None, None, None,
                None)


def avg_block_ragged_ok(mod, seg, inputs) -> bool:
    C = inputs.shape[-1]
    a0, a1 = _bn_args(mod.bn_fc0), _bn_args(mod.bn_fc1)
    return bool(a0[6] and a1[6]) and inputs.dtype == torch.float32 and mod.bn_fc0.fc.weight.shape == (C, 2 * C) and \
        mod.bn_fc1.fc.weight.shape == (C, 2 * C) and kernels.avg_stage_ragged_supported(C, C, seg) and \
        seg.rows == inputs.shape[0] * inputs.shape[1]


def avg_block_ragged(mod, seg, inputs):
    """AvgResNet2 on a packed (1, sum V_i, C) batch as one autograd node at half width."""
    B, V, C = inputs.shape
    rows = B * V
    out, nxt = _AvgBlockRagged.apply(inputs.reshape(rows, C), seg, take_activated(inputs, rows, C), *_bn_args(mod.bn_fc0),
                                     *_bn_args(mod.bn_fc1))
    return attach_activated(out.view(B, V, C), nxt)


class _EluConv(torch.autograd.Function):
    """GraphConv1x1("pre")(F.elu(v)) — the models' last layer (src/as_rigid_as_possible/models.py:148-150) — as one node:
    elu(v) is the activated hand-off of the preceding block when there is one (no ELU pass), and the backward runs the
    BatchNorm tail and the activation derivative as one pass."""

    @staticmethod
    def forward(ctx, v, pre, g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0):
        v = _rows2d(v)
        C = v.shape[1]
        cat = _activated(v, pre)
        part = getattr(cat, "_sn_part", None)        # statistics of elu(v) left by the GEMM that wrote it
        pre_stats = kernels.colstats_from_part(part, v.shape[0]) if (part is not None and tr0 and C == 128) else None
        y, st = bnlin_forward(cat[:, :C], g0, b0, W0, c0, rm0, rv0, tr0, mo0, ep0, pre_stats=pre_stats)
        stash(ctx, st)
        return y

    @staticmethod
    def backward(ctx, dy):
        (st,) = unstash(ctx)
        g_v, dg, db, dW, dc = bnlin_backward_elu_input(st, dy)
        if not ctx.needs_input_grad[0]:
            g_v = None
        return g_v, None, dg, db, dW, dc, None, None, None, None, None


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
    out, nxt = _PropagateBlock.apply(inputs.reshape(rows, C), op, None, None, 0, take_activated(inputs, rows, C), avg_next is not False,
                                     *_bn_args(mod.bn_fc0), *_bn_args(mod.bn_fc1))
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
    a0, a1 = _bn_args(mod.bn_fc0), _bn_args(mod.bn_fc1)
    if a0[6] and a1[6] and kernels.avg_stage_supported(C, mod.bn_fc0.fc.weight.shape[0], V) and \
            mod.bn_fc0.fc.weight.shape[1] == 2 * C and mod.bn_fc1.fc.weight.shape[0] == C:
        out, nxt = _AvgBlock.apply(inputs.reshape(rows, C), mask_rows, inv_count, B, take_activated(inputs, rows, C), *a0, *a1)
    else:
        out, nxt = _PropagateBlock.apply(inputs.reshape(rows, C), None, mask_rows, inv_count, B, take_activated(inputs, rows, C),
                                         False, *a0, *a1)
    return attach_activated(out.view(B, V, C), nxt)
