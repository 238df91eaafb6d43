"""TEST-ONLY: oracle-backed stand-ins for surfacenetworks_amd.kernels so that the product's HOST logic (autograd
Functions, modules, samplers, data-parallel layer) can be exercised on CPU tensors in this GPU-less container.
Installed with pytest's monkeypatch (see the `cpu_kernels` fixture in conftest.py); never imported by the package."""
import numpy as np
import torch

from oracle import c_oracle


def _np(t):
    return t.detach().cpu().numpy()


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def spmm_csr(rowptr, colind, vals, M, K, x, y, group=1):
    N = y.shape[1] // group
    assert x.stride(1) == 1 and y.stride(1) == 1
    c_oracle.spmm_csr_raw(_np(rowptr), _np(colind), _np(vals), M, x.data_ptr(), _ld(x), group, N, y.data_ptr(), _ld(y), group)


def spmm_bsr4(b_rowptr, b_colind, b_vals, Mb, Kb, x, y, group=1):
    # expand the blocks back to CSR (explicit zeros kept) and reuse the CSR oracle: same arithmetic order
    brp, bci, bv = _np(b_rowptr), _np(b_colind), _np(b_vals).reshape(-1, 4, 4)
    counts = np.diff(brp)
    rowptr = np.zeros(4 * Mb + 1, np.int32)
    rowptr[1:] = np.cumsum(np.repeat(counts * 4, 4))
    colind = np.empty(rowptr[-1], np.int32)
    vals = np.empty(rowptr[-1], np.float32)
    for br in range(Mb):
        blk = slice(brp[br], brp[br + 1])
        for q in range(4):
            dst = slice(rowptr[4 * br + q], rowptr[4 * br + q + 1])
            colind[dst] = (4 * bci[blk, None] + np.arange(4)[None]).ravel()
            vals[dst] = bv[blk, q, :].ravel()
    spmm_csr(torch.from_numpy(rowptr), torch.from_numpy(colind), torch.from_numpy(vals), 4 * Mb, 4 * Kb, x, y, group)


def _elubwd_epilogue(y, e, g):
    """y <- y * elu'(e) + g in place, through the C oracle's elu_bwd (same element-wise arithmetic as the fused store)."""
    tmp = y.clone()
    elu_bwd(tmp, e, y, False, None, g)


def spmm_csr_elubwd(rowptr, colind, vals, M, K, x, e, g, y, group=1):
    spmm_csr(rowptr, colind, vals, M, K, x, y, group)
    _elubwd_epilogue(y, e, g)


def spmm_bsr4_elubwd(b_rowptr, b_colind, b_vals, Mb, Kb, x, e, g, y, group=1):
    spmm_bsr4(b_rowptr, b_colind, b_vals, Mb, Kb, x, y, group)
    _elubwd_epilogue(y, e, g)


def _q3_to_bsr4(q_blk):
    q = _np(q_blk)
    p1, p2, p3 = q[:, 0], q[:, 1], q[:, 2]
    col = q[:, 3].copy().view(np.int32)
    z = np.zeros_like(p1)
    blocks = np.stack([np.stack([z, p1, p2, p3], 1), np.stack([-p1, z, p3, -p2], 1), np.stack([-p2, -p3, z, p1], 1),
                       np.stack([-p3, p2, -p1, z], 1)], 1).astype(np.float32)
    return torch.from_numpy(col), torch.from_numpy(blocks.reshape(-1))


def spmm_q3(b_rowptr, q_blk, Mb, Kb, x, y, group=1, e=None, g=None, want_absmax=False):
    bci, bv = _q3_to_bsr4(q_blk)
    spmm_bsr4(b_rowptr, bci, bv, Mb, Kb, x, y, group)
    if e is not None:
        _elubwd_epilogue(y, e, g)


def bsr4_to_q3(b_colind, b_vals):
    b = _np(b_vals).reshape(-1, 4, 4)
    p = b[:, 0, 1:4]
    q = np.zeros((b.shape[0], 4), np.float32)
    q[:, :3] = p
    q[:, 3] = _np(b_colind).astype(np.int32).view(np.float32)
    back = _q3_to_bsr4(torch.from_numpy(q))[1].numpy().reshape(-1, 4, 4)
    flag = 0 if np.array_equal(back, b) else 1
    return torch.from_numpy(q), torch.tensor([flag], dtype=torch.int32)


def coo_to_csr(idx_batch, idx_row, idx_col, B, R, Kb):
    rp, ci = c_oracle.coo_to_csr(None if idx_batch is None else _np(idx_batch), _np(idx_row), _np(idx_col), B, R, Kb)
    return torch.from_numpy(rp), torch.from_numpy(ci)


def csr_transpose(rowptr, colind, vals, M, K):
    return tuple(torch.from_numpy(a) for a in c_oracle.csr_transpose(_np(rowptr), _np(colind), _np(vals), K))


def csr_to_bsr4(rowptr, colind, vals, M, K):
    return tuple(torch.from_numpy(a) for a in c_oracle.csr_to_bsr4(_np(rowptr), _np(colind), _np(vals)))


def blockdiag_concat(pool_rowptr, pool_colind, pool_vals, desc, size0, size1, total, vpe=1):
    if vpe == 4:                                   # Q3 records: the block column lives in the 4th word
        rp, d = _np(pool_rowptr), _np(desc)
        recs = _np(pool_vals).reshape(-1, 4)
        out_rp = np.zeros(d.shape[0] * size0 + 1, np.int32)
        out = np.zeros((total, 4), np.float32)
        for b, (rp_off, e_off, nrows, out_off) in enumerate(d):
            local = rp[rp_off: rp_off + nrows + 1]
            cnt = int(local[-1])
            row_ptr = np.concatenate([local[:-1], np.full(size0 - nrows, cnt, np.int32)]) + out_off
            out_rp[b * size0: (b + 1) * size0] = row_ptr
            r = recs[e_off: e_off + cnt].copy()
            r[:, 3] = (r[:, 3].copy().view(np.int32) + b * size1).view(np.float32)
            out[out_off: out_off + cnt] = r
        out_rp[-1] = total
        return torch.from_numpy(out_rp), None, torch.from_numpy(out.reshape(-1))
    out = c_oracle.blockdiag_concat(_np(pool_rowptr), _np(pool_colind), _np(pool_vals), _np(desc), size0, size1, total, vpe)
    return tuple(torch.from_numpy(a) for a in out)


def blockdiag_concat_ragged(pool_rowptr, pool_colind, pool_vals, desc, total_rows, total_cols, total, vpe=1):
    """numpy restatement of sn_blockdiag_concat_ragged_i32 (include/sn_spmm.h): mesh b owns output rows
    [desc[b][4], desc[b+1][4]), its columns are shifted by desc[b][5]."""
    rp, d = _np(pool_rowptr), _np(desc)
    B = d.shape[0]
    vals = _np(pool_vals).reshape(-1, vpe)
    out_rp = np.zeros(total_rows + 1, np.int32)
    out_v = np.zeros((total, vpe), np.float32)
    out_c = None if vpe == 4 else np.zeros(total, np.int32)
    ci = None if vpe == 4 else _np(pool_colind)
    for b, (rp_off, e_off, nrows, out_off, row0, cshift) in enumerate(d):
        row1 = int(d[b + 1][4]) if b + 1 < B else total_rows
        local = rp[rp_off: rp_off + nrows + 1]
        cnt = int(local[-1])
        out_rp[row0: row1] = np.concatenate([local[:-1], np.full(row1 - row0 - nrows, cnt, np.int32)]) + out_off
        v = vals[e_off: e_off + cnt].copy()
        if vpe == 4:
            v[:, 3] = (v[:, 3].copy().view(np.int32) + np.int32(cshift)).view(np.float32)
        else:
            out_c[out_off: out_off + cnt] = ci[e_off: e_off + cnt] + np.int32(cshift)
        out_v[out_off: out_off + cnt] = v
    out_rp[-1] = total
    return torch.from_numpy(out_rp), (None if out_c is None else torch.from_numpy(out_c)), torch.from_numpy(out_v.reshape(-1))


def validate_csr(rowptr, colind, vals, M, K):
    rp, ci, va = _np(rowptr), _np(colind), _np(vals)
    bad = 0
    if M and rp[0] != 0:
        bad |= 1
    if (np.diff(rp) < 0).any():
        bad |= 2
    if M and rp[-1] != len(ci):
        bad |= 4
    if len(ci) and ((ci < 0) | (ci >= K)).any():
        bad |= 8
    for r in range(M):
        if (np.diff(ci[rp[r]: rp[r + 1]]) <= 0).any():
            bad |= 16
    if not np.isfinite(va).all():
        bad |= 32
    return bad


def segment_colsum_ragged(x, tiles, seg_tile_ptr, nseg, scale=None):
    t = _np(tiles)
    out = np.zeros((nseg, x.shape[1]), np.float64)
    xx = x.detach().double().numpy()
    for g, r0, n in t:
        out[g] += xx[r0: r0 + n].sum(0)
    if scale is not None:
        out *= _np(scale).astype(np.float64)[:, None]
    return torch.from_numpy(out.astype(np.float32))


def bcast_rows_ragged(src, tiles, dst):
    for g, r0, n in _np(tiles):
        dst[r0: r0 + n] = src[g]


def elu_into(src, dst):
    c_oracle.elu_raw(src.data_ptr(), _ld(src), dst.data_ptr(), _ld(dst), src.shape[0], src.shape[1])


def elu_bwd(gdst, out, gsrc, accumulate, gdst2=None, gadd=None):
    c_oracle.elu_bwd_raw(gdst.data_ptr(), _ld(gdst), out.data_ptr(), _ld(out), gsrc.data_ptr(), _ld(gsrc), out.shape[0],
                         out.shape[1], accumulate, None if gdst2 is None else gdst2.data_ptr(), 0 if gdst2 is None else _ld(gdst2),
                         None if gadd is None else gadd.data_ptr(), 0 if gadd is None else _ld(gadd))


def colstats(x):
    return torch.from_numpy(c_oracle.colstats_raw(x.data_ptr(), _ld(x), x.shape[0], x.shape[1]))


def wgrad_supported(J, C):
    return C in (128, 256) and J <= 128 and J % 4 == 0


def avg_merged_supported(J, C, nseg, which=1):
    return False          # the host twins keep the separate steps of a global-average stage


def clear_absmax():
    pass


def absmax_wanted():
    return False          # the host twins have one weight gradient (exact): no bounds are produced or consumed


def tile_sums_supported():
    return False          # the host twins always take the statistics pass over the operand


def new_tile_sums(rows, device):
    return None


def avg_stats_from_tiles(*a, **k):
    raise RuntimeError("host twins: no tile sums")


def avg_stats_from_tiles_ragged(*a, **k):
    raise RuntimeError("host twins: no tile sums")


def note_absmax(t, maxima):
    pass


def take_absmax(t):
    return None


def wgrad(dy, x, center=None, want_colsum=False, bounds=None):
    G = c_oracle.wgrad_raw(dy.data_ptr(), _ld(dy), x.data_ptr(), _ld(x), x.shape[0], dy.shape[1], x.shape[1],
                           None if center is None else _np(center))
    G = torch.from_numpy(G.astype(np.float32))
    return (G, dy.detach().double().sum(0)) if want_colsum else G


def affine_cols_acc(dx, x, B, Cc, center=None):
    c_oracle.affine_cols_acc_raw(dx.data_ptr(), _ld(dx), x.data_ptr(), _ld(x), _np(B), _np(Cc), x.shape[0], x.shape[1],
                                 None if center is None else _np(center))


def affine_cols_elu_bwd(dx, x, B=None, Cc=None, center=None):
    d = dx.double()
    if B is not None:
        d = d + (x.double() - (0.0 if center is None else center.double())) * B.double() + Cc.double()
    d = d * torch.where(x > 0, torch.ones_like(x), x + 1).double()
    dx.copy_(d.float())


def bn_fold(stats, rows, gamma, beta, W, b, eps, momentum, training, running_mean, running_var, num_batches_tracked=None):
    """numpy/double restatement of what nn.BatchNorm1d + the weight folding compute (checker for sn_bn_fold_f32)."""
    if training and num_batches_tracked is not None:
        with torch.no_grad():
            num_batches_tracked.add_(1)
    g, be, Wd = _np(gamma).astype(np.float64), _np(beta).astype(np.float64), _np(W).astype(np.float64)
    if training:
        st = _np(stats)
        mean = st[0] / rows
        var = np.maximum(st[1] / rows - mean * mean, 0.0)
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(torch.from_numpy((momentum * mean).astype(np.float32)))
                running_var.mul_(1 - momentum).add_(torch.from_numpy((momentum * var * rows / max(rows - 1, 1)).astype(np.float32)))
    else:
        mean, var = _np(running_mean).astype(np.float64), _np(running_var).astype(np.float64)
    invstd = 1.0 / np.sqrt(var + eps)
    s = g * invstd
    t = be - mean * s
    s32, t32 = s.astype(np.float32), t.astype(np.float32)
    Wf = (_np(W) * s32[None, :]).astype(np.float32)
    bf = ((0.0 if b is None else _np(b).astype(np.float64)) + Wd @ t32.astype(np.float64)).astype(np.float32)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return f(mean), f(invstd), f(s32), f(t32), f(Wf), f(bf)


def bn_bwd_coeffs(Gc, dystats, W, s, invstd, beta, rows, has_bias):
    G, Wd = _np(Gc).astype(np.float64), _np(W).astype(np.float64)
    sdy = _np(dystats)
    sdy = sdy[0] if sdy.ndim == 2 else sdy
    sd, isd, be = _np(s).astype(np.float64), _np(invstd).astype(np.float64), _np(beta).astype(np.float64)
    a = sdy @ Wd
    p = (Wd * G).sum(0)
    dg = isd * p
    dW = G * sd[None, :] + np.outer(sdy, be)
    f = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return f(dW), (f(sdy) if has_bias else None), f(dg), f(a), f(-(sd * isd * dg) / rows), f(-(sd * a) / rows)


def segment_colsum(x, mask, rows_per_seg, nseg):
    xv = x.detach().double().reshape(nseg, rows_per_seg, x.shape[1])
    if mask is not None:
        xv = xv * mask.detach().double().reshape(nseg, rows_per_seg, 1)
    return xv.sum(1).float()


def bcast_rows(src, dst, rows_per_seg):
    dst.copy_(src.repeat_interleave(rows_per_seg, dim=0))


def elu_bwd_bcast(gdst, out, bias, mask, gsrc, rows_per_seg, gadd=None):
    b = bias.repeat_interleave(rows_per_seg, dim=0)
    m = 1.0 if mask is None else mask.reshape(-1, 1)
    res = torch.addcmul(gdst, b, torch.ones_like(b) * m) * torch.where(out > 0, torch.ones_like(out), out + 1)
    gsrc.copy_(res if gadd is None else res + gadd)


def dirac_from_mesh(V, F):
    """Host restatement through mesh_ops.dirac (pinned to the reference) + the oracle's CSR->BSR4."""
    import scipy.sparse as sp
    from surfacenetworks_amd import mesh_ops

    Vn, Fn = _np(V).astype(np.float64), _np(F).astype(np.int64)
    D, DA = mesh_ops.dirac(Vn, Fn)
    outs = []
    for A in (D, DA.T.tocsr(), DA, D.T.tocsr()):
        A = A.astype(np.float32).tocsr()
        A.sort_indices()
        outs.append(tuple(torch.from_numpy(a) for a in c_oracle.csr_to_bsr4(A.indptr, A.indices, A.data)))
    return tuple(outs)


def laplacian_from_mesh(V, F):
    from surfacenetworks_amd import mesh_ops

    L = mesh_ops.laplacian(_np(V).astype(np.float64), _np(F).astype(np.int64)).astype(np.float32).tocsr()
    L.sort_indices()
    return torch.from_numpy(L.indptr.astype(np.int32)), torch.from_numpy(L.indices.astype(np.int32)), torch.from_numpy(L.data)


def linear_fwd_supported(K, J):
    return J == 128 and K in (128, 256)


def _fill_part(part, y_elu):
    if part is not None:
        part.zero_()
        w = y_elu.shape[1]                       # (64-output layers fill the first 64 of the kernel's 128 columns)
        part[0, 0, :w] = y_elu.double().sum(0)
        part[0, 1, :w] = (y_elu.double() ** 2).sum(0)


def elu_stats_supported():
    return True


def new_elu_stats_part(rows, device):
    return torch.zeros((1, 2, 128), dtype=torch.float64)


def colstats_from_part(part, rows):
    return part.sum(0)


def avg_stats_ragged(m, seg, part, nblk):
    s1 = part[:nblk].sum(0)
    md = m.to(torch.float64)
    ln = torch.from_numpy(np.asarray(seg.lengths, dtype=np.float64))[:, None]
    return torch.cat([s1, torch.stack([(md * ln).sum(0), (md * md * ln).sum(0)])], 1)


def colstats_merge_into(part, out, offset):
    out[:, offset:offset + part.shape[2]] = part.sum(0)


def colstats_into(x, out, offset):
    C = x.shape[1]
    out[:, offset:offset + C] = colstats(x)
    return out


def colstats_halves(x, part, part_hi=None):
    C = x.shape[1] // 2
    out = torch.empty((2, 2 * C), dtype=torch.float64)
    out[:, :C] = part.sum(0)[:, :C] if part is not None else colstats(x[:, :C])
    out[:, C:] = part_hi.sum(0) if part_hi is not None else colstats(x[:, C:])
    return out


def linear_fwd_stats_blocks(rows):
    return 1


def colstats_partial(x):
    return colstats(x).reshape(1, 2, x.shape[1]), 1


def spmm_q3_stats_supported(N, group):
    return N in (16, 32) and group == 4


def spmm_q3_stats(b_rowptr, q_blk, Mb, Kb, x, y, group=4):
    spmm_q3(b_rowptr, q_blk, Mb, Kb, x, y, group)
    part = torch.zeros((1, 2, y.shape[1]), dtype=torch.float64)
    part[0, 0] = y.double().sum(0)
    part[0, 1] = (y.double() ** 2).sum(0)
    return part


def spmm_csr_stats_supported(N, group):
    return N == 128 and group == 1


def spmm_csr_stats(rowptr, colind, vals, M, K, x, y):
    spmm_csr(rowptr, colind, vals, M, K, x, y, 1)
    part = torch.zeros((1, 2, 128), dtype=torch.float64)
    part[0, 0] = y.double().sum(0)
    part[0, 1] = (y.double() ** 2).sum(0)
    return part


def csr_to_rb4(rowptr, colind, vals, M, K):
    """numpy restatement of sn_rb4_count / sn_rb4_fill: rows 4b..4b+3 share one sorted column list, 4 coefficients each."""
    rp, ci, va = _np(rowptr), _np(colind), _np(vals)
    Mb = (M + 3) // 4
    b_ptr = np.zeros(Mb + 1, np.int32)
    cols, coefs = [], []
    for b in range(Mb):
        rows = [r for r in range(4 * b, min(4 * b + 4, M))]
        u = np.unique(np.concatenate([ci[rp[r]: rp[r + 1]] for r in rows])) if rows else np.zeros(0, np.int32)
        blk = np.zeros((len(u), 4), np.float32)
        for q, r in enumerate(rows):
            blk[np.searchsorted(u, ci[rp[r]: rp[r + 1]]), q] = va[rp[r]: rp[r + 1]]
        cols.append(u.astype(np.int32))
        coefs.append(blk)
        b_ptr[b + 1] = b_ptr[b] + len(u)
    nnz = len(ci)
    b_col = np.zeros(nnz, np.int32)
    b_val = np.zeros((nnz, 4), np.float32)
    tot = int(b_ptr[-1])
    if tot:
        b_col[:tot] = np.concatenate(cols)
        b_val[:tot] = np.concatenate(coefs)
    return torch.from_numpy(b_ptr), torch.from_numpy(b_col), torch.from_numpy(b_val)


def spmm_rb4_supported(N, group):
    return group == 1 and N in (64, 128)


def _rb4_to_csr(b_ptr, b_col, b_val, M):
    """Expand RB4 back to CSR with the explicit zeros kept: the oracle then runs the kernel's own arithmetic order."""
    bp, bc, bv = _np(b_ptr), _np(b_col), _np(b_val)
    cnt = np.diff(bp)
    rowptr = np.zeros(M + 1, np.int32)
    rowptr[1:] = np.cumsum(np.repeat(cnt, 4)[:M])
    colind = np.empty(rowptr[-1], np.int32)
    vals = np.empty(rowptr[-1], np.float32)
    for r in range(M):
        sl = slice(bp[r // 4], bp[r // 4 + 1])
        colind[rowptr[r]: rowptr[r + 1]] = bc[sl]
        vals[rowptr[r]: rowptr[r + 1]] = bv[sl, r % 4]
    return torch.from_numpy(rowptr), torch.from_numpy(colind), torch.from_numpy(vals)


def spmm_rb4(b_ptr, b_col, b_val, M, K, x, y, e=None, g=None, want_absmax=False):
    rp, ci, va = _rb4_to_csr(b_ptr, b_col, b_val, M)
    spmm_csr(rp, ci, va, M, K, x, y, 1)
    if e is not None:
        _elubwd_epilogue(y, e, g)


def spmm_rb4_stats(b_ptr, b_col, b_val, M, K, x, y):
    spmm_rb4(b_ptr, b_col, b_val, M, K, x, y)
    part = torch.zeros((1, 2, 128), dtype=torch.float64)
    part[0, 0] = y.double().sum(0)
    part[0, 1] = (y.double() ** 2).sum(0)
    return part


def spmm_ring_supported(N, group, M, K):
    from surfacenetworks_amd import kernels

    return group == 1 and N in (64, 128) and M == K and M >= kernels.RING_MIN_ROWS


def ring_half_window():
    return 160


def csr_band(rowptr, colind, M, K):
    rp, ci = _np(rowptr).astype(np.int64), _np(colind).astype(np.int64)
    if len(ci) == 0:
        return 0, 0, 0
    cnt = np.diff(rp)
    rows = np.repeat(np.arange(M), cnt)
    far = np.abs(ci - rows)
    return int(far.max()), int(cnt.max()), int(np.unique(rows[far > ring_half_window()]).size)


def spmm_ring(rowptr, colind, vals, M, K, x, y, e=None, g=None, want_absmax=False):
    spmm_csr(rowptr, colind, vals, M, K, x, y, 1)                   # (bit-identical to the CSR kernel by contract)
    if e is not None:
        _elubwd_epilogue(y, e, g)


def spmm_ring_stats(rowptr, colind, vals, M, K, x, y):
    spmm_csr(rowptr, colind, vals, M, K, x, y, 1)
    part = torch.zeros((1, 2, 128), dtype=torch.float64)
    part[0, 0] = y.double().sum(0)
    part[0, 1] = (y.double() ** 2).sum(0)
    return part


def linear_fwd(x, W, bias, residual=None, y_elu=None, want_y=True, elu_stats=None, tile_sums=None):
    y = (x.double() @ W.double().t() + bias.double()).float()
    if residual is not None:
        y = y + residual
    if y_elu is not None:
        y_elu.copy_(torch.nn.functional.elu(y))
        _fill_part(elu_stats, y_elu)
    return y if (want_y or y_elu is None) else None


def linear_dgrad_supported(J, C):
    return J == 128 and C in (128, 256)


def linear_dgrad(dy, W, x=None, center=None, B=None, Cc=None):
    dx = (dy.double() @ W.double()).float()
    if B is not None:
        xc = x if center is None else x - center
        dx = dx + torch.addcmul(Cc.expand_as(xc), xc, B.expand_as(xc))
    return dx


def linear_dgrad_elu_supported(J, C):
    return J == 128 and C in (128, 256)


def linear_dgrad_elu(dy, W, x, center, B, Cc, gadd=None):
    dx = linear_dgrad(dy, W, x, center, B, Cc)
    h = W.shape[1] // 2
    gact = torch.empty((dy.shape[0], h), dtype=torch.float32)
    elu_bwd(dx[:, :h], x[:, :h], gact, False, None, gadd)
    return dx[:, h:].contiguous(), gact


def avg_stage_supported(C, J, rows_per_seg):
    return C == 128 and J == 128 and rows_per_seg >= 32


def avg_fwd_prep(segsum, inv_count, rows_per_seg, stats1):
    m = (segsum * inv_count.reshape(-1, 1)).float()
    md = m.double()
    stats = torch.cat([stats1, torch.stack([rows_per_seg * md.sum(0), rows_per_seg * (md * md).sum(0)])], 1)
    return m, stats.contiguous()


def avg_stats(e, mask, inv_count, rows_per_seg, nseg):
    return avg_fwd_prep(segment_colsum(e, mask, rows_per_seg, nseg), inv_count, rows_per_seg, colstats(e))


def seg_affine(A, W, bias):
    out = A.double() @ W.double().t()
    return (out + bias.double() if bias is not None else out).float()


def avg_bwd_gc(G1, seg_dy, m, mu2):
    return torch.cat([G1, (seg_dy.double().t() @ (m.double() - mu2.double())).float()], 1).contiguous()


def avg_bwd_segvec(seg_dy, Wf2, m, mu2, B2, C2, inv_count, rows_per_seg):
    v = seg_dy.double() @ Wf2.double() + rows_per_seg * ((m.double() - mu2.double()) * B2.double() + C2.double())
    return (v * inv_count.double().reshape(-1, 1)).float()


def wgrad_seg(dy, x, center, rows_per_seg, bounds=None):
    G, sdy = wgrad(dy, x, center, want_colsum=True)
    nseg = dy.shape[0] // rows_per_seg
    return G, sdy, dy.double().reshape(nseg, rows_per_seg, -1).sum(1).float()


def wgrad_thin_supported(J, C):
    return 1 <= C <= 8 and J % 4 == 0 and 256 % (J // 4) == 0


def linear_thin_fwd(x, W, bias, y_elu=None):
    y = torch.addmm(bias, x, W.t()) if bias is not None else x.mm(W.t())       # fp32, like the kernel's FMA chain
    if y_elu is not None:
        elu_into(y, y_elu)
    return y


def wgrad_thin(dy, x, want_bias=True):
    G = (dy.double().t() @ x.double()).float()
    return G, (dy.double().sum(0).float() if want_bias else None)


def gather_segments(src, base, rows_per_item, row_stride, length):
    flat = src.reshape(-1)
    idx = base.reshape(-1, 1, 1) + torch.arange(rows_per_item).reshape(1, -1, 1) * row_stride + torch.arange(length).reshape(1, 1, -1)
    return flat[idx]


def gather_segments_ragged(src, base, seg, row_stride, length):
    flat = src.reshape(-1)
    lens = [int(v) for v in seg.lengths]
    parts = [flat[(int(b) + torch.arange(n).reshape(-1, 1) * row_stride + torch.arange(length).reshape(1, -1))] for b, n in zip(base, lens)]
    return torch.cat(parts, 0)


def pair_argmin(GA, pa, GB, pb):
    return torch.argmin(GA[:pb.numel()][:, pa] + GB[pb][:, :pa.numel()], dim=1)


def pair_ce_fwd(S, target, NA, NB):
    x = S[:NA, :NB]
    lse = torch.logsumexp(x, dim=1)
    return lse, lse - x.gather(1, target[:, None]).squeeze(1)


def pair_ce_bwd(S, target, lse, gloss, NA, NB):
    dS = torch.zeros_like(S)
    p = torch.exp(S[:NA, :NB] - lse[:, None])
    p.scatter_add_(1, target[:, None], -torch.ones_like(p[:, :1]))
    dS[:NA, :NB] = p * (gloss.reshape(()) / NA)
    return dS


def pair_fused_fwd(FA, FB, target, NA, NB):
    S = (FA.double() @ FB.double().t()).float()
    lse, rowloss = pair_ce_fwd(S, target, NA, NB)
    return lse, rowloss, (FA, FB)


def pair_fused_bwd(target, lse, gloss, ws, NA, NB, rowsA, rowsB, K):
    FA, FB = ws
    S = (FA.double() @ FB.double().t()).float()
    dS = pair_ce_bwd(S, target, lse, gloss, NA, NB).double()
    return (dS @ FB.double()).float(), (dS.t() @ FA.double()).float()


def masked_smooth_l1_fwd(out2d, target2d, rowmask, scale):
    d = out2d.double() * (1.0 if rowmask is None else rowmask.double().reshape(-1, 1)) - target2d.double()
    a = d.abs()
    return (torch.where(a < 1, 0.5 * d * d, a - 0.5).sum() * scale).float()


def masked_smooth_l1_bwd(out2d, target2d, rowmask, scale, gloss):
    m = torch.ones(out2d.shape[0], 1) if rowmask is None else rowmask.reshape(-1, 1)
    d = out2d * m - target2d
    return (gloss * scale) * m * d.clamp(-1, 1)


def linear_fwd_segbias(x, W, segbias, rows_per_seg, residual=None, y_elu=None, want_y=True, elu_stats=None, tile_sums=None):
    seg = torch.arange(x.shape[0]) // rows_per_seg
    y = (x.double() @ W.double().t()).float() + segbias[seg]
    if residual is not None:
        y = y + residual
    if y_elu is not None:
        elu_into(y, y_elu)
        _fill_part(elu_stats, y_elu)
    return y if want_y else None


def linear_dgrad_eluseg(dy, W, x, center, B, Cc, segvec, rows_per_seg, rowmask=None, gadd=None):
    dx = linear_dgrad(dy, W, x, center, B, Cc)
    seg = torch.arange(x.shape[0]) // rows_per_seg
    add = segvec[seg] if rowmask is None else segvec[seg] * rowmask.reshape(-1, 1)
    pre = (dx + add).contiguous()
    gact = torch.empty_like(pre)
    elu_bwd(pre, x, gact, False, None, gadd)
    return gact


def _seg_of_rows(seg):
    return torch.from_numpy(np.repeat(np.arange(seg.nseg), seg.lengths))


def avg_stage_ragged_supported(C, J, seg):
    return C == 128 and J == 128 and seg.min_len >= 32


def wgrad_slabs(dy, x, center, seg, bounds=None):
    G, sdy = wgrad(dy, x, center, want_colsum=True)
    out = torch.zeros((seg.nseg, dy.shape[1]), dtype=torch.float64)
    out.index_add_(0, _seg_of_rows(seg), dy.double())
    return G, sdy, out.float()


def avg_bwd_segvec_ragged(seg_dy, Wf2, m, mu2, B2, C2, seg):
    lens = torch.from_numpy(seg.lengths.astype(np.float64)).reshape(-1, 1)
    v = seg_dy.double() @ Wf2.double() + lens * ((m.double() - mu2.double()) * B2.double() + C2.double())
    return (v * seg.inv_count.double().reshape(-1, 1)).float()


def linear_fwd_segbias_ragged(x, W, segbias, seg, residual=None, y_elu=None, want_y=True, elu_stats=None, tile_sums=None):
    y = (x.double() @ W.double().t()).float() + segbias[_seg_of_rows(seg)]
    if residual is not None:
        y = y + residual
    if y_elu is not None:
        elu_into(y, y_elu)
        _fill_part(elu_stats, y_elu)
    return y if want_y else None


def linear_dgrad_eluseg_ragged(dy, W, x, center, B, Cc, segvec, seg, gadd=None):
    dx = linear_dgrad(dy, W, x, center, B, Cc)
    pre = (dx + segvec[_seg_of_rows(seg)]).contiguous()
    gact = torch.empty_like(pre)
    elu_bwd(pre, x, gact, False, None, gadd)
    return gact


def install(monkeypatch=None):
    """Patch surfacenetworks_amd.kernels in place (monkeypatch=None: permanent, for spawned worker processes)."""
    from surfacenetworks_amd import kernels

    for name in kernels.__all__:
        fn = globals()[name]
        if monkeypatch is not None:
            monkeypatch.setattr(kernels, name, fn)
        else:
            setattr(kernels, name, fn)
