"""Sliding-window ("ring") Laplacian SpMM (sn_spmm_csr_ring_*: the products at src/utils/utils_pt.py:167,176) against the
C oracle, bit for bit: banded batches, ragged / empty rows, closed meshes (wrap-around rows outside the window: mixed rounds),
rows of more than 8 / 16 / 32 entries, operators with no band at all (every column outside the window), both widths, operands
inside concat buffers, the fused ELU-backward epilogue and the statistics variant; then the dispatch rule of the product."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _banded(M, rng, band, lens_hi=9, long_rows=(), empty_every=0):
    """Square CSR operator whose entries lie within `band` of the diagonal; `long_rows`: (row, entries); rows that are
    multiples of `empty_every` are empty."""
    lens = rng.integers(1, lens_hi, size=M)
    if empty_every:
        lens[::empty_every] = 0
    for r, n in long_rows:
        lens[r] = n
    rows, cols = [], []
    for r in range(M):
        lo, hi = max(0, r - band), min(M, r + band + 1)
        n = min(int(lens[r]), hi - lo)
        if r == band and r < M:
            n = max(n, 1)
        if n:
            c = rng.choice(np.arange(lo, hi), size=n, replace=False)
            if r == band:
                c[0] = 0 if 0 not in c[1:] else c[0]                   # one entry exactly `band` away from the diagonal
            cols.append(np.sort(c))
            rows.append(np.full(n, r))
    rows = np.concatenate(rows) if rows else np.zeros(0, np.int64)
    cols = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    A = sp.csr_matrix((rng.standard_normal(len(rows)).astype(np.float32), (rows, cols)), shape=(M, M))
    A.sort_indices()
    return A


def _mesh_batch(kind, rng):
    from surfacenetworks_amd import mesh_ops

    mats = []
    if kind == "cloth":                     # open grids, ragged sizes (bands 13 .. 60)
        for n, m in [(12, 13), (40, 60), (25, 31), (9, 50), (33, 33)]:
            V, F = mesh_ops.grid_cloth(n, m, rng)
            mats.append(mesh_ops.laplacian(V, F).astype(np.float32))
    elif kind == "torus":                   # closed meshes: the first / last grid rows reach across the whole mesh
        for n, m in [(20, 30), (31, 17), (24, 41)]:
            V, F = mesh_ops.torus_grid(n, m, rng)
            mats.append(mesh_ops.laplacian(V, F).astype(np.float32))
    elif kind == "permuted":                # no band at all: every row gathers from anywhere in its mesh
        for n, m in [(30, 30), (21, 40)]:
            V, F = mesh_ops.grid_cloth(n, m, rng, permute=True)
            mats.append(mesh_ops.laplacian(V, F).astype(np.float32))
    else:                                   # Delaunay discs: valence up to ~10, arbitrary order
        for nv in (150, 400, 233):
            V, F = mesh_ops.delaunay_disc(nv, rng)
            mats.append(mesh_ops.laplacian(V, F).astype(np.float32))
    return mats


def _check(A, rng, N, tag):
    from surfacenetworks_amd import kernels

    M = A.shape[0]
    x = rng.standard_normal((M, N)).astype(np.float32)
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.ravel(), N).reshape(M, N)
    rp, ci, va = dev(A.indptr.astype(np.int32)), dev(A.indices.astype(np.int32)), dev(A.data.astype(np.float32))
    # operands in halves of a concat buffer (row stride 2N), NaN around them
    xcat = torch.full((M, 2 * N), float("nan"), device=DEV)
    xcat[:, :N] = dev(x)
    ycat = torch.full((M, 2 * N), float("nan"), device=DEV)
    kernels.spmm_ring(rp, ci, va, M, M, xcat[:, :N], ycat[:, N:])
    assert np.array_equal(ycat[:, N:].cpu().numpy(), want) and torch.isnan(ycat[:, :N]).all(), (tag, N, "plain")
    e = rng.standard_normal((M, N)).astype(np.float32)
    g = rng.standard_normal((M, N)).astype(np.float32)
    d = np.where(e > 0, np.float32(1), e + np.float32(1)).astype(np.float32)
    for gg in (g, None):
        ye = torch.full((M, N), float("nan"), device=DEV)
        kernels.spmm_ring(rp, ci, va, M, M, dev(x), ye, dev(e), dev(gg) if gg is not None else None)
        w = want * d if gg is None else want * d + gg                  # (a x) * elu'(e) + g, each step rounded to fp32
        assert np.array_equal(ye.cpu().numpy(), w.astype(np.float32)), (tag, N, "epilogue", gg is not None)
        # the same launch leaving the maxima of |y| (one per compute wave; every slot written): the bound the two-piece weight
        # gradient of the layer below takes for its dy operand
        ya = torch.full((M, N), float("nan"), device=DEV)
        am = kernels.spmm_ring(rp, ci, va, M, M, dev(x), ya, dev(e), dev(gg) if gg is not None else None, want_absmax=True)
        if A.nnz > 0:
            assert am is not None and torch.equal(ya, ye), (tag, N, "epilogue with maxima")
            assert bool(torch.isfinite(am).all()) and float(am.min()) >= 0.0
            assert float(am.max()) == float(np.abs(w.astype(np.float32)).max()), (tag, N, "maximum")
    if N == 128:
        ys = torch.full((M, N), float("nan"), device=DEV)
        part = kernels.spmm_ring_stats(rp, ci, va, M, M, dev(x), ys)
        assert np.array_equal(ys.cpu().numpy(), want), (tag, "stats y")
        got = part.sum(0).cpu().numpy()
        w64 = want.astype(np.float64)
        ref = np.stack([w64.sum(0), (w64 * w64).sum(0)])
        scale = np.stack([np.abs(w64).sum(0), (w64 * w64).sum(0)]) + 1e-30
        assert (np.abs(got - ref) / scale).max() < 1e-6, (tag, "stats")


@pytest.mark.parametrize("N", [128, 64])
def test_ring_product_is_the_oracles_on_banded_and_unbanded_operators(N):
    rng = np.random.default_rng(17 + N)
    cases = [
        ("tiny", _banded(5, rng, 3)),
        ("one step", _banded(64, rng, 20)),
        ("step + 1", _banded(65, rng, 20, empty_every=7)),
        ("band 160 (the window exactly)", _banded(2000, rng, 160)),
        ("band 161 (one past the window)", _banded(2000, rng, 161)),
        ("band 400 (mostly outside)", _banded(1500, rng, 400, empty_every=11)),
        ("long rows: 9, 17, 33, 70 entries", _banded(1200, rng, 150, long_rows=((3, 9), (64, 17), (65, 33), (700, 70), (1199, 12)))),
        ("dense rows past the entry buffer", _banded(700, rng, 120, lens_hi=3, long_rows=tuple((r, 100) for r in range(200, 264)))),
        ("all rows empty but two", sp.csr_matrix((np.array([2.0, -1.0], np.float32), (np.array([0, 299]), np.array([1, 298]))), shape=(300, 300))),
    ]
    for kind in ("cloth", "torus", "permuted", "delaunay"):
        mats = _mesh_batch(kind, rng)
        size = max(m.shape[0] for m in mats) + 5                       # padded as the reference batches (empty rows between meshes)
        blocks = [sp.block_diag([m, sp.csr_matrix((size - m.shape[0], size - m.shape[0]), dtype=np.float32)]) for m in mats]
        cases.append((kind + " padded", sp.block_diag(blocks).tocsr()))
        cases.append((kind + " packed", sp.block_diag(mats).tocsr()))
    for tag, A in cases:
        A = A.tocsr().astype(np.float32)
        A.sort_indices()
        _check(A, rng, N, tag)


def test_band_of_an_operator_and_the_dispatch_rule(monkeypatch):
    from surfacenetworks_amd import functional as snF, kernels
    from surfacenetworks_amd.operators import OperatorPool, SparseOperator

    rng = np.random.default_rng(5)
    A = _banded(3000, rng, 97, long_rows=((10, 21),))
    op = SparseOperator.from_scipy(A, DEV)
    counts = np.diff(A.indptr)
    rows = np.repeat(np.arange(A.shape[0]), counts)
    assert op.band() == (int(np.abs(A.indices - rows).max()), int(counts.max()), 0) == kernels.csr_band(op.rowptr, op.colind, 3000, 3000)
    assert kernels.ring_half_window() == 160
    assert not op.ring_ok(128)                                          # too few rows for persistent strips
    monkeypatch.setattr(kernels, "RING_MIN_ROWS", 1024)
    assert op.ring_ok(128) and op.ring_ok(64) and not op.ring_ok(32)
    Aw = _banded(3000, rng, 400)
    wide = SparseOperator.from_scipy(Aw, DEV)
    cw = np.diff(Aw.indptr)
    rw = np.repeat(np.arange(3000), cw)
    n_out = int(np.unique(rw[np.abs(Aw.indices - rw) > 160]).size)
    assert wide.band() == (400, int(cw.max()), n_out) and n_out > 0.05 * 3000 and not wide.ring_ok(128)
    few = SparseOperator.from_scipy(_banded(3000, rng, 161), DEV)          # a handful of rows one column past the window
    assert few.band()[0] == 161 and 0 < few.band()[2] <= 0.05 * 3000 and few.ring_ok(128)
    assert not SparseOperator.from_scipy(_banded(3000, rng, 50, long_rows=((7, 33),)), DEV).ring_ok(128)
    # an operator built straight from UNSORTED CSR arrays (columns descending inside every row) must not take the ring kernel —
    # it bounds a row by its first and last entry; the probe reports INT32_MAX as the longest row and the product, through the
    # generic kernels, equals the sorted operator's up to the summation order
    rev = np.concatenate([A.indices[A.indptr[r]:A.indptr[r + 1]][::-1] for r in range(3000)]).astype(np.int32)
    rvals = np.concatenate([A.data[A.indptr[r]:A.indptr[r + 1]][::-1] for r in range(3000)]).astype(np.float32)
    uns = SparseOperator(torch.from_numpy(A.indptr.astype(np.int32)).to(DEV), torch.from_numpy(rev).to(DEV),
                         torch.from_numpy(rvals).to(DEV), (3000, 3000))
    assert uns.band()[1] == 0x7fffffff and uns.band()[0] == op.band()[0] and not uns.ring_ok(128)
    xu = torch.randn(3000, 128, device=DEV)
    snF.set_laplacian_format("ring")
    torch.testing.assert_close(snF.spmm(uns, xu), snF.spmm(op, xu), rtol=1e-5, atol=1e-5)
    # pools hand the band of a batch over without a device pass: the maximum over the selected meshes
    mats = _mesh_batch("cloth", rng)
    pool = OperatorPool(mats, DEV)
    sel = np.array([1, 3, 0])
    for b in (pool.assemble(sel), pool.assemble(sel, 2500, 2500)):
        assert b._band is not None and b._t._band is not None
        attached = b._band
        b._band = None
        assert b.band() == attached                                     # == what the device kernel measures on the assembled batch
    # the product takes the ring kernel for such a batch and the row-blocked form otherwise; same values either way
    big = pool.assemble(np.arange(len(mats)))
    assert big.ring_ok(128)
    x = torch.randn(big.shape[1], 128, device=DEV)
    with snF.SpmmTimer() as timer:
        y_ring = snF.spmm(big, x)
        snF.set_laplacian_format("rb4")
        try:
            y_rb4 = snF.spmm(big, x)
        finally:
            snF.set_laplacian_format("ring")
        tags = [t[0] for t in timer.results()]
    assert tags[0].endswith("/ring") and tags[1].endswith("/rb4") and torch.equal(y_ring, y_rb4)


def test_ring_argument_checks():
    from surfacenetworks_amd import _lib

    lib = _lib.load()
    one = torch.zeros(256, device=DEV)
    onei = torch.zeros(8, dtype=torch.int32, device=DEV)
    SN_E_NULL, SN_E_RANGE, SN_E_UNSUPPORTED = -1, -3, -7
    p = lambda t: t.data_ptr()
    f = lib.sn_spmm_csr_ring_f32
    assert f(p(onei), p(onei), p(one), 4, 5, 4, p(one), 128, 128, p(one), 128, None) == SN_E_UNSUPPORTED      # not square
    assert f(p(onei), p(onei), p(one), 4, 4, 4, p(one), 32, 32, p(one), 32, None) == SN_E_UNSUPPORTED         # N = 32
    assert f(p(onei), p(onei), p(one), 2 ** 31, 2 ** 31, 4, p(one), 128, 128, p(one), 128, None) == SN_E_RANGE
    assert f(None, p(onei), p(one), 4, 4, 4, p(one), 128, 128, p(one), 128, None) == SN_E_NULL
    assert f(p(onei), p(onei), p(one), 0, 0, 0, p(one), 128, 128, p(one), 128, None) == 0
    assert lib.sn_spmm_csr_ring_elubwd_f32(p(onei), p(onei), p(one), 4, 4, 4, p(one), 128, 128, None, 128, None, 0, p(one), 128, None) == SN_E_NULL
