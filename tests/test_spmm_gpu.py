"""GPU parity of the HIP kernels (through the C-ABI) against the C oracle: bit-exact for SpMM and the
integer/index work, 1e-6 relative for ELU (expm1f implementations differ by an ulp)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from helpers import mesh_fixture, random_csr, rel_err
from oracle import c_oracle

pytestmark = pytest.mark.gpu

from surfacenetworks_amd import functional as snF, kernels  # noqa: E402
from surfacenetworks_amd.operators import OperatorPool, SparseOperator, as_operator  # noqa: E402

DEV = "cuda"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def csr_dev(A):
    return dev(A.indptr.astype(np.int32)), dev(A.indices.astype(np.int32)), dev(A.data.astype(np.float32))


@pytest.mark.parametrize("N", [16, 32, 64, 128, 1, 3, 20, 48])
@pytest.mark.parametrize("shape", [(257, 190), (64, 64), (1, 7), (1000, 40)])
def test_spmm_csr_bit_exact(N, shape):
    M, K = shape
    A = random_csr(M, K, 0.05, seed=M + N, empty_rows=[0, M // 2, M - 1])
    X = np.random.default_rng(N).standard_normal((K, N)).astype(np.float32)
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, X.ravel(), N).reshape(M, N)
    y = torch.full((M, N), float("nan"), device=DEV)          # every element must be written
    kernels.spmm_csr(*csr_dev(A), M, K, dev(X), y, 1)
    got = y.cpu().numpy()
    assert np.array_equal(got, want)
    assert rel_err(got, A.astype(np.float64) @ X.astype(np.float64)) < 1e-6


@pytest.mark.parametrize("N", [16, 32, 64, 128])
@pytest.mark.parametrize("kind", ["cloth", "cloth_perm", "torus", "delaunay"])
@pytest.mark.parametrize("which", ["Di", "DiA"])
def test_dirac_csr_and_bsr4_group_layouts(N, kind, which):
    """Quaternion (group=4) view, contiguous and embedded in a 2C-wide concat buffer, CSR and BSR4."""
    _, _, ops = mesh_fixture(kind)
    A = ops[which]
    M, K = A.shape
    C = 4 * N
    rng = np.random.default_rng(1)
    xcat = rng.standard_normal((K // 4, 2 * C)).astype(np.float32)      # X lives in the FIRST half
    X = np.ascontiguousarray(xcat[:, :C])
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, X.ravel(), N).reshape(M // 4, C)
    # oracle with the strided addressing must agree with the contiguous one
    ycat_o = np.zeros((M // 4, 2 * C), np.float32)
    c_oracle.spmm_csr(A.indptr, A.indices, A.data, xcat.ravel(), N, ldx=2 * C, xg=4, Y=ycat_o.reshape(-1)[C:],
                      ldy=2 * C, yg=4, M=M)
    assert np.array_equal(ycat_o[:, C:], want)
    rp, ci, va = csr_dev(A)
    xc = dev(xcat)
    ycat = torch.full((M // 4, 2 * C), float("nan"), device=DEV)
    # CSR, strided in and out (second half of the output buffer)
    kernels.spmm_csr(rp, ci, va, M, K, xc[:, :C], ycat[:, C:], 4)
    assert np.array_equal(ycat[:, C:].cpu().numpy(), want)
    assert torch.isnan(ycat[:, :C]).all()                                 # the other half is untouched
    # BSR4 built on the device vs the oracle's conversion, then the product
    b = kernels.csr_to_bsr4(rp, ci, va, M, K)
    bo = c_oracle.csr_to_bsr4(A.indptr, A.indices, A.data)
    for g_, o_ in zip(b, bo):
        assert np.array_equal(g_.cpu().numpy(), o_)
    y2 = torch.full((M // 4, C), float("nan"), device=DEV)
    kernels.spmm_bsr4(b[0], b[1], b[2], M // 4, K // 4, xc[:, :C], y2, 4)
    assert np.array_equal(y2.cpu().numpy(), want)
    # group = 1 view of the same contiguous data gives the same numbers
    y3 = torch.empty((M, N), device=DEV)
    kernels.spmm_bsr4(b[0], b[1], b[2], M // 4, K // 4, dev(X).view(K, N), y3, 1)
    assert np.array_equal(y3.cpu().numpy().reshape(M // 4, C), want)
    assert rel_err(want.reshape(M, N), A.astype(np.float64) @ X.reshape(K, N).astype(np.float64)) < 1e-6


@pytest.mark.parametrize("N", [64, 128])
@pytest.mark.parametrize("kind", ["cloth", "delaunay"])
def test_laplacian_strided(N, kind):
    _, _, ops = mesh_fixture(kind)
    L = ops["L"]
    V = L.shape[0]
    xcat = np.random.default_rng(2).standard_normal((V, 2 * N)).astype(np.float32)
    want = c_oracle.spmm_csr(L.indptr, L.indices, L.data, np.ascontiguousarray(xcat[:, :N]).ravel(), N).reshape(V, N)
    cat = dev(xcat)
    kernels.spmm_csr(*csr_dev(L), V, V, cat[:, :N], cat[:, N:], 1)
    assert np.array_equal(cat[:, N:].cpu().numpy(), want)
    assert np.array_equal(cat[:, :N].cpu().numpy(), xcat[:, :N])


def test_transpose_matches_oracle_and_scipy():
    for kind in ["cloth", "delaunay"]:
        _, _, ops = mesh_fixture(kind)
        for name in ["L", "Di", "DiA"]:
            A = ops[name]
            M, K = A.shape
            t = kernels.csr_transpose(*csr_dev(A), M, K)
            to = c_oracle.csr_transpose(A.indptr, A.indices, A.data, K)
            for g_, o_ in zip(t, to):
                assert np.array_equal(g_.cpu().numpy(), o_)
            T = A.T.tocsr()
            T.sort_indices()
            assert np.array_equal(t[0].cpu().numpy(), T.indptr) and np.array_equal(t[1].cpu().numpy(), T.indices)
    # rectangular with empty rows/cols
    A = random_csr(300, 77, 0.03, seed=5, empty_rows=[0, 10, 299])
    t = kernels.csr_transpose(*csr_dev(A), 300, 77)
    to = c_oracle.csr_transpose(A.indptr, A.indices, A.data, 77)
    for g_, o_ in zip(t, to):
        assert np.array_equal(g_.cpu().numpy(), o_)


def test_large_scan_path():
    """Transpose of an operator with > 2048*256 columns exercises the multi-block scan."""
    K = 700_000
    rng = np.random.default_rng(3)
    M = 5000
    cols = np.sort(rng.choice(K, size=(M, 6)), axis=1)
    A = sp.csr_matrix((rng.standard_normal(M * 6).astype(np.float32), cols.ravel(), np.arange(0, 6 * M + 1, 6)), shape=(M, K))
    A.sum_duplicates()
    A.sort_indices()
    t = kernels.csr_transpose(*csr_dev(A), M, K)
    to = c_oracle.csr_transpose(A.indptr, A.indices, A.data, K)
    for g_, o_ in zip(t, to):
        assert np.array_equal(g_.cpu().numpy(), o_)


def test_coo_to_csr_2d_3d_and_empty_rows():
    _, _, ops = mesh_fixture("cloth")
    A = ops["Di"].tocoo()
    order = np.lexsort((A.col, A.row))
    r, c = A.row[order].astype(np.int64), A.col[order].astype(np.int64)
    rp, ci = kernels.coo_to_csr(None, dev(r), dev(c), 1, A.shape[0], A.shape[1])
    rpo, cio = c_oracle.coo_to_csr(None, r, c, 1, A.shape[0], A.shape[1])
    assert np.array_equal(rp.cpu().numpy(), rpo) and np.array_equal(ci.cpu().numpy(), cio)
    # 3-D batched with interior empty rows (the case batch_csr.cu gets wrong)
    B, R, Kb = 3, 40, 25
    rng = np.random.default_rng(4)
    ent = []
    for b in range(B):
        for rr in range(R):
            if rr % 7 == 3 or (b == 1 and rr < 5):
                continue
            for cc in np.sort(rng.choice(Kb, 3, replace=False)):
                ent.append((b, rr, cc))
    ent = np.array(ent, dtype=np.int64)
    rp, ci = kernels.coo_to_csr(dev(ent[:, 0]), dev(ent[:, 1]), dev(ent[:, 2]), B, R, Kb)
    rpo, cio = c_oracle.coo_to_csr(ent[:, 0], ent[:, 1], ent[:, 2], B, R, Kb)
    assert np.array_equal(rp.cpu().numpy(), rpo) and np.array_equal(ci.cpu().numpy(), cio)
    S = sp.csr_matrix((np.ones(len(ent), np.float32), (ent[:, 0] * R + ent[:, 1], ent[:, 0] * Kb + ent[:, 2])), shape=(B * R, B * Kb))
    S.sort_indices()
    assert np.array_equal(rpo, S.indptr) and np.array_equal(cio, S.indices)


def test_pool_assemble_matches_blockdiag():
    meshes = [mesh_fixture(k)[2] for k in ["cloth", "delaunay", "torus", "cloth_perm"]]
    for name, bsr in [("L", False), ("Di", True), ("DiA", True)]:
        mats = [m[name] for m in meshes]
        pool = OperatorPool(mats, DEV, want_bsr4=bsr)
        sel = [2, 0, 3, 0, 1]
        s0 = max(m.shape[0] for m in mats) + (4 if bsr else 3)
        s1 = max(m.shape[1] for m in mats) + (8 if bsr else 1)
        op = pool.assemble(sel, s0, s1)
        blocks = []
        for i in sel:
            P = sp.lil_matrix((s0, s1), dtype=np.float32)
            P[: mats[i].shape[0], : mats[i].shape[1]] = mats[i]
            blocks.append(P.tocsr())
        want = sp.block_diag(blocks, format="csr")
        want.sort_indices()
        got = op.to_scipy()
        assert got.shape == want.shape
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        assert np.array_equal(got.data, want.data)
        gt, wt = op.t().to_scipy(), want.T.tocsr()
        wt.sort_indices()
        assert np.array_equal(gt.indptr, wt.indptr) and np.array_equal(gt.indices, wt.indices) and np.array_equal(gt.data, wt.data)
        if bsr:
            bo = c_oracle.csr_to_bsr4(want.indptr, want.indices, want.data)
            for g_, o_ in zip(op.bsr4(), bo):
                assert np.array_equal(g_.cpu().numpy(), o_)
            bt = c_oracle.csr_to_bsr4(wt.indptr, wt.indices, wt.data)
            for g_, o_ in zip(op.t().bsr4(), bt):
                assert np.array_equal(g_.cpu().numpy(), o_)


def test_elu_kernels():
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((333, 128)) * 3).astype(np.float32)
    x[0, :4] = [0.0, -0.0, 1e-8, -1e-8]
    cat = torch.zeros((333, 256), device=DEV)
    kernels.elu_into(dev(x), cat[:, :128])
    want = c_oracle.elu(x)
    got = cat[:, :128].cpu().numpy()
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)
    assert torch.equal(cat[:, 128:], torch.zeros_like(cat[:, 128:]))
    g = rng.standard_normal((333, 128)).astype(np.float32)
    acc0 = rng.standard_normal((333, 128)).astype(np.float32)
    out = torch.empty((333, 128), device=DEV)
    kernels.elu_bwd(dev(g), dev(want), out, False)
    assert np.allclose(out.cpu().numpy(), c_oracle.elu_bwd(g, want), rtol=1e-6, atol=1e-7)
    acc = dev(acc0)
    kernels.elu_bwd(dev(g), dev(want), acc, True)
    assert np.allclose(acc.cpu().numpy(), c_oracle.elu_bwd(g, want, acc0), rtol=1e-6, atol=1e-7)
    # two gradient inputs in one pass
    g2 = rng.standard_normal((333, 128)).astype(np.float32)
    kernels.elu_bwd(dev(g), dev(want), out, False, dev(g2))
    assert np.allclose(out.cpu().numpy(), c_oracle.elu_bwd(g + g2, want), rtol=1e-6, atol=1e-6)
    # ... plus a term added after the derivative (residual-path gradient)
    g3 = rng.standard_normal((333, 128)).astype(np.float32)
    kernels.elu_bwd(dev(g), dev(want), out, False, dev(g2), dev(g3))
    assert np.allclose(out.cpu().numpy(), c_oracle.elu_bwd(g + g2, want) + g3, rtol=1e-6, atol=1e-6)
    # odd channel count takes the scalar kernel
    x3 = x[:, :7].copy()
    o3 = torch.empty((333, 7), device=DEV)
    kernels.elu_into(dev(x3), o3)
    assert np.allclose(o3.cpu().numpy(), c_oracle.elu(x3), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("fmt", ["csr", "bsr4", "q3"])
def test_autograd_spmm_forward_backward(fmt):
    snF.set_dirac_format(fmt)
    try:
        _, _, ops = mesh_fixture("cloth")
        A = ops["DiA"]
        M, K = A.shape
        N = 32
        rng = np.random.default_rng(7)
        x = rng.standard_normal((K // 4, 4 * N)).astype(np.float32)
        g = rng.standard_normal((M // 4, 4 * N)).astype(np.float32)
        xt = dev(x).requires_grad_(True)
        op = SparseOperator.from_scipy(A, DEV)
        y = snF.spmm(op, xt, group=4)
        y.backward(dev(g))
        want_y = c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.ravel(), N).reshape(M // 4, 4 * N)
        tr = c_oracle.csr_transpose(A.indptr, A.indices, A.data, K)
        want_g = c_oracle.spmm_csr(tr[0], tr[1], tr[2], g.ravel(), N).reshape(K // 4, 4 * N)
        assert np.array_equal(y.detach().cpu().numpy(), want_y)
        assert np.array_equal(xt.grad.cpu().numpy(), want_g)
        # same through a torch sparse COO operator (the reference drivers' type), and vs torch.mm on the device
        coo = A.tocoo()
        Ac = torch.sparse_coo_tensor(np.stack([coo.row, coo.col]).astype(np.int64), coo.data, A.shape).coalesce().to(DEV)
        xt2 = dev(x).requires_grad_(True)
        y2 = snF.spmm(Ac, xt2, group=4)
        y2.backward(dev(g))
        assert np.array_equal(y2.detach().cpu().numpy(), want_y) and np.array_equal(xt2.grad.cpu().numpy(), want_g)
        assert as_operator(Ac) is as_operator(Ac)            # converted once per tensor
    finally:
        snF.set_dirac_format("q3")


def test_cpu_tensors_raise():
    A = random_csr(8, 8, 0.3, 1)
    op = SparseOperator.from_scipy(A, "cpu")
    with pytest.raises(RuntimeError, match="MI355X only"):
        snF.spmm(op, torch.zeros(8, 4))


def test_error_codes():
    from surfacenetworks_amd import _lib

    lib = _lib.load()
    y = torch.zeros(4, 16, device=DEV)
    rp = torch.zeros(5, dtype=torch.int32, device=DEV)
    assert lib.sn_spmm_csr_f32(None, None, None, 4, 4, 0, y.data_ptr(), 16, 1, 16, y.data_ptr(), 16, 1, None) == -1
    assert lib.sn_spmm_csr_f32(rp.data_ptr(), None, None, 4, 4, 0, y.data_ptr(), 8, 1, 16, y.data_ptr(), 16, 1, None) == -4
    assert lib.sn_spmm_csr_f32(rp.data_ptr(), None, None, -1, 4, 0, y.data_ptr(), 16, 1, 16, y.data_ptr(), 16, 1, None) == -2
    assert lib.sn_spmm_csr_f32(rp.data_ptr(), None, None, 2**31, 4, 0, y.data_ptr(), 16, 1, 16, y.data_ptr(), 16, 1, None) == -3
    assert lib.sn_bsr4_count(rp.data_ptr(), None, 6, 8, rp.data_ptr(), rp.data_ptr(), 64, None) == -7
    # empty operator: all-zero output, every element written
    y.fill_(float("nan"))
    kernels.spmm_csr(rp, torch.zeros(0, dtype=torch.int32, device=DEV), torch.zeros(0, device=DEV), 4, 4,
                     torch.ones(4, 16, device=DEV), y, 1)
    assert torch.equal(y, torch.zeros_like(y))


def test_timing_facility_matches_event_bracketing():
    """sn_timing_*: per-launch kernel durations via hipExtLaunchKernelGGL; count/order/metadata and plausibility."""
    _, _, ops = mesh_fixture("cloth")
    A = ops["Di"]
    M, K = A.shape
    op = SparseOperator.from_scipy(A, DEV)
    x = torch.randn(K // 4, 128, device=DEV, requires_grad=True)
    with snF.SpmmTimer() as t:
        for _ in range(3):
            y = snF.spmm(op, x, 4)
        y.sum().backward()
    recs = t.results()
    assert [r[0] for r in recs] == ["fwd/q3"] * 3 + ["bwd/q3"]        # quaternion-packed form is the default for Dirac operators
    assert all(r[1:3] == (M, K) for r in recs[:3]) and recs[3][1:3] == (K, M)
    assert all(r[3] == A.nnz and r[4] == 32 for r in recs)
    assert all(1e-4 < r[5] < 5.0 for r in recs)
    # facility is off again: nothing recorded
    snF.spmm(op, x, 4)
    from surfacenetworks_amd import _lib
    assert _lib.load().sn_timing_count() == 0


def _elubwd_want(prod, e, g):
    """(A·x) * elu'(e) + g with the derivative expressed through the activation output, fp32 like the kernel."""
    f = np.where(e > 0, np.float32(1), e + np.float32(1)).astype(np.float32)
    out = prod * f
    return out if g is None else out + g


@pytest.mark.parametrize("with_g", [True, False])
@pytest.mark.parametrize("N", [32, 128])
@pytest.mark.parametrize("which", ["Di", "DiA"])
def test_fused_elu_backward_epilogue_dirac(which, N, with_g):
    """sn_spmm_*_elubwd_f32 == the plain product followed by the ELU backward, group-4 layout with E a strided half of a
    concat buffer (the way blocks.py calls it); both formats."""
    _, _, ops = mesh_fixture("cloth")
    A = ops[which].T.tocsr()                                 # the backward multiplies by the transpose
    A.sort_indices()
    M, K = A.shape
    C = 4 * N
    rng = np.random.default_rng(7)
    X = rng.standard_normal((K // 4, C)).astype(np.float32)
    ecat = rng.standard_normal((M // 4, 2 * C)).astype(np.float32)
    G = rng.standard_normal((M // 4, C)).astype(np.float32) if with_g else None
    prod = c_oracle.spmm_csr(A.indptr, A.indices, A.data, X.ravel(), N).reshape(M // 4, C)
    want = _elubwd_want(prod, ecat[:, :C], G)
    rp, ci, va = csr_dev(A)
    e_d, g_d = dev(ecat), (dev(G) if with_g else None)
    y = torch.full((M // 4, C), float("nan"), device=DEV)
    kernels.spmm_csr_elubwd(rp, ci, va, M, K, dev(X), e_d[:, :C], g_d, y, 4)
    assert np.allclose(y.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    b = kernels.csr_to_bsr4(rp, ci, va, M, K)
    y2 = torch.full((M // 4, C), float("nan"), device=DEV)
    kernels.spmm_bsr4_elubwd(b[0], b[1], b[2], M // 4, K // 4, dev(X), e_d[:, :C], g_d, y2, 4)
    assert np.allclose(y2.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    assert np.array_equal(y.cpu().numpy(), y2.cpu().numpy())          # the two formats agree bit for bit


def test_fused_elu_backward_epilogue_laplacian():
    _, _, ops = mesh_fixture("delaunay")
    L = ops["L"].T.tocsr()
    L.sort_indices()
    V, N = L.shape[0], 128
    rng = np.random.default_rng(8)
    X = rng.standard_normal((V, N)).astype(np.float32)
    ecat = rng.standard_normal((V, 2 * N)).astype(np.float32)
    G = rng.standard_normal((V, N)).astype(np.float32)
    want = _elubwd_want(c_oracle.spmm_csr(L.indptr, L.indices, L.data, X.ravel(), N).reshape(V, N), ecat[:, :N], G)
    y = torch.empty((V, N), device=DEV)
    kernels.spmm_csr_elubwd(*csr_dev(L), V, V, dev(X), dev(ecat)[:, :N], dev(G), y, 1)
    assert np.allclose(y.cpu().numpy(), want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("N", [16, 32, 64, 128])
@pytest.mark.parametrize("kind", ["cloth", "cloth_perm", "torus", "delaunay"])
@pytest.mark.parametrize("which", ["Di", "DiA", "DiT", "DiAT"])
def test_quaternion_packed_dirac_product_is_bit_exact(N, kind, which):
    """sn_spmm_q3_f32: every Dirac-type operator (and its transpose) packs to three floats per block, and the product equals
    the CSR oracle bit for bit, contiguous and strided, plain and with the fused ELU-backward epilogue."""
    _, _, ops = mesh_fixture(kind)
    A = ops[which[:-1]].T.tocsr() if which.endswith("T") else ops[which]
    A.sort_indices()
    M, K = A.shape
    C = 4 * N
    rng = np.random.default_rng(3)
    xcat = rng.standard_normal((K // 4, 2 * C)).astype(np.float32)
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, np.ascontiguousarray(xcat[:, :C]).ravel(), N).reshape(M // 4, C)
    rp, ci, va = csr_dev(A)
    b = kernels.csr_to_bsr4(rp, ci, va, M, K)
    q, flag = kernels.bsr4_to_q3(b[1], b[2])
    assert int(flag.item()) == 0
    assert q.shape == (b[1].numel(), 4) and np.array_equal(q[:, 3].contiguous().view(torch.int32).cpu().numpy(), b[1].cpu().numpy())
    ycat = torch.full((M // 4, 2 * C), float("nan"), device=DEV)
    kernels.spmm_q3(b[0], q, M // 4, K // 4, dev(xcat)[:, :C], ycat[:, C:], 4)
    assert np.array_equal(ycat[:, C:].cpu().numpy(), want)
    assert torch.isnan(ycat[:, :C]).all()
    e = rng.standard_normal((M // 4, C)).astype(np.float32)
    g = rng.standard_normal((M // 4, C)).astype(np.float32)
    y2 = torch.empty((M // 4, C), device=DEV)
    kernels.spmm_q3(b[0], q, M // 4, K // 4, dev(xcat)[:, :C], y2, 4, dev(e), dev(g))
    y3 = torch.empty((M // 4, C), device=DEV)
    kernels.spmm_bsr4_elubwd(b[0], b[1], b[2], M // 4, K // 4, dev(xcat)[:, :C], dev(e), dev(g), y3, 4)
    assert np.array_equal(y2.cpu().numpy(), y3.cpu().numpy())


def test_non_quaternion_blocks_are_detected_and_fall_back():
    rng = np.random.default_rng(5)
    A = random_csr(64, 48, 0.2, 5)                        # generic operator: 4x4 blocks without any structure
    op = SparseOperator.from_scipy(A, DEV)
    assert op.q3() is None
    x = torch.from_numpy(rng.standard_normal((12, 64)).astype(np.float32)).to(DEV)      # group-4 view: K/4 rows
    y = snF.spmm(op, x, group=4)
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.cpu().numpy().ravel(), 16).reshape(16, 64)
    assert np.array_equal(y.cpu().numpy(), want)


def test_pool_assembles_quaternion_packed_batches():
    _, _, o1 = mesh_fixture("cloth")
    _, _, o2 = mesh_fixture("delaunay")
    mats = [o1["Di"], o2["Di"], o1["Di"]]
    pool = OperatorPool(mats, DEV, want_bsr4=True)
    s0 = max(m.shape[0] for m in mats)
    s1 = max(m.shape[1] for m in mats)
    op = pool.assemble([2, 1, 0], s0, s1)
    assert op._q3 is not None and op._bsr4 is None and op._csr is None            # only the packed form was assembled
    want = sp.block_diag([sp.csr_matrix((m.data, m.indices, m.indptr), shape=m.shape).tocsr() if m.shape == (s0, s1) else
                          sp.bmat([[m, None], [None, sp.csr_matrix((s0 - m.shape[0], s1 - m.shape[1]))]]) for m in
                          (mats[2], mats[1], mats[0])]).tocsr()
    got = op.to_scipy()
    assert (abs(got - want)).max() == 0
    gt = op.t().to_scipy()
    assert (abs(gt - want.T)).max() == 0


@pytest.mark.parametrize("kind", ["cloth", "cloth_perm", "torus", "delaunay"])
@pytest.mark.parametrize("which", ["Di", "DiA"])
@pytest.mark.parametrize("N", [32, 16])
def test_spmm_leaves_the_column_statistics_of_its_output(kind, which, N):
    """sn_spmm_q3_stats_f32: Y bit-identical to sn_spmm_q3_f32 (hence to the CSR oracle), and the partials it leaves add up
    to the column sums / sums of squares of Y — contiguous and into one half of a concat buffer, ragged tails included; at 128
    channels (N = 32) and at 64 (N = 16, the Mesh-MNIST models)."""
    _, _, ops = mesh_fixture(kind)
    A = ops[which]
    A.sort_indices()
    M, K = A.shape
    C = 4 * N
    rng = np.random.default_rng(7)
    xcat = (rng.standard_normal((K // 4, 2 * C)) * 2 + 0.5).astype(np.float32)
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, np.ascontiguousarray(xcat[:, :C]).ravel(), N).reshape(M // 4, C)
    rp, ci, va = csr_dev(A)
    b = kernels.csr_to_bsr4(rp, ci, va, M, K)
    q, _ = kernels.bsr4_to_q3(b[1], b[2])
    for strided in (False, True):
        ybuf = torch.full((M // 4, 2 * C), float("nan"), device=DEV)
        y = ybuf[:, C:] if strided else torch.empty((M // 4, C), device=DEV)
        part = kernels.spmm_q3_stats(b[0], q, M // 4, K // 4, dev(xcat)[:, :C], y, 4)
        assert part.shape[1:] == (2, C)
        assert np.array_equal(y.cpu().numpy(), want)
        got = part.sum(0).cpu().numpy()
        w64 = want.astype(np.float64)
        ref = np.stack([w64.sum(0), (w64 * w64).sum(0)])
        scale = np.stack([np.abs(w64).sum(0), (w64 * w64).sum(0)]) + 1e-30
        assert (np.abs(got - ref) / scale).max() < 1e-6            # fp32 over 32 rows, fp64 above
        if strided:
            assert torch.isnan(ybuf[:, :C]).all()
            st = torch.full((2, 2 * C), float("nan"), dtype=torch.float64, device=DEV)
            kernels.colstats_merge_into(part, st, C)
            assert np.allclose(st[:, C:].cpu().numpy(), got, rtol=1e-14) and torch.isnan(st[:, :C]).all()


def test_csr_rows_kernel_ragged_long_and_empty_rows():
    """The generic CSR kernel (several row passes per wave, the wave's entries staged in LDS) against the C oracle, bit for
    bit: ragged and empty rows, rows longer than the wave's LDS slice (tiled path), every N, both operand layouts, the fused
    ELU-backward epilogue and the statistics variant (tests/csr_rows_check.py)."""
    import csr_rows_check

    csr_rows_check.main()


def test_csr_rows_kernel_large_batch_multi_pass():
    """A batch large enough for the library to pick two passes per wave (300k rows at N = 128): bit-exact vs the oracle, and
    the row-blocked form of the same operator agrees."""
    rng = np.random.default_rng(12)
    import scipy.sparse as sp

    M = K = 300_007
    lens = rng.integers(3, 12, size=M)
    rows = np.repeat(np.arange(M), lens)
    cols = (rows + rng.integers(-2000, 2000, size=len(rows))) % K
    A = sp.csr_matrix((rng.standard_normal(len(rows)).astype(np.float32), (rows, cols)), shape=(M, K))
    A.sum_duplicates()
    A.sort_indices()
    N = 128
    x = rng.standard_normal((K, N)).astype(np.float32)
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.ravel(), N).reshape(M, N)
    rp, ci, va = csr_dev(A)
    y = torch.empty((M, N), device=DEV)
    kernels.spmm_csr(rp, ci, va, M, K, dev(x), y, 1)
    assert np.array_equal(y.cpu().numpy(), want)
    part = kernels.spmm_csr_stats(rp, ci, va, M, K, dev(x), y)
    assert np.array_equal(y.cpu().numpy(), want)
    b = kernels.csr_to_rb4(rp, ci, va, M, K)                      # the 4x1 row-blocked form of the same operator
    y2 = torch.empty((M, N), device=DEV)
    part2 = kernels.spmm_rb4_stats(b[0], b[1], b[2], M, K, dev(x), y2)
    assert torch.equal(y2, y)
    w64 = want.astype(np.float64)
    ref = np.stack([w64.sum(0), (w64 * w64).sum(0)])
    for p_ in (part, part2):
        assert np.allclose(p_.sum(0).cpu().numpy(), ref, rtol=1e-6, atol=1e-6 * np.abs(w64).sum(0).max())


def test_device_library_matches_host_twins():
    """The C-ABI call table of tests/abi_cases.py on the device: the same status as the host-pointer twins for every invalid
    call, and bit-identical outputs for every valid one (ELU: expm1f implementations may differ by an ulp)."""
    import abi_cases as ac

    host, devb = ac.Backend("host"), ac.Backend("device")
    for (what, got_h, want), (_, got_d, _) in zip(ac.invalid_calls(host), ac.invalid_calls(devb)):
        assert got_h == want and got_d == want, (what, got_h, got_d, want)
    oh, od = ac.run_valid(host), ac.run_valid(devb)
    assert set(oh) == set(od)
    for k in oh:
        a, b = oh[k], od[k]
        for x_, y_ in zip(a if isinstance(a, list) else [a], b if isinstance(b, list) else [b]):
            if k == "elu":
                assert np.allclose(x_[:, :8], y_[:, :8], rtol=1e-6, atol=1e-7) and np.isnan(y_[:, 8:]).all()
            else:
                assert np.array_equal(x_, y_), k


def test_non_finite_inputs_per_storage_form():
    """Documented caveat: BSR4 and RB4 store explicit zeros, so a non-finite X entry reaches every row of a
    4-row group whose block (BSR4) or column list (RB4) references it (0 * inf = NaN), where CSR only reaches the rows with a
    non-zero coefficient; the quaternion-packed form multiplies by no structural zero and behaves like CSR.  Everything
    outside those groups is bit-identical to CSR, and stays finite."""
    _, F, ops = mesh_fixture("cloth")
    rng = np.random.default_rng(3)
    # ---- Dirac: Di (4F x 4V), N = 32; poison component 0 of vertex j, column 3 ----
    Di = ops["Di"].tocsr()
    Di.sort_indices()
    M, K = Di.shape
    N, j = 32, 17
    x = rng.standard_normal((K // 4, 4 * N)).astype(np.float32)
    x[j, 0 * N + 3] = np.inf
    rp, ci, va = csr_dev(Di)
    y_csr = torch.empty((M // 4, 4 * N), device=DEV)
    kernels.spmm_csr(rp, ci, va, M, K, dev(x), y_csr, 4)
    y_csr = y_csr.cpu().numpy().reshape(M // 4, 4, N)
    touched_rows = np.zeros((M // 4, 4), bool)                     # CSR: rows with a non-zero coefficient on column 4j+0
    col = Di[:, 4 * j].toarray().ravel() != 0
    touched_rows[:] = col.reshape(M // 4, 4)
    assert np.array_equal(~np.isfinite(y_csr[:, :, 3]), touched_rows) and np.isfinite(np.delete(y_csr, 3, axis=2)).all()
    faces = np.flatnonzero((F == j).any(1))                        # block rows (faces) whose block list contains vertex j
    b = kernels.csr_to_bsr4(rp, ci, va, M, K)
    q, flag = kernels.bsr4_to_q3(b[1], b[2])
    assert int(flag.item()) == 0
    for name, run in (("bsr4", lambda y: kernels.spmm_bsr4(b[0], b[1], b[2], M // 4, K // 4, dev(x), y, 4)),
                      ("q3", lambda y: kernels.spmm_q3(b[0], q, M // 4, K // 4, dev(x), y, 4))):
        y = torch.empty((M // 4, 4 * N), device=DEV)
        run(y)
        y = y.cpu().numpy().reshape(M // 4, 4, N)
        bad = ~np.isfinite(y[:, :, 3])
        if name == "bsr4":           # explicit zeros of the 4x4 block: the whole 4-row group of every incident face
            assert bad[faces].all() and not np.delete(bad, faces, axis=0).any(), name
        else:                        # Q3 multiplies by no structural zero (M(p) has none off its diagonal, and the diagonal
            assert np.array_equal(bad, touched_rows), name     # is skipped): exactly CSR's rows (no p component is 0 here)
        other = np.delete(np.arange(M // 4), faces)
        assert np.array_equal(y[other], y_csr[other]) and np.array_equal(np.delete(y, 3, axis=2), np.delete(y_csr, 3, axis=2)), name
    # ---- Laplacian: RB4 (4x1 row blocks), N = 128; poison row j, column 5 ----
    L = ops["L"].tocsr()
    L.sort_indices()
    V = L.shape[0]
    xl = rng.standard_normal((V, 128)).astype(np.float32)
    xl[j, 5] = -np.inf
    rp, ci, va = csr_dev(L)
    y_csr = torch.empty((V, 128), device=DEV)
    kernels.spmm_csr(rp, ci, va, V, V, dev(xl), y_csr, 1)
    y_csr = y_csr.cpu().numpy()
    assert np.array_equal(~np.isfinite(y_csr[:, 5]), L[:, j].toarray().ravel() != 0)
    r = kernels.csr_to_rb4(rp, ci, va, V, V)
    y = torch.empty((V, 128), device=DEV)
    kernels.spmm_rb4(r[0], r[1], r[2], V, V, dev(xl), y)
    y = y.cpu().numpy()
    groups = np.unique(np.flatnonzero(L[:, j].toarray().ravel() != 0) // 4)
    rows = np.concatenate([np.arange(4 * g_, min(4 * g_ + 4, V)) for g_ in groups])
    bad = ~np.isfinite(y[:, 5])
    assert bad[rows].all() and not np.delete(bad, rows).any()
    other = np.delete(np.arange(V), rows)
    assert np.array_equal(y[other], y_csr[other]) and np.array_equal(np.delete(y, 5, axis=1), np.delete(y_csr, 5, axis=1))


def test_entry_points_from_two_host_threads_on_two_streams():
    """include/sn_spmm.h: "re-entrant; safe from several host threads on different streams".  Two Python threads (ctypes
    releases the GIL across the foreign call), each on its own stream with its own buffers, launch the Dirac / Laplacian
    products, a statistics product (workspace + two kernels per call) and a fused Linear a few hundred times; every result
    equals the one computed alone on the default stream, bit for bit."""
    import threading

    V, F, ops = mesh_fixture("delaunay", 3)
    pools = {"Di": OperatorPool([ops["Di"].astype(np.float32)] * 24, DEV, want_bsr4=True),
             "L": OperatorPool([ops["L"].astype(np.float32)] * 24, DEV)}
    sel = np.arange(24)
    Di, L = pools["Di"].assemble(sel), pools["L"].assemble(sel)
    Di.q3(), L.rb4()                                       # derived forms built before the threads start
    g = torch.Generator(device=DEV).manual_seed(11)
    xv = torch.randn(Di.shape[1] // 4, 128, device=DEV, generator=g)
    xl = torch.randn(L.shape[1], 128, device=DEV, generator=g)
    W = torch.randn(128, 128, device=DEV, generator=g) * 0.1
    b = torch.randn(128, device=DEV, generator=g)

    def work(out):
        yd = torch.empty(Di.shape[0] // 4, 128, device=DEV)
        snF._launch(Di, xv, yd, 4)
        yl = torch.empty(L.shape[0], 128, device=DEV)
        part = snF._launch(L, xl, yl, 1, stats=True)
        y = kernels.linear_fwd(xl, W, b)
        out.append((yd, yl, None if part is None else part.clone(), y))

    ref = []
    work(ref)
    torch.cuda.synchronize()
    results, errors = {0: [], 1: []}, []

    def runner(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(150):
                    work(results[k])
                s.synchronize()
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=runner, args=(k,)) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in (0, 1):
        assert len(results[k]) == 150
        for got in results[k][::7] + results[k][-3:]:
            for a, r in zip(got, ref[0]):
                assert (a is None and r is None) or torch.equal(a, r)


# ---- the parity chain closed on the device: HIP product vs the REFERENCE's stored results ------------------------------------------
@pytest.mark.parametrize("mesh", ["cube", "delaunay150"])
@pytest.mark.parametrize("fmt", ["csr", "bsr4", "q3", "rb4", "ring"])
def test_hip_products_match_the_references_stored_results(golden_dir, mesh, fmt, monkeypatch):
    """tests/golden/spmm_reference.npz holds X, G and the reference's own Y = torch.mm(A_coo, X), GX = its autograd backward
    (written by tests/golden/make_golden.py from the imported reference).  The HIP products in every storage form, through
    functional.spmm, against those numbers directly (<= 1e-6 relative) — not only through the oracle."""
    import os

    import scipy.sparse as sp

    from helpers import rel_err
    from surfacenetworks_amd import functional as snF, kernels
    from surfacenetworks_amd.operators import SparseOperator

    g = np.load(os.path.join(golden_dir, "spmm_reference.npz"))
    z = np.load(os.path.join(golden_dir, f"ops_{mesh}.npz"))
    monkeypatch.setattr(kernels, "RING_MIN_ROWS", 8)                 # (the sliding-window kernel is chosen from 131 072 rows on)
    dirac = fmt in ("bsr4", "q3")
    ran = hits = 0
    for k, Ns, group in (("L", (64, 128), 1), ("Di", (16, 32), 4), ("DiA", (16, 32), 4)):
        if (group == 4) != dirac and fmt != "csr":
            continue
        A = sp.csr_matrix((z[f"{k}_data"], z[f"{k}_indices"], z[f"{k}_indptr"]), shape=tuple(z[f"{k}_shape"]))
        M, K = A.shape
        for N in Ns:
            t = f"{mesh}_{k}_N{N}"
            X, Y, G, GX = g[f"{t}_X"], g[f"{t}_Y"], g[f"{t}_G"], g[f"{t}_GX"]
            op = SparseOperator.from_scipy(A, "cuda")
            op.format = fmt
            kind = snF.product_form(op, group, group * N)[0]
            hits += kind == fmt                                      # (e.g. the 8-row cube has no row-blocked form: falls back)
            assert kind in (fmt, "csr", "rb4", "bsr4"), (fmt, kind)
            x = torch.from_numpy(X.reshape(K // group, group * N)).cuda().requires_grad_(True)
            y = snF.spmm(op, x, group=group)
            y.backward(torch.from_numpy(G.reshape(M // group, group * N)).cuda())
            assert rel_err(y.detach().cpu().numpy().reshape(M, N), Y) <= 1e-6, (t, fmt, kind)
            assert rel_err(x.grad.cpu().numpy().reshape(K, N), GX) <= 1e-6, (t, fmt, kind)
            ran += 1
    assert ran >= 2 and (hits > 0 or mesh == "cube"), (ran, hits)      # the 150-vertex mesh takes every requested form
