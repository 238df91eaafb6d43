"""Device-side Dirac construction (sn_dirac_bsr4_from_mesh) against mesh_ops.dirac, which is pinned bit-for-bit to the
reference's mesh.dirac by the golden fixtures: values must be IDENTICAL in fp32 (fp64 geometry, reference op order)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from helpers import rel_err
from oracle import c_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"

from surfacenetworks_amd import functional as snF, mesh_ops  # noqa: E402
from surfacenetworks_amd.operators import OperatorPool, dirac_operators_from_mesh, laplacian_operator_from_mesh  # noqa: E402


def meshes():
    rng = np.random.default_rng(11)
    yield "cube", mesh_ops.read_ply_ascii.__globals__["np"].array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=float), \
        np.array([[3, 2, 1], [1, 0, 3], [6, 7, 4], [4, 5, 6], [5, 1, 2], [2, 6, 5], [0, 4, 7], [7, 3, 0], [6, 2, 3], [3, 7, 6], [5, 4, 0], [0, 1, 5]])
    yield "cloth", *mesh_ops.grid_cloth(17, 13, rng)
    yield "cloth_perm", *mesh_ops.grid_cloth(12, 15, rng, permute=True)
    yield "torus", *mesh_ops.torus_grid(10, 14, rng)
    yield "delaunay", *mesh_ops.delaunay_disc(300, rng)
    yield "big", *mesh_ops.grid_cloth(71, 71, rng)


@pytest.mark.parametrize("name,V,F", list(meshes()), ids=[m[0] for m in meshes()])
def test_device_dirac_bit_identical_to_reference_builder(name, V, F):
    V32 = V.astype(np.float32)
    Di, DiA = dirac_operators_from_mesh(torch.from_numpy(V32).to(DEV), torch.from_numpy(F.astype(np.int32)).to(DEV))
    D, DA = mesh_ops.dirac(V32.astype(np.float64), F)
    for op, ref in ((Di, D), (DiA, DA), (Di.t(), D.T), (DiA.t(), DA.T)):
        r = ref.astype(np.float32).tocsr()
        r.sort_indices()
        bo = c_oracle.csr_to_bsr4(r.indptr, r.indices, r.data)
        for g_, o_ in zip(op.bsr4(), bo):
            assert np.array_equal(g_.cpu().numpy(), o_)
        g = op.to_scipy()                                         # lazy CSR expansion drops the explicit zeros
        assert np.array_equal(g.indptr, r.indptr) and np.array_equal(g.indices, r.indices) and np.array_equal(g.data, r.data)
        assert op.nnz == r.nnz


def test_batched_build_equals_pool_and_products_match():
    rng = np.random.default_rng(2)
    V, F = mesh_ops.grid_cloth(21, 19, rng)
    B = 5
    Vb = np.stack([V + 0.01 * rng.standard_normal(V.shape) for _ in range(B)]).astype(np.float32)
    Di, DiA = dirac_operators_from_mesh(torch.from_numpy(Vb).to(DEV), torch.from_numpy(F.astype(np.int32)).to(DEV))
    mats = [mesh_ops.dirac(Vb[b].astype(np.float64), F) for b in range(B)]
    nv, nf = V.shape[0], F.shape[0]
    pDi = OperatorPool([m[0].astype(np.float32) for m in mats], DEV, want_bsr4=True).assemble(np.arange(B), 4 * nf, 4 * nv)
    pDiA = OperatorPool([m[1].astype(np.float32) for m in mats], DEV, want_bsr4=True).assemble(np.arange(B), 4 * nv, 4 * nf)
    for a, b in ((Di, pDi), (DiA, pDiA), (Di.t(), pDi.t()), (DiA.t(), pDiA.t())):
        for x, y in zip(a.bsr4(), b.bsr4()):
            assert torch.equal(x, y)
    x = torch.randn(B * nv, 128, device=DEV, requires_grad=True)
    y1 = snF.spmm(Di, x, 4)
    y1.sum().backward()
    g1 = x.grad.clone()
    x.grad = None
    y2 = snF.spmm(pDi, x, 4)
    y2.sum().backward()
    assert torch.equal(y1, y2) and torch.equal(g1, x.grad)


def test_arap_device_operators_train_step():
    from surfacenetworks_amd import arap

    ds_d = arap.ClothSequences([(9, 8)] * 3, frames=44, op_frames=2, seed=4, device=DEV, model="dir", operators="device")
    ds_p = arap.ClothSequences([(9, 8)] * 3, frames=44, op_frames=2, seed=4, device=DEV, model="dir", operators="pool")
    ids, off = np.array([2, 0, 1]), np.array([0, 0, 0])
    bd = ds_d.sample_batch(3, None, seq_ids=ids, offsets=off)
    bp = ds_p.sample_batch(3, None, seq_ids=ids, offsets=off)
    assert torch.equal(bd.inputs, bp.inputs) and torch.equal(bd.targets, bp.targets)
    # pool operators come from the fp64 coordinates, device ones from the stored fp32 coordinates: same pattern,
    # values equal to coordinate round-off
    a, b = bd.Di.to_scipy(), bp.Di.to_scipy()
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    assert abs(a - b).max() <= 1e-4 * abs(b).max()
    torch.manual_seed(0)
    m = arap.DirModel().to(DEV)
    loss = arap.train_step(m, arap.make_optimizer(m), bd)
    assert torch.isfinite(loss)


@pytest.mark.parametrize("name,V,F", list(meshes()), ids=[m[0] for m in meshes()])
def test_device_laplacian_bit_identical_to_reference_builder(name, V, F):
    V32 = V.astype(np.float32)
    L = laplacian_operator_from_mesh(torch.from_numpy(V32).to(DEV), torch.from_numpy(F.astype(np.int32)).to(DEV))
    ref = mesh_ops.laplacian(V32.astype(np.float64), F).astype(np.float32).tocsr()
    ref.sort_indices()
    got = L.to_scipy()
    assert got.shape == ref.shape
    # same pattern up to explicit zeros, identical values
    assert (got != ref).nnz == 0
    g2, r2 = got.copy(), ref.copy()
    g2.eliminate_zeros(); r2.eliminate_zeros()
    assert np.array_equal(g2.indptr, r2.indptr) and np.array_equal(g2.indices, r2.indices) and np.array_equal(g2.data, r2.data)


def test_device_laplacian_batched_and_product():
    rng = np.random.default_rng(5)
    V, F = mesh_ops.grid_cloth(15, 14, rng)
    B = 3
    Vb = np.stack([V * (1 + 0.1 * b) for b in range(B)]).astype(np.float32)
    L = laplacian_operator_from_mesh(torch.from_numpy(Vb).to(DEV), torch.from_numpy(F.astype(np.int32)).to(DEV))
    blocks = [mesh_ops.laplacian(Vb[b].astype(np.float64), F).astype(np.float32) for b in range(B)]
    want = sp.block_diag(blocks, format="csr")
    assert (L.to_scipy() != want).nnz == 0
    x = torch.randn(B * V.shape[0], 128, device=DEV)
    y = snF.spmm(L, x, 1)
    assert rel_err(y.cpu().numpy(), want.astype(np.float64) @ x.cpu().numpy().astype(np.float64)) < 1e-6
