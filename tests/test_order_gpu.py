"""Datasets stored in the locality numbering (mesh_ops.MeshOrder) on the device.

SpMM through the reorder: the HIP product of the STORED operator A' = P_r A P_c^T with the stored operand P_c x is bit-identical
to the C oracle on (A', P_c x) and — mapped back with P_r^T — equal to the oracle's product of the dataset-order operator up to
the rounding of another summation order.  Models: the ARAP step on shuffled meshes gives the same loss and, through
`to_dataset_order`, the same per-vertex outputs whether the dataset is stored as it came or renumbered; the ring kernel
becomes eligible.  The product at src/utils/utils_pt.py:167,176,202,214; dataset order: src/utils/mesh.py:35-64."""
import numpy as np
import pytest
import torch

from helpers import rel_err
from oracle import c_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("which,fmt", [("Di", "q3"), ("Di", "bsr4"), ("Di", "csr"), ("DiA", "q3"), ("DiA", "bsr4"), ("DiA", "csr"), ("L", "csr")])
def test_spmm_through_the_reorder_is_bit_exact_against_the_oracle(fmt, which):
    from surfacenetworks_amd import functional as snF, mesh_ops as mo
    from surfacenetworks_amd.operators import SparseOperator

    rng = np.random.default_rng(8)
    V, F = mo.grid_cloth(23, 19, rng, permute="both")
    o = mo.MeshOrder.of_mesh(F, V.shape[0], True)
    ops = mo.mesh_operators(V, F)
    A = ops[which]
    group = 1 if which == "L" else 4
    rows = {"Di": o.forder, "DiA": o.vorder, "L": o.vorder}[which]
    cols = {"Di": o.vorder, "DiA": o.forder, "L": o.vorder}[which]
    Ap = mo.permute_operator(A, rows, cols, group).astype(np.float32)
    M, K = A.shape
    N = 32 if group == 4 else 128
    x = rng.standard_normal((K // group, group * N)).astype(np.float32)
    xp = x[cols]                                                        # P_c x: stored row k is dataset row cols[k]
    want_p = c_oracle.spmm_csr(Ap.indptr, Ap.indices, Ap.data, xp.ravel(), N).reshape(M // group, group * N)
    snF.set_dirac_format(fmt)
    try:
        op = SparseOperator.from_scipy(Ap, DEV)
        xt = dev(xp).requires_grad_(True)
        y = snF.spmm(op, xt, group=group)
        g = rng.standard_normal(want_p.shape).astype(np.float32)
        y.backward(dev(g))
    finally:
        snF.set_dirac_format("q3")
    got = y.detach().cpu().numpy()
    assert np.array_equal(got, want_p)                                   # bit-exact on the stored operator
    tr = c_oracle.csr_transpose(Ap.indptr, Ap.indices, Ap.data, K)
    want_g = c_oracle.spmm_csr(tr[0], tr[1], tr[2], g.ravel(), N).reshape(K // group, group * N)
    assert np.array_equal(xt.grad.cpu().numpy(), want_g)
    # mapped back: row rows[k] of the dataset-order product is stored row k (another summation order: rounding only)
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.ravel(), N).reshape(M // group, group * N)
    back = np.empty_like(got)
    back[rows] = got
    assert rel_err(back, want) < 1e-6
    assert rel_err(back.reshape(M, N), A.astype(np.float64) @ x.reshape(K, N).astype(np.float64)) < 1e-6


def test_arap_step_is_the_same_on_a_renumbered_dataset():
    """Shuffled meshes, stored as they came vs stored in the locality numbering: same sample, same weights -> the same loss
    and (through to_dataset_order) the same per-vertex predictions, up to the rounding of other summation orders."""
    from surfacenetworks_amd import arap

    grids = [(14, 11), (9, 16), (12, 12)]
    kw = dict(frames=arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2, op_frames=2, seed=21, device=DEV, model="dir", permute="both")
    ds_off = arap.ClothSequences(grids, reorder=False, **kw)
    ds_on = arap.ClothSequences(grids, reorder=True, **kw)
    assert all(o.identity for o in ds_off.orders) and not any(o.identity for o in ds_on.orders)
    ids, offs = np.array([2, 0, 1]), np.zeros(3, dtype=np.int64)
    b_off = ds_off.sample_batch(3, None, seq_ids=ids, offsets=offs)
    b_on = ds_on.sample_batch(3, None, seq_ids=ids, offsets=offs)
    # the stored inputs are the dataset's inputs, renumbered
    assert torch.equal(ds_on.to_dataset_order(b_on.inputs, ids), b_off.inputs)
    assert torch.equal(ds_on.from_dataset_order(b_off.targets, ids), b_on.targets)
    torch.manual_seed(7)
    model = arap.DirModel().to(DEV).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    l_off, out_off = arap.forward_loss(model, b_off)
    l_off.backward()
    g_off = [p.grad.clone() for p in model.parameters()]
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    l_on, out_on = arap.forward_loss(model, b_on)
    l_on.backward()
    assert abs(l_on.item() - l_off.item()) <= 2e-5 * abs(l_off.item())
    back = ds_on.to_dataset_order(out_on.detach(), ids)
    mask = b_off.mask
    scale = float((out_off.detach() * mask).abs().max())
    assert float(((back - out_off.detach()) * mask).abs().max()) <= 2e-4 * scale
    gn = max(float(g.norm()) for g in g_off)
    for p, g0 in zip(model.parameters(), g_off):
        assert float((p.grad - g0).norm()) <= 1e-3 * float(g0.norm()) + 1e-5 * gn


def test_reordered_laplacian_pool_takes_the_ring_kernel():
    """A batch of shuffled meshes large enough for the sliding-window kernel: as stored no row lies in the window (gather
    kernels); renumbered, the band is the grid's and the ring kernel runs — same product, bit for bit against the oracle."""
    from surfacenetworks_amd import functional as snF, mesh_ops as mo
    from surfacenetworks_amd.operators import OperatorPool

    rng = np.random.default_rng(5)
    mats, mats_r = [], []
    for n, m in [(150, 150)] * 6:
        V, F = mo.grid_cloth(n, m, rng, permute="both")
        o = mo.MeshOrder.of_mesh(F, V.shape[0], "auto")
        assert not o.identity
        mats.append(mo.laplacian(V, F).astype(np.float32))
        V2, F2 = o.mesh(V, F)
        mats_r.append(mo.laplacian(V2, F2).astype(np.float32))
    sel = np.arange(len(mats))
    op = OperatorPool(mats, DEV).assemble(sel)
    op_r = OperatorPool(mats_r, DEV).assemble(sel)
    assert not op.ring_ok(128) and op_r.ring_ok(128)
    assert op_r.band()[0] <= 151 and op.band()[0] > 10000
    x = torch.randn(op_r.shape[1], 128, device=DEV)
    y = torch.empty(op_r.shape[0], 128, device=DEV)
    timer = snF.SpmmTimer()
    with timer:
        snF._launch(op_r, x, y, 1, "t")
    assert "/ring" in timer.results()[0][0]
    A = op_r.to_scipy()
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.cpu().numpy().ravel(), 128).reshape(-1, 128)
    assert np.array_equal(y.cpu().numpy(), want)


def test_files_in_arbitrary_order_are_stored_renumbered(tmp_path):
    """datasets.arap_from_files on a sequence whose file keeps a shuffled numbering: coordinates and the file's own operators
    are renumbered once; a batch equals the one a loader without reordering hands out, relabeled."""
    from surfacenetworks_amd import arap, datasets, mesh_ops as mo

    rng = np.random.default_rng(4)
    V, F = mo.grid_cloth(12, 10, rng, permute="both")
    T = arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 1
    Vt = np.repeat(V[None], T, 0).copy()
    Vt[:, :, 2] += 0.01 * np.sin(np.arange(T))[:, None] * V[None, :, 0]
    path = str(tmp_path / "seq.npy")
    datasets.write_arap_sequence(path, Vt, F, op_frames=2)
    ds0 = datasets.arap_from_files([path], DEV, "dir", reorder=False)
    ds1 = datasets.arap_from_files([path], DEV, "dir", reorder="auto")
    assert ds0.orders[0].identity and not ds1.orders[0].identity
    ids, offs = np.array([0]), np.array([0])
    b0 = ds0.sample_batch(1, None, seq_ids=ids, offsets=offs)
    b1 = ds1.sample_batch(1, None, seq_ids=ids, offsets=offs)
    assert torch.equal(ds1.to_dataset_order(b1.inputs, ids), b0.inputs)
    o = ds1.orders[0]
    for name, rows, cols in (("Di", o.forder, o.vorder), ("DiA", o.vorder, o.forder)):
        A0, A1 = getattr(b0, name).to_scipy(), getattr(b1, name).to_scipy()
        assert abs(mo.permute_operator(A0, rows, cols, 4) - A1).max() == 0
    assert mo.edge_span(np.asarray(o.vrank[F[o.forder]]))[1] <= 13


def test_faust_pair_loss_is_invariant_to_the_stored_numbering():
    """dense correspondence (src/dense_correspondence/main.py:229-240): frames whose file numbering is shuffled, stored as they
    came vs renumbered (coordinates, Laplacian, label permutations and the geodesic matrix together): the same pair loss and the
    same parameter gradients up to summation order."""
    from helpers import deterministic_init
    from surfacenetworks_amd import dense_correspondence as dc, mesh_ops as mo

    rng = np.random.default_rng(9)
    frames = []
    for n, m in [(9, 11), (11, 9)]:                # (FAUST bodies share one label space: equal vertex counts)
        V, F = mo.torus_grid(n, m, rng, permute="both")
        nv = V.shape[0]
        label = rng.permutation(nv)
        Vt = torch.from_numpy(V.astype(np.float32)).to(DEV)
        frames.append({"V": Vt, "F": torch.from_numpy(F).to(DEV), "L": mo.laplacian(V, F).astype(np.float32), "Di": None, "DiA": None,
                       "label": torch.from_numpy(label).to(DEV), "label_inv": torch.from_numpy(np.argsort(label)).to(DEV),
                       "G": torch.cdist(Vt, Vt)})
    ds_off = dc.FaustFrames(frames, model="lap", pad_to=112, device=DEV, reorder=False)
    ds_on = dc.FaustFrames(frames, model="lap", pad_to=112, device=DEV, reorder=True)
    assert not any(o.identity for o in ds_on.orders)
    o = ds_on.orders[0]
    fr = ds_on.frames[0]
    assert torch.equal(fr["label_inv"][fr["label"].long()].cpu(), torch.arange(o.vorder.size))      # still inverse permutations
    assert torch.equal(fr["G"].cpu(), frames[0]["G"].cpu()[o.vorder][:, o.vorder])
    res = []
    for ds in (ds_off, ds_on):
        model = deterministic_init(dc.SiameseModel("lap", 3), 12).to(DEV).train()
        loss = dc.forward_pair_loss(model, ds, 0, 1)
        loss.backward()
        res.append((loss.item(), torch.cat([p.grad.reshape(-1) for p in model.parameters()])))
    assert abs(res[0][0] - res[1][0]) <= 1e-4 * abs(res[0][0]), (res[0][0], res[1][0])
    assert float((res[0][1] - res[1][1]).norm() / res[0][1].norm()) < 2e-3


def test_device_built_operators_follow_the_stored_numbering():
    """ClothSequences(operators="device") on shuffled meshes stored in the locality numbering: the Dirac operators built on the
    GPU from the stored coordinates and faces are the pooled (host-built) operators of the same renumbered meshes — same
    pattern, values equal to coordinate round-off — and banded; a packed batch of the renumbered pool trains."""
    from surfacenetworks_amd import arap, mesh_ops as mo

    kw = dict(frames=44, op_frames=2, seed=6, device=DEV, model="dir", permute="both", reorder=True)
    ds_d = arap.ClothSequences([(11, 9)] * 3, operators="device", **kw)
    ds_p = arap.ClothSequences([(11, 9)] * 3, operators="pool", **kw)
    assert not any(o.identity for o in ds_d.orders)
    ids, off = np.array([1, 2, 0]), np.zeros(3, dtype=np.int64)
    bd = ds_d.sample_batch(3, None, seq_ids=ids, offsets=off)
    bp = ds_p.sample_batch(3, None, seq_ids=ids, offsets=off)
    assert torch.equal(bd.inputs, bp.inputs) and torch.equal(bd.targets, bp.targets)
    for name in ("Di", "DiA"):
        a, b = getattr(bd, name).to_scipy(), getattr(bp, name).to_scipy()
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
        assert abs(a - b).max() <= 1e-4 * abs(b).max()
    # banded: a face's three vertices lie within a dozen positions of each other (11 x 9 grid: row-major band 10)
    blk = bp.Di.to_scipy()[: 4 * int(ds_p.num_faces[1]), : 4 * int(ds_p.num_vertices[1])].tocoo()
    assert (np.abs(blk.col // 4 - np.round(blk.row // 4 / 2.0)) <= 3 * 11).all()
    torch.manual_seed(0)
    m = arap.DirModel().to(DEV)
    opt = arap.make_optimizer(m)
    assert torch.isfinite(arap.train_step(m, opt, bd))
    assert torch.isfinite(arap.train_step(m, opt, ds_p.sample_batch(3, None, seq_ids=ids, offsets=off, packed=True)))
