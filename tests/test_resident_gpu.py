"""The fast path behind the reference's batching names (utils_pt.sp_sparse_to_pt_sparse / sparse_diag_cat / sparse_cat,
src/utils/utils_pt.py:21-69; callers src/as_rigid_as_possible/main.py:156-185, src/mesh_mnist/main.py:100-117,
src/dense_correspondence/main.py:180-190): batches assembled on the device from resident copies equal the reference's
host-built coalesced tensors, the residual blocks give the same results on either, and everything the reference's API
promises of the returned tensors still holds (they materialise on demand)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _meshes(rng, grids, permute=False):
    from surfacenetworks_amd import mesh_ops as mo

    out = []
    for n, m in grids:
        V, F = mo.grid_cloth(n, m, rng, permute=permute)
        ops = mo.mesh_operators(V, F)
        out.append((V.astype(np.float32), F, ops))
    return out


def _dense(t):
    return t.to_dense().cpu().numpy()


@pytest.mark.parametrize("which,group", [("Di", 4), ("DiA", 4), ("L", 1)])
def test_diag_cat_of_handles_is_the_references_tensor(which, group):
    import surfacenetworks_amd.utils_pt as U
    from surfacenetworks_amd.operators import SparseOperator, as_operator
    from surfacenetworks_amd.resident import LazySparse, reference_diag_cat, reference_sp_to_coo, resident_cache

    rng = np.random.default_rng(1)
    ms = _meshes(rng, [(7, 6), (9, 8), (5, 5)])
    mats = [m[2][which] for m in ms]
    s0 = max(a.shape[0] for a in mats) + 4 * (group == 4)
    s1 = max(a.shape[1] for a in mats)
    sel = [2, 0, 1, 0]                                                 # a mesh twice in one batch
    want = reference_diag_cat([reference_sp_to_coo(mats[i]) for i in sel], s0, s1)
    handles = [U.sp_sparse_to_pt_sparse(mats[i]) for i in sel]
    assert all(isinstance(h, LazySparse) and h.device.type == "cpu" and not h.is_coalesced() for h in handles)
    cache = resident_cache()
    m0 = cache.misses
    got = U.sparse_diag_cat(handles, s0, s1)
    assert isinstance(got, LazySparse) and got.device.type == "cpu" and got.is_coalesced() and got.coalesce() is got
    assert tuple(got.shape) == tuple(want.shape) and got.dtype == want.dtype and got.layout == torch.sparse_coo
    assert cache.misses - m0 <= 3
    g = got.cuda()
    assert g.is_cuda and g.cuda() is g and isinstance(g._sn_operator, SparseOperator)
    op = as_operator(g)
    assert op is g._sn_operator and op.shape == want.shape and op.batch == len(sel)
    A = op.to_scipy()
    W = want.to_dense().numpy()
    assert np.array_equal(np.asarray(A.todense()), W)                   # the device-assembled batch IS the reference's operator
    # a second step over the same dataset objects: cache hits only, no conversion
    m1, h1 = cache.misses, cache.hits
    U.sparse_diag_cat([U.sp_sparse_to_pt_sparse(mats[i]) for i in sel], s0, s1)
    assert cache.misses == m1 and cache.hits == h1 + len(sel)
    # anything else reads the real tensor, built as the reference builds it
    assert np.array_equal(_dense(got), W) and np.array_equal(_dense(g), W)
    assert torch.equal(got._indices(), want._indices()) and torch.equal(got._values(), want._values())
    assert got._nnz() == want._nnz()
    x = torch.randn(want.shape[1], 8)
    assert torch.equal(torch.mm(got, x), torch.mm(want, x))
    # transposed product through the operator (what the blocks' backward uses)
    N = 32 if group == 4 else 128
    xd = torch.randn(op.shape[1] // group, group * N, device=DEV, requires_grad=True)
    from surfacenetworks_amd import functional as snF

    y = snF.spmm(op, xd, group=group)
    ref_op = SparseOperator.from_torch_coo(want.to(DEV))
    xr = xd.detach().clone().requires_grad_(True)
    yr = snF.spmm(ref_op, xr, group=group)
    assert torch.equal(y, yr)
    gy = torch.randn_like(y)
    y.backward(gy)
    yr.backward(gy)
    assert torch.equal(xd.grad, xr.grad)


def test_sparse_cat_handles_and_variable():
    """The Mesh-MNIST driver converts its dataset at load (main.py:58-72), batches with sparse_cat and wraps in Variable()."""
    import surfacenetworks_amd.utils_pt as U
    from torch.autograd import Variable

    from surfacenetworks_amd.operators import as_operator
    from surfacenetworks_amd.resident import LazySparse, reference_cat, reference_sp_to_coo

    rng = np.random.default_rng(2)
    ms = _meshes(rng, [(6, 6), (7, 5)])
    for which, group in (("L", 1), ("Di", 4)):
        mats = [m[2][which] for m in ms]
        s0 = max(a.shape[0] for a in mats)
        s1 = max(a.shape[1] for a in mats)
        handles = [U.sp_sparse_to_pt_sparse(a) for a in mats]          # kept in the dataset, as convert() does
        got = Variable(U.sparse_cat(handles, s0, s1)).cuda()
        want = reference_cat([reference_sp_to_coo(a) for a in mats], s0, s1)
        assert isinstance(got, LazySparse) and got.dim() == 3 and tuple(got.size()) == tuple(want.size())
        op = as_operator(got)
        assert op.batch == 2 and op.shape == (2 * s0, 2 * s1)
        D = np.asarray(op.to_scipy().todense())
        Wd = want.to_dense().numpy()
        for b in range(2):
            assert np.array_equal(D[b * s0:(b + 1) * s0, b * s1:(b + 1) * s1], Wd[b])
        assert np.array_equal(_dense(got), Wd)


def test_blocks_agree_on_resident_and_host_built_batches():
    """One Dirac block + one Laplacian block, forward and backward: the batch from the resident path against the same batch
    built by the reference's host arithmetic (what the package did before: real COO tensors converted on the device)."""
    import surfacenetworks_amd.utils_pt as U
    from surfacenetworks_amd.resident import reference_diag_cat, reference_sp_to_coo

    rng = np.random.default_rng(3)
    ms = _meshes(rng, [(9, 7), (8, 8), (6, 10)], permute=True)
    nv = max(m[0].shape[0] for m in ms)
    nf = max(m[1].shape[0] for m in ms)
    B, C = len(ms), 128
    torch.manual_seed(0)
    v0 = torch.randn(B, nv, C, device=DEV)
    f0 = torch.randn(B, nf, C, device=DEV)

    def run(resident: bool):
        torch.manual_seed(1)
        dblk, lblk = U.DirResNet2(C).to(DEV).train(), U.LapResNet2(C).to(DEV).train()
        if resident:
            mk = lambda key, a, b: U.sparse_diag_cat([U.sp_sparse_to_pt_sparse(m[2][key]) for m in ms], a, b).cuda()
        else:
            mk = lambda key, a, b: reference_diag_cat([reference_sp_to_coo(m[2][key]) for m in ms], a, b).to(DEV)
        Di, DiA, L = mk("Di", 4 * nf, 4 * nv), mk("DiA", 4 * nv, 4 * nf), mk("L", nv, nv)
        v = v0.clone().requires_grad_(True)
        f = f0.clone().requires_grad_(True)
        vo, fo = dblk(Di, DiA, v, f)
        xo = lblk(L, None, vo)
        (xo.square().mean() + fo.square().mean()).backward()
        grads = [p.grad.clone() for p in list(dblk.parameters()) + list(lblk.parameters())]
        return xo.detach(), fo.detach(), v.grad.clone(), f.grad.clone(), grads

    a, b = run(True), run(False)
    for x, y in zip(a[:4], b[:4]):
        assert torch.equal(x, y)
    for x, y in zip(a[4], b[4]):
        assert torch.equal(x, y)


def test_members_that_cannot_be_resident_take_the_host_path():
    import scipy.sparse as sp

    import surfacenetworks_amd.utils_pt as U
    from surfacenetworks_amd.resident import LazySparse, reference_diag_cat, reference_sp_to_coo

    rng = np.random.default_rng(4)
    A64 = sp.random(12, 10, density=0.3, format="csr", dtype=np.float64, random_state=1)
    A32 = sp.random(12, 10, density=0.3, format="coo", dtype=np.float32, random_state=2)      # not CSR: converted, still resident
    h64, h32 = U.sp_sparse_to_pt_sparse(A64), U.sp_sparse_to_pt_sparse(A32)
    assert h64.dtype == torch.float64 and h32.dtype == torch.float32
    out = U.sparse_diag_cat([h64, h64], 12, 10)                        # float64: the reference's host path, a real tensor
    assert not isinstance(out, LazySparse)
    assert torch.equal(out.to_dense(), reference_diag_cat([reference_sp_to_coo(A64)] * 2, 12, 10).to_dense())
    out32 = U.sparse_diag_cat([h32, h32], 12, 10)
    assert isinstance(out32, LazySparse)
    assert np.array_equal(np.asarray(out32.cuda()._sn_operator.to_scipy().todense()),
                          reference_diag_cat([reference_sp_to_coo(A32)] * 2, 12, 10).to_dense().numpy())
    real = reference_sp_to_coo(A32.tocsr())
    mixed = U.sparse_diag_cat([h32, real], 12, 10)                     # a real tensor among the members: host path
    assert not isinstance(mixed, LazySparse)
    # a Dirac operator and a Laplacian in one list (different packings): host path, same result
    ms = _meshes(rng, [(5, 5)])
    hs = [U.sp_sparse_to_pt_sparse(ms[0][2]["Di"]), U.sp_sparse_to_pt_sparse(sp.random(64, 100, 0.1, "csr", np.float32, random_state=3))]
    odd = U.sparse_diag_cat(hs, 64, 100)
    assert not isinstance(odd, LazySparse) and odd.shape == (128, 200)


def test_in_place_replacement_of_a_dataset_matrix_is_noticed():
    import surfacenetworks_amd.utils_pt as U

    rng = np.random.default_rng(6)
    ms = _meshes(rng, [(6, 6)])
    L = ms[0][2]["L"].copy()
    a = U.sparse_diag_cat([U.sp_sparse_to_pt_sparse(L)], L.shape[0], L.shape[1]).cuda()._sn_operator.to_scipy()
    L.data = L.data * 2                                                # new array object: noticed through its address
    b = U.sparse_diag_cat([U.sp_sparse_to_pt_sparse(L)], L.shape[0], L.shape[1]).cuda()._sn_operator.to_scipy()
    assert np.array_equal(b.data, 2 * a.data)
    L.data *= 3                                                        # the SAME array edited in place: noticed through the value probes
    c = U.sparse_diag_cat([U.sp_sparse_to_pt_sparse(L)], L.shape[0], L.shape[1]).cuda()._sn_operator.to_scipy()
    assert np.array_equal(c.data, 6 * a.data)


def test_an_in_place_edit_between_the_probes_is_found_by_the_round_robin_checksum():
    """Eight probed values per array cannot see every edit; one member's FULL checksum is re-verified per call, round-robin, so
    the edit is noticed within as many calls as the batch has members — deterministically."""
    import surfacenetworks_amd.utils_pt as U
    from surfacenetworks_amd.resident import ResidentCache, resident_cache

    rng = np.random.default_rng(8)
    ms = _meshes(rng, [(6, 6), (6, 6), (6, 6)])
    Ls = [m[2]["L"].copy() for m in ms]
    cache = resident_cache()
    cache.clear()
    batch = lambda: U.sparse_diag_cat([U.sp_sparse_to_pt_sparse(L) for L in Ls], Ls[0].shape[0], Ls[0].shape[1]).cuda()._sn_operator.to_scipy()
    a = batch()
    sig = ResidentCache._signature(Ls[1])
    Ls[1].data[1] *= 5.0                                               # between the probes (index 1 of a >= 16-entry array)
    assert ResidentCache._signature(Ls[1]) == sig
    stale0 = cache.stale
    seen = [batch() for _ in range(len(Ls))]                           # every member's checksum has been re-verified once by now
    assert cache.stale == stale0 + 1
    want = a.copy()
    n0 = Ls[0].nnz
    want.data[n0 + 1] *= 5.0
    assert np.array_equal(seen[-1].data, want.data) and np.array_equal(batch().data, want.data)


def test_freeze_makes_in_place_edits_raise_and_dead_sources_leave_the_index():
    import gc

    import surfacenetworks_amd.utils_pt as U
    from surfacenetworks_amd.resident import resident_cache

    rng = np.random.default_rng(9)
    ms = _meshes(rng, [(6, 6), (7, 5)])
    cache = resident_cache()
    cache.clear()
    cache.freeze = True
    try:
        L = ms[0][2]["L"].copy()
        U.sparse_diag_cat([U.sp_sparse_to_pt_sparse(L)], L.shape[0], L.shape[1])
        with pytest.raises(ValueError, match="read-only"):
            L.data *= 2
    finally:
        cache.freeze = False
    M = ms[1][2]["L"].copy()
    U.sparse_diag_cat([U.sp_sparse_to_pt_sparse(M)], M.shape[0], M.shape[1])
    n, dead = len(cache.index), cache.dead_bytes
    key = id(M)
    assert key in cache.index
    del M
    gc.collect()
    assert key not in cache.index and len(cache.index) == n - 1 and cache.dead_bytes > dead


def test_resident_pools_grow_and_start_over_when_the_budget_is_spent():
    """Operators arrive a few at a time (a driver sampling at random from a large dataset): the pools grow in place (arenas
    that double; earlier members keep their offsets) and every batch — old members, new members, mixed — equals the
    reference's tensor.  With a budget of a few hundred KB the cache starts over in between (generational reset) and the
    batches are still right."""
    import surfacenetworks_amd.utils_pt as U
    from surfacenetworks_amd.resident import reference_diag_cat, reference_sp_to_coo, resident_cache

    rng = np.random.default_rng(11)
    ms = _meshes(rng, [(6 + i % 5, 5 + (3 * i) % 7) for i in range(14)])
    cache = resident_cache()
    cache.clear()
    old_budget = cache.max_bytes
    try:
        for budget in (old_budget, 300_000):
            cache.clear()
            cache.max_bytes = budget
            resets0 = cache.resets
            for step in range(8):
                sel = rng.integers(0, 2 + 2 * step if step < 6 else len(ms), size=5) % len(ms)     # the population grows step by step
                for which, group in (("DiA", 4), ("L", 1)):
                    mats = [ms[i][2][which] for i in sel]
                    s0, s1 = max(a.shape[0] for a in mats), max(a.shape[1] for a in mats)
                    got = U.sparse_diag_cat([U.sp_sparse_to_pt_sparse(a) for a in mats], s0, s1).cuda()
                    want = reference_diag_cat([reference_sp_to_coo(a) for a in mats], s0, s1)
                    assert got._sn_operator is not None
                    assert np.array_equal(np.asarray(got._sn_operator.to_scipy().todense()), want.to_dense().numpy()), (budget, step, which)
                    t = got._sn_operator.t().to_scipy()
                    assert np.array_equal(np.asarray(t.todense()), want.to_dense().numpy().T)
            if budget < old_budget:
                assert cache.resets > resets0                      # the small budget did force the cache to start over
    finally:
        cache.max_bytes = old_budget
        cache.clear()


def test_driver_style_step_equals_the_products_own_step(tmp_path):
    """The ARAP step as the reference's driver assembles it (src/as_rigid_as_possible/main.py:98-185,217-232, restated: frames
    from the sequence files, inputs / targets / mask filled on the host, `sp_sparse_to_pt_sparse` per sample, `sparse_diag_cat`,
    `.cuda()`, `outputs * mask`, `smooth_l1_loss(reduction='sum') / batch_size`) on the device, against the product's own
    sampler and train step on the same files: the same loss and gradients — the batch operators are assembled from the same
    matrices by the same launch on both sides, so bit for bit."""
    import torch.nn.functional as F

    import surfacenetworks_amd.utils_pt as utils
    from helpers import deterministic_init
    from surfacenetworks_amd import arap, datasets, mesh_ops as mo
    from surfacenetworks_amd.resident import LazySparse

    rng = np.random.default_rng(14)
    T = arap.INPUT_FRAMES + arap.OUTPUT_FRAMES + 2
    paths = []
    for k, (n, m) in enumerate([(9, 8), (7, 6), (8, 8)]):
        V, Fc = mo.grid_cloth(n, m, rng)
        Vt = np.repeat(V[None], T, 0).copy()
        Vt[:, :, 2] += 0.02 * np.sin(0.3 * np.arange(T))[:, None] * (1 + V[None, :, 0])
        paths.append(str(tmp_path / f"seq{k}.npy"))
        datasets.write_arap_sequence(paths[-1], Vt, Fc, op_frames=3)
    sequences = [datasets.load_arap_sequence(p) for p in paths]              # what main.py:58-94 keeps: dicts of numpy / scipy objects
    ds = datasets.arap_from_files(paths, DEV, "dir", reorder=False)          # the product's resident dataset of the same files
    indices, offsets = [2, 0, 1, 0], [1, 0, 0, 1]
    B = len(indices)
    nv = max(sequences[i][0]["V"].shape[0] for i in indices)
    nf = max(sequences[i][0]["F"].shape[0] for i in indices)
    inputs, targets, mask = torch.zeros(B, nv, 6), torch.zeros(B, nv, 120), torch.zeros(B, nv, 1)
    Di, DiA = [], []
    for b, (ind, off) in enumerate(zip(indices, offsets)):
        n_ = sequences[ind][0]["V"].shape[0]
        for i in range(2):
            inputs[b, :n_, 3 * i:3 * (i + 1)] = torch.from_numpy(sequences[ind][i + off]["V"])
        for i in range(40):
            targets[b, :n_, 3 * i:3 * (i + 1)] = torch.from_numpy(sequences[ind][i + off + 2]["V"])
        mask[b, :n_] = 1
        Di.append(utils.sp_sparse_to_pt_sparse(sequences[ind][off + 1]["Di"]))
        DiA.append(utils.sp_sparse_to_pt_sparse(sequences[ind][off + 1]["DiA"]))
    Di = utils.sparse_diag_cat(Di, 4 * nf, 4 * nv)
    DiA = utils.sparse_diag_cat(DiA, 4 * nv, 4 * nf)
    inputs, targets, mask, Di, DiA = inputs.cuda(), targets.cuda(), mask.cuda(), Di.cuda(), DiA.cuda()
    assert isinstance(Di, LazySparse) and Di._sn_operator is not None and not Di._sn_payload.real       # nothing was materialised
    own = ds.sample_batch(B, None, seq_ids=np.array(indices), offsets=np.array(offsets))
    assert torch.equal(own.inputs, inputs) and torch.equal(own.targets, targets) and torch.equal(own.mask, mask)
    for a, b_ in ((Di._sn_operator, own.Di), (DiA._sn_operator, own.DiA)):
        assert abs(a.to_scipy() - b_.to_scipy()).max() == 0
    res = []
    for kind in ("driver", "product"):
        model = deterministic_init(arap.DirModel(), 9).to(DEV).train()
        if kind == "driver":
            outputs = model(Di, DiA, mask, inputs)
            outputs = outputs * mask.expand_as(outputs)
            loss = F.smooth_l1_loss(outputs, targets, reduction="sum") / B
        else:
            loss, _ = arap.forward_loss(model, own)
        loss.backward()
        res.append((loss.detach().clone(), [p.grad.clone() for p in model.parameters()]))
    assert abs(res[0][0].item() - res[1][0].item()) <= 1e-6 * abs(res[1][0].item())     # (torch's loss against the fused loss kernel)
    gn = max(float(g.norm()) for g in res[1][1])
    for g0, g1 in zip(res[0][1], res[1][1]):
        assert float((g0 - g1).norm()) <= 1e-5 * float(g1.norm()) + 1e-6 * gn
    assert not Di._sn_payload.real and not DiA._sn_payload.real                 # the whole step ran without the index arrays
