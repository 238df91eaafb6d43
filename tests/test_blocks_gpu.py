"""GPU parity of the product blocks / models / samplers (real HIP kernels through the C-ABI) against the
reference-generated golden fixtures and the fp64 oracle: the same assertions as tests/test_product_host.py."""
import numpy as np
import pytest
import torch

import product_checks as pc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _block_cases():
    """(block, width, operator kind, Dirac format): the format switch only concerns Dirac blocks, the operator kind only blocks
    with a sparse operator."""
    out = []
    for cname, C in pc.BLOCKS:
        sparse = cname not in ("AvgResNet2", "MlpResNet2")
        for opkind in (("pool", "coo2d", "coo3d") if sparse else ("pool",)):
            for fmt in (("q3", "bsr4", "csr") if cname == "DirResNet2" else ("q3",)):
                out.append((cname, C, opkind, fmt))
    return out


@pytest.mark.parametrize("cname,C,opkind,fmt", _block_cases())
def test_blocks_match_reference(golden_dir, cname, C, opkind, fmt):
    from surfacenetworks_amd import functional as snF

    snF.set_dirac_format(fmt)
    try:
        pc.check_block(golden_dir, cname, C, opkind, DEV)
    finally:
        snF.set_dirac_format("q3")


@pytest.mark.parametrize("tag", ["arap_dir", "arap_lap", "mnist_lap", "mnist_dir", "faust_lap"])
def test_models_match_reference(golden_dir, tag):
    pc.check_model(golden_dir, tag, DEV)


@pytest.mark.parametrize("tag", ["arap_dir", "arap_lap", "mnist_lap", "mnist_dir", "faust_lap"])
def test_model_layers_match_reference_layer_by_layer(golden_dir, tag):
    """Each layer given the reference's own input of that layer: <= 1e-5 relative (no compounding over 15 layers)."""
    pc.check_model_layers(golden_dir, tag, DEV)


@pytest.mark.parametrize("tag", ["arap_dir", "mnist_dir"])
def test_models_with_reference_driver_operator_types(golden_dir, tag):
    """Same models fed with the torch sparse COO operators the reference drivers build (2-D block-diag)."""
    pc.check_model(golden_dir, tag, DEV, opkind="coo2d")


def test_unfused_spmm_function():
    pc.check_spmm_autograd(DEV)


def test_arap_sampler():
    pc.check_arap_sampler(DEV)


def test_mnist_sampler():
    pc.check_mnist_sampler(DEV)


@pytest.mark.parametrize("fmt", ["q3", "bsr4", "csr"])
def test_pool_packed_assembly(golden_dir, fmt):
    """Ragged (unpadded) batches: sn_blockdiag_concat_ragged_i32 + the product kernels on packed operands."""
    from surfacenetworks_amd import functional as snF

    snF.set_dirac_format(fmt)
    try:
        pc.check_pool_packed(golden_dir, DEV)
    finally:
        snF.set_dirac_format("q3")


def test_reference_format_files_feed_training(golden_dir):
    """SURVEY.md §8f-3 on the device: the reference's pickled .npy / .np / .npz layouts -> device pools -> train steps."""
    pc.check_dataset_files(golden_dir, DEV)


def test_streamed_faust_loss_on_device():
    """the FAUST correspondence loss streamed over row blocks on the GPU, fp32, against the materialised
    (N, N) score matrix; the golden faust_lap loss goes through it in test_models_match_reference[faust_lap]."""
    pc.check_streamed_faust_loss(DEV, N=3000)


def test_model_variants_match_reference(golden_dir):
    pc.check_model_variants(golden_dir, DEV)


def test_inplace_edit_drops_the_activated_handoff(golden_dir):
    pc.check_inplace_edit_drops_handoff(golden_dir, DEV)


def test_fused_block_equals_unfused_composition(golden_dir):
    """DirResNet2 (fused stages) == the same block composed from F.elu + spmm() + torch.cat, forward and backward."""
    import torch.nn.functional as F

    import surfacenetworks_amd.utils_pt as U
    from helpers import deterministic_init, det_tensor, rel_err
    from surfacenetworks_amd import functional as snF

    rb, ops = pc.batch_operators(golden_dir, "pool", DEV)
    B, nv, nf, C = rb["mask"].shape[0], int(rb["nv"]), int(rb["nf"]), 128
    blk = deterministic_init(U.DirResNet2(C), 5).train().to(DEV)
    res = []
    for fused in (True, False):
        blk.zero_grad()
        v = torch.from_numpy(det_tensor((B, nv, C), 1) * rb["mask"]).to(DEV).requires_grad_(True)
        f = torch.from_numpy(det_tensor((B, nf, C), 2)).to(DEV).requires_grad_(True)
        if fused:
            vo, fo = blk(ops["Di"], ops["DiA"], v, f)
        else:
            x_in, f_in = F.elu(v), F.elu(f)
            y = snF.spmm(ops["Di"], x_in.reshape(B * nv, C), 4).view(B, nf, C)
            fo = blk.bn_fc0(torch.cat([f_in, y], 2))
            z = snF.spmm(ops["DiA"], F.elu(fo).reshape(B * nf, C), 4).view(B, nv, C)
            vo = v + blk.bn_fc1(torch.cat([x_in, z], 2))
        (vo.sum() + (fo * fo).sum()).backward()
        res.append([t.detach().cpu().numpy() for t in (vo, fo, v.grad, f.grad, blk.bn_fc0.fc.weight.grad)])
    for a, b in zip(*res):
        assert rel_err(a, b) < 2e-6


def test_size_independent_properties_at_config_scale():
    """BASELINE config-3 scale (64 cloth meshes of 71x71, C=128): linearity, adjoint identity <A x, g> == <x, A^T g>,
    CSR == BSR4 bit-for-bit, block-diagonal independence of the meshes."""
    from surfacenetworks_amd import functional as snF, kernels, mesh_ops
    from surfacenetworks_amd.operators import OperatorPool

    rng = np.random.default_rng(3)
    mats = []
    for _ in range(4):
        V, F_ = mesh_ops.grid_cloth(71, 71, rng)
        mats.append(mesh_ops.dirac(V, F_)[0].astype(np.float32))
    pool = OperatorPool(mats, DEV, want_bsr4=True)
    sel = np.arange(64) % 4
    op = pool.assemble(sel, mats[0].shape[0], mats[0].shape[1])
    M, K = op.shape
    assert (M, K) == (64 * 39200, 64 * 20164)
    N = 32
    g = torch.Generator(device=DEV).manual_seed(0)
    x1 = torch.randn(K // 4, 4 * N, device=DEV, generator=g)
    x2 = torch.randn(K // 4, 4 * N, device=DEV, generator=g)
    gy = torch.randn(M // 4, 4 * N, device=DEV, generator=g)
    y1, y2, y12 = snF.spmm(op, x1, 4), snF.spmm(op, x2, 4), snF.spmm(op, x1 + 2 * x2, 4)
    assert ((y12 - (y1 + 2 * y2)).abs().max() / y12.abs().max()).item() < 1e-5
    gx = snF.spmm(op.t(), gy, 4)
    lhs, rhs = (y1.double() * gy.double()).sum().item(), (x1.double() * gx.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), abs(rhs), 1.0) * 100
    yc = torch.empty_like(y1)
    kernels.spmm_csr(op.rowptr, op.colind, op.vals, M, K, x1, yc, 4)
    assert torch.equal(yc, y1)                                 # y1 came from the quaternion-packed (q3) kernel
    # mesh b of the batch only sees its own slice of x: same mesh + same slice => identical rows
    rows = M // 4 // 64
    xs = x1.clone()
    xs[4 * (K // 4 // 64): 5 * (K // 4 // 64)] = x1[: K // 4 // 64]
    ys = snF.spmm(op, xs, 4)
    assert torch.equal(ys[4 * rows: 5 * rows], y1[:rows]) and torch.equal(ys[:rows], y1[:rows])


@pytest.mark.parametrize("cname", ["LapResNet2", "DirResNet2", "AvgResNet2"])
def test_whole_block_node_equals_per_stage_functions(golden_dir, cname):
    """blocks.py (one autograd node per block, activated hand-off, fused residual/accumulation) == the per-stage
    functions of functional.py, on a two-block chain so that the cross-block hand-off is exercised."""
    import surfacenetworks_amd.utils_pt as U
    from helpers import deterministic_init, det_tensor, rel_err

    rb, ops = pc.batch_operators(golden_dir, "pool", DEV)
    B, nv, nf, C = rb["mask"].shape[0], int(rb["nv"]), int(rb["nf"]), 128
    mask = torch.from_numpy(rb["mask"]).to(DEV)
    res = []
    for whole in (True, False):
        U.USE_WHOLE_BLOCKS = whole
        try:
            b1 = deterministic_init(getattr(U, cname)(C), 3).train().to(DEV)
            b2 = deterministic_init(getattr(U, cname)(C), 4).train().to(DEV)
            v = torch.from_numpy(det_tensor((B, nv, C), 1) * rb["mask"]).to(DEV).requires_grad_(True)
            if cname == "DirResNet2":
                f = torch.from_numpy(det_tensor((B, nf, C), 2)).to(DEV).requires_grad_(True)
                v1, f1 = b1(ops["Di"], ops["DiA"], v, f)
                v2, f2 = b2(ops["Di"], ops["DiA"], v1, f1)
                loss = (v2 * v2).sum() + f2.sum() + v1.mean()
                outs = [v2, f2]
                gins = [v, f]
            else:
                arg = ops["L"] if cname == "LapResNet2" else None
                v1 = b1(arg, mask, v)
                v2 = b2(arg, mask, v1)
                loss = (v2 * v2).sum() + v1.mean()
                outs, gins = [v2], [v]
            loss.backward()
            res.append([t.detach().cpu().numpy() for t in outs] + [t.grad.cpu().numpy() for t in gins] +
                       [b1.bn_fc0.fc.weight.grad.cpu().numpy(), b2.bn_fc1.bn.weight.grad.cpu().numpy(),
                        b1.bn_fc1.bn.running_var.cpu().numpy()])
        finally:
            U.USE_WHOLE_BLOCKS = True
    # AvgResNet2's whole-block node is the half-width form: the mean-path terms are evaluated analytically (per-mesh algebra in
    # fp64) instead of through a broadcast second half, so the two paths differ by fp32 rounding of different summations
    tol = 2e-5 if cname == "AvgResNet2" else 5e-6
    for a, b in zip(*res):
        assert rel_err(a, b) < tol


def test_unwritten_face_features_are_a_loud_placeholder(golden_dir):
    """DirResNet2(..., f_out_needed=False): the chained result is unchanged, the pre-activation face features are not
    materialised, and what is returned in their place is NaN (any use other than as the next block's `f` is visible)."""
    import surfacenetworks_amd.utils_pt as U
    from helpers import deterministic_init, det_tensor

    rb, ops = pc.batch_operators(golden_dir, "pool", DEV)
    B, nv, nf, C = rb["mask"].shape[0], int(rb["nv"]), int(rb["nf"]), 128
    b1 = deterministic_init(U.DirResNet2(C), 3).train().to(DEV)
    b2 = deterministic_init(U.DirResNet2(C), 4).train().to(DEV)
    res = []
    for needed in (True, False):
        v = torch.from_numpy(det_tensor((B, nv, C), 1)).to(DEV).requires_grad_(True)
        f = torch.from_numpy(det_tensor((B, nf, C), 2)).to(DEV).requires_grad_(True)
        v1, f1 = b1(ops["Di"], ops["DiA"], v, f, f_out_needed=needed)
        if not needed:
            assert f1.shape == (B, nf, C) and torch.isnan(f1).all() and f1.stride() == (0, 0, 0)
        v2, f2 = b2(ops["Di"], ops["DiA"], v1, f1)
        (v2.square().sum() + f2.sum()).backward()
        res.append([v2.detach(), f2.detach(), v.grad.clone(), f.grad.clone(), b1.bn_fc0.fc.weight.grad.clone()])
        b1.zero_grad(), b2.zero_grad()
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("train", [True, False])
def test_zero_face_features_are_never_materialised(golden_dir, train):
    """DirResNet2(Di, DiA, v, None, num_faces=F) == DirResNet2(Di, DiA, v, zeros(B, F, C)) — the first Dirac block of
    every model of the reference (as_rigid_as_possible/models.py:138) — in outputs, gradients, running statistics and batch
    counter, with the face stage at half width (no zero tensor, no statistics / GEMM columns for it)."""
    import copy

    import surfacenetworks_amd.utils_pt as U
    from helpers import deterministic_init, det_tensor, rel_err

    rb, ops = pc.batch_operators(golden_dir, "pool", DEV)
    B, nv, nf, C = rb["mask"].shape[0], int(rb["nv"]), int(rb["nf"]), 128
    base = deterministic_init(U.DirResNet2(C), 3).to(DEV)
    with torch.no_grad():                                        # non-trivial BatchNorm parameters for the zero columns
        base.bn_fc0.bn.weight.copy_(torch.linspace(0.5, 1.5, 2 * C))
        base.bn_fc0.bn.bias.copy_(torch.linspace(-0.3, 0.4, 2 * C))
        base.bn_fc0.bn.running_mean.copy_(torch.linspace(-0.1, 0.1, 2 * C))
        base.bn_fc0.bn.running_var.copy_(torch.linspace(0.5, 2.0, 2 * C))
    res = []
    for zero_path in (False, True):
        blk = copy.deepcopy(base).train(train)
        v = torch.from_numpy(det_tensor((B, nv, C), 1)).to(DEV).requires_grad_(True)
        if zero_path:
            v1, f1 = blk(ops["Di"], ops["DiA"], v, None, num_faces=nf)
        else:
            v1, f1 = blk(ops["Di"], ops["DiA"], v, torch.zeros(B, nf, C, device=DEV))
        (v1.square().sum() + (f1 * 0.37).sum()).backward()
        res.append([v1.detach(), f1.detach(), v.grad.clone()] + [p.grad.clone() for p in blk.parameters()] +
                   [b.clone().float() for b in blk.buffers()])
    for a, b in zip(*res):
        assert a.shape == b.shape
        assert rel_err(b.cpu().numpy(), a.cpu().numpy()) < 2e-5, (a.shape,)
    # what the half-width form may not touch: the batch counter still counts one batch per BatchNorm
    if train:
        assert all(int(b) == 1 for b in res[1][-1:])


@pytest.mark.parametrize("cname,C", [("LapResNet2", 8), ("LapResNet2", 32), ("LapResNet2", 256), ("DirResNet2", 16), ("DirResNet2", 32),
                                     ("DirResNet2", 256)])
def test_blocks_at_widths_without_vector_kernels(golden_dir, cname, C):
    """Channel counts outside the reference models' 64 / 128: the block kernels exist for N in {16, 32, 64,
    128} dense columns; every other width must take the generic CSR kernel and the unfused epilogue — not raise — and agree
    with the oracle block (CPU torch.sparse path) forward and backward."""
    import surfacenetworks_amd.utils_pt as U
    from helpers import deterministic_init, det_tensor, rel_err
    from oracle import ref_blocks as OB

    rb, ops = pc.batch_operators(golden_dir, "pool", DEV)
    B, nv, nf = rb["mask"].shape[0], int(rb["nv"]), int(rb["nf"])
    cpu_ops = {k: torch.sparse_coo_tensor(torch.from_numpy(rb[f"{k}_bd_indices"]), torch.from_numpy(rb[f"{k}_bd_values"]),
                                          tuple(rb[f"{k}_bd_shape"])).coalesce() for k in ("L", "Di", "DiA")}
    res = []
    for lib, dev in ((U, DEV), (OB, "cpu")):
        blk = deterministic_init(getattr(lib, cname)(C), 9).train().to(dev)
        v = (torch.from_numpy(det_tensor((B, nv, C), 1)) * (0.02 if cname == "LapResNet2" else 1.0)).to(dev).requires_grad_(True)
        if cname == "DirResNet2":
            f = torch.from_numpy(det_tensor((B, nf, C), 2)).to(dev).requires_grad_(True)
            o = ops if dev == DEV else cpu_ops
            vo, fo = blk(o["Di"], o["DiA"], v, f)
            (vo.square().mean() + fo.square().mean()).backward()
            res.append([t.detach().cpu().numpy() for t in (vo, fo, v.grad, f.grad, blk.bn_fc0.fc.weight.grad)])
        else:
            vo = blk((ops if dev == DEV else cpu_ops)["L"], None, v)
            vo.square().mean().backward()
            res.append([t.detach().cpu().numpy() for t in (vo, v.grad, blk.bn_fc1.fc.weight.grad)])
    for a, b in zip(*res):
        assert rel_err(a, b) < 3e-5, (cname, C, rel_err(a, b))


def test_operator_moves_between_devices_with_its_transpose():
    """SparseOperator.to(): the transpose link is carried over without recursing through it."""
    from helpers import mesh_fixture
    from surfacenetworks_amd.operators import SparseOperator

    _, _, ops = mesh_fixture("cloth")
    op = SparseOperator.from_scipy(ops["Di"], DEV)
    t = op.t()
    assert t.t() is op
    host = op.to("cpu")
    assert host.device.type == "cpu" and host._t is not None and host._t.device.type == "cpu" and host._t._t is host
    assert host.nnz == op.nnz and abs(host.to_scipy() - ops["Di"]).max() == 0
    back = host.to(DEV)
    assert back.is_cuda and abs(back.t().to_scipy() - ops["Di"].T).max() == 0
    assert op.to(DEV) is op


def test_models_on_packed_batches():
    """Ragged batches without padding through whole models (PackedSegments, ragged global-average kernels)."""
    pc.check_packed_model(DEV)


@pytest.mark.parametrize("C", [128, 64, 120, 3])
def test_ragged_segment_kernels(C):
    """sn_segment_colsum_ragged_f32 / sn_bcast_rows_ragged_f32 against numpy: meshes shorter and longer than a 256-row tile,
    strided operands, with and without the per-mesh scale."""
    from surfacenetworks_amd import kernels
    from surfacenetworks_amd.operators import PackedSegments

    seg = PackedSegments([1, 256, 257, 1000, 7, 513], DEV)
    rng = np.random.default_rng(C)
    buf = rng.standard_normal((seg.rows, 2 * C + 4)).astype(np.float32)
    xb = torch.from_numpy(buf).to(DEV)
    x = xb[:, :C]
    want = np.stack([buf[seg.offsets[i]: seg.offsets[i + 1], :C].astype(np.float64).sum(0) for i in range(seg.nseg)])
    got = kernels.segment_colsum_ragged(x, seg.tiles, seg.seg_tile_ptr, seg.nseg).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6 * np.abs(buf).sum(0).max())
    got_m = seg.mean(x).cpu().numpy()
    assert np.allclose(got_m, want / seg.lengths[:, None], rtol=1e-6, atol=1e-7)
    if C % 4 == 0:
        dst = torch.full((seg.rows, 2 * C), float("nan"), device=DEV)
        src = torch.from_numpy(rng.standard_normal((seg.nseg, C)).astype(np.float32)).to(DEV)
        kernels.bcast_rows_ragged(src, seg.tiles, dst[:, C:])
        ids = np.repeat(np.arange(seg.nseg), seg.lengths)
        assert np.array_equal(dst[:, C:].cpu().numpy(), src.cpu().numpy()[ids]) and torch.isnan(dst[:, :C]).all()


@pytest.mark.parametrize("lengths", [[40, 300, 33, 64, 1000], [5041] * 3, [32, 32, 4097]])
def test_ragged_average_block_at_half_width_matches_the_full_width_composition(lengths, monkeypatch):
    """AvgResNet2 on a PACKED batch (PackedSegments, ragged meshes): the half-width autograd node (per-mesh bias / vector
    by mesh offsets in the GEMM epilogues, weight gradient over caller-defined slabs) against the full-width composition
    ([elu(x) | per-mesh mean] materialised, 2C-wide GEMMs) — outputs, input gradient, parameter gradients, running stats."""
    import copy

    from surfacenetworks_amd import blocks as snB, utils_pt as utils
    from surfacenetworks_amd.operators import PackedSegments

    torch.manual_seed(3)
    seg = PackedSegments(lengths, DEV)
    blk_a = utils.AvgResNet2(128).to(DEV).train()
    with torch.no_grad():
        for p in blk_a.parameters():
            p.copy_(torch.randn_like(p) * (0.1 if p.dim() > 1 else 0.5) + (1.0 if p.dim() == 1 else 0.0))
    blk_b = copy.deepcopy(blk_a)
    x = torch.randn(1, seg.rows, 128, device=DEV)
    w = torch.randn(1, seg.rows, 128, device=DEV)
    res = []
    for blk, half in ((blk_a, True), (blk_b, False)):
        if not half:
            monkeypatch.setattr(snB, "avg_block_ragged_ok", lambda *a: False)
        else:
            assert snB.avg_block_ragged_ok(blk, seg, x)
        xi = x.clone().requires_grad_(True)
        y = blk(None, seg, xi)
        (y * w).sum().backward()
        res.append((y.detach(), xi.grad, [p.grad.clone() for p in blk.parameters()],
                    [b.clone() for b in blk.buffers() if b.dtype.is_floating_point]))
    (ya, ga, pa, ba), (yb, gb, pb, bb) = res

    def rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    assert rel(ya, yb) < 2e-5 and rel(ga, gb) < 2e-4, (rel(ya, yb), rel(ga, gb))
    for a, b in zip(pa, pb):
        assert rel(a, b) < 5e-4, rel(a, b)
    for a, b in zip(ba, bb):
        assert rel(a, b) < 1e-5


def test_siamese_gradients_meet_once():
    pc.check_siamese_gradients_meet_once(DEV)


def test_mnist_driver_evaluation_mode():
    pc.check_mnist_evaluation_mode(DEV)


@pytest.mark.parametrize("cname", ["LapResNet2", "DirResNet2", "AvgResNet2"])
def test_blocks_survive_degenerate_column_statistics(golden_dir, cname):
    """Feature columns a trained network does produce and random tests do not: units saturated behind the ELU (exactly -1 in
    fp32), a large offset with a small spread, an exactly constant column.  The two-piece fp16 kernels scale their operands by
    bounds derived from BatchNorm's statistics; padded rows, zero variance and |mean| >> spread must neither overflow nor turn
    into NaN (a padded row did, tests/test_dense_gpu.py::test_two_piece_weight_gradient_ignores_the_rows_past_the_end).
    Two blocks deep; whole-block nodes (bounded two-piece products) and per-stage functions (three-piece products) are each held
    to the model criterion: as close to the float64 oracle as the reference composition in float32 is (x 4) — the problem
    itself is ill-conditioned here (BatchNorm of a nearly constant column), so a flat tolerance would measure the data."""
    import surfacenetworks_amd.utils_pt as U
    from helpers import deterministic_init, det_tensor, rel_err
    from oracle import ref_blocks as OB

    rb, ops = pc.batch_operators(golden_dir, "pool", DEV)
    B, nv, nf, C = rb["mask"].shape[0], int(rb["nv"]), int(rb["nf"]), 128
    base = det_tensor((B, nv, C), 1)
    hostile = base.copy()
    hostile[:, :, 0::4] = -30.0 + 1e-3 * base[:, :, 0::4]            # saturated behind the activation
    hostile[:, :, 1::4] = 1000.0 + 0.5 * base[:, :, 1::4]            # |mean| >> spread
    hostile[:, :, 2::4] = 3.0                                        # constant: zero variance
    hostile = hostile * rb["mask"]
    fh = det_tensor((B, nf, C), 2)
    fh[:, :, 0::2] = -25.0 + 1e-3 * fh[:, :, 0::2]

    def run(make, dtype, dev, opsd, mask):
        b1 = deterministic_init(make(C), 3).train().to(dev).to(dtype)
        b2 = deterministic_init(make(C), 4).train().to(dev).to(dtype)
        v = torch.from_numpy(hostile).to(dev).to(dtype).requires_grad_(True)
        if cname == "DirResNet2":
            f = torch.from_numpy(fh).to(dev).to(dtype).requires_grad_(True)
            v1, f1 = b1(opsd["Di"], opsd["DiA"], v, f)
            v2, f2 = b2(opsd["Di"], opsd["DiA"], v1, f1)
            loss = (v2 * v2).mean() + f2.sum() * 1e-3 + v1.mean()
            outs, gins = [v2, f2], [v, f]
        else:
            arg = opsd["L"] if cname == "LapResNet2" else None
            v1 = b1(arg, mask, v)
            v2 = b2(arg, mask, v1)
            loss = (v2 * v2).mean() + v1.mean()
            outs, gins = [v2], [v]
        loss.backward()
        return [t.detach().double().cpu().numpy() for t in outs] + [t.grad.double().cpu().numpy() for t in gins] + \
            [b1.bn_fc0.fc.weight.grad.double().cpu().numpy(), b2.bn_fc1.fc.weight.grad.double().cpu().numpy(),
             b2.bn_fc0.bn.weight.grad.double().cpu().numpy(), b1.bn_fc1.bn.running_var.double().cpu().numpy()]

    def cpu_ops(dt):
        return {k: torch.sparse_coo_tensor(torch.from_numpy(rb[f"{k}_bd_indices"]), torch.from_numpy(rb[f"{k}_bd_values"]).to(dt),
                                           tuple(rb[f"{k}_bd_shape"])).coalesce() for k in ("L", "Di", "DiA")}

    truth = run(getattr(OB, cname), torch.float64, "cpu", cpu_ops(torch.float64), torch.from_numpy(rb["mask"]).double())
    ref32 = run(getattr(OB, cname), torch.float32, "cpu", cpu_ops(torch.float32), torch.from_numpy(rb["mask"]))
    mask = torch.from_numpy(rb["mask"]).to(DEV)
    for whole in (True, False):
        U.USE_WHOLE_BLOCKS = whole
        try:
            got = run(getattr(U, cname), torch.float32, DEV, ops, mask)
        finally:
            U.USE_WHOLE_BLOCKS = True
        for k, (a, r, t) in enumerate(zip(got, ref32, truth)):
            assert np.isfinite(a).all(), (whole, k)
            ea, er = rel_err(a, t), rel_err(r, t)
            assert ea <= max(4 * er, 2e-5), (whole, k, "product", ea, "reference fp32", er)


def test_packed_model_with_and_without_tile_sums(monkeypatch):
    """The ARAP Dirac model on a PACKED batch with the tile-sum hand-off through the ragged global-average stages (forced: it is
    used from 32 768 rows on) and with the pass over every stage's operand: same loss, gradients equal up to what the fp32 tile
    sums do to the per-mesh means (1e-7 of a mesh's mean |e|, tests/test_dense_gpu.py) — which this toy batch amplifies: BatchNorm
    of the broadcast half normalises over FOUR meshes' means here (padded form of the same batch: 3e-5, packed: 4e-4)."""
    from helpers import deterministic_init
    from surfacenetworks_amd import arap, blocks as snB, kernels

    ds = arap.ClothSequences([(12, 11), (9, 13), (10, 10)], frames=45, op_frames=2, seed=3, device=DEV, model="dir")
    seq, off = np.array([0, 1, 2, 1]), np.array([0, 0, 0, 0])       # (op_frames = 2: the operator of the last input frame exists for start 0)
    monkeypatch.setattr(snB, "_TILE_SUMS_MIN_ROWS", 0)
    res = []
    for on in (True, False):
        monkeypatch.setattr(kernels, "tile_sums_supported", (lambda: True) if on else (lambda: False))
        model = deterministic_init(arap.DirModel(), 4).to(DEV).train()
        b = ds.sample_batch(4, None, seq_ids=seq, offsets=off, packed=True)
        loss, _ = arap.forward_loss(model, b, 4)
        loss.backward()
        res.append((loss.item(), torch.cat([p.grad.reshape(-1) for p in model.parameters()])))
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(res[1][0])
    assert float((res[0][1] - res[1][1]).norm() / res[1][1].norm()) < 2e-3
