"""Size-independent properties at the FULL sizes of BASELINE.json's configs 2, 4 and 5 (config 3 lives in
tests/test_blocks_gpu.py): linearity, the adjoint identity <A x, g> == <x, A^T g>, every storage form of an operator
bit-identical to the generic CSR kernel (and, on a slice, to the C oracle), independence of the diagonal blocks; the fused
epilogues against their unfused compositions; whole residual blocks at config-3 / config-4 size against the unfused
composition; the int32 index limit reported as SN_E_RANGE; the debug validator."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _props(op, group, N, seed, check_forms=True):
    """Linearity, adjoint identity and form equivalence of y = op x at N dense columns."""
    from surfacenetworks_amd import functional as snF, kernels

    M, K = op.shape
    g = torch.Generator(device=DEV).manual_seed(seed)
    x1 = torch.randn(K // group, group * N, device=DEV, generator=g)
    x2 = torch.randn(K // group, group * N, device=DEV, generator=g)
    gy = torch.randn(M // group, group * N, device=DEV, generator=g)
    y1, y2, y12 = snF.spmm(op, x1, group), snF.spmm(op, x2, group), snF.spmm(op, x1 + 2 * x2, group)
    assert ((y12 - (y1 + 2 * y2)).abs().max() / y12.abs().max()).item() < 1e-5
    gx = snF.spmm(op.t(), gy, group)
    lhs, rhs = (y1.double() * gy.double()).sum().item(), (x1.double() * gx.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * (y1.double().abs() * gy.double().abs()).sum().item()
    if check_forms:
        for o, xin, yref in ((op, x1, y1), (op.t(), gy, gx)):
            yc = torch.empty_like(yref)
            kernels.spmm_csr(o.rowptr, o.colind, o.vals, o.shape[0], o.shape[1], xin, yc, group)
            assert torch.equal(yc, yref)                       # default form (q3 / rb4) == generic CSR kernel, bit for bit
            if group == 4:
                b = o.bsr4()
                yb = torch.empty_like(yref)
                kernels.spmm_bsr4(b[0], b[1], b[2], o.shape[0] // 4, o.shape[1] // 4, xin, yb, 4)
                assert torch.equal(yb, yref)
    return x1, y1, gy, gx


def _oracle_slice(op, x, y, group, N, rows):
    """The first `rows` operator rows against the C oracle (bit-exact)."""
    A = sp.csr_matrix((op.vals.cpu().numpy(), op.colind.cpu().numpy(), op.rowptr.cpu().numpy()), shape=tuple(op.shape))[:rows]
    A.sort_indices()
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.cpu().numpy().ravel(), N)
    assert np.array_equal(y[: rows // group].cpu().numpy().ravel(), want)


def test_config2_mesh_mnist_dirac_batch512():
    """BASELINE configs[1]: 512 meshes of ~150 vertices, C = 64 (N = 16), Dirac operators, padded to the batch maximum."""
    from surfacenetworks_amd import mesh_ops
    from surfacenetworks_amd.operators import OperatorPool

    rng = np.random.default_rng(2)
    ops = [mesh_ops.mesh_operators(*mesh_ops.delaunay_disc(150, rng)) for _ in range(64)]
    sel = np.arange(512) % 64
    for name in ("Di", "DiA"):
        mats = [o[name] for o in ops]
        pool = OperatorPool(mats, DEV, want_bsr4=True)
        s0, s1 = max(m.shape[0] for m in mats), max(m.shape[1] for m in mats)
        op = pool.assemble(sel, s0, s1)
        assert op.shape == (512 * s0, 512 * s1)
        x1, y1, gy, gx = _props(op, 4, 16, 20)
        _oracle_slice(op, x1, y1, 4, 16, 8 * s0)
        # block independence: mesh 64 is mesh 0 again; give it mesh 0's input slice -> identical output rows
        xs = x1.clone()
        xs[64 * (s1 // 4): 65 * (s1 // 4)] = x1[: s1 // 4]
        from surfacenetworks_amd import functional as snF

        ys = snF.spmm(op, xs, 4)
        assert torch.equal(ys[64 * (s0 // 4): 65 * (s0 // 4)], y1[: s0 // 4]) and torch.equal(ys[: s0 // 4], y1[: s0 // 4])


def test_config4_faust_laplacian_7000_padded():
    """BASELINE configs[3]: 6890-vertex closed meshes padded to 7000 rows (dense_correspondence/main.py:193), Laplacian,
    C = 128.  RB4 (default) == CSR kernels == C oracle; fused ELU-backward epilogue and statistics epilogue == their
    unfused compositions; padding rows come out as exact zeros."""
    from surfacenetworks_amd import functional as snF, kernels, mesh_ops
    from surfacenetworks_amd.operators import OperatorPool

    rng = np.random.default_rng(4)
    mats = [mesh_ops.laplacian(*mesh_ops.torus_grid(65, 106, rng)).astype(np.float32) for _ in range(4)]
    B = 16
    pool = OperatorPool(mats, DEV)
    op = pool.assemble(np.arange(B) % 4, 7000, 7000)
    M, K = op.shape
    assert (M, K) == (B * 7000, B * 7000) and op.rb4() is not None
    x1, y1, gy, gx = _props(op, 1, 128, 40)
    _oracle_slice(op, x1, y1, 1, 128, 7000)
    assert not y1.view(B, 7000, 128)[:, 6890:].any()            # empty padding rows are written as zeros
    r = op.rb4()
    e = torch.randn(M, 128, device=DEV)
    g = torch.randn(M, 128, device=DEV)
    fused = torch.empty(M, 128, device=DEV)
    kernels.spmm_rb4(r[0], r[1], r[2], M, K, x1, fused, e, g)
    fused_csr = torch.empty(M, 128, device=DEV)
    kernels.spmm_csr_elubwd(op.rowptr, op.colind, op.vals, M, K, x1, e, g, fused_csr, 1)
    unfused = torch.empty(M, 128, device=DEV)
    kernels.elu_bwd(y1, e, unfused, False, None, g)              # y1 * elu'(e) + g
    assert torch.equal(fused, unfused) and torch.equal(fused_csr, unfused)
    ys = torch.empty(M, 128, device=DEV)
    part = kernels.spmm_rb4_stats(r[0], r[1], r[2], M, K, x1, ys)
    assert torch.equal(ys, y1)
    ref = kernels.colstats(y1)                                   # (2, 128) float64: column sums and sums of squares
    scale = torch.stack([y1.double().abs().sum(0), ref[1]]) + 1e-30
    assert ((part.sum(0) - ref).abs() / scale).max().item() < 1e-6      # fp32 over <= 256 rows per workgroup, fp64 above
    # N = 64 (the Mesh-MNIST width) through the same forms
    _props(op, 1, 64, 41)


def test_config1_mesh_mnist_laplacian_batch32_on_the_gpu():
    """BASELINE configs[0] (the reference's CPU-runnable case) at its own shape on the GPU: 32 meshes of 140 .. 230 vertices,
    cotangent Laplacians padded to the batch maximum, C = 64: properties, RB4 (default at this size) == generic CSR kernel,
    every mesh against the C oracle, padding rows exactly zero."""
    from surfacenetworks_amd import mesh_ops
    from surfacenetworks_amd.operators import OperatorPool

    rng = np.random.default_rng(1)
    nvs = rng.integers(140, 231, size=32)
    mats = [mesh_ops.laplacian(*mesh_ops.delaunay_disc(int(nv), rng)).astype(np.float32) for nv in nvs]
    pool = OperatorPool(mats, DEV)
    size = int(max(nvs))
    op = pool.assemble(np.arange(32), size, size)
    assert op.shape == (32 * size, 32 * size) and not op.ring_ok(64)          # (4 800-7 360 rows: the row-blocked kernel)
    x1, y1, gy, gx = _props(op, 1, 64, 10)
    _oracle_slice(op, x1, y1, 1, 64, 32 * size)
    yb = y1.view(32, size, 64)
    for b, nv in enumerate(nvs):
        assert not yb[b, int(nv):].any()
    _props(op, 1, 128, 11)


def test_config5_laplacian_batch_takes_the_ring_kernel():
    """The Laplacians of BASELINE configs[4]'s meshes (128 ragged grids, 1.3 M rows, C = 128): the product's default is the
    sliding-window kernel; properties, == generic CSR kernel == RB4 bit for bit, first meshes against the C oracle, the fused
    epilogue and the statistics variant == their unfused compositions."""
    from surfacenetworks_amd import functional as snF, kernels, mesh_ops
    from surfacenetworks_amd.operators import OperatorPool

    rng = np.random.default_rng(5)
    vs = rng.integers(1000, 20001, size=128)
    Ls = []
    for v in vs:
        n = int(np.sqrt(v))
        Ls.append(mesh_ops.laplacian(*mesh_ops.grid_cloth(n, int(v) // n, rng)).astype(np.float32))
    op = OperatorPool(Ls, DEV).assemble(np.arange(128))
    M, K = op.shape
    assert op.ring_ok(128) and op.t().ring_ok(128) and op.band()[0] <= 160
    with snF.SpmmTimer() as timer:                       # (only launches that go through the product's dispatch are tagged)
        x1, y1, gy, gx = _props(op, 1, 128, 60, check_forms=False)
        tags = {t[0].split("/")[-1] for t in timer.results()}
    assert tags == {"ring"}
    for o, xin, yref in ((op, x1, y1), (op.t(), gy, gx)):
        yc = torch.empty_like(yref)
        kernels.spmm_csr(o.rowptr, o.colind, o.vals, o.shape[0], o.shape[1], xin, yc, 1)
        assert torch.equal(yc, yref)                    # ring == generic CSR kernel, bit for bit
    _oracle_slice(op, x1, y1, 1, 128, int(op.row_offsets[3]))
    r = op.rb4()
    yr = torch.empty_like(y1)
    kernels.spmm_rb4(r[0], r[1], r[2], M, K, x1, yr)
    assert torch.equal(yr, y1)
    e, g = torch.randn(M, 128, device=DEV), torch.randn(M, 128, device=DEV)
    fused = torch.empty(M, 128, device=DEV)
    kernels.spmm_ring(op.rowptr, op.colind, op.vals, M, K, x1, fused, e, g)
    unfused = torch.empty(M, 128, device=DEV)
    kernels.elu_bwd(y1, e, unfused, False, None, g)
    assert torch.equal(fused, unfused)
    ys = torch.empty(M, 128, device=DEV)
    part = kernels.spmm_ring_stats(op.rowptr, op.colind, op.vals, M, K, x1, ys)
    assert torch.equal(ys, y1)
    ref = kernels.colstats(y1)
    scale = torch.stack([y1.double().abs().sum(0), ref[1]]) + 1e-30
    assert ((part.sum(0) - ref).abs() / scale).max().item() < 1e-6


def test_config5_ragged_packed_batch_and_int32_limit():
    """BASELINE configs[4]: 128 meshes with 1 000 .. 20 000 vertices, Dirac operators, C = 128 (N = 32), as a PACKED batch
    (no padding): properties + the padded batch of the same meshes gives the same rows; sizes past int32 are refused."""
    from surfacenetworks_amd import _lib, functional as snF, mesh_ops
    from surfacenetworks_amd.operators import OperatorPool

    rng = np.random.default_rng(5)
    vs = rng.integers(1000, 20001, size=128)
    Dis, DiAs = [], []
    for v in vs:
        n = int(np.sqrt(v))
        V, F_ = mesh_ops.grid_cloth(n, int(v) // n, rng)
        Di, DiA = mesh_ops.dirac(V, F_)
        Dis.append(Di.astype(np.float32))
        DiAs.append(DiA.astype(np.float32))
    sel = np.arange(128)
    # DiA (and DiA^T through the adjoint identity) at the ragged full size: properties, every storage form == the generic CSR
    # kernel bit for bit, the first meshes against the C oracle
    poolA = OperatorPool(DiAs, DEV, want_bsr4=True)
    opA = poolA.assemble(sel)
    assert opA.shape == (int(poolA.rows.sum()), int(poolA.cols.sum())) and opA.nnz == sum(m.nnz for m in DiAs)
    xa, ya, _, _ = _props(opA, 4, 32, 51)
    _oracle_slice(opA, xa, ya, 4, 32, int(opA.row_offsets[2]))
    del opA, poolA, xa, ya
    pool = OperatorPool(Dis, DEV, want_bsr4=True)
    op = pool.assemble(sel)                                      # packed
    assert op.shape == (int(pool.rows.sum()), int(pool.cols.sum())) and op.nnz == sum(m.nnz for m in Dis)
    x1, y1, gy, gx = _props(op, 4, 32, 50)
    _oracle_slice(op, x1, y1, 4, 32, int(op.row_offsets[2]))
    # padded batch (the reference's layout) of the same meshes: identical rows, zeros in the padding
    pad = pool.assemble(sel, int(pool.rows.max()), int(pool.cols.max()))
    s0, s1 = int(pool.rows.max()) // 4, int(pool.cols.max()) // 4
    xp = torch.zeros(128, s1, 128, device=DEV)
    ro, co = op.row_offsets // 4, op.col_offsets // 4
    for b in range(128):
        xp[b, : co[b + 1] - co[b]] = x1[co[b]: co[b + 1]]
    yp = snF.spmm(pad, xp.view(-1, 128), 4).view(128, s0, 128)
    for b in (0, 1, 17, 127):
        n = int(ro[b + 1] - ro[b])
        assert torch.equal(yp[b, :n], y1[ro[b]: ro[b + 1]]) and not yp[b, n:].any()
    # ---- the int32 index limit: one row / entry past it is refused with SN_E_RANGE before anything is touched ----
    lib = _lib.load()
    big = 2**31
    SN_E_RANGE = -3
    assert lib.sn_spmm_q3_f32(None, None, big // 4, 10, 10, None, 128, 4, 32, None, 128, 4, None) == SN_E_RANGE
    assert lib.sn_spmm_q3_f32(None, None, 10, 10, big, None, 128, 4, 32, None, 128, 4, None) == SN_E_RANGE
    assert lib.sn_spmm_csr_f32(None, None, None, big, 10, 10, None, 32, 1, 32, None, 32, 1, None) == SN_E_RANGE
    assert lib.sn_spmm_csr_f32(None, None, None, 10, 10, big, None, 32, 1, 32, None, 32, 1, None) == SN_E_RANGE
    assert lib.sn_blockdiag_concat_ragged_i32(None, None, None, None, 1, big, 10, 10, 1, None, None, None, None) == SN_E_RANGE
    assert lib.sn_blockdiag_concat_i32(None, None, None, None, 1024, 4 * 600_000, 10, 10, 1, None, None, None, None) == SN_E_RANGE
    assert lib.sn_spmm_rb4_f32(None, None, None, big, 10, 10, None, 128, 128, None, 128, None) == SN_E_RANGE
    assert lib.sn_spmm_csr_ring_f32(None, None, None, big, big, 10, None, 128, 128, None, 128, None) == SN_E_RANGE
    # ... and the largest sizes that still fit are accepted as far as the argument checks go (null operands: SN_E_NULL)
    assert lib.sn_spmm_csr_f32(None, None, None, big - 2, 10, 10, None, 32, 1, 32, None, 32, 1, None) == -1


@pytest.mark.parametrize("kind", ["dir_c3", "lap_c4"])
def test_whole_blocks_at_config_size_equal_unfused_composition(kind):
    """One DirResNet2 at config-3 size (64 x 71x71 cloth, C = 128) and one LapResNet2 at config-4 size (FAUST-sized meshes
    padded to 7000): forward + backward of the fused whole-block path against the unfused composition (F.elu + spmm +
    torch.cat + BatchNorm1d + Linear), outputs, input gradients, parameter gradients and running statistics."""
    import copy

    import torch.nn.functional as F

    import surfacenetworks_amd.utils_pt as U
    from helpers import deterministic_init, rel_err
    from surfacenetworks_amd import functional as snF, mesh_ops
    from surfacenetworks_amd.operators import OperatorPool

    rng = np.random.default_rng(8)
    C = 128
    gen = torch.Generator(device=DEV).manual_seed(1)
    if kind == "dir_c3":
        B = 64
        meshes = [mesh_ops.dirac(*mesh_ops.grid_cloth(71, 71, rng)) for _ in range(2)]
        nv, nf = 5041, 9800
        Di = OperatorPool([m[0].astype(np.float32) for m in meshes], DEV, want_bsr4=True).assemble(np.arange(B) % 2, 4 * nf, 4 * nv)
        DiA = OperatorPool([m[1].astype(np.float32) for m in meshes], DEV, want_bsr4=True).assemble(np.arange(B) % 2, 4 * nv, 4 * nf)
        base = deterministic_init(U.DirResNet2(C), 5)
    else:
        B, nv = 8, 7000
        mats = [mesh_ops.laplacian(*mesh_ops.torus_grid(65, 106, rng)).astype(np.float32) for _ in range(2)]
        L = OperatorPool(mats, DEV).assemble(np.arange(B) % 2, nv, nv)
        # (mass-normalised cotangent operators have entries ~1e3-1e4: scale the input so the block stays well conditioned)
        base = deterministic_init(U.LapResNet2(C), 5)
    res = []
    for fused in (True, False):
        blk = copy.deepcopy(base).to(DEV).train()
        v = (0.1 * torch.randn(B, nv, C, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))).requires_grad_(True)
        if kind == "dir_c3":
            f = (0.1 * torch.randn(B, nf, C, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))).requires_grad_(True)
            if fused:
                vo, fo = blk(Di, DiA, v, f)
            else:
                x_in, f_in = F.elu(v), F.elu(f)
                y = snF.spmm(Di, x_in.reshape(B * nv, C), 4).view(B, nf, C)
                fo = blk.bn_fc0.fc(blk.bn_fc0.bn(torch.cat([f_in, y], 2).view(-1, 2 * C))).view(B, nf, C)
                z = snF.spmm(DiA, F.elu(fo).reshape(B * nf, C), 4).view(B, nv, C)
                vo = v + blk.bn_fc1.fc(blk.bn_fc1.bn(torch.cat([x_in, z], 2).view(-1, 2 * C))).view(B, nv, C)
            (vo.square().mean() + fo.square().mean()).backward()
            outs = [vo, fo, v.grad, f.grad]
        else:
            if fused:
                vo = blk(L, None, v)
            else:
                x = F.elu(v)
                x = blk.bn_fc0.fc(blk.bn_fc0.bn(torch.cat([x, snF.spmm(L, x.reshape(B * nv, C), 1).view(B, nv, C)], 2).view(-1, 2 * C)))
                x = F.elu(x.view(B, nv, C))
                x = blk.bn_fc1.fc(blk.bn_fc1.bn(torch.cat([x, snF.spmm(L, x.reshape(B * nv, C), 1).view(B, nv, C)], 2).view(-1, 2 * C)))
                vo = x.view(B, nv, C) + v
            vo.square().mean().backward()
            outs = [vo, v.grad]
        res.append([t.detach().float().cpu().numpy() for t in outs] +
                   [p.grad.detach().cpu().numpy() for p in blk.parameters()] +
                   [b_.detach().float().cpu().numpy() for n_, b_ in blk.named_buffers() if "running" in n_])
        del blk, v, outs
        torch.cuda.empty_cache()
    # fp32 both ways; the fused path sums BatchNorm statistics in fp64 and forms the Linear products on the split-bf16 matrix
    # pipe, torch's composition uses its own reductions: agreement to a few 1e-5 of the largest entry of each tensor
    for a, b in zip(*res):
        assert a.shape == b.shape and rel_err(a, b) < 5e-5, (kind, a.shape, rel_err(a, b))


def test_debug_validator_flags_malformed_operators():
    from surfacenetworks_amd import kernels

    rp = torch.tensor([0, 2, 2, 5], dtype=torch.int32, device=DEV)
    ci = torch.tensor([0, 3, 1, 2, 4], dtype=torch.int32, device=DEV)
    va = torch.ones(5, device=DEV)
    assert kernels.validate_csr(rp, ci, va, 3, 5) == 0
    assert kernels.validate_csr(rp, ci, va, 3, 4) & 8                                  # column 4 >= K
    bad = ci.clone()
    bad[1] = 0
    assert kernels.validate_csr(rp, bad, va, 3, 5) & 16                                # duplicate / unsorted row
    assert kernels.validate_csr(torch.tensor([0, 3, 2, 5], dtype=torch.int32, device=DEV), ci, va, 3, 5) & 2
    assert kernels.validate_csr(torch.tensor([1, 2, 2, 5], dtype=torch.int32, device=DEV), ci, va, 3, 5) & 1
    assert kernels.validate_csr(torch.tensor([0, 2, 2, 4], dtype=torch.int32, device=DEV), ci, va, 3, 5) & 4
    nf = va.clone()
    nf[2] = float("inf")
    assert kernels.validate_csr(rp, ci, nf, 3, 5) & 32


def test_debug_validation_switch_checks_operators_as_they_are_built(monkeypatch):
    """SN_DEBUG_VALIDATE=1 (operators._DEBUG_VALIDATE): the constructor itself runs the validator on device CSR arrays — a
    well-formed operator is built and multiplies as usual (transposes and batches included), a malformed one raises."""
    from surfacenetworks_amd import operators
    from surfacenetworks_amd.functional import spmm

    monkeypatch.setattr(operators, "_DEBUG_VALIDATE", True)
    rp = torch.tensor([0, 2, 2, 5], dtype=torch.int32, device=DEV)
    ci = torch.tensor([0, 3, 1, 2, 4], dtype=torch.int32, device=DEV)
    va = torch.arange(1, 6, dtype=torch.float32, device=DEV)
    op = operators.SparseOperator(rp, ci, va, (3, 5))
    x = torch.randn(5, 8, device=DEV)
    dense = torch.zeros(3, 5, device=DEV)
    dense[[0, 0, 2, 2, 2], [0, 3, 1, 2, 4]] = va
    assert torch.allclose(spmm(op, x), dense @ x, atol=1e-6)
    assert torch.allclose(spmm(op.t(), torch.ones(3, 8, device=DEV)), dense.t() @ torch.ones(3, 8, device=DEV), atol=1e-6)
    L = sp.random(40, 40, 0.2, "csr", np.float32, random_state=2)
    L.sort_indices()
    pool = operators.OperatorPool([L, L], DEV)
    batch = pool.assemble([0, 1])
    assert tuple(batch.shape) == (80, 80) and batch.t().nnz == 2 * L.nnz
    with pytest.raises(ValueError, match="column index out of range"):
        operators.SparseOperator(rp, ci, va, (3, 4))
    bad = ci.clone()
    bad[1] = 0
    with pytest.raises(ValueError, match="row not sorted"):
        operators.SparseOperator(rp, bad, va, (3, 5))
