"""GPU parity of the BatchNorm+Linear helper kernels (colstats fp64, fp32-MFMA weight gradient, affine epilogue) and of
the fused bn_linear Function against nn.BatchNorm1d + nn.Linear (the layers GraphConv1x1 is made of in the reference,
src/utils/utils_pt.py:83-99)."""
import numpy as np
import pytest
import torch

from helpers import rel_err
from oracle import c_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"

from surfacenetworks_amd import functional as snF, kernels  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("rows,C", [(1, 256), (63, 256), (1000, 128), (20011, 256), (4097, 64), (300, 120), (77, 3), (5000, 6)])
def test_colstats(rows, C):
    rng = np.random.default_rng(rows + C)
    x = (rng.standard_normal((rows, C)) * 3 + 100.0 * rng.standard_normal(C)).astype(np.float32)   # |mean| >> std
    want = c_oracle.colstats_raw(x.ctypes.data, C, rows, C)
    got = kernels.colstats(dev(x)).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    # strided view (first half of a wider buffer)
    if C % 4 == 0:
        wide = np.concatenate([x, x[:, ::-1]], 1).copy()
        got2 = kernels.colstats(dev(wide)[:, :C]).cpu().numpy()
        assert np.allclose(got2, want, rtol=1e-12, atol=0)


@pytest.mark.parametrize("rows", [1, 2, 7, 255, 256, 1000, 33333, 400001])
@pytest.mark.parametrize("J,C", [(128, 256), (128, 128), (120, 128), (64, 128), (4, 256)])
def test_wgrad_mfma(rows, J, C):
    rng = np.random.default_rng(rows * 7 + J + C)
    dy = rng.standard_normal((rows, J)).astype(np.float32)
    x = rng.standard_normal((rows, C)).astype(np.float32)
    # asymmetric operands catch transposed / permuted tiles
    x[:, 0] += 3.0
    dy[:, min(1, J - 1)] -= 2.0
    want = dy.astype(np.float64).T @ x.astype(np.float64)
    got = kernels.wgrad(dev(dy), dev(x)).cpu().numpy()
    assert got.shape == (J, C)
    assert rel_err(got, want) < 2e-6
    got_g, got_s = kernels.wgrad(dev(dy), dev(x), None, want_colsum=True)
    assert np.array_equal(got_g.cpu().numpy(), got)
    assert rel_err(got_s.cpu().numpy(), dy.astype(np.float64).sum(0)) < 2e-6
    # centred operand (BatchNorm backward passes the batch mean)
    cen = (rng.standard_normal(C) * 5).astype(np.float32)
    got_c = kernels.wgrad(dev(dy), dev(x), dev(cen)).cpu().numpy()
    assert rel_err(got_c, dy.astype(np.float64).T @ (x - cen).astype(np.float64)) < 2e-6
    # x as a strided half of a concat buffer
    if rows > 1:
        wide = dev(np.concatenate([x, np.zeros_like(x)], 1))
        got2 = kernels.wgrad(dev(dy), wide[:, :C]).cpu().numpy()
        assert np.array_equal(got2, got)


def test_wgrad_exact_tile_mapping():
    """dy = one-hot rows, x = distinct integers: G must reproduce x rows exactly at the right (j, c)."""
    rows, J, C = 64, 128, 256
    dy = np.zeros((rows, J), np.float32)
    for r in range(rows):
        dy[r, (5 * r + 3) % J] = 1.0
    x = (np.arange(rows * C, dtype=np.float32).reshape(rows, C) % 4093)
    want = dy.T @ x
    got = kernels.wgrad(dev(dy), dev(x)).cpu().numpy()
    assert np.array_equal(got, want)


def test_affine_cols_acc():
    rng = np.random.default_rng(0)
    for rows, C in [(1001, 256), (17, 7)]:
        dx = rng.standard_normal((rows, C)).astype(np.float32)
        x = rng.standard_normal((rows, C)).astype(np.float32)
        B = rng.standard_normal(C).astype(np.float32)
        Cc = rng.standard_normal(C).astype(np.float32)
        for cen in (None, rng.standard_normal(C).astype(np.float32)):
            want = dx.copy()
            c_oracle.affine_cols_acc_raw(want.ctypes.data, C, x.ctypes.data, C, B, Cc, rows, C, cen)
            d = dev(dx)
            kernels.affine_cols_acc(d, dev(x), dev(B), dev(Cc), None if cen is None else dev(cen))
            assert np.array_equal(d.cpu().numpy(), want)


@pytest.mark.parametrize("rows,C,J,train", [(5000, 256, 128, True), (5000, 256, 128, False), (3001, 128, 120, True),
                                             (2000, 128, 64, True), (999, 64, 64, True), (50, 6, 128, True)])
def test_bn_linear_matches_torch_layers(rows, C, J, train):
    """Fused function vs nn.BatchNorm1d + nn.Linear in float64 (the exact answer) and float32 (the reference's layers)."""
    torch.manual_seed(rows + C + J)
    bn = torch.nn.BatchNorm1d(C).to(DEV)
    fc = torch.nn.Linear(C, J).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-1, 1)
        bn.running_var.uniform_(0.5, 2)
    import copy

    bn64, fc64 = copy.deepcopy(bn).double(), copy.deepcopy(fc).double()
    bn32, fc32 = copy.deepcopy(bn), copy.deepcopy(fc)
    for m in (bn, bn64, bn32):
        m.train(train)
    x0 = torch.randn(rows, C, device=DEV) * 2 + 5 * torch.randn(C, device=DEV)
    gy = torch.randn(rows, J, device=DEV)
    outs = {}
    for tag, b, f, dt in (("fused", bn, fc, torch.float32), ("t32", bn32, fc32, torch.float32), ("t64", bn64, fc64, torch.float64)):
        x = x0.to(dt).clone().requires_grad_(True)
        y = snF.bn_linear(x, b, f) if tag == "fused" else f(b(x))
        y.backward(gy.to(dt))
        outs[tag] = [y.detach(), x.grad, b.weight.grad, b.bias.grad, f.weight.grad, f.bias.grad, b.running_mean.clone(),
                     b.running_var.clone()]
    names = ["y", "dx", "dgamma", "dbeta", "dW", "db", "running_mean", "running_var"]
    for n, a, b32, b64 in zip(names, outs["fused"], outs["t32"], outs["t64"]):
        e_f = rel_err(a.double().cpu().numpy(), b64.cpu().numpy())
        e_t = rel_err(b32.double().cpu().numpy(), b64.cpu().numpy())
        assert e_f <= max(4 * e_t, 2e-6), (n, "fused err", e_f, "torch fp32 err", e_t)
    assert int(bn.num_batches_tracked) == int(bn32.num_batches_tracked)


def test_segment_kernels():
    import cpu_kernels as ck

    rng = np.random.default_rng(5)
    for per, nseg, C in [(301, 5, 128), (64, 3, 64), (5041, 4, 128), (7, 9, 16)]:
        rows = per * nseg
        x = rng.standard_normal((rows, 2 * C)).astype(np.float32)
        mask = (rng.random(rows) < 0.8).astype(np.float32)
        xd, md = dev(x), dev(mask)
        for m_np, m_d in ((mask, md), (None, None)):
            want = ck.segment_colsum(torch.from_numpy(x[:, :C].copy()), None if m_np is None else torch.from_numpy(m_np), per, nseg)
            got = kernels.segment_colsum(xd[:, :C], m_d, per, nseg)
            assert rel_err(got.cpu().numpy(), want.numpy()) < 1e-6
        src = rng.standard_normal((nseg, C)).astype(np.float32)
        dst = torch.zeros(rows, 2 * C, device=DEV)
        kernels.bcast_rows(dev(src), dst[:, C:], per)
        assert np.array_equal(dst[:, C:].cpu().numpy(), np.repeat(src, per, 0)) and float(dst[:, :C].abs().sum()) == 0
        g = rng.standard_normal((rows, C)).astype(np.float32)
        out = c_oracle.elu(rng.standard_normal((rows, C)).astype(np.float32) * 2)
        gs = torch.empty(rows, C, device=DEV)
        kernels.elu_bwd_bcast(dev(g), dev(out), dev(src), md, gs, per)
        want = torch.empty(rows, C)
        ck.elu_bwd_bcast(torch.from_numpy(g), torch.from_numpy(out), torch.from_numpy(src), torch.from_numpy(mask), want, per)
        assert np.allclose(gs.cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 1000, 40001])
@pytest.mark.parametrize("K", [128, 256])
def test_linear_fwd_mfma(rows, K):
    rng = np.random.default_rng(rows + K)
    J = 128
    xw = rng.standard_normal((rows, K + 64)).astype(np.float32)        # x is a strided view
    W = (rng.standard_normal((J, K)) / np.sqrt(K)).astype(np.float32)
    W[5, 7] = 3.0                                                        # asymmetric marker
    b = rng.standard_normal(J).astype(np.float32)
    res = rng.standard_normal((rows, J)).astype(np.float32)
    x = dev(xw)[:, :K]
    want = xw[:, :K].astype(np.float64) @ W.astype(np.float64).T + b
    y = kernels.linear_fwd(x, dev(W), dev(b))
    assert rel_err(y.cpu().numpy(), want) < 2e-6
    cat = torch.zeros(rows, 2 * J, device=DEV)
    y2 = kernels.linear_fwd(x, dev(W), dev(b), residual=dev(res), y_elu=cat[:, :J])
    want2 = want + res
    assert rel_err(y2.cpu().numpy(), want2) < 2e-6
    assert np.allclose(cat[:, :J].cpu().numpy(), np.where(want2 > 0, want2, np.expm1(want2)), rtol=1e-5, atol=1e-5)
    assert float(cat[:, J:].abs().sum()) == 0


@pytest.mark.parametrize("rows", [1, 33, 1000, 40001])
@pytest.mark.parametrize("C", [128, 256])
def test_linear_dgrad_mfma(rows, C):
    rng = np.random.default_rng(rows + C)
    J = 128
    dy = rng.standard_normal((rows, J)).astype(np.float32)
    W = (rng.standard_normal((J, C)) / np.sqrt(J)).astype(np.float32)
    W[3, 11] = -2.5
    x = rng.standard_normal((rows, C)).astype(np.float32)
    cen, B, Cc = [rng.standard_normal(C).astype(np.float32) for _ in range(3)]
    want = dy.astype(np.float64) @ W.astype(np.float64)
    got = kernels.linear_dgrad(dev(dy), dev(W))
    assert rel_err(got.cpu().numpy(), want) < 2e-6
    got2 = kernels.linear_dgrad(dev(dy), dev(W), dev(x), dev(cen), dev(B), dev(Cc))
    want2 = want + (x.astype(np.float64) - cen) * B + Cc
    assert rel_err(got2.cpu().numpy(), want2) < 2e-6


# ---- the exact three-piece bf16 split behind the Linear GEMMs (sn_gemm.hip, wgrad_x3_k) ------------------------------
def _three_piece(rng, shape, pieces, keep=None):
    """fp32 values whose bf16 pieces (8 bits each, by truncation) are small non-zero integers scaled by 1, 2^-8, 2^-16:
    `pieces` says which of (h, m, l) are present; `keep` leaves only that many non-zero entries per row."""
    v = np.zeros(shape, np.float64)
    for p, scale in zip(pieces, (1.0, 2.0 ** -8, 2.0 ** -16)):
        if p:
            v += rng.integers(1, 4, size=shape) * scale       # all pieces positive: truncation splits them as built
    if keep is not None:
        mask = np.zeros(shape, bool)
        for r in range(shape[0]):
            mask[r, rng.choice(shape[1], keep, replace=False)] = True
        v *= mask
    return v.astype(np.float32)


@pytest.mark.parametrize("xp,wp", [((1, 1, 1), (1, 0, 0)), ((1, 0, 0), (1, 1, 1)), ((1, 1, 0), (1, 1, 0)),
                                   ((0, 1, 0), (1, 0, 0)), ((0, 0, 1), (1, 0, 0)), ((1, 0, 0), (0, 0, 1))])
def test_split_bf16_keeps_every_partial_product(xp, wp):
    """Each of the six retained products (hh, hm, mh, hl, lh, mm) is needed for these operands to come out EXACT; the three
    dropped ones (ml, lm, ll) are absent by construction.  At most 16 non-zero terms of at most 4·4 per output keep every
    sum below 2^8 with a resolution of 2^-16, i.e. exactly representable in fp32."""
    rng = np.random.default_rng(sum(xp) * 7 + sum(wp))
    rows, K, J = 96, 256, 128
    x = _three_piece(rng, (rows, K), xp, keep=16)
    W = _three_piece(rng, (J, K), wp)
    b = np.zeros(J, np.float32)
    want = x.astype(np.float64) @ W.astype(np.float64).T
    assert np.array_equal(want.astype(np.float32).astype(np.float64), want)          # the test's own premise
    got = kernels.linear_fwd(dev(x), dev(W), dev(b)).cpu().numpy().astype(np.float64)
    assert np.array_equal(got, want)
    # dgrad: dx = dy·W with dy (rows, J), W (J, C)
    dy = _three_piece(rng, (rows, J), xp, keep=16)
    Wd = _three_piece(rng, (J, K), wp)
    assert np.array_equal(kernels.linear_dgrad(dev(dy), dev(Wd)).cpu().numpy().astype(np.float64),
                          dy.astype(np.float64) @ Wd.astype(np.float64))
    # wgrad: G = dyᵀ·x, contraction over 16 rows
    dyg = _three_piece(rng, (16, J), xp)
    xg = _three_piece(rng, (16, K), wp)
    assert np.array_equal(kernels.wgrad(dev(dyg), dev(xg)).cpu().numpy().astype(np.float64),
                          dyg.astype(np.float64).T @ xg.astype(np.float64))


def test_split_bf16_is_as_accurate_as_an_fp32_fma_chain():
    """Random operands with a wide dynamic range, errors against fp64.  Forward / dgrad keep the correction products in
    their own accumulators and must be at least as accurate as a plain fp32 dot product (numpy fp32 matmul).  The weight
    gradient adds all six partial products of a term into one accumulator per tile (8 tiles per wave leave no registers
    for more), which costs up to ~2.5x the rounding of a single chain — still below the library fp32 GEMM the reference
    would call (torch matmul), which is the bound asserted here next to 3x the numpy figure."""
    rng = np.random.default_rng(99)
    rows, K, J = 4096, 256, 128
    x = (rng.standard_normal((rows, K)) * np.exp(rng.standard_normal((rows, K)))).astype(np.float32)
    W = (rng.standard_normal((J, K)) / 16).astype(np.float32)
    ref = x.astype(np.float64) @ W.astype(np.float64).T
    err_fp32 = np.abs((x @ W.T).astype(np.float64) - ref).max()
    got = kernels.linear_fwd(dev(x), dev(W), dev(np.zeros(J, np.float32))).cpu().numpy().astype(np.float64)
    assert np.abs(got - ref).max() <= 1.01 * err_fp32
    dy = rng.standard_normal((rows, J)).astype(np.float32)
    refd = dy.astype(np.float64) @ W.astype(np.float64)
    err_d = np.abs((dy @ W).astype(np.float64) - refd).max()
    assert np.abs(kernels.linear_dgrad(dev(dy), dev(W)).cpu().numpy().astype(np.float64) - refd).max() <= 1.01 * err_d
    refg = dy.astype(np.float64).T @ x.astype(np.float64)
    err_g = np.abs((dy.T @ x).astype(np.float64) - refg).max()
    err_lib = np.abs((dev(dy).t() @ dev(x)).cpu().numpy().astype(np.float64) - refg).max()
    err_k = np.abs(kernels.wgrad(dev(dy), dev(x)).cpu().numpy().astype(np.float64) - refg).max()
    assert err_k <= 3.0 * err_g and err_k <= err_lib


@pytest.mark.parametrize("with_gadd", [True, False])
@pytest.mark.parametrize("rows", [1, 33, 1000, 40001])
@pytest.mark.parametrize("C", [128, 256])
def test_linear_dgrad_through_elu(rows, C, with_gadd):
    """sn_linear_dgrad_elu_f32 == sn_linear_dgrad_f32 followed by the ELU backward on the first half of the columns."""
    if not kernels.linear_dgrad_elu_supported(128, C):
        pytest.skip("fused epilogues exist in the split-bf16 kernels only (SN_GEMM_VARIANT=0 is the A/B baseline)")
    rng = np.random.default_rng(rows + C)
    J, h = 128, C // 2
    dy = rng.standard_normal((rows, J)).astype(np.float32)
    W = (rng.standard_normal((J, C)) / np.sqrt(J)).astype(np.float32)
    x = rng.standard_normal((rows, C)).astype(np.float32)          # first half plays the activation output e
    cen, B, Cc = [rng.standard_normal(C).astype(np.float32) for _ in range(3)]
    gadd = rng.standard_normal((rows, h)).astype(np.float32) if with_gadd else None
    full = kernels.linear_dgrad(dev(dy), dev(W), dev(x), dev(cen), dev(B), dev(Cc)).cpu().numpy()
    f = np.where(x[:, :h] > 0, np.float32(1), x[:, :h] + np.float32(1)).astype(np.float32)
    want_act = full[:, :h] * f + (gadd if with_gadd else 0)
    dx_hi, gact = kernels.linear_dgrad_elu(dev(dy), dev(W), dev(x), dev(cen), dev(B), dev(Cc), dev(gadd) if with_gadd else None)
    assert np.array_equal(dx_hi.cpu().numpy(), full[:, h:])        # the untouched half: same kernel arithmetic
    assert np.allclose(gact.cpu().numpy(), want_act, rtol=1e-6, atol=1e-6)


# ---- half-width global-average stage kernels ---------------------------------------------------------------------------
@pytest.mark.parametrize("nseg,per", [(3, 150), (5, 32), (2, 5041)])
def test_avg_stage_small_kernels(nseg, per):
    rng = np.random.default_rng(nseg * 1000 + per)
    C, J = 128, 128
    ssum = rng.standard_normal((nseg, C)).astype(np.float32) * per
    inv = (1.0 / rng.integers(per // 2 + 1, per + 1, size=nseg)).astype(np.float32)
    stats1 = rng.standard_normal((2, C))
    m, stats = kernels.avg_fwd_prep(dev(ssum), dev(inv), per, dev(stats1))
    m_want = ssum * inv[:, None]
    assert np.array_equal(m.cpu().numpy(), m_want)
    md = m_want.astype(np.float64)
    want = np.concatenate([stats1, np.stack([per * md.sum(0), per * (md * md).sum(0)])], 1)
    assert np.allclose(stats.cpu().numpy(), want, rtol=1e-13, atol=0)
    Wf = rng.standard_normal((J, 2 * C)).astype(np.float32)
    bf = rng.standard_normal(J).astype(np.float32)
    segb = kernels.seg_affine(m, dev(Wf)[:, C:], dev(bf)).cpu().numpy()
    assert rel_err(segb, md @ Wf[:, C:].astype(np.float64).T + bf) < 1e-6
    Sg = rng.standard_normal((nseg, J)).astype(np.float32)
    G1 = rng.standard_normal((J, C)).astype(np.float32)
    mu2, B2, C2 = [rng.standard_normal(C).astype(np.float32) for _ in range(3)]
    Gc = kernels.avg_bwd_gc(dev(G1), dev(Sg), m, dev(mu2)).cpu().numpy()
    assert np.array_equal(Gc[:, :C], G1)
    assert rel_err(Gc[:, C:], Sg.astype(np.float64).T @ (md - mu2)) < 1e-6
    sv = kernels.avg_bwd_segvec(dev(Sg), dev(Wf)[:, C:], m, dev(mu2), dev(B2), dev(C2), dev(inv), per).cpu().numpy()
    want_sv = (Sg.astype(np.float64) @ Wf[:, C:].astype(np.float64) + per * ((md - mu2) * B2 + C2)) * inv[:, None]
    assert rel_err(sv, want_sv) < 1e-6


@pytest.mark.parametrize("nseg,per", [(3, 150), (7, 33), (2, 5041)])
def test_linear_fwd_with_per_mesh_bias_and_dgrad_through_elu(nseg, per):
    if not kernels.avg_stage_supported(128, 128, per):
        pytest.skip("fused epilogues exist in the split-bf16 kernels only (SN_GEMM_VARIANT=0 is the A/B baseline)")
    rng = np.random.default_rng(per)
    rows, C, J = nseg * per, 128, 128
    xw = rng.standard_normal((rows, 2 * C)).astype(np.float32)         # e = first half of a wider buffer
    W = (rng.standard_normal((J, 2 * C)) / 16).astype(np.float32)      # only the first C columns are multiplied
    segb = rng.standard_normal((nseg, J)).astype(np.float32)
    res = rng.standard_normal((rows, J)).astype(np.float32)
    seg = np.arange(rows) // per
    want = xw[:, :C].astype(np.float64) @ W[:, :C].astype(np.float64).T + segb[seg] + res
    cat = torch.zeros(rows, 2 * J, device=DEV)
    y = kernels.linear_fwd_segbias(dev(xw)[:, :C], dev(W)[:, :C], dev(segb), per, dev(res), cat[:, :J])
    assert rel_err(y.cpu().numpy(), want) < 2e-6
    assert np.allclose(cat[:, :J].cpu().numpy(), np.where(want > 0, want, np.expm1(want)), rtol=1e-5, atol=1e-5)
    # ELU copy only (no y)
    cat2 = torch.zeros(rows, 2 * J, device=DEV)
    assert kernels.linear_fwd_segbias(dev(xw)[:, :C], dev(W)[:, :C], dev(segb), per, dev(res), cat2[:, :J], want_y=False) is None
    assert torch.equal(cat2, cat)
    # input gradient through the activation, all C columns, per-mesh vector (masked per row) added before elu'
    dy = rng.standard_normal((rows, J)).astype(np.float32)
    Wd = (rng.standard_normal((J, 2 * C)) / np.sqrt(J)).astype(np.float32)
    cen, B, Cc = [rng.standard_normal(C).astype(np.float32) for _ in range(3)]
    segv = rng.standard_normal((nseg, C)).astype(np.float32)
    mask = (rng.random(rows) > 0.2).astype(np.float32)
    gadd = rng.standard_normal((rows, C)).astype(np.float32)
    e = xw[:, :C]
    pre = dy.astype(np.float64) @ Wd[:, :C].astype(np.float64) + (e.astype(np.float64) - cen) * B + Cc + mask[:, None] * segv[seg]
    want_g = pre * np.where(e > 0, 1.0, e.astype(np.float64) + 1.0) + gadd
    got = kernels.linear_dgrad_eluseg(dev(dy), dev(Wd)[:, :C], dev(xw)[:, :C], dev(cen), dev(B), dev(Cc), dev(segv), per,
                                      dev(mask), dev(gadd)).cpu().numpy()
    assert rel_err(got, want_g) < 2e-6
    got2 = kernels.linear_dgrad_eluseg(dev(dy), dev(Wd)[:, :C], dev(xw)[:, :C], dev(cen), dev(B), dev(Cc), dev(segv), per).cpu().numpy()
    pre2 = pre - mask[:, None] * segv[seg] + segv[seg]
    assert rel_err(got2, pre2 * np.where(e > 0, 1.0, e.astype(np.float64) + 1.0)) < 2e-6


@pytest.mark.parametrize("nseg,per", [(3, 150), (7, 33), (2, 5041), (300, 40)])
def test_wgrad_with_per_mesh_column_sums(nseg, per):
    """sn_wgrad_seg_f32: mesh-aligned row slabs — same G and colsum(dy) as the flat split, plus per-mesh sums of dy."""
    if not kernels.linear_dgrad_elu_supported(128, 128):
        pytest.skip("split-bf16 kernels only (SN_GEMM_VARIANT=0 is the A/B baseline)")
    rng = np.random.default_rng(nseg + per)
    rows, J, C = nseg * per, 128, 128
    dy = rng.standard_normal((rows, J)).astype(np.float32)
    x = rng.standard_normal((rows, C)).astype(np.float32)
    cen = rng.standard_normal(C).astype(np.float32)
    G, sdy, seg = kernels.wgrad_seg(dev(dy), dev(x), dev(cen), per)
    assert rel_err(G.cpu().numpy(), dy.astype(np.float64).T @ (x - cen).astype(np.float64)) < 2e-6
    assert rel_err(sdy.cpu().numpy(), dy.astype(np.float64).sum(0)) < 2e-6
    assert rel_err(seg.cpu().numpy(), dy.astype(np.float64).reshape(nseg, per, J).sum(1)) < 2e-6


@pytest.mark.parametrize("nseg,per", [(3, 150), (5, 33), (2, 5041), (1, 7000), (4, 17), (1, 1), (64, 300)])
def test_avg_stats_single_pass(nseg, per):
    """sn_avg_stats_f32 == sn_segment_colsum_f32 + sn_colstats_f32 + sn_avg_fwd_prep_f32 (fp64 accumulation everywhere)."""
    rng = np.random.default_rng(nseg * 31 + per)
    rows, C = nseg * per, 128
    wide = (rng.standard_normal((rows, 2 * C)) + 3.0).astype(np.float32)
    e = dev(wide)[:, :C]
    mask = (rng.random(rows) > 0.3).astype(np.float32)
    inv = (1.0 / np.maximum(mask.reshape(nseg, per).sum(1), 1)).astype(np.float32)
    m, stats = kernels.avg_stats(e, dev(mask), dev(inv), per, nseg)
    ssum = kernels.segment_colsum(e, dev(mask), per, nseg)
    m2, stats2 = kernels.avg_fwd_prep(ssum, dev(inv), per, kernels.colstats(e))
    assert np.array_equal(m.cpu().numpy(), m2.cpu().numpy())
    assert np.allclose(stats.cpu().numpy(), stats2.cpu().numpy(), rtol=1e-12, atol=0)


# ---- first-layer Linear (a handful of input channels): one-pass weight / bias gradient -----------------------------------
@pytest.mark.parametrize("rows", [1, 7, 63, 64, 1000, 33333, 322624])
@pytest.mark.parametrize("J,C", [(128, 6), (128, 3), (64, 3), (16, 8), (256, 1)])
def test_wgrad_thin(rows, J, C):
    rng = np.random.default_rng(rows * 3 + J + C)
    dy = rng.standard_normal((rows, J)).astype(np.float32)
    xw = (rng.standard_normal((rows, C + 3)) + 2.0).astype(np.float32)          # x = a strided view, row stride C + 3
    want = dy.astype(np.float64).T @ xw[:, :C].astype(np.float64)
    G, db = kernels.wgrad_thin(dev(dy), dev(xw)[:, :C])
    scale = np.sqrt(rows) * 3.0
    # fp32 products and <= 64-row fp32 partial sums, fp64 above: error ~ eps * sqrt(64) per partial
    assert np.abs(G.cpu().numpy() - want).max() <= 2e-6 * scale + 1e-6
    assert np.abs(db.cpu().numpy() - dy.astype(np.float64).sum(0)).max() <= 2e-6 * scale + 1e-6
    G2, none = kernels.wgrad_thin(dev(dy), dev(xw)[:, :C], want_bias=False)
    assert none is None and torch.equal(G2, G)                                    # deterministic


def test_wgrad_thin_rejects_what_it_does_not_cover():
    assert not kernels.wgrad_thin_supported(120, 6) and not kernels.wgrad_thin_supported(128, 9)
    with pytest.raises(RuntimeError):
        kernels.wgrad_thin(torch.zeros(10, 120, device=DEV), torch.zeros(10, 6, device=DEV))
    with pytest.raises(RuntimeError):
        kernels.wgrad_thin(torch.zeros(10, 128, device=DEV), torch.zeros(10, 9, device=DEV))
    G, db = kernels.wgrad_thin(torch.zeros(0, 128, device=DEV), torch.zeros(0, 6, device=DEV))     # empty batch: zeros
    assert G.shape == (128, 6) and not G.any() and not db.any()


@pytest.mark.parametrize("cin,cout,bias", [(6, 128, True), (3, 64, True), (3, 128, False)])
def test_first_layer_matches_nn_linear(cin, cout, bias):
    """GraphConv1x1(cin, cout, batch_norm=None) (utils_pt.py:99) through _ThinLinear == nn.Linear, forward and backward."""
    from surfacenetworks_amd import utils_pt as snU

    torch.manual_seed(cin + cout)
    conv = snU.GraphConv1x1(cin, cout, batch_norm=None).to(DEV)
    if not bias:
        conv.fc.bias = None
    ref = torch.nn.Linear(cin, cout, bias=bias).to(DEV).double()
    ref.weight.data.copy_(conv.fc.weight.data)
    if bias:
        ref.bias.data.copy_(conv.fc.bias.data)
    x = torch.randn(5, 700, cin, device=DEV, requires_grad=True)
    xr = x.detach().double().requires_grad_(True)
    g = torch.randn(5, 700, cout, device=DEV)
    y = conv(x)
    y.backward(g)
    yr = ref(xr)
    yr.backward(g.double())
    assert rel_err(y.detach().cpu().numpy(), yr.detach().cpu().numpy()) < 1e-5
    assert rel_err(x.grad.cpu().numpy(), xr.grad.cpu().numpy()) < 1e-5
    assert rel_err(conv.fc.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy()) < 1e-5
    if bias:
        assert rel_err(conv.fc.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy()) < 1e-5


# ---- masked smooth-L1 loss of the ARAP harness (main.py:225-226) ---------------------------------------------------------
@pytest.mark.parametrize("B,V,C,masked", [(3, 50, 120, True), (2, 33, 7, True), (1, 1, 4, False), (4, 5041, 120, True)])
def test_masked_smooth_l1_matches_the_torch_composition(B, V, C, masked):
    import torch.nn.functional as F

    from surfacenetworks_amd import arap

    torch.manual_seed(B * V + C)
    out = (torch.randn(B, V, C, device=DEV) * 1.5).requires_grad_(True)       # |d| on both sides of 1
    tgt = torch.randn(B, V, C, device=DEV)
    mask = (torch.rand(B, V, 1, device=DEV) > 0.3).float() if masked else torch.ones(B, V, 1, device=DEV)
    loss = arap.loss_fn(out, tgt, mask, B)
    (loss * 3.0).backward()                                                     # a non-unit upstream gradient
    o64 = out.detach().double().requires_grad_(True)
    ref = F.smooth_l1_loss(o64 * mask.double(), tgt.double(), reduction="sum") / B
    (ref * 3.0).backward()
    assert abs(loss.item() - ref.item()) <= 1e-6 * abs(ref.item())
    assert rel_err(out.grad.cpu().numpy(), o64.grad.cpu().numpy()) < 1e-6
    # and bit-for-bit the gradient torch's own fp32 chain produces when the scale is a power of two
    if B in (1, 2, 4):
        o32 = out.detach().clone().requires_grad_(True)
        (F.smooth_l1_loss(o32 * mask.expand_as(o32), tgt, reduction="sum") / B).backward()
        out.grad = None
        arap.loss_fn(out, tgt, mask, B).backward()
        assert torch.equal(out.grad, o32.grad)


def test_masked_smooth_l1_empty_batch():
    loss = kernels.masked_smooth_l1_fwd(torch.zeros(0, 120, device=DEV), torch.zeros(0, 120, device=DEV), None, 1.0)
    assert loss.item() == 0.0


# ---- column statistics of the ELU output from the forward GEMM's epilogue ------------------------------------------------
@pytest.mark.parametrize("rows,K,res,segbias", [(rows, K, res, sb) for rows in (1, 31, 33, 1000, 8191, 40001, 627200)
                                                for K, res, sb in ((256, False, False), (256, True, False), (128, True, True), (128, False, False))
                                                if not (sb and rows < 64)])           # (a per-mesh bias needs meshes of >= 32 rows)
def test_forward_gemm_leaves_the_statistics_of_its_elu_output(rows, K, res, segbias):
    if not kernels.elu_stats_supported():
        pytest.skip("16-bit matrix-pipe kernels only (SN_GEMM_VARIANT=0 is the A/B baseline)")
    torch.manual_seed(rows + K)
    x = torch.randn(rows, K, device=DEV)
    W = torch.randn(128, K, device=DEV) / np.sqrt(K)
    b = torch.randn(128, device=DEV)
    r = torch.randn(rows, 128, device=DEV) if res else None
    cat = torch.empty(rows, 256, device=DEV)
    part = kernels.new_elu_stats_part(rows, DEV)
    part.fill_(float("nan"))                                     # every entry the merge reads must have been written
    if segbias:
        per = rows // 2 if rows % 2 == 0 else rows
        segb = torch.randn(rows // per, 128, device=DEV)
        kernels.linear_fwd_segbias(x, W, segb, per, r, cat[:, :128], False, part)
    else:
        kernels.linear_fwd(x, W, b, r, cat[:, :128], False, part)
    cat[:, 128:] = torch.randn(rows, 128, device=DEV) * 2 + 1
    got = kernels.colstats_halves(cat, part).cpu().numpy()
    want = kernels.colstats(cat).cpu().numpy()                   # one fp64 pass over the whole buffer
    assert np.isfinite(got).all()
    assert np.allclose(got, want, rtol=1e-12, atol=1e-9)
    # without the request nothing changes and the plain call still works
    cat2 = torch.empty_like(cat)
    kernels.linear_fwd(x, W, b, r, cat2[:, :128], False) if not segbias else \
        kernels.linear_fwd_segbias(x, W, segb, per, r, cat2[:, :128], False)
    assert torch.equal(cat2[:, :128], cat[:, :128])


# ---- sampler gather ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,vmax,f3,nv,length,shift", [(3, 7, 135, 6, 6, 0), (5, 40, 135, 33, 120, 6), (2, 5041, 132, 5041, 120, 6),
                                                        (1, 1, 9, 1, 3, 0)])
def test_gather_segments(n, vmax, f3, nv, length, shift):
    torch.manual_seed(n + vmax)
    src = torch.randn(n, vmax, f3, device=DEV)
    B = 4
    sid = torch.randint(0, n, (B,), device=DEV)
    off = torch.randint(0, (f3 - shift - length) // 3 + 1, (B,), device=DEV)
    base = (sid * vmax) * f3 + 3 * off + shift
    got = kernels.gather_segments(src, base, nv, f3, length)
    assert got.shape == (B, nv, length)
    for b in range(B):
        c0 = 3 * int(off[b]) + shift
        assert torch.equal(got[b], src[int(sid[b]), :nv, c0:c0 + length])
    assert kernels.gather_segments(src, base[:0], nv, f3, length).shape == (0, nv, length)


@pytest.mark.parametrize("rows", [1, 9, 1000, 40001])
@pytest.mark.parametrize("J,C,bias", [(128, 6, True), (128, 3, True), (64, 3, False), (16, 8, True)])
def test_linear_thin_fwd(rows, J, C, bias):
    torch.manual_seed(rows + J + C)
    xw = torch.randn(rows, C + 2, device=DEV)
    x = xw[:, :C]                                                   # strided rows
    W = torch.randn(J, C, device=DEV)
    b = torch.randn(J, device=DEV) if bias else None
    cat = torch.full((rows, 2 * J), float("nan"), device=DEV)
    y = kernels.linear_thin_fwd(x, W, b, cat[:, :J])
    want = x.double() @ W.double().t() + (b.double() if bias else 0.0)
    assert rel_err(y.cpu().numpy(), want.cpu().numpy()) < 1e-6
    assert torch.equal(cat[:, :J], torch.where(y > 0, y, torch.expm1(y))) or \
        rel_err(cat[:, :J].cpu().numpy(), torch.nn.functional.elu(want).cpu().numpy()) < 1e-6
    assert torch.isnan(cat[:, J:]).all()                            # the other half is not touched
    assert torch.equal(kernels.linear_thin_fwd(x, W, b), y)


# ---- last layer: conv(F.elu(v)) as one node -------------------------------------------------------------------------------
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("C,J", [(128, 120), (128, 128), (64, 10)])
def test_elu_conv_matches_torch_layers(C, J, train):
    """utils.elu_conv1x1(conv, x) == conv(F.elu(x)) with nn.BatchNorm1d + nn.Linear in fp64 (models.py:148-150): output, input
    gradient, parameter gradients, running statistics — also when elu(x) arrives as the activated hand-off of a block."""
    import torch.nn.functional as F

    from surfacenetworks_amd import blocks as snB, utils_pt as U

    torch.manual_seed(C + J)
    B, N = 3, 333
    conv = U.GraphConv1x1(C, J, batch_norm="pre").to(DEV).train(train)
    with torch.no_grad():
        conv.bn.weight.uniform_(0.5, 1.5), conv.bn.bias.uniform_(-0.3, 0.3)
        conv.bn.running_mean.uniform_(-0.2, 0.2), conv.bn.running_var.uniform_(0.5, 2.0)
    bn64 = torch.nn.BatchNorm1d(C).to(DEV).double().train(train)
    fc64 = torch.nn.Linear(C, J).to(DEV).double()
    bn64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in conv.bn.state_dict().items()})
    fc64.load_state_dict({k: v.double() for k, v in conv.fc.state_dict().items()})
    g = torch.randn(B, N, J, device=DEV)
    x0 = torch.randn(B, N, C, device=DEV) * 1.5
    x64 = x0.double().requires_grad_(True)
    y64 = fc64(bn64(F.elu(x64).reshape(B * N, C))).view(B, N, J)
    y64.backward(g.double())
    for handoff in (False, True):
        conv.zero_grad()
        x = x0.clone().requires_grad_(True)
        xin = x
        if handoff:
            if C != 128:
                continue
            cat = torch.empty(B * N, 2 * C, device=DEV)
            cat[:, :C] = F.elu(x0).reshape(B * N, C)
            xin = snB.attach_activated(x.view(B, N, C), cat)
        y = U.elu_conv1x1(conv, xin)
        y.backward(g)
        assert rel_err(y.detach().cpu().numpy(), y64.detach().cpu().numpy()) < 1e-5
        assert rel_err(x.grad.cpu().numpy(), x64.grad.cpu().numpy()) < 1e-5
        assert rel_err(conv.fc.weight.grad.cpu().numpy(), fc64.weight.grad.cpu().numpy()) < 1e-5
        assert rel_err(conv.bn.weight.grad.cpu().numpy(), bn64.weight.grad.cpu().numpy()) < 1e-5
        assert rel_err(conv.bn.bias.grad.cpu().numpy(), bn64.bias.grad.cpu().numpy()) < 1e-5


def test_split_fp16_row_and_column_scaling_covers_the_fp32_range():
    """The default Linear kernels split every operand into two fp16 pieces after scaling each data row / weight column by an
    exact power of two (sn_gemm.hip).  fp16 has 5 exponent bits, so the scaling is what keeps the result fp32-accurate for
    rows of any magnitude: rows at 1e-30, 1e-6 (gradients), 1, 1e5 (activations behind a cotangent Laplacian), 1e30, an
    all-zero row, rows whose elements span 2^40, and weight columns from 1e-20 to 1e20 — each output within a few fp32
    roundings of the fp64 result, relative to sum |x||w| of its own row and column."""
    rng = np.random.default_rng(2024)
    rows, K, J = 640, 256, 128
    x = rng.standard_normal((rows, K)).astype(np.float64)
    mags = np.array([1e-30, 1e-12, 1e-6, 1e-3, 1.0, 1e3, 1e5, 1e12, 1e30, 0.0])
    x *= mags[np.arange(rows) % len(mags)][:, None]
    x[7] *= np.exp2(rng.integers(-20, 20, K))                     # one row spanning 2^40
    x = x.astype(np.float32)
    W = rng.standard_normal((J, K)) * np.array([1e-20, 1e-5, 1.0, 1e4, 1e20])[np.arange(J) % 5][:, None]
    W[3] = 0.0
    W = W.astype(np.float32)
    b = np.zeros(J, np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        ref = x.astype(np.float64) @ W.astype(np.float64).T
        scale = np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64).T
    got = kernels.linear_fwd(dev(x), dev(W), dev(b)).cpu().numpy().astype(np.float64)
    ok = (scale < 1e38) & (scale > 1e-30)                        # outputs that neither overflow nor underflow fp32 itself
    assert ok.mean() > 0.7 and np.isfinite(got[ok]).all()
    err = np.abs(got - ref)[ok] / np.maximum(scale[ok], 1e-300)
    assert err.max() <= 4 * 2.0 ** -24 * np.sqrt(K), err.max()     # a few roundings of the K-term fp32 accumulation
    assert not got[:, 3].any() and not got[9::10].any()            # zero weight column / zero rows stay exactly zero
    # the input gradient takes the same path with the roles of rows / columns kept: tiny gradients, huge weights
    dy = (rng.standard_normal((rows, J)) * np.array([1e-25, 1e-9, 1e-4, 1.0, 1e8])[np.arange(rows) % 5][:, None]).astype(np.float32)
    Wd = (rng.standard_normal((J, K)) * np.array([1e-6, 1.0, 1e6])[np.arange(K) % 3][None, :]).astype(np.float32)
    refd = dy.astype(np.float64) @ Wd.astype(np.float64)
    scaled = np.abs(dy).astype(np.float64) @ np.abs(Wd).astype(np.float64)
    gotd = kernels.linear_dgrad(dev(dy), dev(Wd)).cpu().numpy().astype(np.float64)
    assert (np.abs(gotd - refd) / np.maximum(scaled, 1e-300)).max() <= 4 * 2.0 ** -24 * np.sqrt(J)


# ---- the Linear kernels address their operands through raw buffer windows (sn_gemm.hip RowWindow, wgrad_u_k): every
# operand below is a strided view inside an arena of NaNs, every output a view inside an arena of canaries ---------------
def _arena(rows, width, pad_rows=3, pad_cols=8, seed=0, fill=float("nan")):
    """(arena, view): `view` = rows x width of random values inside a poisoned arena (rows before / after, columns on both
    sides; the view's leading dimension is width + 2·pad_cols, 16-byte aligned)."""
    a = torch.full((rows + 2 * pad_rows, width + 2 * pad_cols), fill, device=DEV)
    v = a[pad_rows:pad_rows + rows, pad_cols:pad_cols + width]
    v.copy_(torch.from_numpy(np.random.default_rng(seed).standard_normal((rows, width)).astype(np.float32)))
    return a, v


def _outside_untouched(arena, view_rows, width, canary, pad_rows=3, pad_cols=8):
    m = torch.ones_like(arena, dtype=torch.bool)
    m[pad_rows:pad_rows + view_rows, pad_cols:pad_cols + width] = False
    return bool((arena[m] == canary).all())


@pytest.mark.parametrize("rows", [1, 31, 32, 33, 95, 1017, 8200])
@pytest.mark.parametrize("K,J", [(128, 128), (256, 128), (128, 120), (256, 4), (128, 64)])
def test_linear_kernels_stay_inside_their_views(rows, K, J):
    """J < 128: the narrow layers (the models' last layer has 120 outputs) — weights, bias, residual and dy of the columns
    that do not exist are never read, nothing is stored to them."""
    from surfacenetworks_amd import _lib
    from surfacenetworks_amd.kernels import _p, _ld, _stream
    rng = np.random.default_rng(rows * 3 + K + J)
    W = dev((rng.standard_normal((J, K)) / np.sqrt(K)).astype(np.float32))
    b = dev(rng.standard_normal(J).astype(np.float32))
    _, x = _arena(rows, K, seed=1)
    _, res = _arena(rows, J, seed=2)
    ya, y = _arena(rows, J, seed=3, fill=7.0)
    ea, e = _arena(rows, J, seed=4, fill=7.0)
    y.fill_(7.0), e.fill_(7.0)
    _lib.call("sn_linear_fwd_f32", _p(x), _ld(x), _p(W), _ld(W), _p(b), _p(res), _ld(res), _p(y), _ld(y), _p(e), _ld(e),
              rows, K, J, None, _stream())
    want = x.double() @ W.double().t() + b.double() + res.double()
    assert torch.isfinite(y).all() and float((y.double() - want).abs().max()) < 1e-4
    assert float((e.double() - torch.where(want > 0, want, torch.expm1(want))).abs().max()) < 1e-4
    assert _outside_untouched(ya, rows, J, 7.0) and _outside_untouched(ea, rows, J, 7.0)
    # input gradient with the BatchNorm tail, first half through the activation (both outputs strided)
    C = K
    Wd = dev((rng.standard_normal((J, C)) / np.sqrt(J)).astype(np.float32))
    cen, B, Cc = [dev(rng.standard_normal(C).astype(np.float32)) for _ in range(3)]
    _, dy = _arena(rows, J, seed=5)
    _, xs = _arena(rows, C, seed=6)
    _, gadd = _arena(rows, C // 2, seed=7)
    ha, dx_hi = _arena(rows, C // 2, seed=8, fill=7.0)
    ga, gact = _arena(rows, C // 2, seed=9, fill=7.0)
    dx_hi.fill_(7.0), gact.fill_(7.0)
    _lib.call("sn_linear_dgrad_elu_f32", _p(dy), _ld(dy), _p(Wd), _ld(Wd), _p(xs), _ld(xs), _p(cen), _p(B), _p(Cc),
              _p(dx_hi), _ld(dx_hi), _p(gact), _ld(gact), _p(gadd), _ld(gadd), rows, J, C, _stream())
    full = dy.double() @ Wd.double() + (xs.double() - cen.double()) * B.double() + Cc.double()
    o = xs[:, :C // 2].double()
    want_g = full[:, :C // 2] * torch.where(o > 0, torch.ones_like(o), o + 1) + gadd.double()
    assert float((dx_hi.double() - full[:, C // 2:]).abs().max()) < 1e-4 and float((gact.double() - want_g).abs().max()) < 1e-4
    assert _outside_untouched(ha, rows, C // 2, 7.0) and _outside_untouched(ga, rows, C // 2, 7.0)
    # plain input gradient into a strided view
    da, dxv = _arena(rows, C, seed=10, fill=7.0)
    dxv.fill_(7.0)
    _lib.call("sn_linear_dgrad_f32", _p(dy), _ld(dy), _p(Wd), _ld(Wd), _p(xs), _ld(xs), _p(cen), _p(B), _p(Cc), _p(dxv),
              _ld(dxv), rows, J, C, _stream())
    assert float((dxv.double() - full).abs().max()) < 1e-4 and _outside_untouched(da, rows, C, 7.0)
    # every column through the activation, no per-mesh vector (the backward of conv(F.elu(v)))
    _, gadd2 = _arena(rows, C, seed=11)
    aa, gall = _arena(rows, C, seed=12, fill=7.0)
    gall.fill_(7.0)
    _lib.call("sn_linear_dgrad_eluseg_f32", _p(dy), _ld(dy), _p(Wd), _ld(Wd), _p(xs), _ld(xs), _p(cen), _p(B), _p(Cc), None, 0,
              None, _p(gall), _ld(gall), _p(gadd2), _ld(gadd2), rows, J, C, _stream())
    oa = xs.double()
    want_all = full * torch.where(oa > 0, torch.ones_like(oa), oa + 1) + gadd2.double()
    assert float((gall.double() - want_all).abs().max()) < 1e-4 and _outside_untouched(aa, rows, C, 7.0)
    # weight gradient: rows past the views are NaN — one of them read would poison every entry
    G, s = kernels.wgrad(dy, xs, cen, want_colsum=True)
    wantG = dy.double().t() @ (xs.double() - cen.double())
    assert torch.isfinite(G).all() and float((G.double() - wantG).abs().max()) < 1e-3 * max(1.0, float(wantG.abs().max()))
    assert float((s - dy.double().sum(0)).abs().max()) < 1e-3


@pytest.mark.parametrize("lengths", [[32, 700, 45, 33, 2000], [5041, 5041], [40] * 37])
def test_ragged_mesh_entry_points(lengths):
    """The per-mesh-vector GEMM epilogues and the slab weight gradient on RAGGED meshes (packed batches) against fp64, and
    the statuses of their argument checks."""
    from surfacenetworks_amd import _lib
    from surfacenetworks_amd._lib import SnError
    from surfacenetworks_amd.kernels import _p, _ld, _stream
    from surfacenetworks_amd.operators import PackedSegments

    seg = PackedSegments(lengths, DEV)
    rows, C, J = seg.rows, 128, 128
    rng = np.random.default_rng(len(lengths) + rows)
    mesh = torch.from_numpy(np.repeat(np.arange(seg.nseg), seg.lengths)).to(DEV)
    x, res, dy, gadd = [dev(rng.standard_normal((rows, C)).astype(np.float32)) for _ in range(4)]
    W = dev((rng.standard_normal((J, C)) / 11).astype(np.float32))
    segb = dev(rng.standard_normal((seg.nseg, J)).astype(np.float32))
    # forward with a per-mesh bias (+ residual, + elu copy with statistics)
    cat = torch.zeros(rows, 2 * J, device=DEV)
    part = kernels.new_elu_stats_part(rows, DEV)
    y = kernels.linear_fwd_segbias_ragged(x, W, segb, seg, res, cat[:, :J], True, part)
    want = x.double() @ W.double().t() + segb.double()[mesh] + res.double()
    assert float((y.double() - want).abs().max()) < 2e-4
    we = torch.where(want > 0, want, torch.expm1(want))
    assert float((cat[:, :J].double() - we).abs().max()) < 2e-4 and float(cat[:, J:].abs().sum()) == 0
    st = kernels.colstats_from_part(part, rows)
    assert torch.allclose(st[0], we.sum(0), rtol=1e-5, atol=1e-3) and torch.allclose(st[1], (we * we).sum(0), rtol=1e-5, atol=1e-3)
    # input gradient through the activation with a per-mesh vector
    cen, B, Cc = [dev(rng.standard_normal(C).astype(np.float32)) for _ in range(3)]
    segv = dev(rng.standard_normal((seg.nseg, C)).astype(np.float32))
    g = kernels.linear_dgrad_eluseg_ragged(dy, W, x, cen, B, Cc, segv, seg, gadd)
    pre = dy.double() @ W.double() + (x.double() - cen.double()) * B.double() + Cc.double() + segv.double()[mesh]
    wantg = pre * torch.where(x.double() > 0, torch.ones_like(pre), x.double() + 1) + gadd.double()
    assert float((g.double() - wantg).abs().max()) < 5e-4
    # weight gradient over the segments' slabs: G, colsum(dy), per-mesh colsum(dy)
    G, sdy, sg = kernels.wgrad_slabs(dy, x, cen, seg)
    G0, sdy0 = kernels.wgrad(dy, x, cen, want_colsum=True)
    refG = dy.double().t() @ (x.double() - cen.double())
    assert float((G.double() - refG).abs().max()) <= 2.0 * float((G0.double() - refG).abs().max()) + 1e-4
    assert torch.allclose(sdy, dy.double().sum(0), rtol=1e-9, atol=1e-3)
    wsg = torch.zeros(seg.nseg, J, dtype=torch.float64, device=DEV).index_add_(0, mesh, dy.double())
    assert float((sg.double() - wsg).abs().max()) < 1e-3
    # argument checks
    lib = _lib.load()
    assert lib.sn_linear_fwd_segbias_ragged_f32(_p(x), _ld(x), _p(W), _ld(W), _p(segb), None, seg.nseg, None, 0, _p(y), J, None, 0,
                                                rows, C, J, None, _stream()) == -1          # SN_E_NULL: no offsets
    assert lib.sn_linear_fwd_segbias_ragged_f32(_p(x), _ld(x), _p(W), _ld(W), _p(segb), _p(seg.off_dev), 0, None, 0, _p(y), J, None,
                                                0, rows, C, J, None, _stream()) == -2        # SN_E_SHAPE: no meshes
    assert lib.sn_linear_dgrad_eluseg_ragged_f32(_p(dy), _ld(dy), _p(W), _ld(W), _p(x), _ld(x), _p(cen), _p(B), _p(Cc), None,
                                                 _p(seg.off_dev), seg.nseg, _p(g), C, None, 0, rows, J, C, _stream()) == -2
    ws = torch.empty(16, dtype=torch.uint8, device=DEV)
    assert lib.sn_wgrad_slabs_f32(_p(dy), _ld(dy), _p(x), _ld(x), _p(cen), rows, _p(seg.slab_off), seg.nslab, _p(seg.seg_slab_ptr),
                                  seg.nseg, J, C, _p(G), _p(sdy), _p(sg), _p(ws), 16, _stream()) == -6   # SN_E_WORKSPACE
    assert lib.sn_wgrad_slabs_f32(_p(dy), _ld(dy), _p(x), _ld(x), _p(cen), rows, None, seg.nslab, _p(seg.seg_slab_ptr),
                                  seg.nseg, J, C, _p(G), _p(sdy), _p(sg), _p(ws), 16, _stream()) == -1   # SN_E_NULL


# ---- target of the dense-correspondence loss (dense_correspondence/main.py:236-237) ---------------------------------------
@pytest.mark.parametrize("NA,NB,kind", [(1, 1, "real"), (5, 9, "real"), (333, 257, "real"), (1030, 700, "ties"), (64, 513, "nan"),
                                        (6890, 6890, "real"), (40, 17000, "real"), (8, 15349, "real"), (8, 15350, "real")])
def test_pair_argmin_is_the_reference_min_over_the_two_gathered_matrices(NA, NB, kind):
    """Bit-exact index parity with numpy's argmin of the fp32 sum the reference forms, on views with a leading dimension
    larger than the row, tie-heavy integer matrices (lowest index wins), rows holding a NaN (which wins, as in torch.min), and
    row widths on both sides of the LDS-staged form's limit (15 360 columns)."""
    rng = np.random.default_rng(NA * 7 + NB)
    ldA, ldB = NA + NB + 3, NB + 5                      # GA has at least max(pa)+1 columns, GB exactly >= NB
    if kind == "ties":
        GA = rng.integers(0, 3, (NA, ldA)).astype(np.float32)
        GB = rng.integers(0, 3, (NA + 2, ldB)).astype(np.float32)
    else:
        GA = rng.random((NA, ldA), dtype=np.float32)
        GB = rng.random((NA + 2, ldB), dtype=np.float32)
    if kind == "nan":
        GB[rng.integers(0, NA + 2, 20), rng.integers(0, NB, 20)] = np.nan
    pa = rng.integers(0, ldA, NB)
    pb = rng.permutation(NA + 2)[:NA]
    total = GA[:, pa] + GB[pb][:, :NB]                   # fp32 + fp32, as the reference adds them
    want = np.argmin(total, axis=1)                      # numpy: first minimum; NaN wins
    got = kernels.pair_argmin(dev(GA), dev(pa), dev(GB)[:, :NB], dev(pb))
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), want)
    if kind != "nan":                                    # and torch.min itself on the device, where it has no NaN to order
        tm = torch.min(dev(GA)[:, dev(pa)] + dev(GB)[dev(pb)][:, :NB], dim=1)[0].cpu().numpy()
        assert np.array_equal(total[np.arange(NA), want], tm)


def test_pair_argmin_rejects_what_it_cannot_index():
    G = torch.zeros(4, 4, device=DEV)
    i = torch.arange(4, device=DEV)
    with pytest.raises(TypeError):
        kernels.pair_argmin(G.double(), i, G, i)
    with pytest.raises(ValueError):
        kernels.pair_argmin(torch.zeros(4, 8, device=DEV)[:, ::2], i, G, i)
    with pytest.raises(ValueError):
        kernels.pair_argmin(G, torch.arange(5, device=DEV), G, i)
    assert kernels.pair_argmin(G, i, G, i[:0]).numel() == 0


@pytest.mark.parametrize("rows,J", [(2, 64), (1000, 64), (77312, 64), (5000, 32), (999, 64)])
def test_weight_gradient_of_a_64_wide_layer_through_the_paired_rows(rows, J):
    """functional._centered_wgrad at C = 64 (classifier head of the Mesh-MNIST models): two consecutive rows side by side
    feed the 128-wide split-K kernel; odd row counts take the library path.  Both against fp64."""
    rng = np.random.default_rng(rows + J)
    x = (rng.standard_normal((rows, 64)) + 3.0).astype(np.float32)
    dy = rng.standard_normal((rows, J)).astype(np.float32)
    mean = x.mean(0).astype(np.float32)
    G, sdy = snF._centered_wgrad(dev(dy), dev(x), dev(mean))
    want = dy.astype(np.float64).T @ (x.astype(np.float64) - mean.astype(np.float64))
    assert G.shape == (J, 64) and rel_err(G.cpu().numpy(), want) < 2e-6
    s = sdy.cpu().numpy().reshape(-1, J)[0]
    # fp32 partial sums inside a slab, fp64 across slabs
    assert np.allclose(s, dy.astype(np.float64).sum(0), rtol=0, atol=1e-6 * np.abs(dy).sum(0).max())


@pytest.mark.parametrize("N,NA,NB", [(7, 7, 7), (80, 63, 70), (1024, 1000, 1021), (7000, 6890, 6890), (9001, 40, 9001)])
def test_pair_cross_entropy_matches_torch(N, NA, NB):
    """sn_pair_ce_fwd/bwd_f32 against F.cross_entropy on the NA x NB corner of a padded score matrix: value, the gradient
    inside the corner and exact zeros in the padding; rows longer than the register-resident form and unaligned views too."""
    import torch.nn.functional as F

    from surfacenetworks_amd import dense_correspondence as dc

    g = torch.Generator().manual_seed(N)
    rows = min(N, NA + 3)
    for off in (0, 1):                                   # off = 1: a view that is not 16-byte aligned
        base = (3.0 * torch.randn(rows, N + 4, generator=g)).to(DEV)
        S = base[:, off:off + N].detach().requires_grad_(True)
        tgt = torch.randint(0, NB, (NA,), generator=g).to(DEV)
        want = F.cross_entropy(S[:NA, :NB], tgt)
        (gw,) = torch.autograd.grad(want * 1.7, S)
        S2 = S.detach().clone().requires_grad_(True) if off == 0 else base[:, off:off + N].detach().requires_grad_(True)
        got = dc.pair_cross_entropy(S2, tgt, NA, NB)
        (gg,) = torch.autograd.grad(got * 1.7, S2)
        assert abs(got.item() - want.item()) <= 2e-6 * abs(want.item())
        assert rel_err(gg.cpu().numpy(), gw.cpu().numpy()) < 5e-6
        assert gg[NA:].abs().max().item() == 0 if rows > NA else True
        assert gg[:, NB:].abs().max().item() == 0 if N > NB else True


@pytest.mark.parametrize("rows,NA,NB,K", [(7, 7, 7, 120), (80, 63, 70, 120), (33, 33, 1, 5), (300, 257, 290, 128), (1024, 1000, 1021, 64),
                                          (7000, 6890, 6890, 120)])
def test_fused_pair_cross_entropy_matches_the_score_matrix_path(rows, NA, NB, K):
    """sn_pair_fused_fwd/bwd_f32 (scores formed on the matrix pipe from three-piece bf16 splits, never written) against the
    cross entropy of the materialised bmm(FA, FBᵀ) (models.py:203, main.py:238-239) in fp64: value, both feature gradients,
    exact zeros for the padding rows, targets on the last column / row of the corner, run-to-run identical results."""
    import torch.nn.functional as F

    from surfacenetworks_amd import dense_correspondence as dc

    g = torch.Generator().manual_seed(rows + K)
    rowsB = rows + 5
    FA = (torch.randn(1, rows, K, generator=g) * 0.7).to(DEV).requires_grad_(True)
    FB = (torch.randn(1, rowsB, K, generator=g) * 0.7).to(DEV).requires_grad_(True)
    tgt = torch.randint(0, NB, (NA,), generator=g)
    tgt[0], tgt[-1] = NB - 1, 0
    tgt = tgt.to(DEV)
    A64, B64 = FA.detach().double().requires_grad_(True), FB.detach().double().requires_grad_(True)
    want = F.cross_entropy(torch.bmm(A64, B64.transpose(1, 2))[0, :NA, :NB], tgt)
    wa, wb = torch.autograd.grad(want * 1.7, (A64, B64))
    assert dc.fused_pair_supported(FA, FB)
    got = dc.fused_pair_cross_entropy(FA, FB, tgt, NA, NB)
    ga, gb = torch.autograd.grad(got * 1.7, (FA, FB))
    assert abs(got.item() - want.item()) <= 2e-6 * abs(want.item())
    assert ga.shape == FA.shape and gb.shape == FB.shape
    assert rel_err(ga.cpu().numpy(), wa.cpu().numpy()) < 5e-6 and rel_err(gb.cpu().numpy(), wb.cpu().numpy()) < 5e-6
    assert rows == NA or ga[0, NA:].abs().max().item() == 0
    assert gb[0, NB:].abs().max().item() == 0
    got2 = dc.fused_pair_cross_entropy(FA, FB, tgt, NA, NB)
    ga2, gb2 = torch.autograd.grad(got2 * 1.7, (FA, FB))
    assert torch.equal(got, got2) and torch.equal(ga, ga2) and torch.equal(gb, gb2)
    # and against the product's other path (library GEMM + sn_pair_ce_*), at fp32 agreement
    out = torch.bmm(FA, FB.transpose(1, 2))
    l2 = dc.pair_cross_entropy(out, tgt, NA, NB)
    assert abs(l2.item() - got.item()) <= 5e-6 * abs(got.item())


def test_fused_pair_argument_checks():
    from surfacenetworks_amd import _lib

    lib = _lib.load()
    f = torch.zeros(64, 120, device=DEV)
    t = torch.zeros(64, dtype=torch.int64, device=DEV)
    o = torch.zeros(64, device=DEV)
    need = lib.sn_pair_fused_workspace_bytes(64, 64)
    assert need == 256 + 2 * (64 * 128 * 2 * 2) * 2 + 8 * 64 * 4 * 4 + 4 * 2 * 64 * 128 * 4      # header, R + T of both sides, partials
    ws = torch.zeros(need, dtype=torch.uint8, device=DEV)
    p = lambda x: x.data_ptr()
    SN_E_NULL, SN_E_SHAPE, SN_E_UNSUPPORTED, SN_E_WORKSPACE = -1, -2, -7, -6
    fw = lib.sn_pair_fused_fwd_f32
    assert fw(p(f), 120, p(f), 120, p(t), 64, 64, 64, 64, 120, p(o), p(o), p(ws), need, None) == 0
    assert fw(p(f), 120, p(f), 120, p(t), 65, 64, 64, 64, 120, p(o), p(o), p(ws), need, None) == SN_E_SHAPE       # NA > rowsA
    assert fw(p(f), 120, p(f), 120, p(t), 64, 64, 64, 64, 129, p(o), p(o), p(ws), need, None) in (SN_E_SHAPE, SN_E_UNSUPPORTED)
    assert fw(p(f), 200, p(f), 200, p(t), 8, 8, 8, 8, 129, p(o), p(o), p(ws), need, None) == SN_E_UNSUPPORTED
    assert fw(p(f), 120, p(f), 120, p(t), 64, 64, 64, 64, 120, p(o), p(o), p(ws), need - 1, None) == SN_E_WORKSPACE
    assert fw(None, 120, p(f), 120, p(t), 64, 64, 64, 64, 120, p(o), p(o), p(ws), need, None) == SN_E_NULL
    bw = lib.sn_pair_fused_bwd_f32
    assert bw(p(t), p(o), p(o), 64, 64, 64, 64, 120, p(f), 120, None, 120, p(ws), need, None) == SN_E_NULL
    assert bw(p(t), p(o), p(o), 64, 64, 64, 64, 120, p(f), 119, p(f), 120, p(ws), need, None) == SN_E_SHAPE


@pytest.mark.parametrize("B", [1, 2])
def test_pair_cross_entropy_takes_the_siamese_output(B):
    """The (B, N, N) output of SiameseModel goes in whole: sample 0 is scored (main.py:238), the gradient comes back in the
    output's shape — as a view of the kernel's result for B = 1, zero for the other samples otherwise."""
    import torch.nn.functional as F

    from surfacenetworks_amd import dense_correspondence as dc

    g = torch.Generator().manual_seed(5 + B)
    N, NA, NB = 96, 90, 93
    out = torch.randn(B, N, N, generator=g).to(DEV).requires_grad_(True)
    tgt = torch.randint(0, NB, (NA,), generator=g).to(DEV)
    (gw,) = torch.autograd.grad(F.cross_entropy(out[0, :NA, :NB], tgt), out)
    (gg,) = torch.autograd.grad(dc.pair_cross_entropy(out, tgt, NA, NB), out)
    assert gg.shape == out.shape and rel_err(gg.cpu().numpy(), gw.cpu().numpy()) < 5e-6
    if B > 1:
        assert gg[1:].abs().max().item() == 0


@pytest.mark.parametrize("rows", [1, 33, 7000, 300000])
def test_both_halves_of_the_statistics_in_one_launch(rows):
    """sn_colstats_merge2_f64 (colstats_halves with partials of both producers) == two sn_colstats_merge_f64 calls, bit for
    bit: the statistics of [e | P·e] from the GEMM's and the SpMM's partials."""
    from surfacenetworks_amd import kernels

    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, 256, generator=g) * 3 + 1).to(DEV)
    nlo = kernels.linear_fwd_stats_blocks(rows)
    lo = torch.randn(max(nlo, 1), 2, 128, generator=g, dtype=torch.float64).to(DEV)       # (any numbers: the reduction is what is tested)
    hi = torch.randn(37, 2, 128, generator=g, dtype=torch.float64).to(DEV)
    got = kernels.colstats_halves(x, lo, hi)
    want = torch.zeros((2, 256), dtype=torch.float64, device=DEV)
    kernels.colstats_merge_into(lo[:nlo], want, 0)
    kernels.colstats_merge_into(hi, want, 128)
    assert torch.equal(got, want)


@pytest.mark.parametrize("rows", [33, 77000])
def test_statistics_of_a_64_channel_concat_buffer_from_both_producers(rows):
    """The 64-channel stages (Mesh-MNIST): the forward GEMM of a 64-output layer leaves its ELU statistics in the kernel's
    128-column partial layout, the quaternion SpMM in its own 64-column one; colstats_halves reads both in one launch and
    equals the statistics pass over the buffer.  The GEMM partials come from the real kernel here."""
    from surfacenetworks_amd import kernels

    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 128, generator=g).to(DEV)
    W = (torch.randn(64, 128, generator=g) * 0.2).to(DEV)
    b = torch.randn(64, generator=g).to(DEV)
    cat = torch.full((rows, 128), float("nan"), device=DEV)
    part = kernels.new_elu_stats_part(rows, DEV)
    kernels.linear_fwd(x, W, b, None, cat[:, :64], False, part)            # elu(y) -> first half, statistics -> part
    hi_vals = (torch.randn(rows, 64, generator=g) * 2 - 0.3).to(DEV)
    cat[:, 64:] = hi_vals
    # partials of the second half in the SpMM's layout: (blocks, 2, 64)
    nb = 5
    bounds = np.linspace(0, rows, nb + 1).astype(int)
    hi = torch.stack([torch.stack([hi_vals[a:z].double().sum(0), (hi_vals[a:z].double() ** 2).sum(0)]) for a, z in zip(bounds[:-1], bounds[1:])])
    got = kernels.colstats_halves(cat, part, hi.contiguous())
    want = kernels.colstats(cat)
    assert got.shape == (2, 128)
    assert torch.allclose(got, want, rtol=1e-9, atol=1e-9 * float(want.abs().max()))



# ---- per-mesh sums of a global-average operand from the producing GEMM's per-tile column sums -----------------------------
@pytest.mark.parametrize("per,nseg,maskkind", [(5041, 8, "none"), (5041, 8, "prefix"), (1000, 5, "random"), (40, 6, "prefix"),
                                               (4096, 4, "none"), (33, 3, "none")])
@pytest.mark.parametrize("segbias", [False, True])
def test_average_stage_statistics_from_tile_sums_match_the_pass_over_the_operand(per, nseg, maskkind, segbias):
    """sn_linear_fwd_tiles_f32 / sn_linear_fwd_segbias_tiles_f32 + sn_avg_stats_from_tiles_f32 against sn_avg_stats_f32 on the
    activated output: same per-mesh means and BatchNorm sums (fp32 tile sums, fp64 across tiles), meshes that do not start on
    tile boundaries, row masks (prefix masks as the samplers build them, arbitrary ones)."""
    if not kernels.tile_sums_supported():
        pytest.skip("split kernels only")
    rows = per * nseg
    torch.manual_seed(per + nseg)
    x = torch.randn(rows, 128, device=DEV)
    W = torch.randn(128, 128, device=DEV) / np.sqrt(128)
    b = torch.randn(128, device=DEV)
    r = torch.randn(rows, 128, device=DEV)
    cat = torch.full((rows, 256), float("nan"), device=DEV)
    part = kernels.new_elu_stats_part(rows, DEV)
    tiles = kernels.new_tile_sums(rows, DEV)
    tiles.fill_(float("nan"))                                    # every tile must be written
    if segbias:
        segb = torch.randn(nseg, 128, device=DEV)
        kernels.linear_fwd_segbias(x, W, segb, per, r, cat[:, :128], False, part, tiles)
    else:
        kernels.linear_fwd(x, W, b, r, cat[:, :128], False, part, tiles)
    assert torch.isfinite(tiles).all()
    e = cat[:, :128]
    if maskkind == "none":
        mask = None
        cnt = torch.full((nseg,), float(per), device=DEV)
    elif maskkind == "prefix":
        valid = torch.randint(max(1, per // 2), per + 1, (nseg,), device=DEV)
        mask = (torch.arange(per, device=DEV)[None, :] < valid[:, None]).float().reshape(-1).contiguous()
        cnt = valid.float()
    else:
        mask = (torch.rand(rows, device=DEV) > 0.2).float()
        cnt = mask.reshape(nseg, per).sum(1)
    inv = (1.0 / cnt).reshape(nseg, 1)
    m_ref, st_ref = kernels.avg_stats(e, mask, inv, per, nseg)
    m, st = kernels.avg_stats_from_tiles(tiles, part, e, mask, inv, per, nseg)
    assert torch.allclose(m, m_ref, rtol=2e-6, atol=2e-6 * float(m_ref.abs().max()))
    assert np.allclose(st.cpu().numpy(), st_ref.cpu().numpy(), rtol=1e-6, atol=1e-6 * float(st_ref.abs().max()))
    # the per-tile sums themselves, against float64
    want = e.double().reshape(-1, 128)
    ntile = (rows + 31) // 32
    pad = torch.zeros(ntile * 32 - rows, 128, dtype=torch.float64, device=DEV)
    want = torch.cat([want, pad]).reshape(ntile, 32, 128).sum(1)
    assert torch.allclose(tiles.double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))


def test_arap_model_with_and_without_tile_sums(monkeypatch):
    """The whole ARAP Dirac model with the tile-sum hand-off (default) and with the statistics pass over every global-average
    operand (SN_TILE_SUMS=0 semantics): same loss and gradients up to the summation order of the per-mesh means."""
    from helpers import deterministic_init
    from surfacenetworks_amd import arap

    ds = arap.ClothSequences([(12, 11), (9, 13), (10, 10)], frames=45, op_frames=2, seed=3, device=DEV, model="dir")
    seq, off = np.array([0, 1, 2, 1]), np.array([0, 1, 0, 0])
    res = []
    from surfacenetworks_amd import blocks as snB

    monkeypatch.setattr(snB, "_TILE_SUMS_MIN_ROWS", 0)            # (the hand-off is used from 32 768 rows on: force it here)
    for on in (True, False):
        monkeypatch.setattr(kernels, "tile_sums_supported", (lambda: True) if on else (lambda: False))
        model = deterministic_init(arap.DirModel(), 4).to(DEV).train()
        b = ds.sample_batch(4, None, seq_ids=seq, offsets=off)
        loss, _ = arap.forward_loss(model, b, 4)
        loss.backward()
        res.append((loss.item(), torch.cat([p.grad.reshape(-1) for p in model.parameters()])))
    assert abs(res[0][0] - res[1][0]) <= 2e-6 * abs(res[1][0])
    assert float((res[0][1] - res[1][1]).norm() / res[1][1].norm()) < 2e-5


# ---- operands of the BatchNorm-backward helper tests ---------------------------------------------------------------------------
def _bn_operands(rng, rows, J, C, Ct):
    dy = dev(rng.standard_normal((rows, J)).astype(np.float32))
    x = dev((rng.standard_normal((rows, C)) * 1.5 + rng.standard_normal(C)).astype(np.float32))
    W = dev((rng.standard_normal((J, Ct)) / 9).astype(np.float32))
    gamma, beta = [dev(rng.standard_normal(Ct).astype(np.float32)) for _ in range(2)]
    return dy, x, W, gamma, beta


@pytest.mark.parametrize("rows", [1, 2, 3, 31, 33, 1000])
def test_two_piece_weight_gradient_ignores_the_rows_past_the_end(rows):
    """The two-piece fp16 weight gradient pads a slab's last block with zero rows; centred they are -mean, which the bound on
    |x - mean| does not cover — a column far from zero with a tiny spread (an ELU unit sitting at -1) scaled past fp16's range
    and met the padded dy rows' exact zeros as inf x 0.  Must agree with the three-piece form for such columns too."""
    rng = np.random.default_rng(rows)
    J = C = 128
    dy = dev(rng.standard_normal((rows, J)).astype(np.float32))
    xn = rng.standard_normal((rows, C)).astype(np.float32)
    xn[:, ::3] = -1.0 + 1e-4 * xn[:, ::3]                 # saturated units
    xn[:, 1::3] = 1000.0 + 0.5 * xn[:, 1::3]              # large offset, small spread
    x = dev(xn)
    mean = x.double().mean(0)
    var = ((x.double() - mean) ** 2).mean(0)
    invstd = (1.0 / torch.sqrt(var + 1e-5)).float()
    meanf = mean.float()
    bounds = (dy.abs().max().reshape(1), invstd, rows)
    G, sdy = kernels.wgrad(dy, x, meanf, want_colsum=True, bounds=bounds)
    G0, sdy0 = kernels.wgrad(dy, x, meanf, want_colsum=True)
    assert bool(torch.isfinite(G).all())
    ref = dy.double().t() @ (x.double() - meanf.double())
    assert float((G.double() - ref).abs().max()) <= 2.0 * float((G0.double() - ref).abs().max()) + 1e-5 * float(ref.abs().max()) + 1e-6
    assert torch.equal(sdy, sdy0)


@pytest.mark.parametrize("nseg,per", [(1, 40), (3, 150), (7, 33), (64, 300), (65, 40), (200, 17)])
def test_global_average_stage_merged_launches(nseg, per):
    """sn_bn_fold_seg_f32 (fold + per-mesh bias) and sn_avg_bn_bwd_f32 (broadcast half of G + BatchNorm coefficients + per-mesh
    vector) against the launches they merge, bit for bit — equal meshes and the ragged form."""
    from surfacenetworks_amd.operators import PackedSegments

    J = C = 128
    rows = nseg * per
    rng = np.random.default_rng(nseg * 100 + per)
    dy, e, W, gamma, beta = _bn_operands(rng, rows, J, C, 2 * C)
    b = dev(rng.standard_normal(J).astype(np.float32))
    m = e.view(nseg, per, C).mean(1).contiguous()
    inv_count = torch.full((nseg,), 1.0 / per, device=DEV)
    st = torch.cat([kernels.colstats(e), torch.stack([(m.double() * per).sum(0), (m.double() ** 2 * per).sum(0)])], 1).contiguous()
    rm, rv = torch.zeros(2 * C, device=DEV), torch.ones(2 * C, device=DEV)
    rm2, rv2 = rm.clone(), rv.clone()
    nbt, nbt2 = torch.zeros(1, dtype=torch.int64, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    want = kernels.bn_fold(st, rows, gamma, beta, W, b, 1e-5, 0.1, True, rm, rv, nbt)
    want_segb = kernels.seg_affine(m, want[4][:, C:], want[5])
    got = kernels.bn_fold_seg(st, rows, gamma, beta, W, b, 1e-5, 0.1, rm2, rv2, m, nbt2)
    for a, w_, name in zip(got[:6], want, ("mean", "invstd", "s", "t", "Wf", "bf")):
        assert torch.equal(a, w_), name
    assert torch.equal(got[6], want_segb) and torch.equal(rm, rm2) and torch.equal(rv, rv2) and int(nbt) == int(nbt2) == 1
    mean, invstd, s, t, Wf, bf = want
    # backward
    G1, sdy, Sg = kernels.wgrad_seg(dy, e, mean[:C], per)
    w6 = kernels.bn_bwd_coeffs(kernels.avg_bwd_gc(G1, Sg, m, mean[C:]), sdy, W, s, invstd, beta, rows, True)
    wv = kernels.avg_bwd_segvec(Sg, Wf[:, C:], m, mean[C:], w6[4][C:], w6[5][C:], inv_count, per)
    g7 = kernels.avg_bn_bwd(G1, sdy, Sg, m, mean[C:], W, s, invstd, beta, rows, True, Wf[:, C:], inv_count, rows_per_seg=per)
    for a, w_, name in zip(g7, (*w6, wv), ("dW", "db", "dgamma", "dbeta", "Bc", "Cc", "segvec")):
        assert torch.equal(a, w_), (name, float((a - w_).abs().max()))
    assert kernels.avg_bn_bwd(G1, sdy, Sg, m, mean[C:], W, s, invstd, beta, rows, False, Wf[:, C:], inv_count, rows_per_seg=per)[1] is None
    # ragged form: the row counts come from the offsets
    lengths = [per + (3 * i) % 7 for i in range(nseg)]
    seg = PackedSegments(lengths, DEV)
    wvr = kernels.avg_bwd_segvec_ragged(Sg, Wf[:, C:], m, mean[C:], w6[4][C:], w6[5][C:], seg)
    g7r = kernels.avg_bn_bwd(G1, sdy, Sg, m, mean[C:], W, s, invstd, beta, rows, True, Wf[:, C:], seg.inv_count, segoff=seg.off_dev)
    for a, w_, name in zip(g7r, (*w6, wvr), ("dW", "db", "dgamma", "dbeta", "Bc", "Cc", "segvec ragged")):
        assert torch.equal(a, w_), (name, float((a - w_).abs().max()))


@pytest.mark.parametrize("lengths", [[32, 700, 45, 33, 2000], [5041] * 3, [40] * 70])
def test_ragged_global_average_statistics_in_one_launch(lengths):
    """sn_avg_prep_ragged_f32 against the composition it replaces (merge of the partials, then len_i m_i / len_i m_i^2 sums
    in float64 by the framework)."""
    from surfacenetworks_amd.operators import PackedSegments

    seg = PackedSegments(lengths, DEV)
    rows, C = seg.rows, 128
    rng = np.random.default_rng(len(lengths))
    x = dev(rng.standard_normal((rows, C)).astype(np.float32))
    W = dev((rng.standard_normal((C, C)) / 11).astype(np.float32))
    cat = torch.zeros(rows, 2 * C, device=DEV)
    part = kernels.new_elu_stats_part(rows, DEV)
    kernels.linear_fwd(x, W, dev(np.zeros(C, np.float32)), None, cat[:, :C], False, part)
    e = cat[:, :C]
    m = seg.mean(e).contiguous()
    md = m.double()
    s1 = kernels.colstats_from_part(part, rows)
    want = torch.cat([s1, torch.stack([(md * seg.len_f64[:, None]).sum(0), (md * md * seg.len_f64[:, None]).sum(0)])], 1)
    got = kernels.avg_stats_ragged(m, seg, part, kernels.linear_fwd_stats_blocks(rows))
    assert torch.allclose(got, want, rtol=1e-13, atol=1e-9)
    got1 = kernels.avg_stats_ragged(m, seg, kernels.colstats(e).reshape(1, 2, C), 1)
    assert torch.allclose(got1[:, C:], want[:, C:], rtol=1e-13, atol=1e-9) and torch.allclose(got1[:, :C], want[:, :C], rtol=1e-11, atol=1e-7)


@pytest.mark.parametrize("lengths", [[32, 700, 45, 33, 2000], [5041] * 3, [40] * 70, [33, 31 + 32, 64, 95]])
def test_ragged_global_average_statistics_from_tile_sums(lengths):
    """sn_avg_stats_from_tiles_ragged_f32 (per-mesh means and BatchNorm statistics from the producing GEMM's tile sums and
    partials, mesh boundaries anywhere inside a tile) against the pass over the operand."""
    from surfacenetworks_amd.operators import PackedSegments

    seg = PackedSegments(lengths, DEV)
    rows, C = seg.rows, 128
    rng = np.random.default_rng(len(lengths) + rows)
    x = dev(rng.standard_normal((rows, C)).astype(np.float32))
    W = dev((rng.standard_normal((C, C)) / 11).astype(np.float32))
    segb = dev(rng.standard_normal((seg.nseg, C)).astype(np.float32))
    for ragged_producer in (False, True):
        cat = torch.zeros(rows, 2 * C, device=DEV)
        part = kernels.new_elu_stats_part(rows, DEV)
        tiles = kernels.new_tile_sums(rows, DEV)
        if ragged_producer:
            kernels.linear_fwd_segbias_ragged(x, W, segb, seg, None, cat[:, :C], False, part, tiles)
        else:
            kernels.linear_fwd(x, W, dev(np.zeros(C, np.float32)), None, cat[:, :C], False, part, tiles)
        e = cat[:, :C]
        m_ref = seg.mean(e)
        want = kernels.avg_stats_ragged(m_ref.contiguous(), seg, part, kernels.linear_fwd_stats_blocks(rows))
        m, stats = kernels.avg_stats_from_tiles_ragged(tiles, part, e, seg)
        assert float((m - m_ref).abs().max()) <= 2e-6 * float(m_ref.abs().max()) + 1e-7
        # the tiles are fp32 sums of 32 rows: a mean is good to ~1e-7 of the mesh's mean |e|, whatever its own size
        mesh = torch.from_numpy(np.repeat(np.arange(seg.nseg), seg.lengths)).to(DEV)
        m64 = torch.zeros(seg.nseg, C, dtype=torch.float64, device=DEV).index_add_(0, mesh, e.double()) * seg.inv_count.double()[:, None]
        a64 = torch.zeros(seg.nseg, C, dtype=torch.float64, device=DEV).index_add_(0, mesh, e.double().abs()) * seg.inv_count.double()[:, None]
        assert bool(((m.double() - m64).abs() <= 4e-7 * a64 + 1e-12).all())
        assert torch.allclose(stats[:, :C], want[:, :C], rtol=1e-13, atol=1e-9)
        assert torch.allclose(stats[:, C:], want[:, C:], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("length", [6, 120, 7])
def test_gather_segments_into_a_packed_batch(length):
    """sn_gather_segments_ragged_f32 against the padded gather followed by the boolean-mask selection it replaces."""
    from surfacenetworks_amd.operators import PackedSegments

    rng = np.random.default_rng(length)
    n, vmax, f3 = 5, 300, 135
    src = dev(rng.standard_normal((n, vmax, f3)).astype(np.float32))
    lengths = [300, 32, 77, 1, 250]
    items = np.array([3, 0, 4, 2, 1])
    seg = PackedSegments([lengths[i] for i in items], DEV)
    base = dev((items * vmax * f3 + 3 * np.array([0, 2, 1, 4, 3])).astype(np.int64))
    nv = max(lengths)
    padded = kernels.gather_segments(src, base, nv, f3, length)
    keep = torch.arange(nv, device=DEV)[None, :] < dev(np.array([lengths[i] for i in items]))[:, None]
    want = padded[keep]
    got = kernels.gather_segments_ragged(src, base, seg, f3, length)
    assert got.shape == (seg.rows, length) and torch.equal(got, want)


def test_operands_spanning_more_than_2_28_along_the_contraction():
    """The documented edge of the two-piece fp16 split (sn_gemm.hip: "elements more than 2^28 below their row's maximum lose
    low-order bits — an error below 2^-37 of the row's scale"): data rows AND weight columns whose elements span 2^60, forward and
    input gradient; and the bounded two-piece weight gradient with columns of x spread over 2^40 about their mean and dy rows
    from 1e-12 to 1e3.  Every output stays within a few fp32 roundings of fp64, relative to sum |a||b| of its own row / column —
    what an fp32 dot product in any order guarantees."""
    rng = np.random.default_rng(77)
    rows, K, J = 257, 256, 128
    x = (rng.standard_normal((rows, K)) * np.exp2(rng.integers(-30, 31, size=(rows, K)))).astype(np.float32)
    W = (rng.standard_normal((J, K)) * np.exp2(rng.integers(-30, 31, size=(J, K)))).astype(np.float32)
    ref = x.astype(np.float64) @ W.astype(np.float64).T
    scale = np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64).T
    got = kernels.linear_fwd(dev(x), dev(W), dev(np.zeros(J, np.float32))).cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all() and (np.abs(got - ref) / scale).max() <= 4 * 2.0 ** -24 * np.sqrt(K)
    dy = (rng.standard_normal((rows, J)) * np.exp2(rng.integers(-30, 31, size=(rows, J)))).astype(np.float32)
    refd = dy.astype(np.float64) @ W.astype(np.float64)
    scaled = np.abs(dy).astype(np.float64) @ np.abs(W).astype(np.float64)
    gotd = kernels.linear_dgrad(dev(dy), dev(W)).cpu().numpy().astype(np.float64)
    assert np.isfinite(gotd).all() and (np.abs(gotd - refd) / scaled).max() <= 4 * 2.0 ** -24 * np.sqrt(J)
    # weight gradient on two fp16 pieces: the bounds are max |dy| and BatchNorm's own statistics of x
    rows = 4099
    spread = np.exp2(rng.integers(-20, 21, size=K))[None, :]            # column c: standard deviation 2^e_c, mean a few of them
    xs = ((rng.standard_normal((rows, K)) + 4 * rng.standard_normal(K)[None, :]) * spread).astype(np.float32)
    dys = (rng.standard_normal((rows, J)) * np.array([1e-12, 1e-6, 1.0, 1e3])[np.arange(rows) % 4][:, None]).astype(np.float32)
    xd, dyd = dev(xs), dev(dys)
    st = kernels.colstats(xd)
    mean = (st[0] / rows).float()
    var = (st[1] / rows - (st[0] / rows) ** 2).clamp_min(0)
    invstd = (1.0 / torch.sqrt(var + 1e-5)).float()
    bound = dyd.abs().max().reshape(1)
    G = kernels.wgrad(dyd, xd, mean, bounds=(bound, invstd, rows)).cpu().numpy().astype(np.float64)
    xc = xs.astype(np.float64) - mean.cpu().numpy().astype(np.float64)[None, :]
    refg = dys.astype(np.float64).T @ xc
    scaleg = np.abs(dys).astype(np.float64).T @ np.abs(xc)
    assert np.isfinite(G).all() and (np.abs(G - refg) / scaleg).max() <= 8 * 2.0 ** -24 * np.sqrt(rows)
    G3 = kernels.wgrad(dyd, xd, mean).cpu().numpy().astype(np.float64)          # (the unbounded three-piece form, same operands)
    assert (np.abs(G3 - refg) / scaleg).max() <= 8 * 2.0 ** -24 * np.sqrt(rows)
