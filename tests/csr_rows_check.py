"""Called by tests/test_spmm_gpu.py: the generic CSR kernel (row passes per wave as the library chooses them) against the C
oracle, bit for bit — ragged and empty rows,
rows longer than the wave's LDS slice (tiled path), every N, both operand layouts, the fused ELU-backward epilogue and the
statistics variant.  Prints OK."""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from oracle import c_oracle  # noqa: E402
from surfacenetworks_amd import kernels  # noqa: E402

DEV = "cuda"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def ragged_csr(M, K, rng, long_rows=()):
    lens = rng.integers(0, 12, size=M)
    lens[rng.integers(0, M, size=max(1, M // 9))] = 0                      # empty rows, also first / last
    lens[0] = lens[-1] = 0
    for r, n in long_rows:
        lens[r] = n
    lens = np.minimum(lens, K)
    rows = np.repeat(np.arange(M), lens)
    cols = np.concatenate([np.sort(rng.choice(K, size=n, replace=False)) for n in lens]) if lens.sum() else np.zeros(0, np.int64)
    A = sp.csr_matrix((rng.standard_normal(len(rows)).astype(np.float32), (rows, cols)), shape=(M, K))
    A.sort_indices()
    return A


def main():
    rng = np.random.default_rng(11)
    for N in (128, 64, 32, 16):
        for (M, K, long_rows) in [(1031, 777, ()), (257, 900, ((5, 700), (6, 3), (130, 900))), (64, 64, ()), (3, 5, ()),
                                  (4099, 4099, ((4098, 600),))]:
            A = ragged_csr(M, K, rng, long_rows)
            x = rng.standard_normal((K, N)).astype(np.float32)
            want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.ravel(), N).reshape(M, N)
            rp, ci, va = dev(A.indptr.astype(np.int32)), dev(A.indices.astype(np.int32)), dev(A.data)
            y = torch.full((M, N), float("nan"), device=DEV)
            kernels.spmm_csr(rp, ci, va, M, K, dev(x), y, 1)
            assert np.array_equal(y.cpu().numpy(), want), ("plain", N, M)
            # into / out of halves of a concat buffer (row stride 2N)
            xcat = torch.full((K, 2 * N), float("nan"), device=DEV)
            xcat[:, :N] = dev(x)
            ycat = torch.full((M, 2 * N), float("nan"), device=DEV)
            kernels.spmm_csr(rp, ci, va, M, K, xcat[:, :N], ycat[:, N:], 1)
            assert np.array_equal(ycat[:, N:].cpu().numpy(), want) and torch.isnan(ycat[:, :N]).all(), ("strided", N, M)
            # fused ELU-backward epilogue: (A x) * elu'(e) + g
            e = rng.standard_normal((M, N)).astype(np.float32)
            g = rng.standard_normal((M, N)).astype(np.float32)
            d = np.where(e > 0, np.float32(1), e + np.float32(1)).astype(np.float32)
            for gg in (g, None):
                ye = torch.full((M, N), float("nan"), device=DEV)
                kernels.spmm_csr_elubwd(rp, ci, va, M, K, dev(x), dev(e), dev(gg) if gg is not None else None, ye, 1)
                w = want * d if gg is None else want * d + gg
                assert np.array_equal(ye.cpu().numpy(), w.astype(np.float32)), ("epi", N, M)
            if N == 128:
                for strided in (False, True):
                    ybuf = torch.full((M, 2 * N), float("nan"), device=DEV)
                    ys = ybuf[:, N:] if strided else torch.empty((M, N), device=DEV)
                    part = kernels.spmm_csr_stats(rp, ci, va, M, K, xcat[:, :N], ys)
                    assert np.array_equal(ys.cpu().numpy(), want), ("stats y", M)
                    got = part.sum(0).cpu().numpy()
                    w64 = want.astype(np.float64)
                    ref = np.stack([w64.sum(0), (w64 * w64).sum(0)])
                    scale = np.stack([np.abs(w64).sum(0), (w64 * w64).sum(0)]) + 1e-30
                    assert (np.abs(got - ref) / scale).max() < 1e-6, ("stats", M)
    # ---- RB4 (4x1 row blocks): conversion == numpy restatement, product == CSR oracle on the original operator ----
    import cpu_kernels as ck

    for N in (128, 64):
        for (M, K, long_rows) in [(1031, 777, ()), (258, 900, ((5, 700), (6, 3), (130, 900))), (64, 64, ()), (3, 5, ()), (1, 9, ()),
                                  (4099, 4099, ((4098, 600),))]:
            A = ragged_csr(M, K, rng, long_rows)
            x = rng.standard_normal((K, N)).astype(np.float32)
            want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.ravel(), N).reshape(M, N)
            rp, ci, va = dev(A.indptr.astype(np.int32)), dev(A.indices.astype(np.int32)), dev(A.data)
            b = kernels.csr_to_rb4(rp, ci, va, M, K)
            wb = ck.csr_to_rb4(rp.cpu(), ci.cpu(), va.cpu(), M, K)
            tot = int(wb[0][-1])
            assert np.array_equal(b[0].cpu().numpy(), wb[0].numpy()), ("rb4 ptr", M)
            assert np.array_equal(b[1].cpu().numpy()[:tot], wb[1].numpy()[:tot]) and np.array_equal(b[2].cpu().numpy()[:tot], wb[2].numpy()[:tot])
            xcat = torch.full((K, 2 * N), float("nan"), device=DEV)
            xcat[:, :N] = dev(x)
            ycat = torch.full((M, 2 * N), float("nan"), device=DEV)
            kernels.spmm_rb4(b[0], b[1], b[2], M, K, xcat[:, :N], ycat[:, N:])
            assert np.array_equal(ycat[:, N:].cpu().numpy(), want) and torch.isnan(ycat[:, :N]).all(), ("rb4", N, M)
            e = rng.standard_normal((M, N)).astype(np.float32)
            g = rng.standard_normal((M, N)).astype(np.float32)
            d = np.where(e > 0, np.float32(1), e + np.float32(1)).astype(np.float32)
            for gg in (g, None):
                ye = torch.full((M, N), float("nan"), device=DEV)
                kernels.spmm_rb4(b[0], b[1], b[2], M, K, dev(x), ye, dev(e), dev(gg) if gg is not None else None)
                w = want * d if gg is None else want * d + gg
                assert np.array_equal(ye.cpu().numpy(), w.astype(np.float32)), ("rb4 epi", N, M)
                ya = torch.full((M, N), float("nan"), device=DEV)          # ... and the maxima of |y| (one per wave)
                am = kernels.spmm_rb4(b[0], b[1], b[2], M, K, dev(x), ya, dev(e), dev(gg) if gg is not None else None, want_absmax=True)
                assert am is not None and torch.equal(ya, ye) and bool(torch.isfinite(am).all()) and float(am.min()) >= 0.0
                assert float(am.max()) == float(np.abs(w.astype(np.float32)).max()), ("rb4 epi maxima", N, M)
            if N == 128:
                ys = torch.full((M, N), float("nan"), device=DEV)
                part = kernels.spmm_rb4_stats(b[0], b[1], b[2], M, K, dev(x), ys)
                assert np.array_equal(ys.cpu().numpy(), want), ("rb4 stats y", M)
                got = part.sum(0).cpu().numpy()
                w64 = want.astype(np.float64)
                ref = np.stack([w64.sum(0), (w64 * w64).sum(0)])
                scale = np.stack([np.abs(w64).sum(0), (w64 * w64).sum(0)]) + 1e-30
                assert (np.abs(got - ref) / scale).max() < 1e-6, ("rb4 stats", M)
    # group-4 (quaternion view) operands through the generic kernel
    M, K, N = 4 * 300, 4 * 211, 32
    A = ragged_csr(M, K, rng)
    x = rng.standard_normal((K // 4, 4 * N)).astype(np.float32)
    want = c_oracle.spmm_csr(A.indptr, A.indices, A.data, x.ravel(), N).reshape(M // 4, 4 * N)
    y = torch.empty((M // 4, 4 * N), device=DEV)
    kernels.spmm_csr(dev(A.indptr.astype(np.int32)), dev(A.indices.astype(np.int32)), dev(A.data), M, K, dev(x), y, 4)
    assert np.array_equal(y.cpu().numpy(), want)
    torch.cuda.synchronize()
    print("OK")


if __name__ == "__main__":
    main()
