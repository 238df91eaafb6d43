"""Prototype measurement: sliding-window (LDS ring) Laplacian SpMM (lap_ring.hip) against the library's RB4 kernel.
Usage: python tools/scratch/lap_ring.py [c5|c4|c3] ...   (several workloads allowed)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import kernels, mesh_ops  # noqa: E402
from surfacenetworks_amd.operators import OperatorPool  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "liblapring.so"))
lib.lr_spmm.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
dev = "cuda"
N = 128


def timeit(f, iters=50, warm=10):
    for _ in range(warm):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def build(which):
    rng = np.random.default_rng(5 if which == "c5" else 3)
    Ls = []
    if which == "c5":
        vs = rng.integers(1000, 20001, size=128)
        for v in vs:
            n = int(np.sqrt(v))
            V, F = mesh_ops.grid_cloth(n, int(v) // n, rng)
            Ls.append(mesh_ops.laplacian(V, F).astype(np.float32))
        pad = None
    elif which == "c4":
        for _ in range(64):
            V, F = mesh_ops.torus_grid(65, 106, rng)
            Ls.append(mesh_ops.laplacian(V, F).astype(np.float32))
        pad = 7000
    else:
        for _ in range(64):
            V, F = mesh_ops.grid_cloth(71, 71, rng)
            Ls.append(mesh_ops.laplacian(V, F).astype(np.float32))
        pad = None
    pool = OperatorPool(Ls, dev)
    sel = np.arange(len(Ls))
    op = pool.assemble(sel) if pad is None else pool.assemble(sel, pad, pad)
    real = int(pool.rows.sum())
    return op, real


def p(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def main():
    for which in sys.argv[1:] or ["c5"]:
        op0, real = build(which)
        for tag, op in (("L", op0), ("LT", op0.t())):
            M, K = op.shape
            x = torch.randn(K, N, device=dev)
            y0 = torch.empty(M, N, device=dev)
            r = op.rb4()
            ms0 = timeit(lambda: kernels.spmm_rb4(r[0], r[1], r[2], M, K, x, y0))
            ab = op.nnz * 8 + (real + 1) * 4 + real * N * 4 * 2
            print(f"{which} {tag}: M={M} nnz={op.nnz} rb4 {ms0:.4f} ms frac={ab / ms0 / 1e-3 / 8e12:.3f}", flush=True)
            e = torch.randn(M, N, device=dev)
            g = torch.randn(M, N, device=dev)
            ye0 = torch.empty(M, N, device=dev)
            kernels.spmm_rb4(r[0], r[1], r[2], M, K, x, ye0, e, g)
            ys0 = torch.empty(M, N, device=dev)
            st0 = kernels.spmm_rb4_stats(r[0], r[1], r[2], M, K, x, ys0)
            st = torch.cuda.current_stream().cuda_stream
            configs = [(512, 512, 32, 128, 896), (1024, 1024, 32, 64, 1792), (512, 512, 64, 128, 1024), (512, 1024, 64, 128, 1024),
                       (256, 512, 64, 256, 512)]
            for (W, NT, CS, nstrips, ecap) in configs:
                y = torch.zeros(M, N, device=dev)

                def run(E=None, G=None, part=None, out=y, mode=0):
                    rc = lib.lr_spmm(p(op.rowptr), p(op.colind), p(op.vals), M, K, p(x), N, p(out), N, N, W, NT, CS, nstrips, ecap,
                                     p(E), N, p(G), N, p(part), mode, C.c_void_p(st))
                    assert rc == 0, rc
                run()
                torch.cuda.synchronize()
                eq = torch.equal(y, y0)
                ms = timeit(run)
                ms1 = timeit(lambda: run(mode=1))
                ms2 = timeit(lambda: run(mode=2))
                line = f"  W={W} NT={NT} CS={CS} strips={nstrips} ecap={ecap}: {ms:.4f} ms frac={ab / ms / 1e-3 / 8e12:.3f} equal={eq} nostore={ms1:.4f} dmaonly={ms2:.4f}"
                if False:
                    ye = torch.zeros(M, N, device=dev)
                    run(e, g, None, ye)
                    mse = timeit(lambda: run(e, g, None, ye))
                    part = torch.zeros(nstrips, 2, N, device=dev)
                    ys = torch.zeros(M, N, device=dev)
                    run(None, None, part, ys)
                    torch.cuda.synchronize()
                    mss = timeit(lambda: run(None, None, part, ys))
                    tot = part.double().sum(0)
                    ref = torch.stack([ys0.double().sum(0), (ys0.double() ** 2).sum(0)])
                    serr = float(((tot - ref).abs() / (ref.abs() + 1e-3))[1].max())
                    line += (f" | epi {mse:.4f} ms close={torch.allclose(ye, ye0, rtol=1e-6, atol=1e-6)} maxdiff={float((ye - ye0).abs().max()):.2e} "
                             f"| stats {mss:.4f} ms equal={torch.equal(ys, ys0)} sq_relerr={serr:.2e}")
                print(line, flush=True)


if __name__ == "__main__":
    main()
