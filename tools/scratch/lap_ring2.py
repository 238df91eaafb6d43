"""Prototype measurement: loader/compute-specialised sliding-window Laplacian SpMM (lap_ring2.hip) against RB4.
Usage: python tools/scratch/lap_ring2.py [c5|c4|c3] ..."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lap_win import build, timeit, p, N, dev  # noqa: E402  (same batches; its library load is unused here)
from surfacenetworks_amd import kernels  # noqa: E402

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblapring2.so"))
lib.lr2_spmm.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                         C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
VARIANTS = {0: ("CS32 W512 R64 H160 D2 L2 q4", 128), 1: ("CS32 W512 R64 H160 D2 L4 q4", 128), 2: ("CS32 W512 R64 H160 D2 L2 q16", 128),
            3: ("CS32 W512 R64 H160 D2 L4 q16", 128), 4: ("CS64 W512 R64 H160 D2 L4 q16", 128), 5: ("CS64 W512 R64 H160 D2 L8 q16", 128),
            6: ("CS64 W512 R64 H160 D2 L4 q4", 128), 7: ("CS64 W512 R64 H128 D3 L4 q16", 128), 8: ("CS32 W1024 R64 H160 D3 L4 q16", 64)}


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
            mse0 = timeit(lambda: kernels.spmm_rb4(r[0], r[1], r[2], M, K, x, ye0, e, g))
            ys0 = torch.empty(M, N, device=dev)
            kernels.spmm_rb4_stats(r[0], r[1], r[2], M, K, x, ys0)
            mss0 = timeit(lambda: kernels.spmm_rb4_stats(r[0], r[1], r[2], M, K, x, ys0))
            st = torch.cuda.current_stream().cuda_stream
            for v, (name, nstrips) in VARIANTS.items():
                y = torch.zeros(M, N, device=dev)

                def run(E=None, G=None, out=y, mode=0, part=None):
                    rc = lib.lr2_spmm(p(op.rowptr), p(op.colind), p(op.vals), M, K, op.nnz, p(x), N, p(out), N, N, v, nstrips,
                                      p(E), N, p(G), N, p(part), mode, C.c_void_p(st))
                    assert rc == 0, rc
                run()
                torch.cuda.synchronize()
                eq = torch.equal(y, y0)
                ms = timeit(run)
                ms1 = timeit(lambda: run(mode=1))
                ms2 = timeit(lambda: run(mode=2))
                ye = torch.zeros(M, N, device=dev)
                run(e, g, ye)
                mse = timeit(lambda: run(e, g, ye))
                part = torch.zeros(nstrips, 2, N, device=dev)
                ys = torch.zeros(M, N, device=dev)
                run(part=part, out=ys)
                torch.cuda.synchronize()
                mss = timeit(lambda: run(part=part, out=ys))
                tot = part.double().sum(0)
                ref = torch.stack([ys0.double().sum(0), (ys0.double() ** 2).sum(0)])
                serr = float(((tot - ref).abs() / (ref.abs() + 1e-3))[1].max())
                print(f"  v{v} {name} strips={nstrips}: {ms:.4f} ms frac={ab / ms / 1e-3 / 8e12:.3f} equal={eq} nostore={ms1:.4f} dmaonly={ms2:.4f} | "
                      f"epi {mse:.4f} ms (rb4 {mse0:.4f}) close={torch.allclose(ye, ye0, rtol=1e-6, atol=1e-6)} | stats {mss:.4f} (rb4 {mss0:.4f}) "
                      f"equal={torch.equal(ys, ys0)} sq_relerr={serr:.1e}", flush=True)


if __name__ == "__main__":
    main()
