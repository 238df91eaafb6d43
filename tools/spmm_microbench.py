"""SpMM roofline microbench (HIP events on the launch stream).
Usage: python tools/spmm_microbench.py [c3|c4|c5|mnist] [perm] ; SN_MB_LAYOUT=packed|padded (default: packed for c5, padded else)
Prints one line per (operator, format): avg ms, algorithmic GB/s (CSR/int32/fp32 bytes), % of 8 TB/s.
The algorithmic bytes ALWAYS come from the meshes' real sizes (sum of V_i, F_i): a padded batch gets no credit for the
padding rows it scans and the zeros it writes (round-1 VERDICT: the padded numerator inflated config 5 by 1.7x)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import kernels, mesh_ops  # noqa: E402
from surfacenetworks_amd.operators import OperatorPool  # noqa: E402

PEAK = 8.0e12


def alg_bytes(M, K, nnz, N):
    return nnz * 8 + (M + 1) * 4 + K * N * 4 + M * N * 4


def time_launch(fn, iters=60, warm=25):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def build_batch(workload, permute=False):
    rng = np.random.default_rng(5 if workload == "c5" else 3)
    t = time.time()
    if workload == "c3":
        sizes = [(71, 71)] * 64
    elif workload == "c4":
        sizes = "torus"
    elif workload == "mnist":
        sizes = None
    else:
        vs = rng.integers(1000, 20001, size=128)
        sizes = []
        for v in vs:
            n = int(np.sqrt(v))
            sizes.append((n, int(v) // n))
    meshes = []
    if sizes is None:
        for i in range(512):
            V, F = mesh_ops.delaunay_disc(150, rng)
            meshes.append(mesh_ops.mesh_operators(V, F))
    elif sizes == "torus":                       # FAUST-sized closed meshes (65 x 106 = 6890 vertices), SURVEY.md §8d C4
        for i in range(int(os.environ.get("SN_MB_C4_MESHES", "64"))):
            V, F = mesh_ops.torus_grid(65, 106, rng)
            meshes.append(mesh_ops.mesh_operators(V, F))
    else:
        cache = {}
        for (n, m) in sizes:
            V, F = mesh_ops.grid_cloth(n, m, rng, permute=permute)
            meshes.append(mesh_ops.mesh_operators(V, F))
    print(f"# built {len(meshes)} meshes in {time.time() - t:.1f}s", flush=True)
    return meshes


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "c3"
    permute = len(sys.argv) > 2 and sys.argv[2] == "perm"
    meshes = build_batch(workload, permute)
    dev = "cuda"
    C = 64 if workload == "mnist" else 128
    only = os.environ.get("SN_MB_ONLY", "")
    for name, group in [("Di", 4), ("DiA", 4), ("L", 1)]:
        if (only == "bsr4" and group != 4) or (only == "L" and group != 1):
            continue
        mats = [m[name] for m in meshes]
        s0 = max(m.shape[0] for m in mats)
        s1 = max(m.shape[1] for m in mats)
        if workload == "c4" and group == 1:
            s0 = s1 = 7000                               # dense_correspondence/main.py:193 pads to 7000
        pool = OperatorPool(mats, dev, want_bsr4=(group == 4))
        layout = os.environ.get("SN_MB_LAYOUT", "packed" if workload == "c5" else "padded")
        op = pool.assemble(np.arange(len(mats))) if layout == "packed" else pool.assemble(np.arange(len(mats)), s0, s1)
        real = {"fwd": (int(pool.rows.sum()), int(pool.cols.sum())), "bwd(T)": (int(pool.cols.sum()), int(pool.rows.sum()))}
        for tag, o in [("fwd", op), ("bwd(T)", op.t())]:
            M, K = o.shape
            N = C // group
            x = torch.randn(K // group, group * N, device=dev)
            y = torch.empty(M // group, group * N, device=dev)
            ab = alg_bytes(real[tag][0], real[tag][1], o.nnz, N)     # real (unpadded) sizes, whatever the layout
            ms = time_launch(lambda: kernels.spmm_csr(o.rowptr, o.colind, o.vals, M, K, x, y, group)) if only != "bsr4" else float("nan")
            print(f"{workload}/{layout} {name:3s} {tag:6s} csr  N={N:3d} M={M} K={K} nnz={o.nnz} algMB={ab / 1e6:.1f} ms={ms:.4f} GB/s={ab / ms / 1e6:.0f} frac={ab / ms / 1e-3 / PEAK:.3f}", flush=True)
            if group == 1:                                   # Laplacian-type: the 4x1 row-blocked form (default of the product)
                r = o.rb4()
                if r is not None:
                    y4 = torch.empty_like(y)
                    ms = time_launch(lambda: kernels.spmm_rb4(r[0], r[1], r[2], M, K, x, y4))
                    tot = int(r[0][-1].item())
                    actual = tot * 20 + (M // 4 + 1) * 4 + real[tag][1] * N * 4 + real[tag][0] * N * 4
                    print(f"{workload}/{layout} {name:3s} {tag:6s} rb4  N={N:3d} listed={tot} ({tot / max(M, 1):.2f}/row vs {o.nnz / max(M, 1):.2f} entries/row) actualMB={actual / 1e6:.1f} ms={ms:.4f} GB/s(alg)={ab / ms / 1e6:.0f} frac={ab / ms / 1e-3 / PEAK:.3f} equal={torch.equal(y, y4)}", flush=True)
            if group == 1 and kernels.spmm_ring_supported(N, 1, M, K):
                band, longest, outside = o.band()
                y5 = torch.empty_like(y)
                ms = time_launch(lambda: kernels.spmm_ring(o.rowptr, o.colind, o.vals, M, K, x, y5))
                print(f"{workload}/{layout} {name:3s} {tag:6s} ring N={N:3d} band={band} longest_row={longest} rows_outside_window={outside} ring_ok={o.ring_ok(N)} ms={ms:.4f} "
                      f"GB/s(alg)={ab / ms / 1e6:.0f} frac={ab / ms / 1e-3 / PEAK:.3f} equal={torch.equal(y, y5)}", flush=True)
            if group == 4:
                b = o.bsr4()
                y2 = torch.empty_like(y)
                ms = time_launch(lambda: kernels.spmm_bsr4(b[0], b[1], b[2], M // 4, K // 4, x, y2, group))
                actual = b[1].numel() * 68 + (M // 4 + 1) * 4 + K * N * 4 + M * N * 4
                if only == "bsr4":
                    kernels.spmm_csr(o.rowptr, o.colind, o.vals, M, K, x, y, group)
                print(f"{workload}/{layout} {name:3s} {tag:6s} bsr4 N={N:3d} blocks={b[1].numel()} actualMB={actual / 1e6:.1f} ms={ms:.4f} GB/s(alg)={ab / ms / 1e6:.0f} frac={ab / ms / 1e-3 / PEAK:.3f} equal={torch.equal(y, y2)}", flush=True)
                q = o.q3()                                   # quaternion-packed form (the default of the product)
                if q is not None:
                    y3 = torch.empty_like(y)
                    ms = time_launch(lambda: kernels.spmm_q3(q[0], q[1], M // 4, K // 4, x, y3, group))
                    actual = q[1].shape[0] * 16 + (M // 4 + 1) * 4 + K * N * 4 + M * N * 4
                    print(f"{workload}/{layout} {name:3s} {tag:6s} q3   N={N:3d} blocks={q[1].shape[0]} actualMB={actual / 1e6:.1f} ms={ms:.4f} GB/s(alg)={ab / ms / 1e6:.0f} frac={ab / ms / 1e-3 / PEAK:.3f} GB/s(actual)={actual / ms / 1e6:.0f} equal={torch.equal(y2, y3)}", flush=True)
    if only:
        return
    # plain copy ceiling on this box for reference
    n = 256 * 1024 * 1024
    a = torch.empty(n, device=dev)
    b_ = torch.empty(n, device=dev)
    ms = time_launch(lambda: b_.copy_(a), iters=10)
    print(f"copy 2x{n * 4 / 1e6:.0f}MB ms={ms:.3f} GB/s={2 * n * 4 / ms / 1e6:.0f}")


if __name__ == "__main__":
    main()
