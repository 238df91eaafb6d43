"""Fixed-cost probe: Dirac BSR4 SpMM time vs batch size (meshes 71x71), warm (back-to-back) and cold (a 1 GiB fill between
launches evicts L2 + Infinity Cache), each launch bracketed by HIP events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import kernels, mesh_ops  # noqa: E402
from surfacenetworks_amd.operators import OperatorPool  # noqa: E402

dev = "cuda"
rng = np.random.default_rng(3)
base = []
for _ in range(8):
    V, F = mesh_ops.grid_cloth(71, 71, rng)
    base.append(mesh_ops.dirac(V, F)[0].astype(np.float32))
pool = OperatorPool(base, dev, want_bsr4=True)
junk = torch.empty(256 * 1024 * 1024, device=dev)
for B in [4, 8, 16, 32, 64, 128, 256]:
    op = pool.assemble(np.arange(B) % 8, base[0].shape[0], base[0].shape[1])
    for tag, o in (("Di", op), ("DiT", op.t())):
        M, K = o.shape
        x = torch.randn(K // 4, 128, device=dev)
        y = torch.empty(M // 4, 128, device=dev)
        b = o.bsr4()
        fn = lambda: kernels.spmm_bsr4(b[0], b[1], b[2], M // 4, K // 4, x, y, 4)
        for _ in range(10):
            fn()
        res = {}
        for mode in ("warm", "cold"):
            ts = []
            for _ in range(20):
                if mode == "cold":
                    junk.fill_(1.0)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); fn(); e.record()
                ts.append((s, e))
            torch.cuda.synchronize()
            res[mode] = float(np.median([a.elapsed_time(c) for a, c in ts])) * 1e3
        actual = b[1].numel() * 68 + (M // 4 + 1) * 4 + (K + M) * 128
        print(f"B={B:4d} {tag:3s} actualMB={actual / 1e6:8.1f} warm={res['warm']:8.1f}us cold={res['cold']:8.1f}us  cold GB/s(actual)={actual / res['cold'] / 1e3:6.0f}", flush=True)
