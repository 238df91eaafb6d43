"""Loop-average vs per-launch (kernel start/stop events) timing of the Laplacian products on the config-5 batch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import functional as snF, kernels, mesh_ops
from surfacenetworks_amd.operators import OperatorPool
dev = "cuda"
rng = np.random.default_rng(5)
vs = rng.integers(1000, 20001, size=128)
Ls = []
for v in vs:
    n = int(np.sqrt(v))
    Ls.append(mesh_ops.laplacian(*mesh_ops.grid_cloth(n, int(v) // n, rng)).astype(np.float32))
op = OperatorPool(Ls, dev).assemble(np.arange(128))
M, K = op.shape
ab = op.nnz * 8 + (M + 1) * 4 + 2 * M * 512
for fmt in (("ring",) if os.environ.get("SN_RING_ONLY") else ("ring", "rb4")):
    snF.set_laplacian_format(fmt)
    for tag, o in (("L", op), ("LT", op.t())):
        x = torch.randn(K, 128, device=dev); y = torch.empty(M, 128, device=dev)
        for _ in range(10): snF._launch(o, x, y, 1, "t")
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(50): snF._launch(o, x, y, 1, "t")
        e.record(); torch.cuda.synchronize()
        loop = s.elapsed_time(e) / 50
        t = snF.SpmmTimer()
        with t:
            for _ in range(50): snF._launch(o, x, y, 1, "t")
        ms = np.array([r[5] for r in t.results()])
        # events again but with a sync between launches (idle GPU before every launch)
        t2 = snF.SpmmTimer()
        with t2:
            for _ in range(20):
                snF._launch(o, x, y, 1, "t"); torch.cuda.synchronize()
        ms2 = np.array([r[5] for r in t2.results()])
        print(f"{fmt} {tag}: loop {loop:.4f} ms ({ab/loop/1e-3/8e12:.3f}) | per-launch events back to back: median {np.median(ms):.4f} min {ms.min():.4f} "
              f"({ab/np.median(ms)/1e-3/8e12:.3f}) | with a sync between launches: median {np.median(ms2):.4f} ({ab/np.median(ms2)/1e-3/8e12:.3f})", flush=True)
