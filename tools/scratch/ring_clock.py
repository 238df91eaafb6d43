"""Per-workgroup durations of the ring kernel on the config-5 Laplacian batch (debug build with -DSN_X_RING_CLOCK: the statistics
variant leaves each workgroup's duration / end time in its partials): how unequal are the persistent workgroups?"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from surfacenetworks_amd import _lib
_lib.LIB_PATH = os.path.join(root, "tools", "scratch", "dbg", "libsn_ringclock.so")
from surfacenetworks_amd import kernels, mesh_ops  # noqa: E402
from surfacenetworks_amd.operators import OperatorPool  # noqa: E402

dev = "cuda"
rng = np.random.default_rng(5)
Ls = []
for v in rng.integers(1000, 20001, size=128):
    n = int(np.sqrt(v)); V, F = mesh_ops.grid_cloth(n, int(v) // n, rng)
    Ls.append(mesh_ops.laplacian(V, F).astype(np.float32))
op = OperatorPool(Ls, dev).assemble(np.arange(128))
M = op.shape[0]
x = torch.randn(M, 128, device=dev); y = torch.empty(M, 128, device=dev)
for rep in range(6):
    part = kernels.spmm_ring_stats(op.rowptr, op.colind, op.vals, M, M, x, y)
    torch.cuda.synchronize()
    p = part.reshape(-1, 2, 128).cpu().numpy()
    dur = np.concatenate([p[:, 0, 0], p[:, 0, 64]]) / 100.0          # us
    end = np.concatenate([p[:, 1, 0], p[:, 1, 64]]) / 100.0
    end = (end - end.min())
    if rep >= 2:
        print(f"run {rep}: {len(dur)} workgroups, duration us min {dur.min():.1f} mean {dur.mean():.1f} max {dur.max():.1f} | "
              f"last - first to finish {end.max() - end.min():.1f} us, mean finish {end.mean():.1f}, idle tail share {(end.max() - end.mean()) / dur.max() * 100:.1f} %")
        if rep == 5:
            xcd = np.arange(len(dur) // 2) % 8
            d0 = p[:, 0, 0] / 100.0
            print("  mean duration by strip % 8 (XCD order of dispatch):", np.round([d0[xcd == k].mean() for k in range(8)], 1))
