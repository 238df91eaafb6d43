"""Launch sequence for PMC passes over the Laplacian product (run under rocprofv3 --pmc ...): the config-5 batch (128
ragged meshes, packed), L at N = 128, 10 launches of the plain kernel, 10 with the statistics epilogue, 10 with the fused
ELU-backward epilogue; then DiA (q3) x 10 for comparison.  Prints the compulsory byte counts."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import kernels, mesh_ops  # noqa: E402
from surfacenetworks_amd.operators import OperatorPool  # noqa: E402

dev = "cuda"
rng = np.random.default_rng(5)
vs = rng.integers(1000, 20001, size=int(os.environ.get("SN_PROBE_MESHES", "128")))
Ls, DiAs = [], []
for v in vs:
    n = int(np.sqrt(v))
    V, F = mesh_ops.grid_cloth(n, int(v) // n, rng)
    Ls.append(mesh_ops.laplacian(V, F).astype(np.float32))
    if os.environ.get("SN_PROBE_DIRAC", "1") == "1":
        DiAs.append(mesh_ops.dirac(V, F)[1].astype(np.float32))
op = OperatorPool(Ls, dev).assemble(np.arange(len(Ls)))
M, K = op.shape
x = torch.randn(K, 128, device=dev)
y = torch.empty(M, 128, device=dev)
e = torch.randn(M, 128, device=dev)
g = torch.randn(M, 128, device=dev)
for _ in range(10):
    kernels.spmm_csr(op.rowptr, op.colind, op.vals, M, K, x, y, 1)
for _ in range(10):
    kernels.spmm_csr_stats(op.rowptr, op.colind, op.vals, M, K, x, y)
for _ in range(10):
    kernels.spmm_csr_elubwd(op.rowptr, op.colind, op.vals, M, K, x, e, g, y, 1)
r = op.rb4()
for _ in range(10):
    kernels.spmm_rb4(r[0], r[1], r[2], M, K, x, y)
print(f"L rb4: listed columns {int(r[0][-1].item())}")
# the sliding-window kernel (round 3): plain, statistics, fused ELU-backward epilogue — 10 launches each
for _ in range(10):
    kernels.spmm_ring(op.rowptr, op.colind, op.vals, M, K, x, y)
for _ in range(10):
    kernels.spmm_ring_stats(op.rowptr, op.colind, op.vals, M, K, x, y)
for _ in range(10):
    kernels.spmm_ring(op.rowptr, op.colind, op.vals, M, K, x, y, e, g)
for _ in range(10):
    kernels.spmm_rb4(r[0], r[1], r[2], M, K, x, y, e, g)
print(f"L ring: band {op.band()}")
print(f"L: M={M} nnz={op.nnz}: compulsory reads {op.nnz * 8 + (M + 1) * 4 + K * 512} B (+ {2 * M * 512} B with E and G), writes {M * 512} B; "
      f"gathered through L1 {op.nnz * 512} B")
if DiAs:
    opd = OperatorPool(DiAs, dev, want_bsr4=True).assemble(np.arange(len(DiAs)))
    Md, Kd = opd.shape
    q = opd.q3()
    xd = torch.randn(Kd // 4, 128, device=dev)
    yd = torch.empty(Md // 4, 128, device=dev)
    for _ in range(10):
        kernels.spmm_q3(q[0], q[1], Md // 4, Kd // 4, xd, yd, 4)
    print(f"DiA q3: Mb={Md // 4} blocks={q[1].shape[0]}: compulsory reads {q[1].shape[0] * 16 + (Md // 4 + 1) * 4 + Kd * 128} B, writes {Md * 128} B; "
          f"gathered through L1 {q[1].shape[0] * 512} B")
torch.cuda.synchronize()
