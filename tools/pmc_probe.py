"""Launch sequence for PMC calibration (run under rocprofv3 --pmc ...): a streaming copy of known size, then the Dirac
SpMM products of the config-3 batch, 10 plain launches and 10 with the fused ELU-backward epilogue each.  Prints the byte counts every launch should move."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import kernels, mesh_ops  # noqa: E402
from surfacenetworks_amd.operators import OperatorPool  # noqa: E402

dev = "cuda"
rng = np.random.default_rng(3)
n = 128 * 1024 * 1024
a = torch.randn(n, device=dev)
b = torch.empty_like(a)
for _ in range(10):
    b.copy_(a)
print(f"copy: reads {n * 4} B writes {n * 4} B per launch")
meshes = []
for _ in range(64):
    V, F = mesh_ops.grid_cloth(71, 71, rng)
    meshes.append(mesh_ops.mesh_operators(V, F))
for name in ("Di", "DiA"):
    mats = [m[name] for m in meshes]
    pool = OperatorPool(mats, dev, want_bsr4=True)
    op = pool.assemble(np.arange(64), mats[0].shape[0], mats[0].shape[1])
    for tag, o in (("fwd", op), ("bwdT", op.t())):
        M, K = o.shape
        x = torch.randn(K // 4, 128, device=dev)
        y = torch.empty(M // 4, 128, device=dev)
        bb = o.bsr4()
        for _ in range(10):
            kernels.spmm_bsr4(bb[0], bb[1], bb[2], M // 4, K // 4, x, y, 4)
        e = torch.randn(M // 4, 128, device=dev)
        g = torch.randn(M // 4, 128, device=dev)
        for _ in range(10):                          # fused ELU-backward epilogue: + E and G reads (M*32*4 bytes each)
            kernels.spmm_bsr4_elubwd(bb[0], bb[1], bb[2], M // 4, K // 4, x, e, g, y, 4)
        qq = o.q3()                                  # quaternion-packed form: 16-byte records
        for _ in range(10):
            kernels.spmm_q3(qq[0], qq[1], M // 4, K // 4, x, y, 4)
        for _ in range(10):
            kernels.spmm_q3(qq[0], qq[1], M // 4, K // 4, x, y, 4, e, g)
        if tag == "fwd":                             # forward launches of the blocks: + the statistics partials (1 KiB / 32 rows)
            for _ in range(10):
                kernels.spmm_q3_stats(qq[0], qq[1], M // 4, K // 4, x, y, 4)
        print(f"{name} {tag} q3: expected reads {qq[1].shape[0] * 16 + (M // 4 + 1) * 4 + K * 128} B (+ {2 * M * 128} B with E and G), writes {M * 128} B")
        rd = bb[1].numel() * 68 + (M // 4 + 1) * 4 + K * 32 * 4
        print(f"{name} {tag}: expected reads {rd} B (operator {bb[1].numel() * 68 + (M // 4 + 1) * 4} + X {K * 128}), writes {M * 128} B; "
              f"algorithmic CSR bytes {o.nnz * 8 + (M + 1) * 4 + K * 128 + M * 128}")
torch.cuda.synchronize()
