import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from surfacenetworks_amd import kernels
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng = np.random.default_rng(99)
for rows in (4096, 65536, 627200):
    K, J = 256, 128
    x = (rng.standard_normal((rows, K)) * np.exp(rng.standard_normal((rows, K)))).astype(np.float32)
    W = (rng.standard_normal((J, K)) / 16).astype(np.float32)
    dy = rng.standard_normal((rows, J)).astype(np.float32)
    n = min(rows, 8192)
    ref = x[:n].astype(np.float64) @ W.astype(np.float64).T
    e_np = np.abs((x[:n] @ W.T).astype(np.float64) - ref).max()
    e_k = np.abs(kernels.linear_fwd(dev(x[:n]), dev(W), dev(np.zeros(J, np.float32))).cpu().numpy().astype(np.float64) - ref).max()
    refg = dy.astype(np.float64).T @ x.astype(np.float64)
    e_gnp = np.abs((dy.T @ x).astype(np.float64) - refg).max()
    e_gt = np.abs((dev(dy).t() @ dev(x)).cpu().numpy().astype(np.float64) - refg).max()
    e_gk = np.abs(kernels.wgrad(dev(dy), dev(x)).cpu().numpy().astype(np.float64) - refg).max()
    print(f"rows={rows}: fwd err kernel {e_k:.3e} numpy-fp32 {e_np:.3e} | wgrad err kernel {e_gk:.3e} numpy-fp32 {e_gnp:.3e} torch(hipBLAS) {e_gt:.3e}  scale {np.abs(refg).max():.3e}", flush=True)
