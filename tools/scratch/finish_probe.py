"""Launch time of the weight gradient + coefficients: separate launches vs kernels.wgrad_bn (env SN_FIN_DEBUG read per process)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from surfacenetworks_amd import kernels

def run(rows, C, J=128, reps=30):
    rng = np.random.default_rng(0)
    dy = torch.randn(rows, J, device="cuda")
    x = torch.randn(rows, C, device="cuda")
    W = torch.randn(J, C, device="cuda") / 9
    gamma, beta = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    mean = x.mean(0)
    invstd = 1 / torch.sqrt(x.var(0, unbiased=False) + 1e-5)
    s = gamma * invstd
    bounds = (dy.abs().max().reshape(1), invstd, rows)
    def old():
        G, sdy = kernels.wgrad(dy, x, mean, want_colsum=True, bounds=bounds)
        return kernels.bn_bwd_coeffs(G, sdy, W, s, invstd, beta, rows, True)
    def new():
        return kernels.wgrad_bn(dy, x, mean, W, s, invstd, beta, rows, True, bounds)
    out = []
    for f in (old, new):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1000 / reps)
    print(f"rows {rows} C {C}: separate {out[0]:.1f} us, two-launch {out[1]:.1f} us")

for rows, C in ((322624, 128), (322624, 256), (627200, 256), (7000, 256)):
    run(rows, C)
