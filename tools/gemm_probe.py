"""GEMM kernel probe (run plain or under rocprofv3 --pmc): forward / dgrad / wgrad of the folded BatchNorm+Linear at the
config-3 shapes, F rows (627200) and V rows (322624), 10 launches each, with timings."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import kernels  # noqa: E402

dev = "cuda"
for rows in (627200, 322624):
    x = torch.randn(rows, 256, device=dev)
    dy = torch.randn(rows, 128, device=dev)
    W = torch.randn(128, 256, device=dev) / 16
    b = torch.randn(128, device=dev)
    mu, B, Cc = torch.randn(256, device=dev), torch.randn(256, device=dev), torch.randn(256, device=dev)
    res = torch.randn(rows, 128, device=dev)
    cat = torch.empty(rows, 256, device=dev)
    # accuracy against fp64 on a slice
    ref = (x[:4096].double() @ W.double().t() + b.double())
    got = kernels.linear_fwd(x[:4096], W, b).double()
    print(f"fwd max|err|/max|ref| = {float((got - ref).abs().max() / ref.abs().max()):.3e}  "
          f"rms = {float(((got - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()):.3e}")
    for name, fn in (("fwd", lambda: kernels.linear_fwd(x, W, b)),
                     ("fwd+res+elu", lambda: kernels.linear_fwd(x, W, b, residual=res, y_elu=cat[:, :128])),
                     ("dgrad+affine", lambda: kernels.linear_dgrad(dy, W, x, mu, B, Cc)),
                     ("wgrad", lambda: kernels.wgrad(dy, x, mu)),
                     ("torch addmm", lambda: torch.addmm(b, x, W.t())),
                     ("torch dgrad mm", lambda: dy.mm(W))):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        flops = 2.0 * rows * 128 * 256
        print(f"rows={rows} {name:15s} {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF/s ({flops / ms / 1e9 / 157.3 * 100:4.1f}% of fp32 MFMA peak)", flush=True)
