"""One GEMM shape, a few launches: run under rocprofv3 --pmc to see what bounds the kernel.
usage: gemm_pmc_probe.py [fwd|fwd_elu|fwd_copy|dgrad|wgrad] [rows]      (fwd_copy: activated copy + statistics only — the
eight-wave kernel gemm_fwd_w8_k at K = 256)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import kernels  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 627200
dev = "cuda"
x = torch.randn(rows, 256, device=dev)
dy = torch.randn(rows, 128, device=dev)
W = torch.randn(128, 256, device=dev) / 16
b = torch.randn(128, device=dev)
res = torch.randn(rows, 128, device=dev)
cat = torch.empty(rows, 256, device=dev)
mu, B, Cc = torch.randn(256, device=dev), torch.randn(256, device=dev), torch.randn(256, device=dev)
fn = {"fwd": lambda: kernels.linear_fwd(x, W, b),
      "fwd_elu": lambda: kernels.linear_fwd(x, W, b, residual=res, y_elu=cat[:, :128]),
      "fwd_copy": lambda: kernels.linear_fwd(x, W, b, y_elu=cat[:, :128], want_y=False, elu_stats=kernels.new_elu_stats_part(rows, dev)),
      "dgrad": lambda: kernels.linear_dgrad(dy, W, x, mu, B, Cc),
      "wgrad": lambda: kernels.wgrad(dy, x, mu)}[what]
for _ in range(5):
    fn()
torch.cuda.synchronize()
