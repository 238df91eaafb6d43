"""Weight-gradient kernel alone at the config-3 shapes (V and F rows, C = 256 and 128): time per launch (torch events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import kernels  # noqa: E402
out = []
for rows in (322624, 627200):
    for C in (256, 128):
        x = torch.randn(rows, C, device="cuda"); dy = torch.randn(rows, 128, device="cuda"); mu = torch.randn(C, device="cuda")
        for _ in range(3): kernels.wgrad(dy, x, mu)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): kernels.wgrad(dy, x, mu)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        out.append("%d/%d %.0f us (%.2f TB/s)" % (rows, C, us, rows * (C + 128) * 4 / us / 1e6))
print(sys.argv[1] if len(sys.argv) > 1 else "", " | ".join(out), flush=True)
