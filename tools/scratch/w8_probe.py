"""GPU duration of the eight-wave forward kernel (K = 256, activated copy + statistics only) and of the four-wave forward with
residual, config-3 shapes; median of 40 launches (the kernel's own start / stop events)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from surfacenetworks_amd import functional as snF, kernels
dev = "cuda"
for rows in (627200, 322624):
    x = torch.randn(rows, 256, device=dev); W = torch.randn(128, 256, device=dev) / 16; b = torch.randn(128, device=dev)
    res = torch.randn(rows, 128, device=dev); cat = torch.empty(rows, 256, device=dev)
    part = kernels.new_elu_stats_part(rows, dev)
    for name, fn in (("copy-only (w8)", lambda: kernels.linear_fwd(x, W, b, y_elu=cat[:, :128], want_y=False, elu_stats=part)),
                     ("res + y + copy (4 waves)", lambda: kernels.linear_fwd(x, W, b, residual=res, y_elu=cat[:, :128], elu_stats=part))):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        t = snF.SpmmTimer()
        with t:
            for _ in range(40): fn()
        torch.cuda.synchronize(); t.results()
        ms = sorted(r[5] for r in t.linear); nb = t.linear[0][4]
        print(f"rows={rows} {name:26s} median {ms[len(ms)//2]*1e3:7.1f} us  min {ms[0]*1e3:7.1f}  {nb/ms[len(ms)//2]/1e9:5.2f} TB/s")
