"""Two-piece fp16 weight gradient (sn_wgrad_bounded_f32) against the bf16 form and float64: time per launch and error.
usage: SN_WGRAD_H=<mode> python tools/scratch/wgrad_h_probe.py [tag]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import kernels  # noqa: E402

torch.manual_seed(0)
out = []
for rows in (322624, 627200):
    for C in (256, 128):
        sig = torch.exp(torch.randn(C, device="cuda") * 1.5)                 # column scales over ~3 decades
        mu0 = torch.randn(C, device="cuda") * 3
        x = torch.randn(rows, C, device="cuda") * sig + mu0
        dy = torch.randn(rows, 128, device="cuda") * 1e-5 * torch.exp(torch.randn(128, device="cuda"))
        dy[torch.randint(0, rows, (64,)), torch.randint(0, 128, (64,))] *= 300.0      # a few outliers
        xd = x.double()
        mean64 = xd.mean(0)
        var64 = (xd - mean64).pow(2).mean(0)
        mean = mean64.float()
        invstd = (1.0 / torch.sqrt(var64 + 1e-5)).float().contiguous()
        dyb = dy.abs().max().reshape(1)
        bounds = (dyb, invstd, rows)
        G64 = dy.double().t() @ (xd - mean.double())
        den = dy.double().abs().t() @ (xd - mean.double()).abs()
        res = {}
        for name, b in (("bf16x3", None), ("fp16x2", bounds)):
            for _ in range(3):
                G = kernels.wgrad(dy, x, mean, bounds=b)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                G = kernels.wgrad(dy, x, mean, bounds=b)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 20 * 1e3
            err = ((G.double() - G64).abs() / den).max().item()
            rel = ((G.double() - G64).norm() / G64.norm()).item()
            res[name] = (us, err, rel)
        out.append("%d/%d " % (rows, C) + " ".join("%s %.0f us err %.1e rel %.1e" % (k, *v) for k, v in res.items()))
        del x, dy, xd, G64, den
print(sys.argv[1] if len(sys.argv) > 1 else "", "\n  ".join([""] + out), flush=True)
