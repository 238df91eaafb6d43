"""Forward GEMM: the 8-wave 16-column kernel (SN_GEMM_W8) against float64 and timing per launch.
usage: SN_GEMM_W8=<0|1|2> python tools/scratch/fwd_w8_probe.py [tag]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import kernels  # noqa: E402

torch.manual_seed(0)
out = []
for rows in (322624, 627200 + 7):
    for K in (256, 128):
        x = torch.randn(rows, K, device="cuda") * torch.exp(torch.randn(rows, 1, device="cuda"))
        W = torch.randn(128, K, device="cuda") * 0.1
        b = torch.randn(128, device="cuda")
        res = torch.randn(rows, 128, device="cuda")
        cat = torch.empty(rows, 256, device="cuda")
        y64 = x.double() @ W.double().t() + b.double()
        den = x.double().abs() @ W.double().abs().t() + b.double().abs()
        for name, r, el, wy in (("plain", None, False, True), ("elu-only", None, True, False), ("res+elu+y", res, True, True)):
            part = kernels.new_elu_stats_part(rows, x.device) if el else None
            args = dict(residual=r, y_elu=cat[:, :128] if el else None, want_y=wy, elu_stats=part)
            for _ in range(3):
                y = kernels.linear_fwd(x, W, b, **args)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                y = kernels.linear_fwd(x, W, b, **args)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 20 * 1e3
            ref = y64 + (r.double() if r is not None else 0)
            errs = []
            if y is not None:
                errs.append("y %.1e" % ((y.double() - ref).abs() / den).max().item())
            if el:
                eref = torch.nn.functional.elu(ref)
                errs.append("elu %.1e" % ((cat[:, :128].double() - eref).abs() / den).max().item())
                st = kernels.colstats_from_part(part, rows)
                errs.append("stats %.1e %.1e" % (((st[0] - eref.sum(0)).abs() / eref.abs().sum(0)).max().item(),
                                                 ((st[1] - (eref * eref).sum(0)).abs() / (eref * eref).sum(0)).max().item()))
            nbytes = rows * 4 * (K + (128 if wy else 0) + (128 if r is not None else 0) + (128 if el else 0))
            out.append("%d/%d %-10s %6.1f us %.2f TB/s  %s" % (rows, K, name, us, nbytes / us / 1e6, " ".join(errs)))
        del x, y64, den, res, cat
print(sys.argv[1] if len(sys.argv) > 1 else "", "\n  ".join([""] + out), flush=True)
