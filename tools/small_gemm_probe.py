"""Back-to-back launch time of the Linear kernels on small operands (the FAUST towers: 7000 rows; the Mesh-MNIST batch: 76 800 /
145 920 rows): how much of a launch is the weight prologue?  GPU duration of the kernel itself (its own start / stop events), median of 100 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surfacenetworks_amd import functional as snF, kernels  # noqa: E402

dev = "cuda"


def timed(fn, n=100, warm=20):
    """GPU duration of the launch itself (the kernel's own start / stop events, hipExtLaunchKernelGGL), median over n."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    timer = snF.SpmmTimer()
    with timer:
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    timer.results()
    ms = sorted(r[5] for r in timer.linear)
    return ms[len(ms) // 2] * 1e3 if ms else float("nan")


for K, J in ((256, 128), (128, 128), (128, 64)):
    W = torch.randn(J, K, device=dev) / 16
    b = torch.randn(J, device=dev)
    for rows in (32, 7000, 76800, 145920):
        x = torch.randn(rows, K, device=dev)
        y = torch.empty(rows, J, device=dev)
        dy = torch.randn(rows, J, device=dev)
        t_f = timed(lambda: kernels.linear_fwd(x, W, b))
        t_d = timed(lambda: kernels.linear_dgrad(dy, W))
        print(f"K={K:3d} J={J:3d} rows={rows:6d}: forward {t_f:6.1f} us   input gradient {t_d:6.1f} us   (streaming time of the bytes at 5 TB/s: "
              f"{rows * (K + J) * 4 / 5e6:5.1f} us)")
