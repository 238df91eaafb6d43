"""Kernel times of the fused correspondence loss (sn_pair_fused_*) at the FAUST size, for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import dense_correspondence as dc  # noqa: E402

rows, NA, NB, K = 7000, 6890, 6890, 120
g = torch.Generator().manual_seed(1)
FA = (torch.randn(1, rows, K, generator=g) * 0.5).cuda().requires_grad_(True)
FB = (torch.randn(1, rows, K, generator=g) * 0.5).cuda().requires_grad_(True)
tgt = torch.randint(0, NB, (NA,), generator=g).cuda()


def fused():
    l = dc.fused_pair_cross_entropy(FA, FB, tgt, NA, NB)
    torch.autograd.grad(l, (FA, FB))


def mat():
    l = dc.pair_cross_entropy(torch.bmm(FA, FB.transpose(1, 2)), tgt, NA, NB)
    torch.autograd.grad(l, (FA, FB))


for name, f in (("fused", fused), ("materialised", mat)):
    for _ in range(5):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(20):
        f()
    e.record()
    torch.cuda.synchronize()
    print(f"{name}: {s.elapsed_time(e) / 20:.3f} ms per fwd+bwd", flush=True)
