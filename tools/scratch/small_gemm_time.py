import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from graph_timer import *  # noqa
dev = "cuda"
for K in (256, 128):
    W = torch.randn(128, K, device=dev); b = torch.randn(128, device=dev)
    for rows in (32, 256, 2048, 7000, 14000, 77000):
        x = torch.randn(rows, K, device=dev); dy = torch.randn(rows, 128, device=dev); mean = torch.zeros(K, device=dev)
        cat = torch.empty(rows, 256, device=dev)
        f1 = t(lambda: kernels.linear_fwd(x, W, b))
        f2 = t(lambda: kernels.linear_fwd(x, W, b, None, cat[:, :128], False))
        d1 = t(lambda: kernels.linear_dgrad(dy, W))
        w1 = t(lambda: kernels.wgrad(dy, x, mean, want_colsum=True))
        print(f"K={K} rows={rows:6d}: fwd {f1:6.1f} us  fwd(elu only) {f2:6.1f}  dgrad(plain) {d1:6.1f}  wgrad+reduce {w1:6.1f}", flush=True)
e = torch.empty(0, device=dev)
a_, b_ = torch.empty(64, 128, device=dev), torch.empty(64, 128, device=dev)
print("near-empty kernel (elu_into on 64 rows):", t(lambda: kernels.elu_into(a_, b_)))
