"""device time of the helper kernels around a BatchNorm+Linear at FAUST / ARAP sizes (graph replay of 50 launches)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from graph_timer import *  # noqa

dev = "cuda"
J, C = 128, 256
for rows in (7000, 77000, 627200):
    x = torch.randn(rows, C, device=dev); dy = torch.randn(rows, J, device=dev)
    W = torch.randn(J, C, device=dev); b = torch.randn(J, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    lo, nlo = kernels.colstats_partial(x[:, :128]); hi, nhi = kernels.colstats_partial(x[:, 128:])
    stats = torch.zeros(2, C, dtype=torch.float64, device=dev)
    t_final = t(lambda: kernels.colstats_merge_into(lo[:nlo], stats, 0))
    t_fold = t(lambda: kernels.bn_fold(stats, rows, gamma, beta, W, b, 1e-5, 0.1, True, rm, rv))
    mean, invstd, s_, t_, Wf, bf = kernels.bn_fold(stats, rows, gamma, beta, W, b, 1e-5, 0.1, True, rm, rv)
    t_wg = t(lambda: kernels.wgrad(dy, x, mean, want_colsum=True))
    G, sdy = kernels.wgrad(dy, x, mean, want_colsum=True)
    t_co = t(lambda: kernels.bn_bwd_coeffs(G, sdy, W, s_, invstd, beta, rows, True))
    print(f"rows={rows}: colstats_final({nlo} partial rows) {t_final:.2f} us  bn_fold {t_fold:.2f}  wgrad+reduce {t_wg:.2f}  bn_bwd_coeffs {t_co:.2f}", flush=True)
