"""Fixed cost per launch of the large Linear kernels: time at three row counts, straight-line fit (intercept = what a launch
costs before / after it streams: weight prologue, pipeline fill, scale reductions, partial stores)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import kernels, _lib  # noqa: E402

dev = "cuda"
ROWS = (160000, 320000, 640000)


def timed(fn, n=20):
    lib = _lib.load()
    for _ in range(3):
        fn()
    lib.sn_timing_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    cnt = int(lib.sn_timing_count())
    import ctypes
    meta = np.zeros((max(cnt, 1), 5), np.int64); ms = np.zeros(max(cnt, 1), np.float64)
    written = ctypes.c_int64(0)
    _lib.call("sn_timing_drain", ms.ctypes.data, meta.ctypes.data, cnt, ctypes.addressof(written))
    lib.sn_timing_enable(0)
    big = ms[: written.value]
    big = big[meta[: written.value, 0] >= 0x100]          # the Linear-layer launches (not their small reductions)
    return float(np.median(big)) * 1e3


def fit(name, f):
    us = [f(r) for r in ROWS]
    A = np.stack([np.array(ROWS, float), np.ones(3)], 1)
    slope, icpt = np.linalg.lstsq(A, np.array(us), rcond=None)[0]
    print("%-34s %s us   per Mrow %.1f us   intercept %.1f us" % (name, " ".join("%7.1f" % u for u in us), slope * 1e6, icpt), flush=True)


def mk(rows, K):
    x = torch.randn(rows, K, device=dev); W = torch.randn(128, K, device=dev) * 0.1; b = torch.randn(128, device=dev)
    return x, W, b


def fwd(K, res, elu, wy):
    def f(rows):
        x, W, b = mk(rows, K)
        r = torch.randn(rows, 128, device=dev) if res else None
        cat = torch.empty(rows, 256, device=dev)
        part = kernels.new_elu_stats_part(rows, dev) if elu else None
        return timed(lambda: kernels.linear_fwd(x, W, b, r, cat[:, :128] if elu else None, wy, part))
    return f


def dgrad_elu(C):
    def f(rows):
        dy = torch.randn(rows, 128, device=dev); W = torch.randn(128, C, device=dev) * 0.1; x = torch.randn(rows, C, device=dev)
        mu = torch.randn(C, device=dev); B = torch.randn(C, device=dev); Cc = torch.randn(C, device=dev)
        return timed(lambda: kernels.linear_dgrad_elu(dy, W, x, mu, B, Cc, None))
    return f


def wgrad(C, bounded, nb):
    def f(rows):
        dy = torch.randn(rows, 128, device=dev) * 1e-4; x = torch.randn(rows, C, device=dev); mu = x.mean(0)
        inv = (1 / x.std(0)).contiguous()
        bounds = (torch.full((nb,), float(dy.abs().max()), device=dev), inv, rows) if bounded else None
        return timed(lambda: kernels.wgrad(dy, x, mu, bounds=bounds))
    return f


fit("fwd K=256 elu-only (w8)", fwd(256, False, True, False))
fit("fwd K=256 res+elu+y", fwd(256, True, True, True))
fit("fwd K=128 res+elu+y", fwd(128, True, True, True))
fit("fwd K=128 elu-only", fwd(128, False, True, False))
fit("dgrad+elu C=256", dgrad_elu(256))
fit("dgrad+elu C=128", dgrad_elu(128))
fit("wgrad bf16 C=256", wgrad(256, False, 0))
fit("wgrad fp16 C=256, 512 maxima", wgrad(256, True, 512))
fit("wgrad fp16 C=256, 19600 maxima", wgrad(256, True, 19600))
fit("wgrad fp16 C=128, 512 maxima", wgrad(128, True, 512))
